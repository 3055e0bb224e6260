#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/r04j_pytest.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/r04j_pytest.txt
tail -5 $OUT/r04j_pytest.txt
python3 scripts/step_times_probe.py > $OUT/r04j_step_times.txt 2>&1; grep -v amdgpu.ids $OUT/r04j_step_times.txt | head -8
