#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_pipeline or separable or golden or nondefault or dog_elision" > $OUT/r04c_pytest.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/r04c_pytest.txt
tail -3 $OUT/r04c_pytest.txt
python3 scripts/small_volume_times.py 256 128 64 > $OUT/r04c_small_volumes.txt 2>&1; cat $OUT/r04c_small_volumes.txt
bash scripts/timeline.sh 512 > $OUT/r04c_timeline.txt 2>&1; tail -8 $OUT/r04c_timeline.txt
timeout -k 10 500 bash scripts/sweep_sched3.sh > $OUT/r04c_sweep.txt 2>&1; cat $OUT/r04c_sweep.txt
