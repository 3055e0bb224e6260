#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_pipeline or separable or golden or nondefault or dog_elision or two_stream" > $OUT/r04d_pytest.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/r04d_pytest.txt
tail -3 $OUT/r04d_pytest.txt
python3 scripts/small_volume_times.py 256 128 64 > $OUT/r04d_small_volumes.txt 2>&1; cat $OUT/r04d_small_volumes.txt
bash scripts/timeline.sh 512 > $OUT/r04d_timeline.txt 2>&1; tail -8 $OUT/r04d_timeline.txt
bash scripts/timeline.sh 128 > $OUT/r04d_timeline128.txt 2>&1; cat $OUT/r04d_timeline128.txt
python3 scripts/ab_full.py > $OUT/r04d_ab_full.txt 2>&1; cat $OUT/r04d_ab_full.txt
