#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profile.sh r01c
# writes gpurun_out/<tag>_*; copy the files into profiles/ afterwards.
set -e
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
# compile-time gate first (no GPU time is spent on sources whose hot kernels spill): resource report + the ring contract of k_describe
python3 $REPO/scripts/kernel_resources.py --check --out $OUT/${TAG}_kernel_resources.txt > /dev/null
python3 $REPO/scripts/check_desc_ring.py > $OUT/${TAG}_desc_ring_check.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
# PMC passes FIRST (separate --pmc runs, no trace options): bench.py reports roofline.traffic / descriptor.roofline only from files
# stamped with the kernel sources it runs on, so they are installed under profiles/ before the bench lines are taken
python3 $REPO/scripts/measure_traffic.py 512 > $OUT/${TAG}_pyramid_traffic_512.json
cp $OUT/${TAG}_pyramid_traffic_512.json $REPO/profiles/pyramid_traffic_512.json
python3 $REPO/scripts/pmc_kernel.py k_describe 512 5 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES > $OUT/${TAG}_pmc_k_describe.json
python3 - $OUT/${TAG}_pmc_k_describe.json $REPO/profiles/pmc_k_describe_512.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
k = [v for n, v in j.items() if n.startswith("k_describe")][0]
k["kernel_source_sha"] = j["kernel_source_sha"]
json.dump(k, open(sys.argv[2], "w"), indent=1)
PY
cp $REPO/profiles/pmc_k_describe_512.json $OUT/${TAG}_pmc_k_describe_512.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $REPO/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $OUT/${TAG}_bench_under_rocprof.json 2> /tmp/prof_$TAG.err || true
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/${TAG}_rocprofv3_kernel_stats.csv
python3 $REPO/bench.py > $OUT/${TAG}_bench.json 2> /dev/null
# the N = 1 record of configs[3] (the bench line's "slab" block) stamped with the kernel sources: a --gpus N run forms slab.speedup_vs_1gpu from it
python3 - $OUT/${TAG}_bench.json $REPO/profiles/slab_1gpu.json $REPO <<'PY'
import importlib, json, sys
sys.path.insert(0, sys.argv[3])
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
sl = j.get("slab") or {}
if "ms_per_step" in sl:
    json.dump({"kernel_source_sha": importlib.import_module("3dsift_amd.capi").kernel_source_sha(), "dims": "1024x1024x512", "ms_per_step": sl["ms_per_step"],
               "Mvoxels_per_s": sl["value"], "keypoints": sl["keypoints"], "steps": sl.get("steps"), "warmup": sl.get("warmup"),
               "stage_ms_last_step": sl.get("last_step_ms")}, open(sys.argv[2], "w"), indent=1)
PY
cp $REPO/profiles/slab_1gpu.json $OUT/${TAG}_slab_1gpu.json
head -12 $OUT/${TAG}_rocprofv3_kernel_stats.csv
S3D_HOOKS=one_stream=1 python3 $REPO/scripts/pmc_kernel.py k_march_level 512 1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES > $OUT/${TAG}_pmc_k_march_level.json
# z-slab workload on one GPU: the plain single-GPU run, the NATIVE driver on 2 / 4 / 8 simulated ranks (each with every rank's solo step:
# sim_rank_alone_ms) and the python driver on 8
cd $REPO
( python3 bench.py --workload slab --steps 5 --warmup 2 2>/dev/null; for r in 2 4 8; do python3 bench.py --workload slab --native --sim-ranks $r --steps 5 --warmup 2 2>/dev/null; done;
  python3 bench.py --workload slab --native --sim-ranks 8 --whole-windows --steps 5 --warmup 2 2>/dev/null; python3 bench.py --workload slab --sim-ranks 8 --steps 3 --warmup 1 2>/dev/null;
  python3 bench.py --workload slab --native --rank-threads 8 --steps 5 --warmup 2 2>/dev/null;
  python3 bench.py --workload slab --native --sim-ranks 8 --ghost --steps 5 --warmup 2 2>/dev/null ) > $OUT/${TAG}_slab_sim.json   # (the last two: 8 rank THREADS on the one GPU over the copy transport; octave 0 on ghost zones)
bash $REPO/scripts/slab_kernel_sums.sh 8 native > $OUT/${TAG}_slab_kernel_sums.txt 2>&1
bash $REPO/scripts/solo_rank_trace.sh 3 > $OUT/${TAG}_solo_rank3.txt 2>&1
FULL=1 bash $REPO/scripts/solo_rank_trace.sh 7 > $OUT/${TAG}_solo_rank7.txt 2>&1   # (the tail rank, every launch)
python3 $REPO/scripts/xfer_probe.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_xfer.txt
bash $REPO/scripts/kernel_times.sh > $OUT/${TAG}_kernel_times.txt 2>&1
bash $REPO/scripts/level_times.sh 512 > $OUT/${TAG}_levels_isolated.txt 2>&1
bash $REPO/scripts/timeline.sh 512 > $OUT/${TAG}_timeline.txt 2>&1
bash $REPO/scripts/timeline_full.sh 512 > $OUT/${TAG}_timeline_full.txt 2>&1
bash $REPO/scripts/match_kernel_times.sh > $OUT/${TAG}_match_kernels.txt 2>&1
cd /tmp
python3 $REPO/scripts/pmc_kernel.py k_mark 512 3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TA_BUSY_avr > $OUT/${TAG}_pmc_k_mark.json
python3 $REPO/scripts/small_volume_times.py 256 128 64 > $OUT/${TAG}_small_volumes.txt 2>/dev/null
python3 $REPO/scripts/step_times_probe.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_step_times.txt
cat $OUT/${TAG}_bench.json
