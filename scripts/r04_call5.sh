#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/r04e_pytest.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/r04e_pytest.txt
tail -5 $OUT/r04e_pytest.txt
timeout -k 10 400 python3 bench.py > $OUT/r04e_bench.json 2> $OUT/r04e_bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/r04e_bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "stage_ms"): print(k, j[k])
print("roofline", {k: j["roofline"][k] for k in ("frac", "seconds", "frac_survey")})
for k in ("thin", "nonaligned", "pipeline2", "slab", "cpu_baseline", "matcher"):
    print(k, json.dumps(j.get(k))[:900])
PY
