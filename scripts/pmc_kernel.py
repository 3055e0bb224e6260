#!/usr/bin/env python3
"""Collect SQ counters for the library's kernels with rocprofv3 (run on the GPU box; --pmc passes only, never mixed
with trace options):

    python3 scripts/pmc_kernel.py <kernel-substring> <N> <stage> COUNTER [COUNTER ...]     (<= 8 SQ counters per pass)

Prints per-kernel sums over the launches of ONE run of `scripts/prof_pyramid.py N 1 stage` as JSON."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sub, n, stage = sys.argv[1], sys.argv[2], sys.argv[3]
counters = sys.argv[4:]
out = collections.defaultdict(lambda: collections.defaultdict(float))
for i in range(0, len(counters), 8):
    group = counters[i:i + 8]
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        env = dict(os.environ, TMPDIR="/tmp")
        subprocess.run(["rocprofv3", "--pmc", *group, "--output-format", "csv", "-d", d, "--", sys.executable,
                        os.path.join(ROOT, "scripts", "prof_pyramid.py"), n, "1", stage], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, env=env, cwd="/tmp")
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        seen = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("(")[0].replace("void s3d::", "")
                out[name][r["Counter_Name"]] += float(r["Counter_Value"])
                seen[name].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
        for k, v in seen.items():
            out[k]["launches"] = len(v)
import importlib
sys.path.insert(0, ROOT)
res = {k: dict(v) for k, v in out.items()}
res["kernel_source_sha"] = importlib.import_module("3dsift_amd.capi").kernel_source_sha()  # ties the counters to the kernel sources
print(json.dumps(res, indent=1))
