#!/usr/bin/env python3
"""Measure the HBM-side traffic of the pyramid build with rocprofv3 PMC counters (run on the GPU box):

    python3 scripts/measure_traffic.py [N=512] > profiles/rNN_pyramid_traffic.json

Two separate --pmc passes (FETCH_SIZE uses 3 of the 4 TCC slots, WRITE_SIZE 2: they do not fit together),
kernel-trace style options are NOT combined with them.  Units and gfx950 corrections follow
/opt/skills/guides/MI355X_MICROARCH.md section HBM: the counters are in KiB; FETCH_SIZE reports exactly half of the
bytes of a wide coalesced streaming read on gfx950 (so it is doubled), WRITE_SIZE is exact for streaming stores.
The sum covers every pyramid kernel of ONE KpSiftAlgorithm stage-1 run (k_march_level, k_downsample, k_conv_axis).
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = sys.argv[1] if len(sys.argv) > 1 else "512"
KERNELS = ("k_march_level", "k_downsample", "k_conv_axis")


def one_pass(counter):
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        env = dict(os.environ, TMPDIR="/tmp")
        subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                        os.path.join(ROOT, "scripts", "prof_pyramid.py"), n, "1"], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, env=env, cwd="/tmp")
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        per = collections.defaultdict(float)
        calls = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in KERNELS):
                name = r["Kernel_Name"].split("(")[0].replace("void s3d::", "")
                per[name] += float(r["Counter_Value"])
                calls[name] += 1
        return per, calls


fetch, calls = one_pass("FETCH_SIZE")
write, _ = one_pass("WRITE_SIZE")
rows = {}
for k in sorted(set(fetch) | set(write)):
    rows[k] = {"launches": calls.get(k, 0), "FETCH_SIZE_KiB": fetch.get(k, 0.0), "WRITE_SIZE_KiB": write.get(k, 0.0),
               "read_bytes_corrected": 2.0 * fetch.get(k, 0.0) * 1024.0, "write_bytes": write.get(k, 0.0) * 1024.0}
tot_r = sum(v["read_bytes_corrected"] for v in rows.values())
tot_w = sum(v["write_bytes"] for v in rows.values())
import importlib
source_sha = importlib.import_module("3dsift_amd.capi").kernel_source_sha()  # ties the numbers to the kernel sources they were measured on
print(json.dumps({"workload": f"{n}^3 fp32, pyramid build (stage 1) of one KpSiftAlgorithm", "kernel_source_sha": source_sha, "kernels": rows,
                  "total_read_bytes": tot_r, "total_write_bytes": tot_w, "total_bytes": tot_r + tot_w,
                  "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), KiB->bytes; WRITE_SIZE exact"}, indent=1))
