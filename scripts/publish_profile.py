#!/usr/bin/env python3
"""Copy the file set `scripts/collect_profile.sh TAG` left in gpurun_out/ into profiles/ (and the traffic file to
profiles/pyramid_traffic_512.json, which bench.py reads), check the traffic stamp against the kernel sources, print the headline numbers.
python3 scripts/publish_profile.py r02c      (CPU; run from the repo root after the gpurun call)"""
import importlib, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
names = ["rocprofv3_kernel_stats.csv", "bench_under_rocprof.json", "bench.json", "pyramid_traffic_512.json", "pmc_k_describe.json", "pmc_k_describe_512.json",
         "pmc_k_march_level.json", "slab_sim.json", "kernel_times.txt", "levels_isolated.txt", "timeline.txt"]
for n in ["timeline_full.txt", "match_kernels.txt", "pmc_k_mark.json", "small_volumes.txt", "slab_1gpu.json", "step_times.txt",
          "kernel_resources.txt", "desc_ring_check.txt", "slab_kernel_sums.txt", "solo_rank3.txt", "solo_rank7.txt", "xfer.txt"]:   # since r03b / r04 / r05 / r06
    if os.path.exists(os.path.join(ROOT, "gpurun_out", f"{tag}_{n}")): names.append(n)
for n in names:
    shutil.copy(os.path.join(ROOT, "gpurun_out", f"{tag}_{n}"), os.path.join(ROOT, "profiles", f"{tag}_{n}"))
shutil.copy(os.path.join(ROOT, "gpurun_out", f"{tag}_pyramid_traffic_512.json"), os.path.join(ROOT, "profiles", "pyramid_traffic_512.json"))
shutil.copy(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_k_describe_512.json"), os.path.join(ROOT, "profiles", "pmc_k_describe_512.json"))
capi = importlib.import_module("3dsift_amd.capi")
d = json.load(open(os.path.join(ROOT, "profiles", "pyramid_traffic_512.json")))
b = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench.json")))
print("traffic stamp", d["kernel_source_sha"], "sources", capi.kernel_source_sha(), "OK" if d["kernel_source_sha"] == capi.kernel_source_sha() else "MISMATCH")
print("reads %.2f GB writes %.2f GB" % (d["total_read_bytes"] / 1e9, d["total_write_bytes"] / 1e9))
print("value %.1f Mvoxel/s  %.2f ms/step  stages %s" % (b["value"], b["ms_per_step"], b["stage_ms"]))
r = b["roofline"]
print("roofline frac (52 B moved) %.3f  every level built (68 B, timed) %s  of copy ceiling %.3f (%.0f GB/s)  traffic %s" % (r["frac"], r.get("frac_every_level_built"), r["frac_of_copy_ceiling"], r["copy_ceiling_GBs"], r["traffic"]))
print("descriptor", {k: v for k, v in b.get("descriptor", {}).items()})
print("nonaligned", b.get("nonaligned"))
print("ctor_ms", b.get("ctor_ms"), "get_keypoints_ms", b.get("get_keypoints_ms"))
m = b["matcher"]
print("matcher %.2f ms  %.1f TFLOP/s  parity %s" % (m["seconds"] * 1e3, m["roofline"]["achieved"], m["parity"]))
print("parity", b["parity"])
for l in open(os.path.join(ROOT, "profiles", f"{tag}_slab_sim.json")):
    if l.startswith("{"):
        j = json.loads(l)
        print("slab", j["config"]["workload"][-60:], "%.1f ms" % j["ms_per_step"], j["slab"].get("sim_rank_alone_ms", ""))
if os.path.exists(os.path.join(ROOT, "gpurun_out", f"{tag}_slab_1gpu.json")):
    shutil.copy(os.path.join(ROOT, "gpurun_out", f"{tag}_slab_1gpu.json"), os.path.join(ROOT, "profiles", "slab_1gpu.json"))
