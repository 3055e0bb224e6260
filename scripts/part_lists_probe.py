"""timing only (wrong results): rank 3's solo step with its own record list alone / the foreign lists alone in the partial-window launches"""
import importlib, os, sys, numpy as np
sys.path.insert(0, '.')
capi = importlib.import_module("3dsift_amd.capi"); synth = importlib.import_module("3dsift_amd.synth")
import torch
vol = synth.blobs_torch((512, 1024, 1024), "cuda", seed=4321).cpu().numpy()
sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=8)
for _ in range(3): sh.KpSiftAlgorithm()
print("S3D_PART_LISTS", os.environ.get("S3D_PART_LISTS"), [round(min(sh.time_rank(r) for _ in range(4)) * 1e3, 2) for r in (0, 3, 7)], flush=True)
