"""the native z-slab driver on 8 simulated ranks at 1024x1024x512: the step with all ranks on the one GPU and every rank's solo step
(sift3d_test_sharded_time_rank); S3D_LIB selects a variant library (a -DS3D_DEV_SWITCHES build reads S3D_PARTIAL_MAX_RANKS)"""
import importlib, os, sys, numpy as np
sys.path.insert(0, '.')
capi = importlib.import_module("3dsift_amd.capi"); synth = importlib.import_module("3dsift_amd.synth")
import torch
nx, ny, nz = 1024, 1024, 512
vol = synth.blobs_torch((nz, ny, nx), "cuda", seed=4321).cpu().numpy()
sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=8)
ts = []
for _ in range(6):
    sh.KpSiftAlgorithm(); ts.append(sh.info()["seconds"] * 1e3)
pr = [round(min(sh.time_rank(r) for _ in range(3)) * 1e3, 2) for r in range(8)]
print(os.path.basename(os.environ.get("S3D_LIB", "default")), "step %.2f ms" % np.median(ts[2:]), "ranks alone", pr, "planes", sh.info()["planes"], flush=True)
