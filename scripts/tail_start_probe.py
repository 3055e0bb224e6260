"""where the tail's pipeline is released on the tail rank (S3D_TAIL_START in a -DS3D_DEV_SWITCHES library): simulated 8-rank step and every rank's solo time"""
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
print("S3D_TAIL_START", os.environ.get("S3D_TAIL_START"), "step %.2f ms" % np.median(ts[2:]), "ranks alone", pr, "planes", sh.info()["planes"], flush=True)
