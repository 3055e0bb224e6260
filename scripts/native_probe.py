import importlib, sys, os, time, faulthandler
faulthandler.dump_traceback_later(45, exit=True)
import numpy as np
sys.path.insert(0, os.getcwd())
capi = importlib.import_module("3dsift_amd.capi"); synth = importlib.import_module("3dsift_amd.synth")
import torch
dims = tuple(int(v) for v in sys.argv[1].split("x")); ranks = int(sys.argv[2])
nx, ny, nz = dims
t = time.time()
vol = synth.blobs_torch((nz, ny, nx), "cuda", seed=4321).cpu().numpy()
print("volume", round(time.time() - t, 2), flush=True)
t = time.time()
sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=ranks)
print("create", round(time.time() - t, 2), sh.info(), flush=True)
for i in range(3):
    t = time.time(); sh.KpSiftAlgorithm(); print("run", i, round(time.time() - t, 3), sh.info()["seconds"], flush=True)
kp, ds = sh.GetKeypoints(); print("keypoints", len(kp), flush=True)
sh.close()
