"""one configuration of the native z-slab driver on simulated ranks against the single-volume extractor, with progress lines (the guard after a fault):
   python scripts/native_repro.py NXxNYxNZ ranks sharded_octaves"""
import faulthandler, importlib, sys, numpy as np
faulthandler.enable()
sys.path.insert(0, '.')
capi = importlib.import_module("3dsift_amd.capi"); synth = importlib.import_module("3dsift_amd.synth")
import torch
nx, ny, nz = (int(v) for v in sys.argv[1].split("x")); ranks = int(sys.argv[2]); octs = int(sys.argv[3])
vol = synth.blobs_torch((nz, ny, nx), "cuda", seed=4321)
ex = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(nz, ny, nx)).KpSiftAlgorithm(); kp, ds = ex.GetKeypoints(); ex.close()
vol = vol.cpu().numpy()
print("single ok", len(kp), flush=True)
sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=ranks, sharded_octaves=octs)
print(sh.info(), flush=True)
sh.KpSiftAlgorithm()
print("run ok", flush=True)
k2, d2 = sh.GetKeypoints()
print("fetched", len(k2), np.array_equal(k2, kp), np.array_equal(d2, ds), flush=True)
for r in range(ranks):
    print("rank", r, "alone %.2f ms" % (sh.time_rank(r) * 1e3), flush=True)
k3, d3 = sh.KpSiftAlgorithm().GetKeypoints()
print("again", np.array_equal(k3, kp), np.array_equal(d3, ds), flush=True)
