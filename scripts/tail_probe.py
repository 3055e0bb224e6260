"""the tail of the sharded 1024x1024x512 volume (octaves >= 2: a seeded extractor of 256x256x128) alone on the GPU: stage times, keypoints"""
import importlib, sys, time, numpy as np
sys.path.insert(0, '.')
capi = importlib.import_module("3dsift_amd.capi"); synth = importlib.import_module("3dsift_amd.synth")
import torch
nx, ny, nz = 1024, 1024, 512
vol = synth.blobs_torch((nz, ny, nx), "cuda", seed=4321); torch.cuda.synchronize()
ex = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(nz, ny, nx)).KpSiftAlgorithm()
kp, _ = ex.GetKeypoints()
print("whole volume:", {k: round(v * 1e3, 3) for k, v in ex.m_timer.items() if v}, "keypoints per octave", np.bincount(kp["octave"]).tolist())
seed = ex.gss(2, 0)
t = capi.SeededCSIFT3D((nz // 4, ny // 4, nx // 4), 2, ex.num_octaves)
t.seed(seed)
for _ in range(5):
    t0 = time.perf_counter(); t.KpSiftAlgorithm(); dt = time.perf_counter() - t0
    print("tail alone: wall %.3f ms" % (dt * 1e3), {k: round(v * 1e3, 3) for k, v in t.m_timer.items() if v}, "keypoints", len(t.GetKeypoints(with_desc=False)[0]))
