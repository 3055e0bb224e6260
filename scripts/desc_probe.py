"""development probe: read the step counters an S3D_EXP=5 build leaves in the descriptor rows"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
capi = importlib.import_module("3dsift_amd.capi")
synth = importlib.import_module("3dsift_amd.synth")
n = 512
vol = synth.blobs_torch((n, n, n), "cuda", seed=1234)
torch.cuda.synchronize()
ex = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(n, n, n))
ex.KpSiftAlgorithm()
kp, ds = ex.GetKeypoints()
steps, lanes, pops = ds[:, 0].astype(np.float64), ds[:, 1].astype(np.float64), ds[:, 2].astype(np.float64)
print("keypoints", len(kp), "per octave", np.bincount(kp["octave"]).tolist(), "per level", np.bincount(kp["level"]).tolist())
print("wave-0 steps total %.3e  mean %.1f ; block lane-steps (visited voxels) total %.3e ; wave-0 pops %.3e" % (steps.sum(), steps.mean(), lanes.sum(), pops.sum()))
print("lane utilisation of the march = visited / (4 waves * steps * 64) ~ %.3f" % (lanes.sum() / (4 * steps.sum() * 64)))
print("active fraction ~ pops*64*4 / visited = %.3f" % (pops.sum() * 64 * 4 / lanes.sum()))
for lv in (1, 2, 3):
    m = kp["level"] == lv
    print("level", lv, "n", m.sum(), "steps/kp", steps[m].mean(), "visited/kp", lanes[m].mean())
print(ex.m_timer)
