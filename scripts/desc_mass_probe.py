"""development probe: an S3D_EXP=6 build of kernels_desc.hip leaves (gradient mass, estimate, passes, fix_scale) in the first four
descriptor columns: how good is the first guess of the fixed-point unit?   S3D_LIB=variants/libsift3d_hip_exp6.so python3 scripts/desc_mass_probe.py"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
capi = importlib.import_module("3dsift_amd.capi")
synth = importlib.import_module("3dsift_amd.synth")
for n, seed in ((512, 1234), (256, 1234), (256, 7)):
    vol = synth.blobs_torch((n, n, n), "cuda", seed=seed)
    torch.cuda.synchronize()
    ex = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(n, n, n))
    ex.KpSiftAlgorithm()
    kp, ds = ex.GetKeypoints()
    mass, est, att, fs = ds[:, 0].astype(np.float64), ds[:, 1].astype(np.float64), ds[:, 2], ds[:, 3].astype(np.float64)
    ratio = mass / np.maximum(est, 1e-30)
    print(n, seed, "keypoints", len(kp), "second passes", int((att > 1).sum()), "ratio mass/estimate: min %.3g  1%% %.3g  median %.3g  99%% %.3g  max %.3g" % (
        ratio.min(), np.quantile(ratio, 0.01), np.median(ratio), np.quantile(ratio, 0.99), ratio.max()))
    print("   log2(fix_scale) histogram", np.unique(np.log2(fs).round().astype(int), return_counts=True), " mass*fix/2^31 max %.3f median %.3f" % ((mass * fs / 2**31).max(), np.median(mass * fs / 2**31)))
