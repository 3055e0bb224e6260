#!/usr/bin/env python3
"""development probe: which keypoints of the 512^3 benchmark volume carry the largest descriptor error against the oracle, and what
do they look like?   python3 scripts/desc_err_probe.py [N=512]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
capi = importlib.import_module("3dsift_amd.capi"); synth = importlib.import_module("3dsift_amd.synth")
import oracle_lib as ol, torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
vol = synth.blobs_torch((n, n, n), "cuda", seed=1234); torch.cuda.synchronize()
ex = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(n, n, n)).KpSiftAlgorithm()
kp, ds = ex.GetKeypoints()
print("counters", ex.debug_counters())
orc = ol.load("orc"); orc.set_threads(max(1, min(64, (os.cpu_count() or 2) // 2)))
okp, od = orc.extractor(vol.cpu().numpy()).run(5).keypoints()
d = ds.astype(np.float64) - od
per = np.sqrt((d * d).mean(1)); mx = np.abs(d).max(1)
order = np.argsort(-mx)[:15]
print("global rms %.3e  per-kp rms: median %.2e p99 %.2e max %.2e | max abs: median %.2e p99 %.2e max %.2e" % (np.sqrt((d * d).mean()), np.median(per), np.percentile(per, 99), per.max(), np.median(mx), np.percentile(mx, 99), mx.max()))
for i in order:
    big = np.nonzero(np.abs(d[i]) > 1e-4)[0]
    print("kp %5d oct %d lvl %d xyz (%4d,%4d,%4d) rms %.2e maxabs %.2e n>1e-4 %3d  sum(d) %+.2e nnz_gpu %d nnz_cpu %d clamp_gpu %d clamp_cpu %d" % (
        i, kp["octave"][i], kp["level"][i], kp["x"][i], kp["y"][i], kp["z"][i], per[i], mx[i], len(big), d[i].sum(), (ds[i] > 0).sum(), (od[i] > 0).sum(),
        (ds[i] >= 0.0333).sum(), (od[i] >= 0.0333).sum()))
    for j in big[:6]:
        c, v = divmod(int(j), 12)
        print("      elem %3d cell (%d,%d,%d) vert %2d gpu %.6f cpu %.6f diff %+.2e" % (j, c & 3, (c >> 2) & 3, c >> 4, v, ds[i, j], od[i, j], d[i, j]))

# hypothesis: the large element errors come from the ROTATION (orientation sums are fp32 sums in another order on the GPU): the
# oracle's describe_one fed with the GPU's rotation must then reproduce the GPU's descriptor
o = orc.extractor(vol.cpu().numpy()).run(2)
for i in order[:8]:
    k = kp[i].copy()
    k["Rotation"] = kp[i]["Rotation"].reshape(3, 3).T.reshape(9)   # GetKeypoints returns it transposed (Src/cSIFT3D.cc:1214)
    lvl = o.gss(int(k["octave"]), int(k["level"]))
    unit = o.level_info(0, int(k["octave"]) * 6 + int(k["level"]))[1][0]
    _, dg = orc.describe_one(k, lvl, unit)
    dR = np.abs(kp[i]["Rotation"] - okp[i]["Rotation"]).max()
    ev = okp[i]["eigvalue"]
    print("kp %5d  |R_gpu - R_cpu| max %.2e  eig ratios %.3f %.3f | oracle(R_gpu) vs gpu: maxabs %.2e | oracle(R_gpu) vs oracle: maxabs %.2e | st rel diff %.2e" % (
        i, dR, abs(ev[0] / ev[1]), abs(ev[1] / ev[2]), np.abs(dg - ds[i]).max(), np.abs(dg - od[i]).max(),
        np.abs(kp[i]["str_tensor"] - okp[i]["str_tensor"]).max() / np.abs(okp[i]["str_tensor"]).max()))
