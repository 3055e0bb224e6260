#!/usr/bin/env python3
"""BASELINE configs[2]: two 512^3 volumes, extract + muBruteMatcher::enhancedMatch on device-resident descriptors.
python3 scripts/bench_match.py [N=512]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
capi = importlib.import_module("3dsift_amd.capi")
synth = importlib.import_module("3dsift_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
exs = []
for shift in ((0, 0, 0), (1.0, 0, 0)):
    v = synth.blobs_torch((n, n, n), "cuda", seed=1234, shift=shift)
    torch.cuda.synchronize()
    e = capi.CSIFT3D(None, device_ptr=v.data_ptr(), shape=(n, n, n))
    e.KpSiftAlgorithm()
    print("extract ms", round(e.m_timer["d_TotalTime"] * 1e3, 2))
    exs.append(e)
(da, xa, na), (db, xb, nb) = exs[0].device_results(), exs[1].device_results()
m = capi.muBruteMatcher()
for mode in ("injectMatch", "enhancedMatch"):
    reps = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = getattr(m, mode)(da, xa, db, xb, 0.85, on_device=True, n=na, m=nb)
        wall = time.perf_counter() - t0
        reps.append(round(m.totalTime * 1e3, 2))
    print(mode, "device ms of every repetition", reps)
    fl = 2.0 * na * nb * 768 * (2 if mode == "enhancedMatch" else 1)
    print(f"{mode}: {na} x {nb} descriptors, device {m.totalTime*1e3:.2f} ms, wall {wall*1e3:.2f} ms, {fl/m.totalTime/1e12:.1f} TFLOP/s fp32, {len(r['pairs'])} pairs")
