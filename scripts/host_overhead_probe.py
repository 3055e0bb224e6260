#!/usr/bin/env python3
"""wall time per KpSiftAlgorithm call against the event time of its stages (512^3): what the host adds around the GPU work"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
capi = importlib.import_module("3dsift_amd.capi"); synth = importlib.import_module("3dsift_amd.synth")
n = 512
vol = synth.blobs_torch((n, n, n), "cuda", seed=1234); torch.cuda.synchronize()
ex = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(n, n, n))
for _ in range(3): ex.KpSiftAlgorithm()
w, d = [], []
for _ in range(20):
    t0 = time.perf_counter(); ex.KpSiftAlgorithm(); w.append(time.perf_counter() - t0); d.append(ex.m_timer["d_TotalTime"])
print("wall %.3f ms  event total %.3f ms  difference %.3f ms (medians of 20)" % (1e3 * np.median(w), 1e3 * np.median(d), 1e3 * (np.median(w) - np.median(d))))
