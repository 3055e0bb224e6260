#!/usr/bin/env python3
"""Can two KpSiftAlgorithm runs share one GPU profitably?  Two handles on two host threads (each handle has its own streams)
versus the same two runs back to back.  python3 scripts/overlap_probe.py [N=512]"""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
capi = importlib.import_module("3dsift_amd.capi")
synth = importlib.import_module("3dsift_amd.synth")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
v1 = synth.blobs_torch((n, n, n), "cuda", seed=1234).cpu().numpy()
v2 = synth.blobs_torch((n, n, n), "cuda", seed=1235).cpu().numpy()
a, b = capi.CSIFT3D(v1), capi.CSIFT3D(v2)
for ex in (a, b): ex.KpSiftAlgorithm(); ex.KpSiftAlgorithm()
def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))
serial = timed(lambda: (a.KpSiftAlgorithm(), b.KpSiftAlgorithm()))
def both():
    t = threading.Thread(target=b.KpSiftAlgorithm); t.start(); a.KpSiftAlgorithm(); t.join()
conc = timed(both)
print("two volumes back to back %.2f ms, concurrently %.2f ms (%.2fx)" % (serial, conc, serial / conc))
