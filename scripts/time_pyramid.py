#!/usr/bin/env python3
"""Median / min time of the pyramid stage (HIP events on the library's stream) over repeated runs: A/B timing of kernel
variants.  python3 scripts/time_pyramid.py [N=512] [reps=12]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("3dsift_amd.capi")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
v = np.random.default_rng(0).random((n, n, n), dtype=np.float32)
ex = capi.CSIFT3D(v)
t = []
for _ in range(reps):
    ex.run_stages(1)
    t.append(ex.m_timer["d_BuildGSS"] * 1e3)
t = np.array(t[2:])
print("pyramid ms: median %.3f  min %.3f  max %.3f" % (np.median(t), t.min(), t.max()))
