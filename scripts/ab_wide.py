#!/usr/bin/env python3
"""A/B in one process on one box: the pyramid stage of a 512^3 volume with the 64 x 32 tiles (hook march_tiles = 0, the product
rule) and with 32 x 32 tiles everywhere (march_tiles = 2), alternating; prints d_BuildGSS medians and the full-step time."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
capi = importlib.import_module("3dsift_amd.capi")
synth = importlib.import_module("3dsift_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
vol = synth.blobs_torch((n, n, n), "cuda").cpu().numpy()
ex = capi.CSIFT3D(vol)
res = {0: [], 2: []}
tot = {0: [], 2: []}
for rnd in range(6):
    for mode in (0, 2):
        with capi.hook("march_tiles", mode):
            for _ in range(4):
                ex.KpSiftAlgorithm()
                t = ex.m_timer
                res[mode].append(t["d_BuildGSS"] * 1e3); tot[mode].append(t["d_TotalTime"] * 1e3)
for mode in (0, 2):
    a = np.array(res[mode][4:]); b = np.array(tot[mode][4:])
    print("march_tiles", mode, "pyramid median %.3f min %.3f ms   step median %.3f ms" % (np.median(a), a.min(), np.median(b)))
