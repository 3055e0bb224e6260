#!/usr/bin/env python3
"""Profiling driver: build the Gaussian/DoG pyramid (stage 1) of a random NxNxN volume a few times.
Used under rocprofv3 (kernel-trace or --pmc passes) so the trace holds only the library's kernels:
    rocprofv3 --pmc SQ_WAVE_CYCLES ... -- python3 scripts/prof_pyramid.py 512 2 [stage]
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
capi = importlib.import_module("3dsift_amd.capi")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
stage = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if stage >= 3:
    synth = importlib.import_module("3dsift_amd.synth")
    import torch

    v = synth.blobs_torch((n, n, n), "cuda").cpu().numpy()
else:
    v = np.random.default_rng(0).random((n, n, n), dtype=np.float32)
ex = capi.CSIFT3D(v)
for _ in range(reps):
    ex.run_stages(stage)
    t = ex.m_timer
print({k: round(x * 1e3, 3) for k, x in t.items()})
