#!/usr/bin/env python3
"""tiny end-to-end run (64^3) printing progress after every stage: first thing to run after touching a kernel's control flow"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("3dsift_amd.capi")
synth = importlib.import_module("3dsift_amd.synth")
vol = synth.blobs((64, 64, 64), seed=1234)
ex = capi.CSIFT3D(vol)
for st in (1, 3, 4, 5):
    ex.run_stages(st); print("stage", st, "ok", flush=True)
print(len(ex.GetKeypoints()[0]), "keypoints", flush=True)
