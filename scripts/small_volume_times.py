#!/usr/bin/env python3
"""stage times of small volumes (launch-latency bound: 256^3 runs at half the Gvoxel/s of 512^3); S3D_LIB selects a variant library.
python3 scripts/small_volume_times.py [N ...]"""
import importlib, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
capi = importlib.import_module("3dsift_amd.capi"); synth = importlib.import_module("3dsift_amd.synth")
for n in ([int(a) for a in sys.argv[1:]] or [256, 128]):
    vol = synth.blobs_torch((n, n, n), "cuda", seed=1234); torch.cuda.synchronize()
    ex = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(n, n, n))
    ts = []
    for _ in range(8):
        ex.KpSiftAlgorithm(); ts.append(dict(ex.m_timer))
    med = {k: round(1e3 * float(np.median([t[k] for t in ts[2:]])), 3) for k in ("d_TotalTime", "d_BuildGSS", "d_Detect", "d_AssignOrientation", "d_Extraction")}
    print(os.path.basename(os.environ.get("S3D_LIB", "default")), n, med, "kp", len(ex.GetKeypoints()[0]), flush=True)
