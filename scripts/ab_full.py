#!/usr/bin/env python3
"""A/B of variant libraries on the whole KpSiftAlgorithm (512^3 blob volume): stage times (median of 6) and a hash of the
descriptors + keypoint records.   python3 scripts/ab_full.py [lib.so ...]"""
import hashlib, importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, ROOT)
    capi = importlib.import_module("3dsift_amd.capi")
    synth = importlib.import_module("3dsift_amd.synth")
    import torch
    n = 512
    vol = synth.blobs_torch((n, n, n), "cuda", seed=1234)
    torch.cuda.synchronize()
    ex = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(n, n, n))
    ts = []
    for _ in range(8):
        ex.KpSiftAlgorithm(); ts.append(dict(ex.m_timer))
    kp, ds = ex.GetKeypoints()
    med = {k: 1e3 * float(np.median([t[k] for t in ts[2:]])) for k in ("d_TotalTime", "d_BuildGSS", "d_Detect", "d_AssignOrientation", "d_Extraction")}
    h = hashlib.sha1(kp.tobytes() + ds.tobytes()).hexdigest()[:12]
    try:
        redo = ex.debug_counters()["desc_second_passes"]
    except Exception:
        redo = -1
    print("%-32s total %.2f pyr %.2f det %.2f ori %.2f desc %.2f ms  kp %d hash %s redo %d" % (os.path.basename(os.environ.get("S3D_LIB") or "default"), med["d_TotalTime"], med["d_BuildGSS"], med["d_Detect"], med["d_AssignOrientation"], med["d_Extraction"], len(kp), h, redo), flush=True)
    sys.exit(0)
for lib in [None] + sys.argv[1:]:
    env = dict(os.environ)
    if lib: env["S3D_LIB"] = os.path.abspath(lib)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=False)
