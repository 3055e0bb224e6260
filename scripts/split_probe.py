#!/usr/bin/env python3
"""split descriptor windows vs the unsplit form: descriptors (bitwise), second passes, times"""
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
capi = importlib.import_module("3dsift_amd.capi"); synth = importlib.import_module("3dsift_amd.synth")
for n in [int(a) for a in sys.argv[1:]] or [64, 96, 128, 192, 256]:
    vol = synth.blobs((n, n, n), seed=1234)
    res = {}
    for name, hk in (("split", 0), ("unsplit", 1)):
        with capi.hook("desc_nosplit", hk):
            ex = capi.CreateCSIFT3D(vol)
            ts = []
            for _ in range(6):
                ex.KpSiftAlgorithm(); ts.append(ex.m_timer["d_Extraction"] * 1e3)
            kp, ds = ex.GetKeypoints()
            res[name] = (ds, float(np.median(ts[2:])), ex.debug_counters()["desc_second_passes"])
            ex.close()
    a, b = res["split"][0], res["unsplit"][0]
    bad = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(axis=1))[0]
    print(n, "kp", len(a), "describe ms split %.3f unsplit %.3f" % (res["split"][1], res["unsplit"][1]), "second passes", res["split"][2], res["unsplit"][2],
          "rows that differ", len(bad), (float(np.abs(a[bad] - b[bad]).max()) if len(bad) else 0.0), bad[:8], flush=True)
