import importlib, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
capi = importlib.import_module("3dsift_amd.capi"); synth = importlib.import_module("3dsift_amd.synth")
for n in (256, 128):
    vol = synth.blobs_torch((n, n, n), "cuda", seed=1234); torch.cuda.synchronize()
    ex = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(n, n, n))
    ts = []
    for _ in range(8):
        ex.KpSiftAlgorithm(); ts.append(dict(ex.m_timer))
    med = {k: round(1e3 * float(np.median([t[k] for t in ts[2:]])), 3) for k in ("d_TotalTime", "d_BuildGSS", "d_Detect", "d_AssignOrientation", "d_Extraction")}
    print(n, med, "kp", len(ex.GetKeypoints()[0]))
