import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
capi = importlib.import_module("3dsift_amd.capi"); synth = importlib.import_module("3dsift_amd.synth")
n = 512; exs = []
for shift in ((0, 0, 0), (1.0, 0, 0)):
    v = synth.blobs_torch((n, n, n), "cuda", seed=1234, shift=shift); torch.cuda.synchronize()
    e = capi.CSIFT3D(None, device_ptr=v.data_ptr(), shape=(n, n, n)); e.KpSiftAlgorithm(); exs.append(e)
(da, xa, na), (db, xb, nb) = exs[0].device_results(), exs[1].device_results()
m = capi.muBruteMatcher()
for mode in ("injectMatch", "enhancedMatch"):
    ts = []
    for _ in range(6):
        r = getattr(m, mode)(da, xa, db, xb, 0.85, on_device=True, n=na, m=nb); ts.append(m.totalTime * 1e3)
    print(os.path.basename(os.environ.get("S3D_LIB", "default")), mode, "min %.3f ms" % min(ts), "exact rows", m.exact_rows, "pairs", len(r["pairs"]))
# identical sets (every row has an exact duplicate): the worst case for the candidate lists
r = m.injectMatch(da, xa, da, xa, 0.85, on_device=True, n=na, m=na); print("self match: %.3f ms, exact rows %d" % (m.totalTime * 1e3, m.exact_rows))
