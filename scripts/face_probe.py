import importlib, sys, os
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
capi = importlib.import_module("3dsift_amd.capi")
rng = np.random.Generator(np.random.PCG64(8))
d = rng.normal(size=(400000, 3)).astype(np.float32)
d[:1000] *= np.float32(1e-2)
fa, ba = capi.face_lookup(d, route=0)
fb, bb = capi.face_lookup(d, route=1)
bad = np.nonzero(fa != fb)[0]
print("face mismatches", len(bad), "of", len(d))
for i in bad[:10]: print(d[i], fa[i], fb[i], ba[i], bb[i])
err = np.abs(ba - bb).max(1)
print("max bary err", err.max(), "at", d[err.argmax()], fa[err.argmax()], ba[err.argmax()], bb[err.argmax()])
print("per face max err", [float(err[fb == f].max()) for f in range(20)])
