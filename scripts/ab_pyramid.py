#!/usr/bin/env python3
"""A/B of kernel-variant libraries (scripts/build_variant.sh): for each library, in its own process, the pyramid time at
512^3 (HIP events, median of 10) and a hash of every GSS/DoG level of a 200x168x136 volume + a 256^3 volume's keypoint-free
pyramid (bit-exactness against the default build).   python3 scripts/ab_pyramid.py [lib.so ...]"""
import hashlib, importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, ROOT)
    capi = importlib.import_module("3dsift_amd.capi")
    h = hashlib.sha1()
    for shape in (() if os.environ.get("S3D_AB_NOHASH") else ((136, 168, 200), (256, 256, 256))):
        v = np.random.default_rng(1).random(shape, dtype=np.float32)
        ex = capi.CSIFT3D(v); ex.run_stages(1)
        for o in range(ex.num_octaves):
            for i in range(6): h.update(ex.gss(o, i).tobytes())
            for i in range(5): h.update(ex.dog(o, i).tobytes())
        del ex
    n = 512
    v = np.random.default_rng(0).random((n, n, n), dtype=np.float32)
    ex = capi.CSIFT3D(v)
    t = []
    for _ in range(12):
        ex.run_stages(1); t.append(ex.m_timer["d_BuildGSS"] * 1e3)
    t = np.array(t[2:])
    print("%-40s pyramid ms median %.3f min %.3f   hash %s" % (os.path.basename(os.environ.get("S3D_LIB", "default")) + " " + os.environ.get("S3D_TAG", ""), np.median(t), t.min(), h.hexdigest()[:12]), flush=True)
    sys.exit(0)
for lib in [None] + sys.argv[1:]:
    env = dict(os.environ)
    if lib: env["S3D_LIB"] = os.path.abspath(lib)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=False)
