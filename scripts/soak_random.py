#!/usr/bin/env python3
"""Soak run (GPU box, not part of the test suite): N random draws of (shape, parameters, hooks) through the whole pipeline against the
oracle -- every level, extrema, keypoints, descriptors -- plus the native z-slab driver (whole / partial windows) on the same volume
against the single-volume result, and every third draw the matcher (three modes) on the keypoints of the volume and of a perturbed copy.   python3 scripts/soak_random.py [N=40] [seed=1] [kinds]
Prints one line per draw and a summary; exits non-zero on the first mismatch (the draw is printed so that it can be replayed)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
capi = importlib.import_module("3dsift_amd.capi")
synth = importlib.import_module("3dsift_amd.synth")
import oracle_lib as ol
from hipcheck import bits, compare_keypoints, extrema_table

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
KINDS = len(sys.argv) > 3 and sys.argv[3] == "kinds"   # other data than the synthetic blobs: a masked body, white noise, piecewise constant boxes
orc = ol.load("orc")
rng = np.random.default_rng(seed)
pool = [24, 32, 33, 40, 47, 48, 56, 64, 65, 70, 72, 80, 96, 100, 128, 130, 160, 192]
t0 = time.time()
nkp = 0
for case in range(N):
    shape = tuple(int(rng.choice(pool)) for _ in range(3))
    levels = int(rng.integers(1, 5))
    sd = float(np.round(rng.uniform(1.2, 2.6), 2))
    params = dict(num_kp_levels=levels, sigma_default=sd, sigma_n_default=float(np.round(rng.uniform(0.5, min(1.15, sd - 0.2)), 2)),
                  peak_thresh=float(np.round(rng.uniform(0.03, 0.25), 3)), max_eig_thres=float(np.round(rng.uniform(0.7, 0.95), 2)),
                  corner_thresh=float(np.round(rng.uniform(0.2, 0.6), 2)))
    hooks = {}
    for name, p in (("march_tiles", 0.4), ("dog_eager", 0.15), ("glast_eager", 0.15), ("lazy_generic", 0.15), ("desc_nosplit", 0.15), ("det_serial", 0.1), ("one_stream", 0.1)):
        if rng.random() < p:
            hooks[name] = 1
    if rng.random() < 0.1:
        hooks["desc_mass_shift"] = int(rng.integers(3, 10))
    if rng.random() < 0.1:
        hooks["list_cap"] = int(rng.integers(64, 600))
    vol = synth.blobs(shape, seed=5000 + seed * 1000 + case, noise=float(rng.choice([0.0, 0.01, 0.03])))
    kind = str(rng.choice(["blobs", "blobs", "masked", "noise", "steps"])) if KINDS else "blobs"
    if kind == "masked":      # a zero background around a body, like CT / MR volumes: flat descriptor windows, exact zeros
        zz, yy, xx = np.meshgrid(*[np.linspace(-1, 1, n) for n in shape], indexing="ij")
        vol = np.where(zz * zz + 0.8 * yy * yy + 1.2 * xx * xx < float(rng.uniform(0.3, 0.9)), vol, 0).astype(np.float32)
    elif kind == "noise":     # white noise: extrema everywhere (list regrow), tiny windows' worth of structure
        vol = np.random.default_rng(7000 + case).random(shape).astype(np.float32)
    elif kind == "steps":     # piecewise constant boxes: sharp edges, plateaus with zero gradient, ties between neighbours
        r2 = np.random.default_rng(8000 + case)
        vol = np.zeros(shape, np.float32)
        for _ in range(int(r2.integers(5, 40))):
            lo = [int(r2.integers(0, n - 2)) for n in shape]; hi = [int(r2.integers(l + 1, n)) for l, n in zip(lo, shape)]
            vol[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] += np.float32(r2.uniform(-1, 1))
    hooks["_kind"] = kind
    tag = (case, shape, params, hooks)
    import contextlib
    with contextlib.ExitStack() as st:
        for k, v in hooks.items():
            if not k.startswith("_"):
                st.enter_context(capi.hook(k, v))
        try:
            g = capi.CreateCSIFT3D(vol, **params).KpSiftAlgorithm()
        except capi.Sift3dError as e:
            if "129 taps" in str(e):
                print("draw", tag, "refused (kernel wider than 129 taps)")
                continue
            raise
        o = orc.extractor(vol, **params).run(5)
        try:
            assert g.num_octaves == o.num_octaves
            for oc in range(g.num_octaves):
                for i in range(levels + 3):
                    assert np.array_equal(bits(g.gss(oc, i)), bits(o.gss(oc, i))), ("gss", oc, i)
                for i in range(levels + 2):
                    assert np.array_equal(bits(g.dog(oc, i)), bits(o.dog(oc, i))), ("dog", oc, i)
            assert np.array_equal(extrema_table(g.extrema()), extrema_table(o.extrema())), "extrema"
            kp, desc = g.GetKeypoints()
            okp, odesc = o.keypoints()
            compare_keypoints(kp, desc, okp, odesc)
            # the native z-slab driver on the same volume (default parameters only take the hooks' paths; its own parameters: the draw's)
            # (r06: up to 8 ranks and three sharded octaves, the driver's own plan / forced partial / forced whole windows; every rank's solo
            # re-run must leave the results as they are)
            ranks = int(rng.integers(2, 9)); octs = int(rng.integers(0, 4)); partial = (None, True, False)[int(rng.integers(0, 3))]
            threads = bool(rng.integers(0, 2))   # (late r06) rank THREADS over the copy transport instead of ranks simulated by one thread
            ghost = bool(rng.integers(0, 2))     # ... and octave 0 on ghost zones
            sh = None; why = ""
            # the slab contexts only take half widths 2 .. 8 and planes of >= 40 voxels: a draw outside that runs the driver with a sigma schedule inside
            # it (against the single-volume extractor with the same parameters; the oracle comparison above stays the draw's own)
            params_n, kp_n, desc_n = params, kp, desc
            if min(shape[1], shape[2]) >= 40 and shape[0] >= 16 and (sd > 1.75 or sd < 1.4 or levels < 2):
                params_n = dict(params, sigma_default=float(np.round(rng.uniform(1.4, 1.75), 2)), num_kp_levels=int(rng.integers(2, 4)),
                                sigma_n_default=min(params["sigma_n_default"], 1.15))
                gn = capi.CreateCSIFT3D(vol, **params_n).KpSiftAlgorithm()
                kp_n, desc_n = gn.GetKeypoints()
                gn.close()
            while sh is None and ranks >= 2:   # (small draws: fewer ranks until the planes suffice; a forced form that is refused: the driver's rule)
                try:
                    if threads:
                        sh = capi.ShardedCSIFT3D(vol, devices=(0,) * ranks, sharded_octaves=octs, partial_windows=partial, transport="copies", ghost_octave0=ghost, **params_n)
                    else:
                        sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=ranks, sharded_octaves=octs, partial_windows=partial, ghost_octave0=ghost, **params_n)
                except capi.Sift3dError as e:
                    why = str(e)[:90]   # (the last refusal is printed with the draw: most small draws have no plane shape / depth the slab contexts take)
                    if partial is True:
                        partial = None
                    else:
                        ranks -= 1
            if sh is not None:
                try:
                    k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
                except capi.Sift3dError as e:   # (a plan that was accepted must run)
                    raise AssertionError(("sharded run failed", str(e), ranks, octs, partial, threads, params_n, sh.info()))
                assert np.array_equal(k2, kp_n), ("sharded keypoints", ranks, octs, partial, threads, params_n, sh.info())
                assert np.array_equal(d2, desc_n), ("sharded descriptors", ranks, octs, partial, threads, params_n, sh.info())
                for r in range(0 if threads else ranks):   # (the solo re-run of a rank: simulated ranks only)
                    sh.time_rank(r)
                k2, d2 = sh.GetKeypoints()
                assert np.array_equal(k2, kp_n) and np.array_equal(d2, desc_n), ("after the solo re-runs", ranks, octs, partial, sh.info())
                k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
                plan = sh.info()["stage_partial"]; planes = sh.info()["planes"]
                sh.close()
                assert np.array_equal(k2, kp_n) and np.array_equal(d2, desc_n), ("second run", ranks, octs, partial)
            # every other draw: the python z-slab driver (3dsift_amd/slab.py, ranks simulated on this GPU) with the draw's parameters
            if case % 2 == 1:
                slab = importlib.import_module("3dsift_amd.slab")
                w2 = int(rng.integers(2, 5)); o2 = int(rng.integers(1, 3)); p2 = bool(rng.integers(0, 2))
                try:
                    ex = slab.SlabExtractor((shape[2], shape[1], shape[0]), slab.SimComm(w2), sharded_octaves=o2, desc_partial=p2, **params)
                except (ValueError, capi.Sift3dError, AssertionError):
                    ex = None
                if ex is not None:
                    ex.load(volume=vol)
                    ex.KpSiftAlgorithm()
                    k3, d3 = ex.GetKeypoints()
                    ex.close()
                    for f in kp.dtype.names:
                        assert np.array_equal(k3[f], kp[f]), ("python slab driver", f, w2, o2, p2)
                    assert np.array_equal(d3, desc), ("python slab driver descriptors", w2, o2, p2)
            # every third draw: the matcher on this volume's keypoints against those of a perturbed copy (all three modes, every output)
            if case % 3 == 0 and len(kp) >= 2:
                vol2 = (vol + synth.blobs(shape, seed=9000 + case, noise=0.0) * np.float32(0.05)).astype(np.float32)
                g2 = capi.CreateCSIFT3D(vol2, **params).KpSiftAlgorithm()
                kpb, descb = g2.GetKeypoints()
                g2.close()
                if len(kpb) >= 1:
                    ax = np.stack([kp["rx"], kp["ry"], kp["rz"]], 1); bx = np.stack([kpb["rx"], kpb["ry"], kpb["rz"]], 1)
                    mt = capi.muBruteMatcher()
                    thr = float(rng.choice([0.7, 0.85, 0.95]))
                    for mode in (1, 2, 3):
                        got = mt._match(desc, ax, descb, bx, thr, mode)
                        want = orc.match(desc, ax, descb, bx, thr, mode)
                        for key in want:
                            assert np.array_equal(got[key], want[key]), ("matcher", mode, key, thr)
        except AssertionError as e:
            print("MISMATCH", tag, e, flush=True)
            sys.exit(1)
        nkp += len(kp)
        pys = "  py-slabs %d/%d%s" % (w2, o2, " partial" if p2 else "") if (case % 2 == 1 and ex is not None) else ""
        print("draw %3d ok  %-16s levels %d sigma %.2f hooks %s  kp %d%s%s" % (case, shape, levels, sd, hooks, len(kp), ("  no slabs: " + why) if sh is None else "  slabs %d/%d %s%s" % (ranks, octs, "".join("p" if p else "w" for p in plan), " threads" if threads else "") + (" ghost" if ghost else "") + " planes %s" % planes, pys), flush=True)
        g.close()
print("soak: %d draws, %d keypoints, %.0f s, all equal" % (N, nkp, time.time() - t0))
