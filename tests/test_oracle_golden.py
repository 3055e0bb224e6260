"""CPU tests: our restatement (oracle/sift3d_oracle.c) against the golden vectors the REAL
reference produced (tests/golden/make_golden.py).  Everything here is bit-exact."""
import hashlib

import numpy as np
import pytest

from conftest import golden


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_g1_taps(orc):
    """G1: the 1-D tap vectors of GaussianSmooth_3D (Src/cSIFT3D.cc:546-572), pinned through the reference's impulse response: a unit
    impulse leaves tap[d] after the x pass, and the y and z passes multiply by the centre tap: line[c+d] = rn(tc * rn(tc * tap[d]))."""
    g = golden("g1_taps.npz")
    i = 0
    while f"sigma_{i}" in g:
        t = orc.gaussian_taps(g[f"sigma_{i}"])
        line = g[f"line_{i}"]
        hw, c = (len(t) - 1) // 2, len(line) // 2
        tc = t[hw]
        want = np.zeros_like(line)
        want[c - hw:c + hw + 1] = (tc * (tc * t).astype(np.float32)).astype(np.float32)  # tap index d+hw multiplies src[p-d]: symmetric
        assert np.array_equal(bits(want), bits(line)), (i, float(g[f"sigma_{i}"]))
        i += 1
    assert i >= 6


def test_g2_smooth(orc):
    g = golden("g2_smooth.npz")
    for name, src in (("v1_s0", "v1"), ("v1_s5", "v1"), ("v2_s2", "v2")):
        out = orc.gaussian_smooth(g[src], g[name + "_sigma"])
        assert np.array_equal(bits(out), bits(g[name])), name


def test_g3_pyramid_bitexact(orc, synth):
    g = golden("g3_pyramid.npz")
    vol = synth.blobs((28, 20, 24), seed=11, noise=0.01, nblobs=12)
    assert np.array_equal(bits(vol), bits(g["vol"])), "synthetic generator drifted"
    ex = orc.extractor(g["vol"]).run(3)
    assert ex.num_octaves == int(g["noct"])
    assert np.array_equal(bits(ex.input()), bits(g["input"]))
    for o in range(ex.num_octaves):
        for i in range(6):
            assert np.array_equal(bits(ex.gss(o, i)), bits(g[f"gss_{o}_{i}"])), (o, i)
        for i in range(5):
            assert np.array_equal(bits(ex.dog(o, i)), bits(g[f"dog_{o}_{i}"])), (o, i)
    e = ex.extrema()
    got = np.stack([e["octave"], e["level"], e["x"].astype(np.int32), e["y"].astype(np.int32), e["z"].astype(np.int32)], 1)
    assert np.array_equal(got, g["extrema"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g4_g6_full_pipeline(orc, synth, tag):
    g4, g6 = golden("g4_hashes.npz"), golden("g6_keypoints.npz")
    shape = tuple(int(v) for v in g4[f"{tag}_shape"])
    vol = synth.blobs(shape, seed=int(g4[f"{tag}_seed"]), noise=float(g4[f"{tag}_noise"]))
    assert sha(vol) == str(g4[f"{tag}_vol_sha"]), "synthetic generator drifted"
    ex = orc.extractor(vol).run(5)
    assert ex.num_octaves == int(g4[f"{tag}_noct"])
    assert sha(ex.input()) == str(g4[f"{tag}_input_sha"])
    want = {}
    for h in g4[f"{tag}_hashes"]:
        name, shell, digest = str(h).split(":")
        want[name] = (int(shell), digest)
    absmax = []
    for o in range(ex.num_octaves):
        for i in range(6):
            lv = ex.gss(o, i)
            shell, digest = want[f"gss_{o}_{i}"]
            assert (sha(lv[1:-1, 1:-1, 1:-1]) if shell else sha(lv)) == digest, ("gss", o, i)
            assert np.float32(ex.level_info(0, o * 6 + i)[2]) == g4[f"{tag}_gss_scales"][o * 6 + i]
        for i in range(5):
            lv = ex.dog(o, i)
            shell, digest = want[f"dog_{o}_{i}"]
            assert (sha(lv[1:-1, 1:-1, 1:-1]) if shell else sha(lv)) == digest, ("dog", o, i)
            absmax.append(np.abs(lv).max())
    # abs-max of the levels the detector thresholds on (1..3) must match bit-for-bit
    am = np.array(absmax, np.float32).reshape(-1, 5)[:, 1:4]
    assert np.array_equal(am, g4[f"{tag}_dog_absmax"].reshape(-1, 5)[:, 1:4])
    e = ex.extrema()
    got = np.stack([e["octave"], e["level"], e["x"].astype(np.int32), e["y"].astype(np.int32), e["z"].astype(np.int32)], 1)
    assert np.array_equal(got, g4[f"{tag}_extrema"])
    assert np.array_equal(e["scale"], g4[f"{tag}_extrema_scale"])
    kp, desc = ex.keypoints()
    gk = g6[f"{tag}_kp"]
    assert len(kp) == len(gk)
    for f in ("x", "y", "z", "scale", "octave", "level", "rx", "ry", "rz", "win", "eigvalue", "Rotation", "str_tensor"):
        assert np.array_equal(kp[f], gk[f]), f
    assert np.array_equal(bits(desc), bits(g6[f"{tag}_desc"]))
    assert np.allclose((desc.astype(np.float64) ** 2).sum(1), 1.0, atol=1e-5)


def test_g5_orientation_every_extremum(orc, synth):
    g4, g5 = golden("g4_hashes.npz"), golden("g5_orient.npz")
    shape = tuple(int(v) for v in g4["a_shape"])
    vol = synth.blobs(shape, seed=int(g4["a_seed"]), noise=float(g4["a_noise"]))
    ex = orc.extractor(vol).run(3)
    e = ex.extrema()
    assert len(e) == len(g5["codes"])
    seen = set()
    for j, k in enumerate(e):
        lvl = ex.gss(int(k["octave"]), int(k["level"]))
        unit = ex.level_info(0, int(k["octave"]) * 6 + int(k["level"]))[1][0]
        code, ko = orc.orient_one(k, lvl, unit, np.float32(1.5) * k["scale"])
        assert code == int(g5["codes"][j]), j
        seen.add(code)
        assert np.array_equal(ko["win"], g5["win"][j])
        assert np.array_equal(ko["str_tensor"], g5["str_tensor"][j])
        if code != -1:
            assert np.array_equal(ko["eigvalue"], g5["eigvalue"][j])
        if code == 1:
            assert np.array_equal(ko["Rotation"], g5["Rotation"][j])
    assert 1 in seen and (-2 in seen or -3 in seen)


def test_g7_mesh_and_face_lookup(orc):
    g = golden("g7_mesh.npz")
    v, idx = orc.mesh()
    assert np.array_equal(bits(v), bits(g["verts"])) and np.array_equal(idx, g["idx"])
    for j, d in enumerate(g["dirs"]):
        f, b = orc.intersect(d)
        assert f == int(g["faces"][j]), j
        if f >= 0:
            assert np.array_equal(bits(b), bits(g["bary"][j])), j
    assert (g["faces"] < 0).any() and len(set(g["faces"].tolist())) == 21


def test_g8_matcher(orc):
    g = golden("g8_match.npz")
    perm = g["perm"]
    sets = {"p": (g["da"], g["xa"], g["db"], g["xb"]), "q": (g["da"], g["xa"], g["db"][perm], g["xb"][perm])}
    zero_hit = False
    for tag, (a, ax, b, bx) in sets.items():
        for mode in (1, 2, 3):
            for thr in (0.85, 0.95):
                r = orc.match(a, ax, b, bx, thr, mode)
                key = f"{tag}_m{mode}_t{int(thr * 100)}"
                for k, v in r.items():
                    assert np.array_equal(v, g[f"{key}_{k}"]), (key, k)
                zero_hit |= bool((g[f"{key}_gIdx"] == 0).any())
    assert zero_hit, "fixture must exercise the index-0 quirk"


def test_edge_cases(orc):
    # too small for a single octave: no pyramid, no keypoints (cSIFT3D.cc:254-255)
    ex = orc.extractor(np.ones((6, 6, 6), np.float32)).run(5)
    assert ex.num_octaves == 0 and len(ex.keypoints()[0]) == 0
    # all-zero volume: 0/0 -> NaN everywhere (cUtil.cc:536-564 has no guard), no keypoints
    ex = orc.extractor(np.zeros((16, 16, 16), np.float32)).run(5)
    assert np.isnan(ex.input()).all() and len(ex.extrema()) == 0 and len(ex.keypoints()[0]) == 0
    # constant volume: flat DoG, nothing passes the strict comparisons
    ex = orc.extractor(np.full((16, 20, 24), 3.0, np.float32)).run(5)
    assert len(ex.extrema()) == 0
    # empty matcher inputs
    r = orc.match(np.zeros((0, 768), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 768), np.float32), np.zeros((0, 3), np.float32))
    assert len(r["pairs"]) == 0
