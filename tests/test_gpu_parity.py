"""-m gpu: the HIP path, called through the C-ABI (3dsift_amd/libsift3d_hip.so), against
 (1) the committed golden vectors produced by the real reference and
 (2) the CPU oracle on seeded inputs.
Bars: Gaussian / DoG pyramid, abs-max thresholds, extrema lists, keypoint coordinates and the
matcher outputs are BIT-EXACT; orientation frames and descriptors are fp32 sums evaluated in a
different order on the GPU -> tolerance 1e-4 RMS on descriptors (BASELINE.json), measured ~1e-7."""
import importlib

import numpy as np
import pytest

from conftest import golden
from hipcheck import bits, compare_keypoints, compare_pyramids, extrema_table

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    m = importlib.import_module("3dsift_amd.capi")
    assert m.device_count() >= 1, "GPU tests need a visible MI355X (no CPU fallback exists)"
    return m


def test_library_is_the_hip_build(capi):
    import os

    assert os.path.exists(capi.LIB_PATH)
    with open("/proc/self/maps") as f:
        assert "libsift3d_hip.so" in f.read()


def test_g2_gaussian_smooth_golden(capi, orc):
    g = golden("g2_smooth.npz")
    for name, src in (("v1_s0", "v1"), ("v1_s5", "v1"), ("v2_s2", "v2")):
        out = capi.gaussian_smooth(g[src], g[name + "_sigma"])
        assert np.array_equal(bits(out), bits(g[name])), name
    rng = np.random.Generator(np.random.PCG64(3))
    for shape, sigma in (((33, 17, 70), 1.9466), ((5, 40, 9), 0.9733), ((64, 64, 64), 2.452547), ((3, 3, 3), 0.5387)):
        v = rng.normal(size=shape).astype(np.float32)
        assert np.array_equal(bits(capi.gaussian_smooth(v, sigma)), bits(orc.gaussian_smooth(v, sigma))), shape


def test_g3_pyramid_golden_bitexact(capi):
    g = golden("g3_pyramid.npz")
    ex = capi.CreateCSIFT3D(g["vol"]).run_stages(3)
    assert ex.num_octaves == int(g["noct"])
    assert np.array_equal(bits(ex.input()), bits(g["input"]))
    for o in range(ex.num_octaves):
        for i in range(6):
            assert np.array_equal(bits(ex.gss(o, i)), bits(g[f"gss_{o}_{i}"])), (o, i)
        for i in range(5):
            assert np.array_equal(bits(ex.dog(o, i)), bits(g[f"dog_{o}_{i}"])), (o, i)
    assert np.array_equal(extrema_table(ex.extrema()), g["extrema"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g4_g6_golden_keypoints(capi, synth, tag):
    g4, g6 = golden("g4_hashes.npz"), golden("g6_keypoints.npz")
    shape = tuple(int(v) for v in g4[f"{tag}_shape"])
    vol = synth.blobs(shape, seed=int(g4[f"{tag}_seed"]), noise=float(g4[f"{tag}_noise"]))
    ex = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
    assert np.array_equal(extrema_table(ex.extrema()), g4[f"{tag}_extrema"])
    kp, desc = ex.GetKeypoints()
    rms = compare_keypoints(kp, desc, g6[f"{tag}_kp"], g6[f"{tag}_desc"])
    assert rms < 1e-5


@pytest.mark.parametrize("shape,seed,noise", [
    ((64, 64, 64), 1234, 0.0),      # BASELINE configs[0]
    ((40, 56, 72), 5, 0.01),        # ragged, 3 octaves
    ((17, 33, 20), 23, 0.05),       # one octave, everything is boundary
    ((128, 96, 80), 9, 0.0),        # 4 octaves, non-cubic
    ((48, 50, 45), 3, 0.01),        # odd width: scalar (non 16-byte) load/store path of the fused kernel
    ((40, 37, 54), 4, 0.0),         # nx % 4 == 2, odd ny
    ((96, 64, 32), 6, 0.01),        # nx = 32: ONE tile column (left and right edge in the same tile), two tile rows; edge fast path
    ((40, 32, 64), 8, 0.0),         # ny = 32: one tile row (top and bottom edge in the same tile), two tile columns
    ((33, 96, 96), 10, 0.01),       # 3 x 3 tiles: every edge class and an interior tile, odd depth
    ((48, 40, 128), 31, 0.02),      # nx = 128: rows of whole ballot words -> k_mark's lean row loop, two words per row (r03)
    ((20, 70, 192), 32, 0.3),       # nx = 192: three words per row (the loop behind the batches of eight), dense: candidates on every border
    ((45, 51, 64), 33, 0.02),       # nx % 4 == 0 with odd ny and nz: the decimation fused into the seed level's kernel drops the odd last row / plane
    ((36, 40, 300), 34, 0.1),       # nx = 300: a partial last ballot word in the lean row loop (clamped lane offsets)
    ((24, 512, 512), 35, 0.01),     # r04: a THIN volume (planes of 256 tiles, 24 of them): the march kernel on short columns; octave 1 is 12 planes: its widest level (hw 6 > (12 - 2) / 2) takes the separable passes
    ((40, 300, 300), 36, 0.02),     # r04: thin and not tile aligned (shifted last tile column / row), 3 octaves: 300 x 300 x 40, 150 x 150 x 20, 75 x 75 x 10
    ((16, 96, 128), 37, 0.01),      # r04: 16 planes: hw 8 never fits, hw 6 and below do (16 >= 2 hw + 2 for hw <= 7)
    ((32, 32, 32), 38, 0.05),       # r04: octave 1 is already the one-workgroup launch, on the chain stream: the early detection must wait for it
    ((20, 40, 35), 39, 0.05),       # late r04: nx = 35 -- a shifted tile would start at x0 = 3, inside its own 4-column left halo (pieces that straddle column 0 are not loaded): hw 2 / 3 take the separable passes, hw >= 4 does not fit 35 either
    ((24, 44, 38), 40, 0.05),       # nx = 38: x0 = 6 -- fine for hw <= 4 (halo 4), inside the 8-column halo of hw 5 / 6
])
def test_full_pipeline_vs_oracle(capi, orc, synth, shape, seed, noise):
    vol = synth.blobs(shape, seed=seed, noise=noise)
    g = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
    o = orc.extractor(vol).run(5)
    assert np.array_equal(bits(g.input()), bits(o.input()))
    compare_pyramids(g, o)
    assert np.array_equal(extrema_table(g.extrema()), extrema_table(o.extrema()))
    kp, desc = g.GetKeypoints()
    okp, odesc = o.keypoints()
    compare_keypoints(kp, desc, okp, odesc)
    t = g.m_timer
    assert t["d_TotalTime"] > 0 and t["d_BuildGSS"] > 0


def test_orientation_codes_vs_oracle(capi, orc, synth):
    vol = synth.blobs((56, 48, 40), seed=7, noise=0.01)
    g = capi.CreateCSIFT3D(vol).run_stages(4)
    o = orc.extractor(vol).run(3)
    e = o.extrema()
    codes = g.orientation_codes()
    assert len(codes) == len(e)
    want = []
    for k in e:
        lvl = o.gss(int(k["octave"]), int(k["level"]))
        unit = o.level_info(0, int(k["octave"]) * 6 + int(k["level"]))[1][0]
        want.append(orc.orient_one(k, lvl, unit, np.float32(1.5) * k["scale"])[0])
    assert np.array_equal(codes, np.array(want, np.int32))
    assert set(want) >= {1, -2} or set(want) >= {1, -3}


def test_256_cubed_config(capi, orc, synth):
    """BASELINE configs[1] size: full KpSiftAlgorithm on 256^3 (the NIfTI container is host-side IO)."""
    vol = synth.blobs((256, 256, 256), seed=1234)
    g = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
    o = orc.extractor(vol).run(5)
    compare_pyramids(g, o)
    assert np.array_equal(extrema_table(g.extrema()), extrema_table(o.extrema()))
    kp, desc = g.GetKeypoints()
    okp, odesc = o.keypoints()
    assert len(kp) > 500
    compare_keypoints(kp, desc, okp, odesc)


def test_edge_cases(capi):
    ex = capi.CreateCSIFT3D(np.ones((6, 6, 6), np.float32)).KpSiftAlgorithm()
    assert ex.num_octaves == 0 and len(ex.GetKeypoints()[0]) == 0
    ex = capi.CreateCSIFT3D(np.zeros((16, 16, 16), np.float32)).KpSiftAlgorithm()
    assert np.isnan(ex.input()).all() and len(ex.extrema()) == 0 and len(ex.GetKeypoints()[0]) == 0
    ex = capi.CreateCSIFT3D(np.full((16, 20, 24), 3.0, np.float32)).KpSiftAlgorithm()
    assert len(ex.extrema()) == 0
    # GetKeypoints before KpSiftAlgorithm returns empty (reference: `filter` is empty)
    ex = capi.CreateCSIFT3D(np.random.default_rng(0).random((16, 16, 16), dtype=np.float32))
    assert len(ex.GetKeypoints()[0]) == 0
    # rerunning an extractor gives the same result (arena reuse)
    vol = np.random.default_rng(1).random((32, 32, 32), dtype=np.float32)
    ex = capi.CreateCSIFT3D(vol)
    k1, d1 = ex.KpSiftAlgorithm().GetKeypoints()
    k2, d2 = ex.KpSiftAlgorithm().GetKeypoints()
    assert k1.tobytes() == k2.tobytes()


def test_g8_matcher_golden(capi):
    g = golden("g8_match.npz")
    perm = g["perm"]
    sets = {"p": (g["da"], g["xa"], g["db"], g["xb"]), "q": (g["da"], g["xa"], g["db"][perm], g["xb"][perm])}
    mt = capi.muBruteMatcher()
    for tag, (a, ax, b, bx) in sets.items():
        for mode, fn in ((1, mt.injectMatch), (2, mt.bijectMatch), (3, mt.enhancedMatch)):
            for thr in (0.85, 0.95):
                r = fn(a, ax, b, bx, thr)
                key = f"{tag}_m{mode}_t{int(thr * 100)}"
                for k, v in r.items():
                    assert np.array_equal(v, g[f"{key}_{k}"]), (key, k)


def test_matcher_vs_oracle_random(capi, orc):
    rng = np.random.Generator(np.random.PCG64(17))

    def descs(n):
        d = np.clip(rng.normal(0.02, 0.03, size=(n, 768)), 0, None).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        return d.astype(np.float32)

    a, b = descs(300), descs(333)
    b[:100] = a[100:200] + rng.normal(0, 0.004, size=(100, 768)).astype(np.float32)   # true correspondences
    b[7] = b[3]                                                                        # exact duplicate rows -> ties
    ax = rng.uniform(0, 100, (300, 3)).astype(np.float32); bx = rng.uniform(0, 100, (333, 3)).astype(np.float32)
    mt = capi.muBruteMatcher()
    for mode in (1, 2, 3):
        got = mt._match(a, ax, b, bx, 0.85, mode)
        want = orc.match(a, ax, b, bx, 0.85, mode)
        for k in want:
            assert np.array_equal(got[k], want[k]), (mode, k)
    # empty / ragged
    e = np.zeros((0, 768), np.float32); ex = np.zeros((0, 3), np.float32)
    assert len(mt.enhancedMatch(e, ex, b, bx)["pairs"]) == 0
    got = mt.enhancedMatch(a, ax, e, ex); want = orc.match(a, ax, e, ex, 0.85, 3)
    for k in want:
        assert np.array_equal(got[k], want[k]), k


def _dev(arr):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr)).cuda()


def test_matcher_device_resident_inputs(capi, orc):
    """on_device=True (device pointers for descriptors and coordinates) against the golden g8 vectors and the oracle."""
    g = golden("g8_match.npz")
    mt = capi.muBruteMatcher()
    a, ax, b, bx = (_dev(g[k]) for k in ("da", "xa", "db", "xb"))
    for mode in (1, 2, 3):
        for thr in (0.85, 0.95):
            r = mt._match(a.data_ptr(), ax.data_ptr(), b.data_ptr(), bx.data_ptr(), thr, mode, on_device=True, n=a.shape[0], m=b.shape[0])
            for k, v in r.items():
                assert np.array_equal(v, g[f"p_m{mode}_t{int(thr * 100)}_{k}"]), (mode, thr, k)
    rng = np.random.Generator(np.random.PCG64(23))
    d = np.clip(rng.normal(0.02, 0.03, size=(500, 768)), 0, None).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a_h, b_h = d[:230].copy(), d[230:].copy()
    b_h[:60] = a_h[40:100] + rng.normal(0, 0.003, size=(60, 768)).astype(np.float32)
    ax_h = rng.uniform(0, 100, (230, 3)).astype(np.float32); bx_h = rng.uniform(0, 100, (270, 3)).astype(np.float32)
    a, ax, b, bx = _dev(a_h), _dev(ax_h), _dev(b_h), _dev(bx_h)
    for mode in (1, 2, 3):
        got = mt._match(a.data_ptr(), ax.data_ptr(), b.data_ptr(), bx.data_ptr(), 0.85, mode, on_device=True, n=230, m=270)
        want = orc.match(a_h, ax_h, b_h, bx_h, 0.85, mode)
        for k in want:
            assert np.array_equal(got[k], want[k]), (mode, k)


def test_matcher_near_ties(capi, orc):
    """Many target rows whose scores against a reference row differ only in the last bits: the fp32 MFMA selection cannot order
    them, the exact fp64 re-score must (Src/cMatcher.cc:52-77).  Needs the near-tie guard of kernels_match.hip."""
    rng = np.random.Generator(np.random.PCG64(31))
    base = np.clip(rng.normal(0.02, 0.03, size=(40, 768)), 0, None).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    a = base.astype(np.float32)
    rows = []
    for i in range(40):
        for _ in range(7):   # seven near-copies of every reference row: perturbations of a few ulp in a handful of components
            r = a[i].copy()
            idx = rng.integers(0, 768, 6)
            r[idx] = np.nextafter(r[idx], np.float32(1.0) if rng.random() < 0.5 else np.float32(-1.0)).astype(np.float32)
            rows.append(r)
    filler = np.clip(rng.normal(0.02, 0.03, size=(200, 768)), 0, None).astype(np.float32)
    filler /= np.linalg.norm(filler, axis=1, keepdims=True)
    b = np.concatenate([np.array(rows, np.float32), filler.astype(np.float32)])
    b = b[rng.permutation(len(b))]
    ax = rng.uniform(0, 100, (len(a), 3)).astype(np.float32); bx = rng.uniform(0, 100, (len(b), 3)).astype(np.float32)
    mt = capi.muBruteMatcher()
    for mode in (1, 2, 3):
        got = mt._match(a, ax, b, bx, 0.85, mode)
        want = orc.match(a, ax, b, bx, 0.85, mode)
        for k in want:
            assert np.array_equal(got[k], want[k]), (mode, k, int((got[k] != want[k]).sum()))


@pytest.mark.parametrize("levels", [1, 2, 4])
def test_nondefault_num_kp_levels(capi, orc, synth, levels):
    """CreateCSIFT3D(..., num_kp_levels) != 3 (Include/cSIFT3D.h:184): other sigma schedule / half widths (levels without a fused
    instantiation take the separable kernels), other count of DoG levels -- the first and the last one are formed on request."""
    vol = synth.blobs((72, 64, 80), seed=11, noise=0.01)
    g = capi.CreateCSIFT3D(vol, num_kp_levels=levels).KpSiftAlgorithm()
    o = orc.extractor(vol, num_kp_levels=levels).run(5)
    assert g.num_octaves == o.num_octaves
    for oc in range(g.num_octaves):
        for i in range(levels + 3):
            assert np.array_equal(bits(g.gss(oc, i)), bits(o.gss(oc, i))), ("gss", oc, i)
        for i in range(levels + 2):
            assert np.array_equal(bits(g.dog(oc, i)), bits(o.dog(oc, i))), ("dog", oc, i)
    assert np.array_equal(extrema_table(g.extrema()), extrema_table(o.extrema()))
    kp, desc = g.GetKeypoints()
    okp, odesc = o.keypoints()
    compare_keypoints(kp, desc, okp, odesc)


@pytest.mark.parametrize("params", [
    dict(sigma_default=2.0),                          # wider kernels: the last level (half width 10) evaluated at the parked candidates by k_lazy_next<25>
    dict(sigma_default=1.7),                          # half width 7: fused since r06
    dict(sigma_default=2.3),                          # half width 9 (separable) and 11 (lazy)
    dict(sigma_default=1.3, sigma_n_default=0.9),     # narrower base blur
    dict(peak_thresh=0.05),                           # many more extrema
    dict(peak_thresh=0.3, max_eig_thres=0.95, corner_thresh=0.2),
    dict(num_kp_levels=2, sigma_default=2.4, peak_thresh=0.08),
])
def test_nondefault_parameters(capi, orc, synth, params):
    """Every CreateCSIFT3D parameter (Include/cSIFT3D.h:184-194) away from its default, against the oracle."""
    vol = synth.blobs((64, 72, 56), seed=21, noise=0.01)
    g = capi.CreateCSIFT3D(vol, **params).KpSiftAlgorithm()
    o = orc.extractor(vol, **params).run(5)
    levels = params.get("num_kp_levels", 3)
    assert g.num_octaves == o.num_octaves
    for oc in range(g.num_octaves):
        for i in range(levels + 3):
            assert np.array_equal(bits(g.gss(oc, i)), bits(o.gss(oc, i))), ("gss", oc, i)
        for i in range(levels + 2):
            assert np.array_equal(bits(g.dog(oc, i)), bits(o.dog(oc, i))), ("dog", oc, i)
    assert np.array_equal(extrema_table(g.extrema()), extrema_table(o.extrema()))
    kp, desc = g.GetKeypoints()
    okp, odesc = o.keypoints()
    compare_keypoints(kp, desc, okp, odesc)


def test_random_parameters_and_shapes_vs_oracle(capi, orc, synth):
    """Nine random draws of (shape, num_kp_levels, sigma_default, sigma_n_default, peak_thresh, tile form): other half widths (hw 2 .. 8 and
    the generic separable kernels), other level counts in the one-workgroup octaves, other window sizes in orientation and descriptor;
    half of the draws with the 64 x 32 tiles forced wherever they fit.  Everything against the oracle."""
    rng = np.random.default_rng(77)
    pool = [40, 48, 56, 64, 66, 70, 72, 80, 96]
    for case in range(9):
        shape = tuple(int(rng.choice(pool)) for _ in range(3))
        levels = int(rng.integers(1, 5))
        sd = float(np.round(rng.uniform(1.2, 2.6), 2))
        params = dict(num_kp_levels=levels, sigma_default=sd, sigma_n_default=float(np.round(rng.uniform(0.5, min(1.15, sd - 0.2)), 2)),
                      peak_thresh=float(np.round(rng.uniform(0.04, 0.25), 3)))
        vol = synth.blobs(shape, seed=300 + case, noise=0.02)
        with capi.hook("march_tiles", case & 1):
            try:
                g = capi.CreateCSIFT3D(vol, **params).KpSiftAlgorithm()
            except capi.Sift3dError as e:   # the one documented limit: Gaussian kernels of more than 129 taps are refused (none of these draws)
                assert "129 taps" in str(e), (params, e)
                raise
            o = orc.extractor(vol, **params).run(5)
            try:
                assert g.num_octaves == o.num_octaves
                for oc in range(g.num_octaves):
                    for i in range(levels + 3):
                        assert np.array_equal(bits(g.gss(oc, i)), bits(o.gss(oc, i))), ("gss", oc, i)
                    for i in range(levels + 2):
                        assert np.array_equal(bits(g.dog(oc, i)), bits(o.dog(oc, i))), ("dog", oc, i)
                assert np.array_equal(extrema_table(g.extrema()), extrema_table(o.extrema()))
                kp, desc = g.GetKeypoints()
                okp, odesc = o.keypoints()
                compare_keypoints(kp, desc, okp, odesc)
            except AssertionError as e:
                raise AssertionError((case, shape, params, e))
            g.close()


def _full_hash(capi, ex, with_dog=False, with_extrema=False):
    import hashlib
    h = hashlib.sha1()
    kp, d = ex.GetKeypoints()
    if with_extrema:
        h.update(np.ascontiguousarray(ex.extrema()).tobytes())
    h.update(kp.tobytes()); h.update(d.tobytes())
    if with_dog:
        for o in range(ex.num_octaves):
            for i in range(5):
                h.update(ex.dog(o, i).tobytes())
    return h.hexdigest(), len(kp)


def test_dog_elision_matches_eager_build(capi, synth):
    """The single-volume path does not write the first / last DoG level of an octave (the extrema test forms those values from
    the Gaussian levels) and does not build the last Gaussian level at all.  With the hooks dog_eager / glast_eager (every level
    written, as round 1 did) the keypoints and every DoG level must be the same bit for bit."""
    vol = synth.blobs((72, 96, 64), seed=3, noise=0.01)
    base = _full_hash(capi, capi.CreateCSIFT3D(vol).KpSiftAlgorithm(), with_dog=True)
    with capi.hook("dog_eager", 1):
        eager = _full_hash(capi, capi.CreateCSIFT3D(vol).KpSiftAlgorithm(), with_dog=True)
    with capi.hook("glast_eager", 1):
        glast = _full_hash(capi, capi.CreateCSIFT3D(vol).KpSiftAlgorithm(), with_dog=True)
    assert base == eager == glast and base[1] > 20


@pytest.mark.parametrize("params,eager", [
    (dict(sigma_default=1.7), 0),   # half width 7 at level 4: fits the 16^3 octave of a 64^3 volume, not its 8^3 octave
    (dict(), 1),                    # default schedule with the last Gaussian level built: half width 8 at the 8^3 octave
    (dict(sigma_default=1.74), 1),
])
def test_small_octave_launch_checks_every_octave(capi, orc, params, eager):
    """the one-workgroup launch of the small octaves (kernels_small.hip) needs hw <= n - 2 on every axis of EVERY octave
    it takes; the octaves behind the first are smaller.  Power-of-two cubes end in an 8^3 octave: every level against the oracle
    (Src/cSIFT3D.cc:745-765 is the boundary term that differed)."""
    rng = np.random.default_rng(5)
    vol = rng.random((64, 64, 64), dtype=np.float32)
    with capi.hook("glast_eager", eager):
        g = capi.CreateCSIFT3D(vol, **params).KpSiftAlgorithm()
        o = orc.extractor(vol, **params).run(5)
        assert g.num_octaves == o.num_octaves == 4
        for oc in range(g.num_octaves):
            for i in range(6):
                assert np.array_equal(bits(g.gss(oc, i)), bits(o.gss(oc, i))), ("gss", oc, i)
            for i in range(5):
                assert np.array_equal(bits(g.dog(oc, i)), bits(o.dog(oc, i))), ("dog", oc, i)
        assert np.array_equal(extrema_table(g.extrema()), extrema_table(o.extrema()))


def test_lazy_last_level_wave_form_matches_the_workgroup_form(capi, orc, synth):
    """The last Gaussian level is evaluated only at the parked candidates of the last keypoint level (Src/cSIFT3D.cc:889-896 reads its
    centre value through DoG[nd-1]): interior candidates one WAVE each (k_lazy_wave, r05: LDS-DMA staged 17^3 block), candidates next to a
    border one workgroup each (k_lazy_next); the hook lazy_generic sends all of them down the workgroup form.  Same extrema either way,
    and equal to the oracle's; shapes with rows shorter than x + 11 (the staged row is shifted) and a noise volume dense in candidates."""
    rng = np.random.default_rng(9)
    # (x = 72, 76, 80: octaves with rows of 18 .. 20 voxels, shorter than or equal to the 20 floats staged per row)
    for vol in (synth.blobs((96, 80, 72), seed=5, noise=0.02), rng.random((64, 56, 120), dtype=np.float32), synth.blobs((40, 128, 44), seed=6, noise=0.01),
                rng.random((80, 72, 76), dtype=np.float32), rng.random((72, 80, 80), dtype=np.float32)):
        g = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
        with capi.hook("lazy_generic", 1):
            h = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
        o = orc.extractor(vol).run(3)
        a, b, c = extrema_table(g.extrema()), extrema_table(h.extrema()), extrema_table(o.extrema())
        assert len(a) > 20 and np.array_equal(a, b) and np.array_equal(a, c)
        lvl3 = int((a[:, 1] == 3).sum())
        assert lvl3 > 0, "no extremum of the last keypoint level: the lazy level was not exercised"


def test_separable_kernels_match_fused(capi, synth):
    """hook separable: every level by the generic three-pass kernels (the path of half widths without a fused instantiation)"""
    vol = synth.blobs((64, 96, 72), seed=13, noise=0.01)
    base = _full_hash(capi, capi.CreateCSIFT3D(vol).KpSiftAlgorithm(), with_dog=True, with_extrema=True)
    with capi.hook("separable", 1):
        sep = _full_hash(capi, capi.CreateCSIFT3D(vol).KpSiftAlgorithm(), with_dog=True, with_extrema=True)
    assert base == sep and base[1] > 20


@pytest.mark.parametrize("shape,seed,noise,eager", [
    ((40, 32, 64), 41, 0.02, 0),     # ONE 64 x 32 tile: left / right / top / bottom edge in the same tile
    ((33, 96, 96), 42, 0.01, 0),     # nx = 96: the second tile column is shifted back by 32 (x0 = 32); three tile rows
    ((48, 40, 128), 43, 0.02, 1),    # two aligned tile columns, shifted last tile row; every DoG level written (ring forms of hw 2 / 3)
    ((36, 70, 300), 44, 0.1, 0),     # five tile columns, the last one shifted by 20; ny = 70: shifted tile row
    ((24, 128, 192), 45, 0.01, 0),   # thin: 24 planes (hw 8 never fits, chunks of a few planes), three columns, four rows
    ((70, 64, 72), 46, 0.05, 0),     # nx = 72 = 64 + 8: the narrowest shifted column (hw 6 <= 8), interior code on a window that is mostly overlap
])
def test_wide_tiles_pyramid_vs_oracle(capi, orc, synth, shape, seed, noise, eager):
    """r04: the big levels take 64 x 32 tiles (eight waves per workgroup, DoG centres partly in registers).  The hook march_tiles = 1
    gives every level whose geometry allows them the wide tiles, so that volumes the oracle finishes in seconds run that kernel:
    every level bit-identical, and the same keypoints / descriptors as the 32 x 32 form (march_tiles = 2)."""
    vol = synth.blobs(shape, seed=seed, noise=noise)
    o = orc.extractor(vol).run(5)
    with capi.hook("march_tiles", 1), capi.hook("dog_eager", eager), capi.hook("glast_eager", eager):
        g = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
        compare_pyramids(g, o)
        assert np.array_equal(extrema_table(g.extrema()), extrema_table(o.extrema()))
        wide = _full_hash(capi, g, with_dog=True, with_extrema=True)
    with capi.hook("march_tiles", 2), capi.hook("dog_eager", eager), capi.hook("glast_eager", eager):
        narrow = _full_hash(capi, capi.CreateCSIFT3D(vol).KpSiftAlgorithm(), with_dog=True, with_extrema=True)
    assert wide == narrow


def test_random_shapes_vs_oracle(capi, orc, synth):
    """Ten volumes whose dimensions are drawn around the limits of the kernels (one tile, one tile + halo, shifted tiles, the small-
    octave launch, columns barely longer than a kernel): the whole pipeline against the oracle.  (late r04: the many-shapes test of the
    wide tiles found two bugs that no hand-picked shape had met; this is the same net under the rest of the pipeline.)"""
    rng = np.random.default_rng(20261004)
    pool = list(range(17, 25)) + list(range(30, 42)) + [47, 48, 49] + list(range(62, 74)) + [80]
    for case in range(10):
        shape = tuple(int(rng.choice(pool)) for _ in range(3))
        vol = synth.blobs(shape, seed=100 + case, noise=float(rng.choice([0.0, 0.02, 0.1])))
        g = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
        o = orc.extractor(vol).run(5)
        try:
            compare_pyramids(g, o)
            assert np.array_equal(extrema_table(g.extrema()), extrema_table(o.extrema()))
            kp, desc = g.GetKeypoints()
            okp, odesc = o.keypoints()
            compare_keypoints(kp, desc, okp, odesc)
        except AssertionError as e:
            raise AssertionError((case, shape, e))
        g.close()


def _pyramid_hash(capi, vol):
    import hashlib
    ex = capi.CSIFT3D(vol)
    ex.run_stages(2)
    h = hashlib.sha1()
    for o in range(ex.num_octaves):
        for i in range(6):
            h.update(ex.gss(o, i).tobytes())
        for i in range(5):
            h.update(ex.dog(o, i).tobytes())
    ex.close()
    return h.hexdigest()


def test_wide_tiles_match_the_32x32_form_on_many_shapes(capi):
    """Shapes around every limit of the 64 x 32 tiles, random data (no flat regions that would hide an indexing slip), every Gaussian and
    DoG level of every octave hashed: the wide form (hook march_tiles = 1) equals the 32 x 32 form (= 2), which the oracle tests pin.
    Widths: 64 (one tile), 64 + hw ... (the narrowest shifted column for hw 2 .. 6: 66 .. 70), 71, 100, 127, 128, 129, 190, 256, 260;
    heights 32, 33 + hw, 50, 64, 96, 97; depths from 14 (hw 6 barely fits: 2 hw + 2) to 130 (several chunks)."""
    rng = np.random.default_rng(7)
    shapes = [(14, 32, 64), (20, 40, 66), (30, 45, 67), (26, 64, 68), (40, 50, 69), (33, 96, 70), (47, 97, 71), (64, 33, 100), (21, 64, 127),
              (130, 64, 128), (35, 70, 129), (50, 48, 190), (96, 96, 256), (18, 128, 260)]
    for shape in shapes:
        vol = rng.random(shape, dtype=np.float32)
        for eager in (0, 1):
            with capi.hook("dog_eager", eager), capi.hook("glast_eager", eager):
                with capi.hook("march_tiles", 1):
                    wide = _pyramid_hash(capi, vol)
                with capi.hook("march_tiles", 2):
                    narrow = _pyramid_hash(capi, vol)
            assert wide == narrow, (shape, eager)


def test_descriptor_chord_cache_matches_recomputed_chords(capi, synth):
    """k_describe keeps the z range of every column of a window in a byte cache in LDS; windows whose ranges do not fit a byte
    (far larger than any default window) recompute them instead.  The hook desc_nocache forces that path: same descriptors, bit
    for bit (the histogram sums are integers: the order in which the columns are visited does not matter)."""
    vol = synth.blobs((96, 80, 72), seed=5, noise=0.01)
    base = _full_hash(capi, capi.CreateCSIFT3D(vol).KpSiftAlgorithm())
    with capi.hook("desc_nocache", 1):
        nocache = _full_hash(capi, capi.CreateCSIFT3D(vol).KpSiftAlgorithm())
    assert base == nocache and base[1] > 20


def test_two_stream_detection_matches_serial(capi, synth):
    """Octaves >= 1 form their extremum masks on a second stream with their own scratch beside octave 0 (context.hip); the ordered
    compaction is serial either way.  The hook det_serial runs everything on one stream with one scratch: same extrema (order
    included), keypoints and descriptors, bit for bit.  one_stream (all octave chains on the handle's stream) likewise."""
    vol = synth.blobs((128, 96, 112), seed=9, noise=0.01)
    ex = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
    base = _full_hash(capi, ex, with_extrema=True)
    assert ex.num_octaves >= 3 and base[1] > 20
    with capi.hook("det_serial", 1):
        assert _full_hash(capi, capi.CreateCSIFT3D(vol).KpSiftAlgorithm(), with_extrema=True) == base
    with capi.hook("one_stream", 1):
        assert _full_hash(capi, capi.CreateCSIFT3D(vol).KpSiftAlgorithm(), with_extrema=True) == base


# ---- the product's rarely taken branches, forced ---------------------------------------------------------

def test_list_overflow_regrow_rerun(capi, orc, synth):
    """The extrema / keypoint lists overflow -> the host regrows them and reruns (context.hip run_impl).  Forced by a tiny initial
    capacity (hook list_cap), on a dense volume (blobs + 30 % noise): the result must equal the oracle's and the regrow must have
    happened (debug counter)."""
    vol = synth.blobs((56, 64, 48), seed=17, noise=0.3)
    with capi.hook("list_cap", 48):
        g = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
    assert g.debug_counters()["list_regrows"] >= 1
    o = orc.extractor(vol).run(5)
    assert len(o.extrema()) > 48
    assert np.array_equal(extrema_table(g.extrema()), extrema_table(o.extrema()))
    kp, desc = g.GetKeypoints()
    okp, odesc = o.keypoints()
    compare_keypoints(kp, desc, okp, odesc)
    # a second run of the same handle keeps the grown lists: no regrow, same result
    k2, d2 = g.KpSiftAlgorithm().GetKeypoints()
    assert g.debug_counters()["list_regrows"] == 0 and np.array_equal(k2, kp) and np.array_equal(d2, desc)


def test_dense_noise_volume_with_regrow(capi, orc):
    """Pure uniform noise is all texture (no blobs): 0.38 % of the voxels are DoG extrema -- the densest input found; the default list
    capacity max(4096, V / 256) sits just above that density, so the regrow is forced with a capacity of 1024 here too.  The parked
    candidates of the lazy last level overflow their list as well (same capacity)."""
    vol = np.random.Generator(np.random.PCG64(77)).random((72, 72, 72), dtype=np.float32)
    with capi.hook("list_cap", 1024):
        g = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
    o = orc.extractor(vol).run(5)
    assert len(o.extrema()) > 1024 and g.debug_counters()["list_regrows"] >= 1
    assert np.array_equal(extrema_table(g.extrema()), extrema_table(o.extrema()))
    kp, desc = g.GetKeypoints()
    okp, odesc = o.keypoints()
    assert len(okp) > 200
    compare_keypoints(kp, desc, okp, odesc)


def test_descriptor_second_pass_with_exact_unit(capi, orc, synth):
    """k_describe picks its fixed-point unit from an ESTIMATE of the window's gradient mass and redoes a keypoint with the exact bound
    when the estimate was too small (or far too large).  Hook desc_mass_shift = 6 divides the estimate by 64: most keypoints overflow
    the first unit and take the second pass.  Descriptors stay within the bars against the oracle and within rounding of the
    normal run (a different power-of-two unit rounds differently: not bitwise)."""
    vol = synth.blobs((80, 72, 88), seed=19, noise=0.02)
    g0 = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
    kp0, d0 = g0.GetKeypoints()
    with capi.hook("desc_mass_shift", 6):
        g = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
        kp, desc = g.GetKeypoints()
        redone = g.debug_counters()["desc_second_passes"]
    assert len(kp) > 30 and redone > len(kp) // 2, (redone, len(kp))
    assert np.array_equal(kp, kp0)
    assert np.abs(desc - d0).max() < 2e-5
    okp, odesc = orc.extractor(vol).run(5).keypoints()
    compare_keypoints(kp, desc, okp, odesc)


def test_descriptor_sparse_volume_coarse_estimate(capi, orc):
    """a sharp structure inside the orientation window with a flat descriptor window (the zero background of CT / MR
    volumes) makes the first mass estimate overshoot 50-100x -> a fixed-point unit that much too coarse.  Such keypoints are redone
    with the exact unit; per-keypoint descriptor error stays at the 1e-4 bar."""
    rng = np.random.Generator(np.random.PCG64(5))
    vol = np.zeros((96, 96, 96), np.float32)
    zz, yy, xx = np.mgrid[0:96, 0:96, 0:96].astype(np.float32)
    for _ in range(60):   # small, sharp, anisotropic blobs on an exactly zero background (91 % of the voxels are 0)
        c = rng.uniform(16, 80, 3); s = rng.uniform(1.0, 3.0, 3); a = rng.uniform(0.4, 1.0)
        vol += (a * np.exp(-0.5 * (((zz - c[0]) / s[0]) ** 2 + ((yy - c[1]) / s[1]) ** 2 + ((xx - c[2]) / s[2]) ** 2))).astype(np.float32)
    vol[vol < 1e-3] = 0.0
    g = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
    kp, desc = g.GetKeypoints()
    okp, odesc = orc.extractor(vol).run(5).keypoints()
    assert len(okp) >= 20
    compare_keypoints(kp, desc, okp, odesc)


def test_matcher_register_staged_form(capi, orc):
    """Matrices of 4 GB and more cannot be addressed by the LDS-DMA staging (32-bit offsets) and take the register-staged form of
    k_scores_top4; the hook match_nodma forces it on any size: every output equals the DMA form's and the oracle's."""
    rng = np.random.Generator(np.random.PCG64(41))
    d = np.clip(rng.normal(0.02, 0.03, size=(700, 768)), 0, None).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a, b = d[:310].copy(), d[310:].copy()
    b[:90] = a[100:190] + rng.normal(0, 0.003, size=(90, 768)).astype(np.float32)
    ax = rng.uniform(0, 100, (len(a), 3)).astype(np.float32); bx = rng.uniform(0, 100, (len(b), 3)).astype(np.float32)
    mt = capi.muBruteMatcher()
    for mode in (1, 2, 3):
        base = mt._match(a, ax, b, bx, 0.85, mode)
        with capi.hook("match_nodma", 1):
            nodma = mt._match(a, ax, b, bx, 0.85, mode)
        want = orc.match(a, ax, b, bx, 0.85, mode)
        for k in want:
            assert np.array_equal(nodma[k], base[k]) and np.array_equal(nodma[k], want[k]), (mode, k)


def test_matcher_exact_row_guard_fires(capi, orc):
    """the near-tie guard's exact re-score (k_exact_rows) must actually run on the near-tie input (debug counter)"""
    rng = np.random.Generator(np.random.PCG64(31))
    a = np.clip(rng.normal(0.02, 0.03, size=(16, 768)), 0, None).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = np.repeat(a, 9, axis=0)   # nine exact copies of every row: more ties than the candidate list holds
    ax = rng.uniform(0, 100, (len(a), 3)).astype(np.float32); bx = rng.uniform(0, 100, (len(b), 3)).astype(np.float32)
    mt = capi.muBruteMatcher()
    got = mt.injectMatch(a, ax, b, bx)
    assert mt.exact_rows >= len(a)
    want = orc.match(a, ax, b, bx, 0.85, 1)
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    assert mt.wallTime >= mt.totalTime > 0


def test_g7_face_lookup_on_device(capi):
    """Check_intersect_faces + cart2bary (Src/cSIFT3D.cc:1542-1637) as k_describe evaluates them, against golden g7 (1000 random
    directions, some below the |g|^2 rejection, + the mesh's vertices, edge midpoints and face centres, where the eps-tolerant
    first-hit rule matters).  Route 1 = the literal ordered scan: face and barycentrics bit for bit.  Route 0 = what the kernel
    takes for ~99 % of the voxels, the lookup by the symmetry of the mesh (falls back to the scan within 6e-6 of an edge): the same
    face for every direction, the barycentrics to rounding (they are the same rational functions of g evaluated by other formulas)."""
    g = golden("g7_mesh.npz")
    f1, b1 = capi.face_lookup(g["dirs"], route=1)
    assert np.array_equal(f1, g["faces"])
    hit = g["faces"] >= 0
    assert np.array_equal(bits(b1[hit]), bits(g["bary"][hit]))
    f0, b0 = capi.face_lookup(g["dirs"], route=0)
    assert np.array_equal(f0, g["faces"])
    assert np.abs(b0[hit] - g["bary"][hit]).max() <= 2e-6
    # many more directions, dense around the edges of the mesh: route 0 must agree with the scan on the FACE everywhere
    rng = np.random.Generator(np.random.PCG64(8))
    d = rng.normal(size=(100000, 3)).astype(np.float32)
    v = g["verts"].reshape(-1, 3, 3)
    e = rng.integers(0, 20, 50000)
    t = rng.random((50000, 1)).astype(np.float32)
    near = (v[e, 0] * t + v[e, 1] * (1 - t) + rng.normal(scale=1e-5, size=(50000, 3))).astype(np.float32)   # within ~3e-5 of an edge
    d = np.concatenate([d, near, near * np.float32(7.5), -near])
    fa, ba = capi.face_lookup(d, route=0)
    fb, bb = capi.face_lookup(d, route=1)
    assert np.array_equal(fa, fb)
    assert np.abs(ba - bb).max() <= 2e-6 and (fa >= 0).all()


@pytest.mark.gpu
def test_free_functions_downsample_and_sub(capi, orc):
    """DownSample_3D / Sub of the reference's header (Include/cSIFT3D.h:210, 218; Src/cSIFT3D.cc:506-533, 849-882) through the
    C-ABI: every second voxel, and (cur - prev) * (-1) in fp32 -- bit for bit; against the oracle's pyramid where it has the same
    operation (octave 1 level 0 is the decimated level 3 of octave 0; DoG[i] = Sub(G[i], G[i + 1]))."""
    rng = np.random.default_rng(5)
    for shape in ((20, 33, 47), (64, 64, 64), (9, 10, 11)):
        v = rng.random(shape, dtype=np.float32)
        half = tuple(s // 2 for s in shape)
        assert np.array_equal(capi.downsample(v), v[::2, ::2, ::2][:half[0], :half[1], :half[2]])
        up = tuple((s + 1) // 2 for s in shape)   # the largest dst that fits
        assert np.array_equal(capi.downsample(v, up), v[::2, ::2, ::2])
        w = rng.random(shape, dtype=np.float32)
        assert np.array_equal(capi.dog_sub(v, w), (w - v) * np.float32(-1.0))
    with pytest.raises(Exception):
        capi.downsample(np.zeros((8, 8, 8), np.float32), (5, 4, 4))   # 2 * (5 - 1) >= 8: does not fit
    vol = np.random.default_rng(6).random((40, 48, 56), dtype=np.float32)
    o = orc.extractor(vol); o.run(2)
    g3, g10 = o.gss(0, 3), o.gss(1, 0)
    assert np.array_equal(capi.downsample(g3, g10.shape), g10)
    assert np.array_equal(capi.dog_sub(o.gss(0, 1), o.gss(0, 2)), o.dog(0, 1))


@pytest.mark.gpu
def test_matcher_awkward_sizes(capi, orc):
    """The score kernel deals HALF tiles (128 rows x 64 columns) to a fixed number of workgroups and the merge kernel recomputes which
    slots were written (r03): sizes around the tile edges -- one row, one column, 63 / 65 / 127 / 129 columns, more row blocks than
    column tiles and the other way round, a set large enough that a workgroup's share crosses row blocks -- all outputs against the
    oracle's matcher (Src/cMatcher.cc:146-228)."""
    rng = np.random.Generator(np.random.PCG64(23))

    def descs(n):
        d = np.clip(rng.normal(0.02, 0.03, size=(n, 768)), 0, None).astype(np.float32)
        d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-12)
        return d.astype(np.float32)

    mt = capi.muBruteMatcher()
    for n, m in ((1, 1), (1, 200), (200, 1), (63, 65), (127, 129), (129, 127), (300, 64), (64, 300), (1500, 40), (40, 1500), (2100, 2300)):
        a, b = descs(n), descs(m)
        k = min(n, m) // 2
        if k:
            b[:k] = a[:k] + rng.normal(0, 0.004, size=(k, 768)).astype(np.float32)
        ax = rng.uniform(0, 100, (n, 3)).astype(np.float32); bx = rng.uniform(0, 100, (m, 3)).astype(np.float32)
        for mode in (1, 3):
            got = mt._match(a, ax, b, bx, 0.85, mode)
            want = orc.match(a, ax, b, bx, 0.85, mode)
            for key in want:
                assert np.array_equal(got[key], want[key]), (n, m, mode, key)


def test_async_run_two_volumes_in_flight(capi, synth):
    """sift3d_run_async / sift3d_wait (r04): one host thread enqueues KpSiftAlgorithm on TWO handles, then waits for both.  Extrema,
    keypoints and descriptors of both volumes equal the blocking runs bit for bit (the descriptor histograms are integer sums, the
    pyramids bit-exact: nothing may depend on what else shares the GPU); accessors complete a run in flight by themselves; a second
    run_async without a wait in between is refused."""
    va = synth.blobs((96, 112, 128), seed=41, noise=0.01)
    vb = synth.blobs((128, 96, 80), seed=42, noise=0.02)
    base = [_full_hash(capi, capi.CreateCSIFT3D(v).KpSiftAlgorithm(), with_extrema=True) for v in (va, vb)]
    assert base[0][1] > 20 and base[1][1] > 20
    ea, eb = capi.CreateCSIFT3D(va), capi.CreateCSIFT3D(vb)
    for rep in range(3):
        ea.KpSiftAlgorithmAsync(); eb.KpSiftAlgorithmAsync()
        with pytest.raises(capi.Sift3dError):
            ea.KpSiftAlgorithmAsync()          # a run is in flight
        eb.Wait(); ea.Wait(); ea.Wait()        # any order; a wait with nothing in flight is a no-op
        assert [_full_hash(capi, e, with_extrema=True) for e in (ea, eb)] == base, rep
    # r06, sift3d_run_async_after: the second volume's pipeline is gated behind the first one's orientation stage; same results; with nothing in
    # flight on the other handle (or the handle itself) it is a plain run_async
    for rep in range(2):
        ea.KpSiftAlgorithmAsync(); eb.KpSiftAlgorithmAsync(after=ea)
        ea.Wait(); eb.Wait()
        assert [_full_hash(capi, e, with_extrema=True) for e in (ea, eb)] == base, rep
    eb.KpSiftAlgorithmAsync(after=ea).Wait(); ea.KpSiftAlgorithmAsync(after=ea).Wait()
    assert [_full_hash(capi, e, with_extrema=True) for e in (ea, eb)] == base
    ea.KpSiftAlgorithmAsync()
    assert len(ea.GetKeypoints()[0]) == base[0][1]   # no Wait(): the accessor completes the run
    with capi.hook("list_cap", 64):   # the regrow + rerun of an overflowing list happens inside the wait
        ec = capi.CreateCSIFT3D(va)
        ec.KpSiftAlgorithmAsync().Wait()
        assert ec.debug_counters()["list_regrows"] >= 1 and _full_hash(capi, ec, with_extrema=True) == base[0]


def test_descriptor_exact_cell_path_everywhere(capi, orc, synth):
    """k_describe forms a voxel's cell coordinates with one fused multiply-add per axis and repeats the reference's arithmetic only for
    voxels within 1e-4 of a discontinuity (a face of the 4x4x4 cube, b = 0): that path recovers the voxel's integer offsets from the
    coordinates.  The hook desc_exact_cells sends EVERY voxel through it -- a wrong recovery would scramble whole descriptors, not one
    voxel in a thousand: the descriptors must stay inside the oracle's bars and within 2e-6 of the default run's (fixed-point units
    that round the other way), with identical keypoint records and gradient masses (same second passes)."""
    vol = synth.blobs((96, 104, 112), seed=21, noise=0.02)
    o = orc.extractor(vol).run(5)
    okp, odesc = o.keypoints()
    ex = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
    kp, ds = ex.GetKeypoints()
    passes = ex.debug_counters()["desc_second_passes"]
    assert len(kp) > 100
    with capi.hook("desc_exact_cells", 1):
        ee = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
        kpe, dse = ee.GetKeypoints()
        assert ee.debug_counters()["desc_second_passes"] == passes
    assert kp.tobytes() == kpe.tobytes()
    compare_keypoints(kpe, dse, okp, odesc)
    compare_keypoints(kp, ds, okp, odesc)
    assert float(np.abs(ds - dse).max()) <= 2e-6, float(np.abs(ds - dse).max())


def test_split_descriptor_windows_match_unsplit(capi, synth):
    """r04: with few keypoints a descriptor window is marched by 8 or 4 workgroups (by the keypoint count), each adding its integer
    histogram into the keypoint's accumulators in global memory; the part that arrives last normalises, and a keypoint whose first
    fixed-point unit fails is repeated by its finisher alone.  The hook desc_nosplit gives every keypoint one workgroup (the form of
    runs with many keypoints): same descriptors bit for bit, same number of second passes -- also when the hook desc_mass_shift sends
    EVERY keypoint through the second pass."""
    for shape, seed in (((64, 64, 64), 1234), ((96, 112, 128), 7), ((160, 160, 160), 8)):   # 40 / ~200 / ~450 keypoints: 8 and 4 parts
        vol = synth.blobs(shape, seed=seed, noise=0.01)
        for shift in (0, 6):
            with capi.hook("desc_mass_shift", shift):
                ex = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
                split = _full_hash(capi, ex); n_split = ex.debug_counters()["desc_second_passes"]
                with capi.hook("desc_nosplit", 1):
                    ex2 = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
                    plain = _full_hash(capi, ex2); n_plain = ex2.debug_counters()["desc_second_passes"]
                # twice on one handle: the accumulators were left clean
                again = _full_hash(capi, ex.KpSiftAlgorithm())
            assert split == plain and split[1] > 20, (shape, shift)
            assert n_split == n_plain and (shift == 0 or n_split > 0), (shape, shift, n_split, n_plain)
            assert again == split, (shape, shift)


def test_rows_wider_than_4096_voxels(capi, orc):
    """k_mark / k_emit walk a row's ballot words in segments of 64: rows of 4 200 and 9 000 voxels (found nowhere else in the suite; the
    soak run scripts/soak_shapes.py has more such shapes), white noise so that extrema sit in every segment"""
    for shape in ((16, 16, 4200), (8, 8, 9000)):
        vol = np.random.default_rng(shape[2]).random(shape).astype(np.float32)
        g = capi.CreateCSIFT3D(vol, peak_thresh=0.05).KpSiftAlgorithm()
        o = orc.extractor(vol, peak_thresh=0.05).run(5)
        assert np.array_equal(extrema_table(g.extrema()), extrema_table(o.extrema())), shape
        assert (g.extrema()["x"] > 4096).sum() > 5
        kp, desc = g.GetKeypoints()
        okp, odesc = o.keypoints()
        assert len(kp) > 100
        compare_keypoints(kp, desc, okp, odesc)
        g.close()


def test_large_window_cut_off_by_the_volume_border(capi, orc, synth):
    """The draw of scripts/soak_random.py that fell outside the descriptor bar in round 5: sigma_default 2.47 gives windows of radius 55 voxels,
    cut off by the border of a 64 x 128 x 80 volume; the first fixed-point unit assumed the whole sphere, was 16x coarser than the window's mass
    allowed and 340 000 contributions of 12 units each lost 2.9e-5 RMS to rounding.  With the estimate scaled by the share of the window inside
    the level: 3.6e-6."""
    vol = synth.blobs((80, 128, 64), seed=6010, noise=0.0)
    params = dict(num_kp_levels=3, sigma_default=2.47, sigma_n_default=0.51, peak_thresh=0.097, max_eig_thres=0.95, corner_thresh=0.3)
    g = capi.CreateCSIFT3D(vol, **params).KpSiftAlgorithm()
    o = orc.extractor(vol, **params).run(5)
    kp, desc = g.GetKeypoints()
    okp, odesc = o.keypoints()
    assert len(kp) > 100 and kp["scale"].max() > 3.9
    compare_keypoints(kp, desc, okp, odesc, rms_tol=8e-6)
    g.close()
