"""-m gpu: the per-axis / per-keypoint free functions of the reference's public header (Include/cSIFT3D.h:214, 224, 228:
GaussianSmooth_3D_Imp, Assign_Orientation_Imp, Extract_Descriptor_Imp) through the C-ABI (sift3d_conv_axis, sift3d_orient_keypoint,
sift3d_describe_keypoint) and through the C++ shell, against the oracle's restatement of the same functions (oracle/sift3d_oracle.c
orient_one / describe_one / gaussian_smooth) and the pipeline's own results; Trilinear_interpolation_over_desc(_debug), a host helper of the
shell, against a numpy restatement of Src/cSIFT3D.cc:1383-1540."""
import importlib
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from hipcheck import bits, descriptor_errors

pytestmark = pytest.mark.gpu

PKG = os.path.dirname(importlib.import_module("3dsift_amd.capi").__file__)


@pytest.fixture(scope="module")
def capi():
    m = importlib.import_module("3dsift_amd.capi")
    assert m.device_count() >= 1, "GPU tests need a visible MI355X (no CPU fallback exists)"
    return m


def _line_rule(line, w):
    """GaussianSmooth_3D_Imp along one line (Src/cSIFT3D.cc:624-788), fp32 like the reference: interior taps, mirrored boundary with the
    0.1-voxel inset on the far side and the lerp between the two neighbours"""
    n, hw = len(line), len(w) // 2
    out = np.zeros(n, np.float32)
    one = np.float32(1.0)
    for p in range(n):
        acc = np.float32(0.0)
        interior = hw <= p <= n - 2 - hw
        for d in range(-hw, hw + 1):
            c = np.float32(p) - np.float32(d)
            if not interior:
                if c < 0:
                    c = np.float32(-1.0) * c
                elif c >= np.float32(n - 1):
                    c = np.float32(2 * (n - 1)) - c - np.float32(0.1)
            lo = int(c)
            fr = np.float32(c - np.float32(lo))
            lo_c, hi_c = min(max(lo, 0), n - 1), min(max(lo + 1, 0), n - 1)
            acc = np.float32(acc + np.float32(w[d + hw] * np.float32(np.float32((one - fr) * line[lo_c]) + np.float32(fr * line[hi_c]))))
        out[p] = acc
    return out


@pytest.mark.parametrize("shape,width", [((7, 9, 12), 5), ((5, 4, 6), 7), ((10, 3, 9), 3), ((6, 8, 20), 13)])
def test_conv_axis_with_arbitrary_taps(capi, shape, width):
    """one pass along each axis with ASYMMETRIC taps (nothing in the pass may rely on the Gaussian's symmetry or normalisation), shapes
    with lines shorter than the kernel (every voxel on the boundary rule): bit-identical to the rule written out in numpy"""
    rng = np.random.default_rng(width * 100 + shape[0])
    vol = rng.standard_normal(shape).astype(np.float32)
    w = rng.uniform(-0.5, 1.0, width).astype(np.float32)
    for dim in range(3):
        got = capi.conv_axis(vol, dim, w)
        want = np.apply_along_axis(_line_rule, 2 - dim, vol, w)   # dim 0 = x = the last numpy axis
        assert np.array_equal(bits(got), bits(want)), (dim, int((bits(got) != bits(want)).sum()))
    with pytest.raises(capi.Sift3dError):
        capi.conv_axis(vol, 0, w[: width - 1])   # even width: the reference would read past its taps


def test_three_passes_equal_gaussian_smooth(capi, orc, synth):
    vol = synth.blobs((40, 36, 44), seed=3, noise=0.02)
    taps = orc.gaussian_taps(1.3)
    a = capi.conv_axis(capi.conv_axis(capi.conv_axis(vol, 0, taps), 1, taps), 2, taps)
    assert np.array_equal(bits(a), bits(orc.gaussian_smooth(vol, 1.3)))


@pytest.fixture(scope="module")
def run(capi, orc, synth):
    vol = synth.blobs((112, 96, 104), seed=11, noise=0.01)
    g = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
    o = orc.extractor(vol).run(5)
    return g, o


def _level_of(o, k):
    oc, lv = int(k["octave"]), int(k["level"])
    return o.gss(oc, lv), o.level_info(0, oc * 6 + lv)[1][0]


@pytest.mark.parametrize("factor", [1.5, 1.1, 2.2])
def test_orient_keypoint_vs_oracle(capi, orc, run, factor):
    """every extremum of a small volume, one call each, for the pipeline's sigma (1.5 scale) and two others (the table is rebuilt):
    the reference's code; for accepted keypoints structure tensor, mean gradient, eigenvalues and rotation bit-identical"""
    g, o = run
    ext = o.extrema()
    assert len(ext) > 100
    codes, n_ok = set(), 0
    for k in ext:
        lvl, unit = _level_of(o, k)
        sigma = np.float32(factor) * k["scale"]
        want, ko = orc.orient_one(k, lvl, unit, sigma)
        rec = np.array([k], dtype=capi.KP_DTYPE)
        got = capi.orient_keypoint(lvl, unit, rec, sigma)
        assert got == want, (k["x"], k["y"], k["z"], got, want)
        codes.add(want)
        if want == 1:
            n_ok += 1
            for f in ("str_tensor", "win", "eigvalue", "Rotation"):
                assert np.array_equal(bits(rec[0][f]), bits(ko[f])), f
    assert n_ok >= 5 and len(codes) >= 2


def test_describe_keypoint_vs_oracle_and_pipeline(capi, orc, run):
    """accepted keypoints one call each: within the descriptor bars of the oracle's describe_one, the rotation transposed like the
    reference leaves it, and BIT-IDENTICAL to the descriptor the pipeline produced for the same keypoint (same record, same integers,
    whatever the box the window was cut out of)"""
    g, o = run
    kp, desc = g.GetKeypoints()
    okp, odesc = o.keypoints()
    assert len(kp) == len(okp) >= 10
    got = np.zeros_like(desc)
    for i, k in enumerate(okp):
        lvl, unit = _level_of(o, k)
        rec = np.array([k], dtype=capi.KP_DTYPE)
        rec[0]["Rotation"] = k["Rotation"].reshape(3, 3).T.reshape(9)   # as the orientation stage left it
        got[i] = capi.describe_keypoint(lvl, unit, rec)
        assert np.array_equal(bits(rec[0]["Rotation"]), bits(k["Rotation"]))
    rms, worst_kp, worst_abs = descriptor_errors(got, odesc)
    assert rms <= 2e-5 and worst_kp <= 2e-5 and worst_abs <= 1e-4, (rms, worst_kp, worst_abs)
    assert np.array_equal(bits(got), bits(desc)), int((bits(got) != bits(desc)).any(axis=1).sum())


def test_single_keypoint_inputs_outside_the_pipeline_are_refused(capi, run):
    g, o = run
    k = o.extrema()[0]
    lvl, unit = _level_of(o, k)
    rec = np.array([k], dtype=capi.KP_DTYPE)
    rec[0]["x"] += np.float32(0.5)
    with pytest.raises(capi.Sift3dError, match="voxel"):
        capi.orient_keypoint(lvl, unit, rec, 1.5 * k["scale"])
    rec = np.array([k], dtype=capi.KP_DTYPE)
    with pytest.raises(capi.Sift3dError, match="power of two"):
        capi.describe_keypoint(lvl, 3.0, rec)


def _trilinear_numpy(mesh_idx, face, bary, vb, grad, desc):
    """Src/cSIFT3D.cc:1383-1448 for one voxel whose face / barycentrics are known"""
    fr = (vb - np.floor(vb)).astype(np.float32)
    cell = vb.astype(np.int32)   # truncation
    mag = np.float32(np.sqrt(np.float32(np.float32(grad[0] * grad[0]) + np.float32(grad[1] * grad[1])) + np.float32(grad[2] * grad[2])))
    for dx in range(2):
        for dy in range(2):
            for dz in range(2):
                c = cell + np.array([dx, dy, dz])
                if (c < 0).any() or (c >= 4).any():
                    continue
                wd = 1.0
                for a, up in enumerate((dx, dy, dz)):
                    wd *= float(fr[a]) if up else 1.0 - float(fr[a])
                w = np.float32(wd)
                b0 = (int(c[0]) + 4 * int(c[1]) + 16 * int(c[2])) * 12
                for v in range(3):
                    desc[b0 + mesh_idx[face][v]] = np.float32(desc[b0 + mesh_idx[face][v]] + np.float32(np.float32(mag * w) * bary[v]))


def test_cpp_shell_free_functions(capi, orc, synth):
    """The reference's own call shapes, compiled against the shell: GET_GSS levels -> Assign_Orientation_Imp + Extract_Descriptor_Imp per
    keypoint reproduce the keypoints KpSiftAlgorithm returned (rotation and descriptor bit for bit); GaussianSmooth_3D_Imp x 3 equals
    GaussianSmooth_3D; Trilinear_interpolation_over_desc(_debug) on random voxels equals the numpy restatement."""
    src = r"""
    #include "Include/cSIFT3D.h"
    #include <cstdio>
    #include <cstring>
    #include <cmath>
    using namespace CPUSIFT;
    int main(int, char** a) {
        CSIFT3D *A = CSIFT3DFactory::CreateCSIFT3D(std::string(a[1]));
        A->KpSiftAlgorithm();
        std::vector<Keypoint> k = A->GetKeypoints();
        std::vector<TexImage> *G = A->GET_GSS();
        Mesh mesh; Initialize_geometry(&mesh);
        int bad_rot = 0, bad_desc = 0, bad_code = 0;
        std::vector<float> d(DESC_NUMEL);
        for (auto& p : k) {
            Keypoint q; memset(&q, 0, sizeof(q));
            q.x = p.x; q.y = p.y; q.z = p.z; q.scale = p.scale; q.octave = p.octave; q.level = p.level; q.desc = d.data();
            TexImage *lvl = &(*G)[p.octave * 6 + p.level];
            int code = Assign_Orientation_Imp(q, lvl, 1.5f * q.scale, 0.9f, 0.4f);
            if (code != 1) { bad_code++; continue; }
            Extract_Descriptor_Imp(q, lvl, &mesh);
            if (memcmp(q.Rotation, p.Rotation, 36)) bad_rot++;
            if (memcmp(d.data(), p.desc, 4 * DESC_NUMEL)) bad_desc++;
        }
        // three single-axis passes with the taps of sigma 1.3 against the three-pass function
        TexImage &L = (*G)[1];
        TexImage t1, t2, t3, ref;
        float w[9]; { float acc = 0; for (int i = 0; i < 9; i++) { float x = (float)((double)(i - 4) / (1.3 + 2.220446049250313e-16)); w[i] = (float)exp(-0.5 * (double)x * (double)x); acc += w[i]; } for (int i = 0; i < 9; i++) w[i] /= acc; }
        GaussianSmooth_3D_Imp(&L, &t1, 0, 1.f, w, 9); GaussianSmooth_3D_Imp(&t1, &t2, 1, 1.f, w, 9); GaussianSmooth_3D_Imp(&t2, &t3, 2, 1.f, w, 9);
        GaussianSmooth_3D(&L, &ref, 1.3f);
        size_t nv = (size_t)L.GetDimX() * L.GetDimY() * L.GetDimZ();
        int bad_blur = memcmp(t3._Data, ref._Data, 4 * nv) ? 1 : 0;
        // the per-voxel scatter: inputs from the file a[3] (n, then n x (vbins[3], grad[3])), results to a[4]
        FILE* f = fopen(a[3], "rb"); int n = 0; if (fread(&n, 4, 1, f) != 1) return 2;
        std::vector<float> in((size_t)n * 6); if (fread(in.data(), 4, in.size(), f) != in.size()) return 2; fclose(f);
        std::vector<float> h1(DESC_NUMEL, 0.f), h2(DESC_NUMEL, 0.f), dv(3 * n), br(3 * n), acc(24 * n, 0.f);
        std::vector<int> face(n), off(24 * n, -1);
        Keypoint k1; memset(&k1, 0, sizeof(k1)); k1.desc = h1.data();
        Keypoint k2; memset(&k2, 0, sizeof(k2)); k2.desc = h2.data();
        for (int i = 0; i < n; i++) {
            Cvec vb(in[6 * i], in[6 * i + 1], in[6 * i + 2]), g(in[6 * i + 3], in[6 * i + 4], in[6 * i + 5]);
            Trilinear_interpolation_over_desc(&mesh, k1, vb, g, i);
            Trilinear_interpolation_over_desc_debug(&mesh, k2, vb, g, i, dv.data(), face.data(), br.data(), off.data(), acc.data(), 1);
        }
        f = fopen(a[4], "wb");
        fwrite(h1.data(), 4, DESC_NUMEL, f); fwrite(h2.data(), 4, DESC_NUMEL, f); fwrite(face.data(), 4, n, f); fwrite(br.data(), 4, 3 * n, f);
        fwrite(dv.data(), 4, 3 * n, f); fwrite(off.data(), 4, 24 * n, f); fwrite(acc.data(), 4, 24 * n, f);
        fclose(f);
        printf("keypoints %zu bad_code %d bad_rot %d bad_desc %d bad_blur %d\n", k.size(), bad_code, bad_rot, bad_desc, bad_blur);
        delete A;
        return 0;
    }"""
    vol = synth.blobs((64, 56, 72), seed=21, noise=0.01)
    rng = np.random.default_rng(5)
    n = 400
    vb = rng.uniform(-0.49, 3.49, (n, 3)).astype(np.float32)
    vb[:40] = np.round(vb[:40])          # voxels exactly on cell boundaries (truncation / floor agree there)
    vb[40:60, 0] = rng.uniform(-0.49, -0.01, 20).astype(np.float32)   # the trunc-vs-floor pairing below zero
    gr = rng.standard_normal((n, 3)).astype(np.float32)
    gr[-3:] = 0                          # vanishing gradients: no face
    with tempfile.TemporaryDirectory() as t:
        with open(os.path.join(t, "v.bin"), "wb") as f:
            f.write(struct.pack("<3i", 72, 56, 64) + vol.tobytes())
        with open(os.path.join(t, "in.bin"), "wb") as f:
            f.write(struct.pack("<i", n) + np.concatenate([vb, gr], 1).astype(np.float32).tobytes())
        open(os.path.join(t, "m.cpp"), "w").write(src)
        subprocess.check_call(["g++", "-std=c++14", "-I" + os.path.join(PKG, "host"), "-o", os.path.join(t, "m"), os.path.join(t, "m.cpp"),
                               "-L" + PKG, "-lsift3d", "-lsift3d_hip", "-Wl,-rpath," + PKG])
        out = subprocess.check_output([os.path.join(t, "m"), os.path.join(t, "v.bin"), "x", os.path.join(t, "in.bin"), os.path.join(t, "out.bin")],
                                      stderr=subprocess.STDOUT).decode()
        raw = np.fromfile(os.path.join(t, "out.bin"), np.uint8)
    last = out.strip().splitlines()[-1].split()
    assert int(last[1]) > 15 and last[2:] == ["bad_code", "0", "bad_rot", "0", "bad_desc", "0", "bad_blur", "0"], out
    p = 0

    def take(count, dt):
        nonlocal p
        a = raw[p:p + 4 * count].view(dt)
        p += 4 * count
        return a
    h1, h2 = take(768, np.float32), take(768, np.float32)
    face, bary, dv = take(n, np.int32), take(3 * n, np.float32).reshape(n, 3), take(3 * n, np.float32).reshape(n, 3)
    off, acc = take(24 * n, np.int32).reshape(n, 8, 3), take(24 * n, np.float32).reshape(n, 8, 3)
    assert np.array_equal(bits(h1), bits(h2))
    _, idx = orc.mesh()
    want = np.zeros(768, np.float32)
    hits = 0
    for i in range(n):
        fo, bo = orc.intersect(gr[i])
        assert fo == face[i]
        if fo < 0:
            continue
        assert np.array_equal(bits(bo), bits(bary[i]))
        assert np.array_equal(bits(dv[i]), bits((vb[i] - np.floor(vb[i])).astype(np.float32)))
        _trilinear_numpy(idx, fo, bo, vb[i], gr[i], want)
        hits += 1
    assert hits > 300 and (face[-3:] == -1).all()
    assert np.array_equal(bits(want), bits(h1)), int((bits(want) != bits(h1)).sum())
    # the trace of the _debug form adds up to the histogram
    tot = np.zeros(768, np.float64)
    np.add.at(tot, off[off >= 0], acc[off >= 0].astype(np.float64))
    assert np.allclose(tot, h1, rtol=1e-5, atol=1e-6)
