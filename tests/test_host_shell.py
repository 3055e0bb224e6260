"""The C++ drop-in shell (3dsift_amd/host, namespace CPUSIFT) and its loaders.

CPU: it builds, the reference's own Example.cpp compiles and links UNCHANGED against our headers
(only where /root/reference exists), readNiiFile / ReadMatrixFromDisk parse files written by numpy.
GPU: the demo program runs the reference's flow end to end (extract x2 -> enhancedMatch)."""
import ctypes as C
import gzip
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3dsift_amd")


@pytest.fixture(scope="module")
def shell():
    subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "-j8"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(PKG, "host")], stdout=subprocess.DEVNULL)
    C.CDLL(os.path.join(PKG, "libsift3d_hip.so"), mode=C.RTLD_GLOBAL)
    return C.CDLL(os.path.join(PKG, "libsift3d.so"))


def test_reference_example_compiles_unchanged(shell):
    src = "/root/reference/3DSIFT/Example.cpp"
    if not os.path.exists(src):
        pytest.skip("reference tree not present on this machine")
    with tempfile.TemporaryDirectory() as t:
        os.symlink(src, os.path.join(t, "Example.cpp"))  # so "Include/..." resolves to OUR headers, not the reference's
        subprocess.check_call(["g++", "-std=c++14", "-I" + os.path.join(PKG, "host"), "-o", os.path.join(t, "a.out"),
                               os.path.join(t, "Example.cpp"), "-L" + PKG, "-lsift3d", "-lsift3d_hip", "-Wl,-rpath," + PKG])


def nifti1(vol, dtype, code, big_endian=False):
    nz, ny, nx = vol.shape
    e = ">" if big_endian else "<"
    h = bytearray(352)
    struct.pack_into(e + "i", h, 0, 348)
    struct.pack_into(e + "8h", h, 40, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into(e + "h", h, 70, code)
    struct.pack_into(e + "h", h, 72, np.dtype(dtype).itemsize * 8)
    struct.pack_into(e + "f", h, 108, 352.0)
    struct.pack_into(e + "f", h, 112, 2.0)   # scl_slope: must be IGNORED (readNii.cpp:17-33)
    struct.pack_into(e + "f", h, 116, 5.0)   # scl_inter
    h[344:348] = b"n+1\0"
    return bytes(h) + vol.astype(np.dtype(dtype).newbyteorder(e)).tobytes()


@pytest.mark.parametrize("dtype,code", [("f4", 16), ("i2", 4), ("u1", 2), ("f8", 64), ("u2", 512)])
def test_read_nii(shell, dtype, code):
    rng = np.random.default_rng(0)
    vol = (rng.random((5, 6, 7)) * 100).astype(dtype)
    fn = shell._Z11readNiiFilePKcRiS1_S1_
    fn.restype = C.POINTER(C.c_float)
    fn.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    with tempfile.TemporaryDirectory() as t:
        for name, blob in (("a.nii", nifti1(vol, dtype, code)), ("b.nii.gz", gzip.compress(nifti1(vol, dtype, code))),
                           ("c.nii", nifti1(vol, dtype, code, big_endian=True))):
            p = os.path.join(t, name)
            open(p, "wb").write(blob)
            nx, ny, nz = C.c_int(), C.c_int(), C.c_int()
            ptr = fn(p.encode(), C.byref(nx), C.byref(ny), C.byref(nz))
            assert ptr and (nx.value, ny.value, nz.value) == (7, 6, 5), name
            got = np.ctypeslib.as_array(ptr, shape=(5, 6, 7)).copy()
            assert np.array_equal(got, vol.astype(np.float32)), name
        assert not fn(os.path.join(t, "missing.nii").encode(), C.byref(nx), C.byref(ny), C.byref(nz))


def test_read_nii_matches_reference_reader(shell):
    """g9: files + the arrays the REFERENCE's readNiiFile (layNii, compiled by `make -C oracle ref`) returned for them."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "g9_nifti.npz"))
    fn = shell._Z11readNiiFilePKcRiS1_S1_
    fn.restype = C.POINTER(C.c_float)
    fn.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    with tempfile.TemporaryDirectory() as t:
        seen = set()
        for name in g["names"]:
            name = str(name)
            if name + "_files" in g.files:   # r05: NIfTI-2 single files and .hdr + .img pairs (NIfTI-1 / NIfTI-2 / ANALYZE), named by either file
                for fn_ in g[name + "_files"]:
                    open(os.path.join(t, str(fn_)), "wb").write(g[name + "_blob_" + str(fn_)].tobytes())
                p = os.path.join(t, str(g[name + "_arg"]))
                seen.add("pair" if len(g[name + "_files"]) == 2 else "nii2")
            else:
                p = os.path.join(t, name + (".nii.gz" if bool(g[name + "_gz"]) else ".nii"))
                open(p, "wb").write(g[name + "_file"].tobytes())
            nx, ny, nz = C.c_int(), C.c_int(), C.c_int()
            ptr = fn(p.encode(), C.byref(nx), C.byref(ny), C.byref(nz))
            want = g[name + "_data"]
            assert ptr and (nz.value, ny.value, nx.value) == want.shape, name
            got = np.ctypeslib.as_array(ptr, shape=want.shape).copy()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
        assert seen == {"pair", "nii2"}


def test_read_nii_rejects_malformed_headers(shell):
    fn = shell._Z11readNiiFilePKcRiS1_S1_
    fn.restype = C.POINTER(C.c_float)
    fn.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    vol = np.arange(5 * 6 * 7, dtype=np.int16).reshape(5, 6, 7)
    good = bytearray(nifti1(vol, "i2", 4))
    bad = {}
    b = bytearray(good); struct.pack_into("<h", b, 42, -7); bad["negative dim"] = b
    b = bytearray(good); struct.pack_into("<h", b, 72, 32); bad["bitpix != datatype size"] = b
    b = bytearray(good); struct.pack_into("<f", b, 108, float("nan")); bad["vox_offset NaN"] = b
    b = bytearray(good); struct.pack_into("<f", b, 108, 3.0e9); bad["vox_offset absurd"] = b
    b = bytearray(good); struct.pack_into("<h", b, 70, 1536); bad["unsupported datatype"] = b
    bad["truncated payload"] = good[:-10]
    b = bytearray(good); b[344:348] = b"ni1\0"; bad["header of a pair in a file that is not *.hdr"] = b
    b = bytearray(good); b[344:348] = b"xyz\0"; bad["unknown magic"] = b
    b = bytearray(544); struct.pack_into("<i", b, 0, 540); b[4:12] = b"n+2\0\r\n\032\n"; struct.pack_into("<2h", b, 12, 4, 16)
    struct.pack_into("<8q", b, 16, 3, 7, 6, 1 << 40, 1, 1, 1, 1); bad["NIfTI-2 with an absurd dimension"] = b + good[352:]
    with tempfile.TemporaryDirectory() as t:
        for why, blob in bad.items():
            p = os.path.join(t, "x.nii")
            open(p, "wb").write(bytes(blob))
            nx, ny, nz = C.c_int(), C.c_int(), C.c_int()
            assert not fn(p.encode(), C.byref(nx), C.byref(ny), C.byref(nz)), why


def test_matrix_io_roundtrip(shell):
    vol = np.random.default_rng(1).random((4, 5, 6)).astype(np.float32)
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "m.bin").encode()
        assert getattr(shell, "_Z17WriteMatrixToDiskPKciiiPKf")(p, 6, 5, 4, vol.ctypes.data_as(C.POINTER(C.c_float))) == 0
        raw = open(p, "rb").read()
        assert struct.unpack("<3i", raw[:12]) == (6, 5, 4) and raw[12:] == vol.tobytes()
        m, n, q = C.c_int(), C.c_int(), C.c_int()
        out = C.POINTER(C.c_float)()
        assert getattr(shell, "_Z18ReadMatrixFromDiskPKcPiS1_S1_PPf")(p, C.byref(m), C.byref(n), C.byref(q), C.byref(out)) == 0
        assert np.array_equal(np.ctypeslib.as_array(out, shape=(4, 5, 6)), vol)


@pytest.mark.gpu
def test_example_program_end_to_end(shell, orc, synth):
    """The reference's Example flow through the C++ shell, checked against the oracle."""
    va = synth.blobs((64, 64, 64), seed=1234)
    vb = synth.blobs((64, 64, 64), seed=1234, shift=(1.0, 0.0, 0.0))
    with tempfile.TemporaryDirectory() as t:
        for name, v in (("a.bin", va), ("b.bin", vb)):
            with open(os.path.join(t, name), "wb") as f:
                f.write(struct.pack("<3i", 64, 64, 64) + v.tobytes())
        out = subprocess.check_output([os.path.join(PKG, "example_sift3d"), "--raw", os.path.join(t, "a.bin"), os.path.join(t, "b.bin")],
                                      stderr=subprocess.STDOUT).decode()
    ka, da = orc.extractor(va).run(5).keypoints()
    kb, db = orc.extractor(vb).run(5).keypoints()
    xa = np.stack([ka["rx"], ka["ry"], ka["rz"]], 1); xb = np.stack([kb["rx"], kb["ry"], kb["rz"]], 1)
    want = orc.match(da, xa, db, xb, 0.85, 3)["pairs"]
    assert f"keypoints: {len(ka)} / {len(kb)}, matched pairs: {len(want)}" in out, out[-600:]
    lines = out.strip().splitlines()[-len(want):] if len(want) else []
    got = np.array([[float(v) for v in ln.replace(";", ",").split(",")] for ln in lines], np.float32).reshape(-1, 6)
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_cpp_matcher_uses_device_resident_results(shell, synth):
    """SURVEY 8f-2 in the C++ shell: keypoint vectors that alias live extractors are matched from the device-resident descriptors
    (muBruteMatcher::usedDeviceResults); deep copies take the host path; both give the same pairs and indices."""
    src = r"""
    #include "Include/cSIFT3D.h"
    #include "Include/cMatcher.h"
    #include "Include/Util/matrixIO3D.h"
    #include <cstdio>
    #include <cstring>
    using namespace CPUSIFT;
    int main(int, char** a) {
        CSIFT3D *A = CSIFT3DFactory::CreateCSIFT3D(std::string(a[1])), *B = CSIFT3DFactory::CreateCSIFT3D(std::string(a[2]));
        A->KpSiftAlgorithm(); B->KpSiftAlgorithm();
        std::vector<Keypoint> ka = A->GetKeypoints(), kb = B->GetKeypoints();
        muBruteMatcher m1, m2;
        std::vector<Cvec> r1, t1, r2, t2;
        m1.enhancedMatch(r1, t1, ka, kb, 0.85);
        // deep copies: descriptor pointers no longer alias the extractors
        std::vector<float> da(ka.size() * 768), db(kb.size() * 768);
        std::vector<Keypoint> ca = ka, cb = kb;
        for (size_t i = 0; i < ca.size(); i++) { memcpy(&da[i * 768], ka[i].desc, 768 * 4); ca[i].desc = &da[i * 768]; }
        for (size_t i = 0; i < cb.size(); i++) { memcpy(&db[i * 768], kb[i].desc, 768 * 4); cb[i].desc = &db[i * 768]; }
        m2.enhancedMatch(r2, t2, ca, cb, 0.85);
        bool same = r1.size() == r2.size() && m1.getGlodenIdx() == m2.getGlodenIdx() && m1.getSilverIdx() == m2.getSilverIdx() &&
                    m1.getGlodenDistSquare() == m2.getGlodenDistSquare();
        for (size_t i = 0; same && i < r1.size(); i++)
            same = r1[i].x == r2[i].x && r1[i].y == r2[i].y && r1[i].z == r2[i].z && t1[i].x == t2[i].x && t1[i].y == t2[i].y && t1[i].z == t2[i].z;
        printf("kp %zu %zu pairs %zu device %d host %d same %d\n", ka.size(), kb.size(), r1.size(), (int)m1.usedDeviceResults, (int)m2.usedDeviceResults, (int)same);
        // ADVICE r02: Keypoint::desc is a mutable float* -- a caller edits descriptors IN PLACE (here: one reference row zeroed, as a
        // row-masking caller would); the matcher must see the edit, i.e. leave the device fast path, and agree with deep copies
        // carrying the same edit
        size_t victim = 0;
        for (size_t i = 0; i < ka.size(); i++) if (m1.getGlodenIdx()[i] >= 0) { victim = i; break; }
        memset(ka[victim].desc, 0, 768 * 4); memset(&da[victim * 768], 0, 768 * 4);
        muBruteMatcher m3, m4;
        std::vector<Cvec> r3, t3, r4, t4;
        m3.enhancedMatch(r3, t3, ka, kb, 0.85);
        m4.enhancedMatch(r4, t4, ca, cb, 0.85);
        bool same2 = r3.size() == r4.size() && m3.getGlodenIdx() == m4.getGlodenIdx() && m3.getGlodenDistSquare() == m4.getGlodenDistSquare();
        printf("edited device %d same %d changed %d\n", (int)m3.usedDeviceResults, (int)same2, (int)(m3.getGlodenIdx() != m1.getGlodenIdx()));
        delete A; delete B;
        return 0;
    }"""
    va = synth.blobs((64, 64, 64), seed=1234)
    vb = synth.blobs((64, 64, 64), seed=1234, shift=(1.0, 0.0, 0.0))
    with tempfile.TemporaryDirectory() as t:
        for name, v in (("a.bin", va), ("b.bin", vb)):
            with open(os.path.join(t, name), "wb") as f:
                f.write(struct.pack("<3i", 64, 64, 64) + v.tobytes())
        open(os.path.join(t, "m.cpp"), "w").write(src)
        subprocess.check_call(["g++", "-std=c++14", "-I" + os.path.join(PKG, "host"), "-o", os.path.join(t, "m"), os.path.join(t, "m.cpp"),
                               "-L" + PKG, "-lsift3d", "-lsift3d_hip", "-Wl,-rpath," + PKG])
        out = subprocess.check_output([os.path.join(t, "m"), os.path.join(t, "a.bin"), os.path.join(t, "b.bin")], stderr=subprocess.STDOUT).decode()
    last = out.strip().splitlines()[-2].split()
    assert int(last[1]) > 20 and int(last[4]) > 5, out
    assert last[5:] == ["device", "1", "host", "0", "same", "1"], out
    assert out.strip().splitlines()[-1].split() == ["edited", "device", "0", "same", "1", "changed", "1"], out


@pytest.mark.gpu
def test_cpp_all_pairs_match(shell, synth):
    """CSIFT3D::AllPairsMatch (r04, configs[4] for a single-process C++ caller): three extractors, the six ordered pairs from their
    device-resident results equal muBruteMatcher::enhancedMatch on the same keypoint vectors, pair by pair; extractors still in
    flight (KpSiftAlgorithmAsync) are completed by the call."""
    src = r"""
    #include "Include/cSIFT3D.h"
    #include "Include/cMatcher.h"
    #include <cstdio>
    using namespace CPUSIFT;
    int main(int, char** a) {
        std::vector<CSIFT3D*> ex;
        for (int k = 1; k <= 3; k++) { ex.push_back(CSIFT3DFactory::CreateCSIFT3D(std::string(a[k]))); ex.back()->KpSiftAlgorithmAsync(); }
        std::vector<CSIFT3D::PairMatch> all = CSIFT3D::AllPairsMatch(ex, 0.85, 3);
        std::vector<std::vector<Keypoint>> kp;
        for (auto e : ex) kp.push_back(e->GetKeypoints());
        int same = (int)all.size() == 6, total = 0;
        for (auto &p : all) {
            muBruteMatcher m; std::vector<Cvec> r, t;
            m.enhancedMatch(r, t, kp[p.ref], kp[p.tar], 0.85);
            same = same && r.size() == p.refMatch.size() && m.getGlodenIdx() == p.glodenIdx;
            for (size_t i = 0; same && i < r.size(); i++) same = r[i].x == p.refMatch[i].x && r[i].z == p.refMatch[i].z && t[i].y == p.tarMatch[i].y && t[i].x == p.tarMatch[i].x;
            total += (int)r.size();
        }
        printf("pairs %zu matched %d same %d\n", all.size(), total, same);
        for (auto e : ex) delete e;
        return 0;
    }"""
    vols = [synth.blobs((64, 64, 64), seed=1234), synth.blobs((64, 64, 64), seed=1234, shift=(1.0, 0.0, 0.0)), synth.blobs((64, 64, 64), seed=99)]
    with tempfile.TemporaryDirectory() as t:
        names = []
        for k, v in enumerate(vols):
            names.append(os.path.join(t, "v%d.bin" % k))
            with open(names[-1], "wb") as f:
                f.write(struct.pack("<3i", 64, 64, 64) + v.tobytes())
        open(os.path.join(t, "m.cpp"), "w").write(src)
        subprocess.check_call(["g++", "-std=c++14", "-I" + os.path.join(PKG, "host"), "-o", os.path.join(t, "m"), os.path.join(t, "m.cpp"),
                               "-L" + PKG, "-lsift3d", "-lsift3d_hip", "-lpthread", "-Wl,-rpath," + PKG])
        out = subprocess.check_output([os.path.join(t, "m")] + names, stderr=subprocess.STDOUT).decode()
    last = out.strip().splitlines()[-1].split()
    assert last[:2] == ["pairs", "6"] and int(last[3]) > 10 and last[4:] == ["same", "1"], out


@pytest.mark.gpu
def test_config1_256_cubed_through_nifti(shell, orc, synth):
    """BASELINE configs[1]: two 256^3 volumes stored as NIfTI-1 files (int16 and gzip-compressed float32), read by readNiiFile, full
    KpSiftAlgorithm + enhancedMatch through the C++ shell; keypoint counts and matched pairs equal the oracle's on the same arrays."""
    import gzip
    va = synth.blobs((256, 256, 256), seed=1234)
    vb = synth.blobs((256, 256, 256), seed=1234, shift=(1.0, 0.0, 0.0))
    qa = np.round(va * 20000).astype(np.int16)            # what an int16 scanner volume looks like; the extractor normalises by max|v|
    fa, fb = qa.astype(np.float32), vb.astype(np.float32)
    with tempfile.TemporaryDirectory() as t:
        open(os.path.join(t, "a.nii"), "wb").write(nifti1(qa, "i2", 4))
        open(os.path.join(t, "b.nii.gz"), "wb").write(gzip.compress(nifti1(vb, "f4", 16), compresslevel=1))
        out = subprocess.check_output([os.path.join(PKG, "example_sift3d"), os.path.join(t, "a.nii"), os.path.join(t, "b.nii.gz")],
                                      stderr=subprocess.STDOUT).decode()
    ka, da = orc.extractor(fa).run(5).keypoints()
    kb, db = orc.extractor(fb).run(5).keypoints()
    xa = np.stack([ka["rx"], ka["ry"], ka["rz"]], 1); xb = np.stack([kb["rx"], kb["ry"], kb["rz"]], 1)
    want = orc.match(da, xa, db, xb, 0.85, 3)["pairs"]
    assert "Dimensions of reference image:256 256 256" in out
    assert f"keypoints: {len(ka)} / {len(kb)}, matched pairs: {len(want)}" in out, out[-400:]


def test_sift_kp_csv_roundtrip(shell):
    """write_sift_kp / read_sift_kp (reference cUtil.cc:938-954, 1002-1016): "%.5lf,%.5lf,%.5lf" lines."""
    src = r"""
    #include "Include/cUtil.h"
    #include <cstdio>
    int main(int, char** a) {
        std::vector<CPUSIFT::Cvec> v = {{1.5f, 2.25f, 3.0f}, {10.123456f, 0.f, -4.5f}}, w;
        CPUSIFT::write_sift_kp(v, a[1]);
        CPUSIFT::read_sift_kp(a[1], w);
        if (w.size() != 2) return 1;
        printf("%.5f %.5f %.5f\n", w[1].x, w[1].y, w[1].z);
        return 0;
    }"""
    with tempfile.TemporaryDirectory() as t:
        open(os.path.join(t, "m.cpp"), "w").write(src)
        subprocess.check_call(["g++", "-std=c++14", "-I" + os.path.join(PKG, "host"), "-o", os.path.join(t, "m"), os.path.join(t, "m.cpp"),
                               "-L" + PKG, "-lsift3d", "-lsift3d_hip", "-Wl,-rpath," + PKG])
        out = subprocess.check_output([os.path.join(t, "m"), os.path.join(t, "kp.csv")]).decode()
        assert open(os.path.join(t, "kp.csv")).read().splitlines() == ["1.50000,2.25000,3.00000", "10.12346,0.00000,-4.50000"]
        assert out.split() == ["10.12346", "0.00000", "-4.50000"]


def test_sift_kp_csv_matches_reference_writer_and_reader(shell):
    """g10: the file the REFERENCE's write_sift_kp produced for a coordinate list (through oracle/_ref, see
    tests/golden/make_golden.py) byte for byte, and what its read_sift_kp read back from it, bit for bit."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "g10_sift_kp.npz"))
    src = r"""
    #include "Include/cUtil.h"
    #include <cstdio>
    #include <cstdlib>
    int main(int, char** a) {
        std::vector<CPUSIFT::Cvec> v, w;
        FILE* f = fopen(a[1], "rb");
        float p[3];
        while (fread(p, 4, 3, f) == 3) v.push_back(CPUSIFT::Cvec(p[0], p[1], p[2]));
        fclose(f);
        CPUSIFT::write_sift_kp(v, a[2]);       // our writer on the reference's input
        CPUSIFT::read_sift_kp(a[3], w);        // our reader on the REFERENCE's file
        f = fopen(a[4], "wb");
        for (auto& c : w) { float q[3] = {c.x, c.y, c.z}; fwrite(q, 4, 3, f); }
        fclose(f);
        return 0;
    }"""
    with tempfile.TemporaryDirectory() as t:
        open(os.path.join(t, "m.cpp"), "w").write(src)
        subprocess.check_call(["g++", "-std=c++14", "-I" + os.path.join(PKG, "host"), "-o", os.path.join(t, "m"), os.path.join(t, "m.cpp"),
                               "-L" + PKG, "-lsift3d", "-lsift3d_hip", "-Wl,-rpath," + PKG])
        open(os.path.join(t, "in.bin"), "wb").write(g["xyz"].tobytes())
        open(os.path.join(t, "ref.csv"), "wb").write(g["csv"].tobytes())
        subprocess.check_call([os.path.join(t, "m"), os.path.join(t, "in.bin"), os.path.join(t, "ours.csv"), os.path.join(t, "ref.csv"),
                               os.path.join(t, "back.bin")], stdout=subprocess.DEVNULL)
        assert open(os.path.join(t, "ours.csv"), "rb").read() == g["csv"].tobytes()
        back = np.frombuffer(open(os.path.join(t, "back.bin"), "rb").read(), np.float32).reshape(-1, 3)
    assert back.shape == g["read_back"].shape
    assert np.array_equal(back.view(np.uint32), g["read_back"].view(np.uint32))


FREE_FN_SRC = r"""
#include "Include/cSIFT3D.h"
#include <cstdio>
#include <cstdlib>
using namespace CPUSIFT;
int main() {
	TexImage a(16, 12, 10), b(16, 12, 10), g, d, h(8, 6, 5);
	a.MallocArrayMemory(); b.MallocArrayMemory(); h.MallocArrayMemory();
	for (int z = 0; z < 10; z++) for (int y = 0; y < 12; y++) for (int x = 0; x < 16; x++) {
		a.SetImageDataWithIdx((float)((x * 7 + y * 3 + z * 11) % 13) / 13.0f, x, y, z);
		b.SetImageDataWithIdx((float)((x * 5 + y * 2 + z) % 7) / 7.0f, x, y, z);
	}
	a.SetImageUnit(2.f, 2.f, 2.f);
	GaussianSmooth_3D(&a, &g, 1.1f);
	Sub(&a, &b, &d);
	DownSample_3D(&a, &h);
	int bad = 0;
	for (int z = 0; z < 10; z++) for (int y = 0; y < 12; y++) for (int x = 0; x < 16; x++)
		if (d.GetImageDataWithIdx(x, y, z) != (b.GetImageDataWithIdx(x, y, z) - a.GetImageDataWithIdx(x, y, z)) * (-1)) bad++;
	for (int z = 0; z < 5; z++) for (int y = 0; y < 6; y++) for (int x = 0; x < 8; x++)
		if (h.GetImageDataWithIdx(x, y, z) != a.GetImageDataWithIdx(2 * x, 2 * y, 2 * z)) bad++;
	double s = 0; for (int z = 0; z < 10; z++) for (int y = 0; y < 12; y++) for (int x = 0; x < 16; x++) s += g.GetImageDataWithIdx(x, y, z);
	printf("bad %d dims %d %d %d unit %g sum %.6f\n", bad, g.GetDimX(), g.GetDimY(), g.GetDimZ(), g.GetUnitX(), s);
	return bad;
}
"""


def _build_free_fn_program(t):
    os.makedirs(os.path.join(t, "Include"), exist_ok=True)
    src = os.path.join(t, "free_fn.cpp")
    open(src, "w").write(FREE_FN_SRC)
    subprocess.check_call(["g++", "-std=c++14", "-I" + os.path.join(PKG, "host"), "-o", os.path.join(t, "free_fn"), src,
                           "-L" + PKG, "-lsift3d", "-lsift3d_hip", "-Wl,-rpath," + PKG])
    return os.path.join(t, "free_fn")


def test_free_functions_of_the_public_header_link(shell):
    """DownSample_3D / GaussianSmooth_3D / Sub (Include/cSIFT3D.h:210-218) are exported by the C++ shell: a user program compiles and links."""
    with tempfile.TemporaryDirectory() as t:
        _build_free_fn_program(t)


@pytest.mark.gpu
def test_free_functions_of_the_public_header_run(shell, orc):
    with tempfile.TemporaryDirectory() as t:
        out = subprocess.check_output([_build_free_fn_program(t)], text=True)
    assert out.startswith("bad 0 dims 16 12 10 unit 2 "), out
    # the smoothed volume against the oracle's GaussianSmooth_3D on the same input
    z, y, x = np.meshgrid(np.arange(10), np.arange(12), np.arange(16), indexing="ij")
    a = (((x * 7 + y * 3 + z * 11) % 13).astype(np.float32) / np.float32(13.0)).astype(np.float32)
    want = orc.gaussian_smooth(a, 1.1).astype(np.float64).sum()
    assert abs(float(out.split("sum")[1]) - want) < 1e-4 * max(1.0, abs(want))


def test_header_surface_helpers_against_golden_g7(shell):
    """The small host utilities of the reference's public header (cart2bary, Check_intersect_faces, Initialize_geometry, Im_permute,
    the TexImage members, the Tri / Mesh / Image / EigenVal records, the templated matrix IO): a program that names them compiles
    against OUR headers, and the mesh / face / barycentric results equal golden g7 -- which the reference itself produced -- bit for bit."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "g7_mesh.npz"))
    with tempfile.TemporaryDirectory() as t:
        exe = os.path.join(t, "helpers_check")
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-I" + os.path.join(PKG, "host"), "-o", exe,
                               os.path.join(PKG, "host", "example", "helpers_check.cpp"), "-L" + PKG, "-lsift3d", "-lsift3d_hip",
                               "-Wl,-rpath," + PKG])
        dirs = np.ascontiguousarray(g["dirs"], np.float32).view(np.uint32)
        txt = "%d\n" % len(dirs) + "\n".join("%08x %08x %08x" % tuple(r) for r in dirs) + "\n"
        out = subprocess.run([exe, t], input=txt, capture_output=True, text=True, check=True).stdout.splitlines()
    tri = [l.split() for l in out if l.startswith("tri ")]
    assert len(tri) == 20
    idx = np.array([[int(v) for v in r[1:4]] for r in tri], np.int32)
    verts = np.array([[int(v, 16) for v in r[4:]] for r in tri], np.uint32).reshape(20, 3, 3)
    assert np.array_equal(idx, g["idx"]) and np.array_equal(verts, g["verts"].view(np.uint32))
    rows = [l.split() for l in out if l.startswith("dir ")]
    faces = np.array([int(r[1]) for r in rows], np.int32)
    bary = np.array([[int(v, 16) for v in r[2:]] for r in rows], np.uint32)
    assert np.array_equal(faces, g["faces"])
    hit = faces >= 0
    assert np.array_equal(bary[hit], g["bary"].view(np.uint32)[hit])
    rest = {l.split()[0]: l.split()[1:] for l in out if not l.startswith(("tri ", "dir "))}
    assert rest["tex"] == ["96", "4", "3", "2", "0", "1"]          # sizing ctor: _numsize in BYTES, scale 0, units 1 (cTexImage.cc:19-33)
    assert rest["view"] == ["23", "13"] and rest["perm"] == ["2", "3", "4", "23"]
    assert rest["nvox"] == ["8"] and rest["size"] == ["0", "1", "2", "3"]   # SetImageSize: _numsize in VOXELS (cTexImage.cc:76)
    assert rest["reset"] == ["32", "2", "1"] and rest["tr"] == ["3", "6", "7", "1", "0"]
    assert rest["dmat"] == ["0", "3", "2", "1", "5.5"] and rest["imat"] == ["0", "1", "2", "3", "6"]
