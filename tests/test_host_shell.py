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
