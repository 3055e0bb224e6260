"""CPU sanitizer runs: the oracle's C restatement and the host-side C++ (file readers / writers,
shell classes) built with ASan + UBSan.  The reference's own heap overflow at Src/cSIFT3D.cc:762 was found this way;
these runs keep our CPU-side code clean of that class of defect.  No GPU needed."""
import gzip
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _asan_runtime():
    p = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    if not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip("no libasan in this toolchain")
    return os.path.realpath(p)


def test_oracle_golden_suite_under_asan_ubsan():
    """tests/test_oracle_golden.py (every golden vector of the reference + the edge cases) against the ASan + UBSan build of
    oracle/sift3d_oracle.c."""
    rt = _asan_runtime()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "oracle", "_asan", "liboracle3dsift.so")
    env = dict(os.environ, LD_PRELOAD=rt, S3D_ORACLE_LIB=lib, OMP_NUM_THREADS="4",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "passed" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def _nifti1(vol, dtype, code, big_endian=False, vox_offset=352.0, hdr_bytes=352):
    nz, ny, nx = vol.shape
    e = ">" if big_endian else "<"
    h = bytearray(352)
    struct.pack_into(e + "i", h, 0, 348)
    struct.pack_into(e + "8h", h, 40, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into(e + "h", h, 70, code)
    struct.pack_into(e + "h", h, 72, np.dtype(dtype).itemsize * 8)
    struct.pack_into(e + "f", h, 108, vox_offset)
    h[344:348] = b"n+1\0"
    return bytes(h[:hdr_bytes]) + vol.astype(np.dtype(dtype).newbyteorder(e)).tobytes()


def test_host_readers_and_shell_under_asan_ubsan():
    """3dsift_amd/host/src/*.cpp with ASan + UBSan: every NIfTI datatype / byte order / gzip, malformed and random headers, truncated
    payloads, raw matrices, key-point CSV files, and the shell classes (no-device error paths here; a real run on a GPU box)."""
    _asan_runtime()
    if not os.path.exists(os.path.join(ROOT, "3dsift_amd", "libsift3d_hip.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "3dsift_amd", "csrc"), "-j8"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "3dsift_amd", "host"), "asan"], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "3dsift_amd", "host", "build_asan", "asan_check")
    rng = np.random.Generator(np.random.PCG64(4))
    base = rng.random((5, 6, 7)) * 200 - 60
    with tempfile.TemporaryDirectory() as t:
        want = {}
        for name, dt, code, be, gz in (("f4le", "f4", 16, False, False), ("f4be", "f4", 16, True, False), ("i2gz", "i2", 4, False, True),
                                       ("u1", "u1", 2, False, False), ("f8be", "f8", 64, True, False), ("u2", "u2", 512, False, False),
                                       ("i4", "i4", 8, False, False), ("i1", "i1", 256, False, False), ("u4gz", "u4", 768, False, True)):
            info = np.iinfo(dt) if np.dtype(dt).kind in "iu" else None
            vol = np.clip(base, info.min, info.max).astype(dt) if info else base.astype(dt)
            blob = _nifti1(vol, dt, code, big_endian=be)
            fn = f"good_{name}.nii" + (".gz" if gz else "")
            open(os.path.join(t, fn), "wb").write(gzip.compress(blob) if gz else blob)
            v = vol.astype(np.float32).ravel().astype(np.float64)
            want[fn] = float((v * ((np.arange(v.size) % 97) + 1)).sum())
        good = bytearray(_nifti1(base.astype("i2"), "i2", 4))
        bad = {}
        b = bytearray(good); struct.pack_into("<h", b, 42, -7); bad["negdim"] = b
        b = bytearray(good); struct.pack_into("<h", b, 42, 32767); struct.pack_into("<h", b, 44, 32767); struct.pack_into("<h", b, 46, 32767); bad["hugedim"] = b
        b = bytearray(good); struct.pack_into("<h", b, 72, 32); bad["bitpix"] = b
        b = bytearray(good); struct.pack_into("<f", b, 108, float("nan")); bad["nanoff"] = b
        b = bytearray(good); struct.pack_into("<f", b, 108, 1.0e8); bad["faroff"] = b
        b = bytearray(good); struct.pack_into("<h", b, 70, 1536); bad["dtype"] = b
        b = bytearray(good); struct.pack_into("<h", b, 40, 9); bad["ndim9"] = b
        bad["trunc"] = good[:-10]
        bad["short"] = good[:100]
        bad["empty"] = b""
        bad["random"] = rng.integers(0, 256, 4096, dtype=np.uint8).tobytes()
        b = bytearray(rng.integers(0, 256, 4096, dtype=np.uint8).tobytes()); struct.pack_into("<i", b, 0, 348); b[344:348] = b"n+1\0"; bad["random348"] = b
        for k, v in bad.items():
            open(os.path.join(t, f"bad_{k}.nii"), "wb").write(bytes(v))
        # r05: NIfTI-2 and two-file images
        vol2 = base.astype("f4")
        h2 = bytearray(544); struct.pack_into("<i", h2, 0, 540); h2[4:12] = b"n+2\0\r\n\032\n"; struct.pack_into("<2h", h2, 12, 16, 32)
        struct.pack_into("<8q", h2, 16, 3, 7, 6, 5, 1, 1, 1, 1); struct.pack_into("<q", h2, 168, 544)
        open(os.path.join(t, "good_v2.nii"), "wb").write(bytes(h2) + vol2.tobytes())
        hp = bytearray(_nifti1(vol2, "f4", 16)[:348]); hp[344:348] = b"ni1\0"; struct.pack_into("<f", hp, 108, 0.0)
        open(os.path.join(t, "good_pair.hdr"), "wb").write(bytes(hp)); open(os.path.join(t, "good_pair.img"), "wb").write(vol2.tobytes())
        v2 = vol2.ravel().astype(np.float64)
        for fn in ("good_v2.nii", "good_pair.hdr", "good_pair.img"):
            want[fn] = float((v2 * ((np.arange(v2.size) % 97) + 1)).sum())
        open(os.path.join(t, "bad_noimg.hdr"), "wb").write(bytes(hp))                                   # the .img is missing
        open(os.path.join(t, "bad_shortimg.hdr"), "wb").write(bytes(hp)); open(os.path.join(t, "bad_shortimg.img"), "wb").write(vol2.tobytes()[:100])
        b = bytearray(rng.integers(0, 256, 4096, dtype=np.uint8).tobytes()); struct.pack_into("<i", b, 0, 540); b[4:12] = b"n+2\0\r\n\032\n"
        open(os.path.join(t, "bad_random540.nii"), "wb").write(bytes(b))
        b = bytearray(rng.integers(0, 256, 4096, dtype=np.uint8).tobytes()); struct.pack_into(">i", b, 0, 540); b[4:12] = b"ni2\0\r\n\032\n"
        open(os.path.join(t, "bad_random540be.hdr"), "wb").write(bytes(b))
        more_bad = ["bad_noimg.hdr", "bad_shortimg.hdr", "bad_shortimg.img", "bad_random540.nii", "bad_random540be.hdr"]
        open(os.path.join(t, "bad_notgz.nii.gz"), "wb").write(b"\x1f\x8b" + rng.integers(0, 256, 200, dtype=np.uint8).tobytes())
        m = rng.random((4, 5, 6)).astype(np.float32)
        open(os.path.join(t, "m.bin"), "wb").write(struct.pack("<3i", 6, 5, 4) + m.tobytes())
        open(os.path.join(t, "m_trunc.bin"), "wb").write(struct.pack("<3i", 6, 5, 4) + m.tobytes()[:-8])
        open(os.path.join(t, "m_bad.bin"), "wb").write(struct.pack("<3i", -6, 5, 4) + m.tobytes())
        open(os.path.join(t, "kp_garbage.csv"), "w").write("1.0,2.0\nabc,def,ghi\n,,\n3,4,5\n" + "9" * 5000 + "\n")
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
        r = subprocess.run([exe, t], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-3000:]
    lines = dict((ln.split()[1], ln.split()[2:]) for ln in r.stdout.splitlines() if ln.startswith("nii "))
    for fn, chk in want.items():
        assert lines[fn][:3] == ["7", "6", "5"], (fn, lines[fn])
        assert abs(float(lines[fn][3]) - chk) <= 1e-6 * max(1.0, abs(chk)), (fn, lines[fn], chk)
    for k in bad:
        if k in ("faroff",):
            continue
        assert lines[f"bad_{k}.nii"] == ["rejected"], (k, lines[f"bad_{k}.nii"])
    for fn in more_bad:
        assert lines[fn] == ["rejected"], (fn, lines[fn])
    out = r.stdout
    want_m = float((m.ravel().astype(np.float64) * ((np.arange(m.size) % 97) + 1)).sum())
    ml = [ln.split() for ln in out.splitlines() if ln.startswith("matrix ") or ln.startswith("matrix2 ")]
    assert len(ml) == 2 and all(x[1:4] == ["6", "5", "4"] and abs(float(x[4]) - want_m) < 1e-6 * want_m for x in ml), ml
    assert "matrix_trunc 1" in out and "matrix_bad 1" in out and "matrix_missing 1" in out  # the reference's convention: 0 ok, 1 failure
    assert "csv 3 10.12346 0.00000 -4.50000" in out and "asan_check done" in out
    assert "scatter 32 voxels sum" in out and "orient off-voxel code 0" in out   # r05: the per-voxel / per-keypoint free functions ran under the sanitizers
