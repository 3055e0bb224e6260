"""ctypes driver for the two CPU checkers that export oracle/oracle_api.h.

TEST INFRASTRUCTURE ONLY -- imported by tests/, tests/golden/make_golden.py,
``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg, never by the product.

  * ``load("orc")``  -> oracle/liboracle3dsift.so      (our C restatement, travels everywhere)
  * ``load("ref")``  -> oracle/_ref/libref3dsift.so    (the real reference compiled from
                         /root/reference by ``make -C oracle ref``; exists only where it was built)
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DESC = 768

KP_DTYPE = np.dtype(
    [
        ("x", "<f4"), ("y", "<f4"), ("z", "<f4"),
        ("scale", "<f4"),
        ("octave", "<i4"), ("level", "<i4"),
        ("rx", "<f4"), ("ry", "<f4"), ("rz", "<f4"),
        ("win", "<f4", (3,)),
        ("eigvalue", "<f4", (3,)),
        ("eigvector", "<f4", (9,)),
        ("Rotation", "<f4", (9,)),
        ("str_tensor", "<f4", (9,)),
    ]
)
assert KP_DTYPE.itemsize == 168

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def _f(a):
    return a.ctypes.data_as(_fp)


def _i(a):
    return a.ctypes.data_as(_ip)


class Checker:
    """Same python API over either prefix (orc_ / ref_)."""

    def __init__(self, prefix, path):
        self.prefix = prefix
        self.lib = C.CDLL(path)
        g = lambda n: getattr(self.lib, prefix + "_" + n)
        self._create = g("create")
        self._create.restype = C.c_void_p
        self._create.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_float] * 5
        self._destroy = g("destroy"); self._destroy.argtypes = [C.c_void_p]; self._destroy.restype = None
        self._set_threads = g("set_threads"); self._set_threads.argtypes = [C.c_int]
        self._run = g("run"); self._run.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]; self._run.restype = None
        self._num_octaves = g("num_octaves"); self._num_octaves.argtypes = [C.c_void_p]; self._num_octaves.restype = C.c_int
        self._level_info = g("level_info"); self._level_info.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _fp, _fp]; self._level_info.restype = None
        self._copy_level = g("copy_level"); self._copy_level.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp]; self._copy_level.restype = None
        self._copy_input = g("copy_input"); self._copy_input.argtypes = [C.c_void_p, _fp]; self._copy_input.restype = None
        self._num_extrema = g("num_extrema"); self._num_extrema.argtypes = [C.c_void_p]; self._num_extrema.restype = C.c_int
        self._copy_extrema = g("copy_extrema"); self._copy_extrema.argtypes = [C.c_void_p, C.c_void_p]; self._copy_extrema.restype = None
        self._num_keypoints = g("num_keypoints"); self._num_keypoints.argtypes = [C.c_void_p]; self._num_keypoints.restype = C.c_int
        self._copy_keypoints = g("copy_keypoints"); self._copy_keypoints.argtypes = [C.c_void_p, C.c_void_p, _fp]; self._copy_keypoints.restype = None
        self._gaussian_smooth = g("gaussian_smooth"); self._gaussian_smooth.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp]; self._gaussian_smooth.restype = None
        self._gaussian_taps = g("gaussian_taps"); self._gaussian_taps.argtypes = [C.c_float, _fp]; self._gaussian_taps.restype = C.c_int
        self._mesh = g("mesh"); self._mesh.argtypes = [_fp, _ip]; self._mesh.restype = C.c_int
        self._intersect = g("intersect"); self._intersect.argtypes = [_fp, _fp]; self._intersect.restype = C.c_int
        self._orient_one = g("orient_one"); self._orient_one.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, C.c_int] + [C.c_float] * 4; self._orient_one.restype = C.c_int
        self._describe_one = g("describe_one"); self._describe_one.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp]; self._describe_one.restype = None
        self._match = g("match")
        self._match.argtypes = [_fp, _fp, C.c_int, _fp, _fp, C.c_int, C.c_double, C.c_int, _ip, _ip, _fp, _fp, _fp]
        self._match.restype = C.c_int

    # ---- extractor -------------------------------------------------------------------
    def set_threads(self, n):
        self._set_threads(int(n))

    def extractor(self, vol, **params):
        return Extractor(self, vol, **params)

    # ---- unit level ------------------------------------------------------------------
    def gaussian_smooth(self, vol, sigma):
        vol = np.ascontiguousarray(vol, dtype=np.float32)
        nz, ny, nx = vol.shape
        out = np.empty_like(vol)
        self._gaussian_smooth(_f(vol), nx, ny, nz, float(sigma), _f(out))
        return out

    def gaussian_taps(self, sigma):
        buf = np.zeros(64, np.float32)
        w = self._gaussian_taps(float(sigma), _f(buf))
        return None if w < 0 else buf[:w].copy()

    def mesh(self):
        v = np.zeros((20, 3, 3), np.float32)
        idx = np.zeros((20, 3), np.int32)
        self._mesh(_f(v), _i(idx))
        return v, idx

    def intersect(self, grad):
        g = np.ascontiguousarray(grad, np.float32)
        b = np.zeros(3, np.float32)
        r = self._intersect(_f(g), _f(b))
        return r, b

    def orient_one(self, kp, level, unit, sigma, max_eig_ratio=0.9, corner_thresh=0.4):
        level = np.ascontiguousarray(level, np.float32)
        nz, ny, nx = level.shape
        k = np.array([kp], dtype=KP_DTYPE)
        r = self._orient_one(k.ctypes.data, _f(level), nx, ny, nz, float(unit), float(sigma), float(max_eig_ratio), float(corner_thresh))
        return r, k[0]

    def describe_one(self, kp, level, unit):
        level = np.ascontiguousarray(level, np.float32)
        nz, ny, nx = level.shape
        k = np.array([kp], dtype=KP_DTYPE)
        d = np.zeros(DESC, np.float32)
        self._describe_one(k.ctypes.data, _f(level), nx, ny, nz, float(unit), _f(d))
        return k[0], d

    def match(self, ref_desc, ref_xyz, tar_desc, tar_xyz, thresh=0.85, mode=3):
        a = np.ascontiguousarray(ref_desc, np.float32); b = np.ascontiguousarray(tar_desc, np.float32)
        ax = np.ascontiguousarray(ref_xyz, np.float32); bx = np.ascontiguousarray(tar_xyz, np.float32)
        n, m = a.shape[0], b.shape[0]
        gi = np.zeros(max(n, 1), np.int32); si = np.zeros(max(n, 1), np.int32)
        gd = np.zeros(max(n, 1), np.float32); sd = np.zeros(max(n, 1), np.float32)
        pairs = np.zeros((max(n, 1), 6), np.float32)
        k = self._match(_f(a), _f(ax), n, _f(b), _f(bx), m, float(thresh), int(mode), _i(gi), _i(si), _f(gd), _f(sd), _f(pairs))
        return dict(gIdx=gi[:n], sIdx=si[:n], gDist=gd[:n], sDist=sd[:n], pairs=pairs[:k].copy())


class Extractor:
    """vol is indexed [z, y, x] (x fastest), like the reference's TexImage (cTexImage.h:5)."""

    def __init__(self, chk, vol, num_kp_levels=3, sigma_default=1.6, sigma_n_default=1.15, peak_thresh=0.1,
                 max_eig_thres=0.9, corner_thresh=0.4):
        self.chk = chk
        vol = np.ascontiguousarray(vol, dtype=np.float32)
        self.shape = vol.shape
        nz, ny, nx = vol.shape
        self.levels = num_kp_levels
        self.h = chk._create(_f(vol), nx, ny, nz, num_kp_levels, sigma_default, sigma_n_default, peak_thresh,
                             max_eig_thres, corner_thresh)
        self.times = None

    def close(self):
        if self.h:
            self.chk._destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, upto=5):
        t = (C.c_double * 6)()
        self.chk._run(self.h, upto, t)
        self.times = dict(zip(["init", "gss", "dog", "detect", "orient", "descr"], list(t)))
        return self

    @property
    def num_octaves(self):
        return self.chk._num_octaves(self.h)

    def level_info(self, is_dog, idx):
        d = np.zeros(3, np.int32); u = np.zeros(3, np.float32); s = np.zeros(1, np.float32)
        self.chk._level_info(self.h, int(is_dog), idx, _i(d), _f(u), _f(s))
        return tuple(int(v) for v in d), tuple(float(v) for v in u), float(s[0])

    def level(self, is_dog, idx):
        (nx, ny, nz), _, _ = self.level_info(is_dog, idx)
        out = np.empty((nz, ny, nx), np.float32)
        self.chk._copy_level(self.h, int(is_dog), idx, _f(out))
        return out

    def gss(self, octave, i):
        return self.level(0, octave * (self.levels + 3) + i)

    def dog(self, octave, i):
        return self.level(1, octave * (self.levels + 2) + i)

    def input(self):
        out = np.empty(self.shape, np.float32)
        self.chk._copy_input(self.h, _f(out))
        return out

    def extrema(self):
        n = self.chk._num_extrema(self.h)
        out = np.zeros(n, KP_DTYPE)
        if n:
            self.chk._copy_extrema(self.h, out.ctypes.data)
        return out

    def keypoints(self):
        n = self.chk._num_keypoints(self.h)
        out = np.zeros(n, KP_DTYPE)
        desc = np.zeros((n, DESC), np.float32)
        if n:
            self.chk._copy_keypoints(self.h, out.ctypes.data, _f(desc))
        return out, desc


_cache = {}


def lib_path(prefix):
    if prefix == "orc":
        # S3D_ORACLE_LIB: the ASan + UBSan build of the restatement (`make -C oracle asan`; tests/test_sanitizers_cpu.py)
        return os.environ.get("S3D_ORACLE_LIB") or os.path.join(ROOT, "oracle", "liboracle3dsift.so")
    return os.path.join(ROOT, "oracle", "_ref", "libref3dsift.so")


def available(prefix):
    return os.path.exists(lib_path(prefix))


def load(prefix="orc"):
    if prefix not in _cache:
        _cache[prefix] = Checker(prefix, lib_path(prefix))
    return _cache[prefix]
