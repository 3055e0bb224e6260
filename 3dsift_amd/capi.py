"""ctypes binding of the C-ABI in include/sift3d_hip.h (lib: 3dsift_amd/libsift3d_hip.so).

This is plumbing for tests/ and bench.py: the product is the HIP library itself and the C++ shell
in 3dsift_amd/host/ (namespace CPUSIFT).  There is NO CPU fallback anywhere in this module: if the
library is missing or no GPU is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("S3D_LIB") or os.path.join(HERE, "libsift3d_hip.so")  # S3D_LIB: kernel-variant A/B builds (scripts/ab_pyramid.py)
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

# every symbol include/sift3d_hip.h and include/sift3d_hip_test.h declare (tests check that the library exports all of them)
SYMBOLS = [
    "sift3d_default_params", "sift3d_create", "sift3d_destroy", "sift3d_run", "sift3d_run_async", "sift3d_run_async_after", "sift3d_wait", "sift3d_run_stages",
    "sift3d_stage_times", "sift3d_num_keypoints", "sift3d_get_keypoints", "sift3d_device_results",
    "sift3d_num_octaves", "sift3d_level_info", "sift3d_copy_level", "sift3d_copy_input", "sift3d_num_extrema",
    "sift3d_get_extrema", "sift3d_get_orientation_codes", "sift3d_gaussian_smooth", "sift3d_downsample", "sift3d_dog_sub", "sift3d_conv_axis",
    "sift3d_orient_keypoint", "sift3d_describe_keypoint", "sift3d_match",
    "sift3d_match_handles",
    "sift3d_device_count", "sift3d_error_string", "sift3d_last_error",
    # multi-GPU sharding (z-slabs of octave 0 + seeded replicated tail)
    "sift3d_slab_min_halo", "sift3d_slab_arena_floats", "sift3d_slab_create", "sift3d_slab_buffer", "sift3d_slab_upload",
    "sift3d_slab_input_absmax", "sift3d_slab_input_scale", "sift3d_slab_level", "sift3d_slab_level_hw", "sift3d_slab_halo_planes",
    "sift3d_slab_sync", "sift3d_slab_get_dogmax", "sift3d_slab_set_dogmax", "sift3d_slab_detect", "sift3d_slab_describe",
    "sift3d_slab_decimate", "sift3d_slab_admits", "sift3d_slab_set_ghost", "sift3d_slab_min_halo_ghost", "sift3d_create_seeded", "sift3d_seed_upload", "sift3d_set_describe_partition",
    "sift3d_export_device", "sift3d_import_descriptors_device", "sift3d_run_partial_orientation",
    "sift3d_export_orientation_device", "sift3d_import_orientation_device", "sift3d_run_describe",
    "sift3d_set_stream", "sift3d_slab_export_dogmax_device", "sift3d_slab_import_dogmax_device", "sift3d_slab_decimate_async",
    # r05: descriptor windows split along z over the ranks (partial integer histograms)
    "sift3d_slab_set_desc_partial", "sift3d_slab_min_halo_partial", "sift3d_slab_record_bytes", "sift3d_slab_desc_reach", "sift3d_slab_orient",
    "sift3d_slab_export_records", "sift3d_slab_describe_partial", "sift3d_slab_describe_finish", "sift3d_slab_orient_launch",
    "sift3d_slab_orient_count",
    # r06: the same stages without host read-backs, the tail's seed level in place, the native driver's plan
    "sift3d_slab_keypoints_launch", "sift3d_slab_keypoints_count", "sift3d_slab_describe_finish_launch", "sift3d_slab_describe_finish_count",
    "sift3d_seed_buffer", "sift3d_sharded_plan", "sift3d_slab_describe_launch", "sift3d_sharded_traffic",
    # test hooks / debug accessors / matcher timing
    # native driver of the z-slab sharding
    "sift3d_sharded_create", "sift3d_sharded_create_ex", "sift3d_sharded_run", "sift3d_sharded_num_keypoints", "sift3d_sharded_get_keypoints", "sift3d_sharded_info",
    "sift3d_sharded_error", "sift3d_sharded_destroy",
    "sift3d_test_hook", "sift3d_debug_counters", "sift3d_debug_face_lookup", "sift3d_match_times", "sift3d_debug_copy_bandwidth",
    "sift3d_match_warmup", "sift3d_test_staging_slice", "sift3d_test_slab_plan", "sift3d_test_sharded_time_rank",
]
HOOKS = {"dog_eager": 0, "glast_eager": 1, "det_serial": 2, "separable": 3, "desc_nocache": 4, "match_nodma": 5, "one_stream": 6,
         "desc_mass_shift": 7, "list_cap": 8, "peer_copy": 9, "desc_nosplit": 10, "march_tiles": 11, "desc_exact_cells": 12, "lazy_generic": 13, "sharded_fail_rank": 14}
ORIENT_WORDS = 34
SHARDED_PARTIAL_WINDOWS = 1   # sift3d_sharded_create_ex flags
SHARDED_WHOLE_WINDOWS = 2
SHARDED_COPY_TRANSPORT = 4
SHARDED_GHOST_OCTAVE0 = 8


class Params(C.Structure):
    _fields_ = [("num_kp_levels", C.c_int), ("sigma_default", C.c_float), ("sigma_n_default", C.c_float),
                ("peak_thresh", C.c_float), ("max_eig_thres", C.c_float), ("corner_thresh", C.c_float)]


class SlabDesc(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("z0", C.c_int), ("z1", C.c_int), ("halo", C.c_int),
                ("noct_total", C.c_int), ("octave", C.c_int)]


class Sift3dError(RuntimeError):
    pass


_lib = None
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def lib():
    """Load the HIP library; fails loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Sift3dError(f"{LIB_PATH} is missing: build it with `make -C 3dsift_amd/csrc` "
                              f"(or __graft_entry__.build()); there is no CPU fallback")
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; a process must not end up with two HIP runtimes (the
        # second one finds no GPU).  Loading torch's first makes this library bind to the same runtime by soname, so
        # torch-allocated device memory (halo arenas, synthetic volumes) and this library share one HIP context.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.sift3d_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Params), C.c_int, C.c_int]
        L.sift3d_destroy.argtypes = [C.c_void_p]
        L.sift3d_run.argtypes = [C.c_void_p]
        L.sift3d_run_async.argtypes = [C.c_void_p]
        L.sift3d_wait.argtypes = [C.c_void_p]
        L.sift3d_run_stages.argtypes = [C.c_void_p, C.c_int]
        L.sift3d_stage_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.sift3d_num_keypoints.argtypes = [C.c_void_p, _ip]
        L.sift3d_get_keypoints.argtypes = [C.c_void_p, C.c_void_p, _fp]
        L.sift3d_device_results.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _ip]
        L.sift3d_num_octaves.argtypes = [C.c_void_p, _ip]
        L.sift3d_level_info.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _fp, _fp]
        L.sift3d_copy_level.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp]
        L.sift3d_copy_input.argtypes = [C.c_void_p, _fp]
        L.sift3d_num_extrema.argtypes = [C.c_void_p, _ip]
        L.sift3d_get_extrema.argtypes = [C.c_void_p, C.c_void_p]
        L.sift3d_get_orientation_codes.argtypes = [C.c_void_p, _ip]
        L.sift3d_gaussian_smooth.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp, C.c_int]
        L.sift3d_downsample.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.sift3d_dog_sub.argtypes = [_fp, _fp, C.c_size_t, _fp, C.c_int]
        L.sift3d_conv_axis.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int]
        L.sift3d_orient_keypoint.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, _ip]
        L.sift3d_describe_keypoint.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, _fp, C.c_int]
        L.sift3d_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int,
                                   C.c_int, C.c_int, _ip, _ip, _fp, _fp, _fp, _ip, C.POINTER(C.c_double)]
        L.sift3d_match_handles.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, _ip, _ip, _fp, _fp, _fp, _ip, C.POINTER(C.c_double)]
        L.sift3d_device_count.argtypes = [_ip]
        _sz = C.POINTER(C.c_size_t)
        L.sift3d_slab_min_halo.argtypes = [C.POINTER(Params), _ip]
        L.sift3d_slab_arena_floats.argtypes = [C.POINTER(SlabDesc), C.POINTER(Params), _sz]
        L.sift3d_slab_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(SlabDesc), C.POINTER(Params), C.c_int, C.c_void_p, C.c_size_t]
        L.sift3d_slab_buffer.argtypes = [C.c_void_p, C.c_int, C.c_int, _sz, _ip, _ip]
        L.sift3d_slab_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.sift3d_slab_input_absmax.argtypes = [C.c_void_p, _fp]
        L.sift3d_slab_input_scale.argtypes = [C.c_void_p, C.c_float]
        L.sift3d_slab_level.argtypes = [C.c_void_p, C.c_int]
        L.sift3d_slab_halo_planes.argtypes = [C.c_void_p, C.c_int, _ip]
        L.sift3d_slab_level_hw.argtypes = [C.c_void_p, C.c_int, _ip]
        L.sift3d_slab_sync.argtypes = [C.c_void_p]
        L.sift3d_slab_get_dogmax.argtypes = [C.c_void_p, _fp]
        L.sift3d_slab_set_dogmax.argtypes = [C.c_void_p, _fp]
        L.sift3d_slab_detect.argtypes = [C.c_void_p]
        L.sift3d_slab_describe.argtypes = [C.c_void_p]
        L.sift3d_slab_set_desc_partial.argtypes = [C.c_void_p, C.c_int]
        L.sift3d_slab_min_halo_partial.argtypes = [C.POINTER(Params), _ip]
        L.sift3d_slab_record_bytes.argtypes = [_ip]
        L.sift3d_slab_desc_reach.argtypes = [C.c_void_p, _ip]
        L.sift3d_slab_orient.argtypes = [C.c_void_p]
        L.sift3d_slab_export_records.argtypes = [C.c_void_p, C.c_void_p]
        L.sift3d_slab_describe_partial.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), _ip, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                   C.POINTER(C.c_void_p), _ip, _ip]
        L.sift3d_slab_describe_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p,
                                                  C.c_int, C.c_void_p, C.c_void_p, _ip]
        L.sift3d_slab_orient_launch.argtypes = [C.c_void_p]
        L.sift3d_slab_orient_count.argtypes = [C.c_void_p, _ip]
        L.sift3d_slab_decimate.argtypes = [C.c_void_p, C.c_void_p]
        L.sift3d_slab_decimate_async.argtypes = [C.c_void_p, C.c_void_p]
        L.sift3d_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.sift3d_slab_export_dogmax_device.argtypes = [C.c_void_p, C.c_void_p]
        L.sift3d_slab_import_dogmax_device.argtypes = [C.c_void_p, C.c_void_p]
        L.sift3d_create_seeded.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Params), C.c_int]
        L.sift3d_seed_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.sift3d_set_describe_partition.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.sift3d_export_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.sift3d_run_partial_orientation.argtypes = [C.c_void_p]
        L.sift3d_export_orientation_device.argtypes = [C.c_void_p, C.c_void_p]
        L.sift3d_import_orientation_device.argtypes = [C.c_void_p, C.c_void_p]
        L.sift3d_run_describe.argtypes = [C.c_void_p]
        L.sift3d_import_descriptors_device.argtypes = [C.c_void_p, C.c_void_p]
        L.sift3d_sharded_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Params), _ip, C.c_int, C.c_int, C.c_int]
        L.sift3d_sharded_create_ex.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Params), _ip, C.c_int, C.c_int, C.c_int, C.c_uint]
        L.sift3d_sharded_run.argtypes = [C.c_void_p]
        L.sift3d_sharded_num_keypoints.argtypes = [C.c_void_p, _ip]
        L.sift3d_sharded_get_keypoints.argtypes = [C.c_void_p, C.c_void_p, _fp]
        L.sift3d_sharded_info.argtypes = [C.c_void_p, _ip, _ip, _ip, C.POINTER(C.c_double)]
        L.sift3d_sharded_error.argtypes = [C.c_void_p]
        L.sift3d_sharded_error.restype = C.c_char_p
        L.sift3d_sharded_destroy.argtypes = [C.c_void_p]
        L.sift3d_sharded_plan.argtypes = [C.c_void_p, _ip, _ip, _ip, _ip]
        L.sift3d_test_hook.argtypes = [C.c_int, C.c_int]
        L.sift3d_debug_counters.argtypes = [C.c_void_p, _ip]
        L.sift3d_debug_face_lookup.argtypes = [_fp, C.c_int, C.c_int, _ip, _fp, C.c_int]
        L.sift3d_debug_copy_bandwidth.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.sift3d_match_times.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.sift3d_error_string.argtypes = [C.c_int]
        L.sift3d_error_string.restype = C.c_char_p
        L.sift3d_last_error.restype = C.c_char_p
        _lib = L
        # measurement scripts: S3D_HOOKS="one_stream=1,det_serial=1" sets test hooks of the library from the environment of the
        # PYTHON driver (the library itself never reads the environment)
        for item in filter(None, os.environ.get("S3D_HOOKS", "").split(",")):
            k, _, v = item.partition("=")
            L.sift3d_test_hook(HOOKS[k.strip()], int(v or 1))
    return _lib


def _check(rc):
    if rc != 0:
        L = lib()
        raise Sift3dError(f"{L.sift3d_error_string(rc).decode()}: {L.sift3d_last_error().decode()}")


def _f(a):
    return a.ctypes.data_as(_fp)


def kernel_source_sha():
    """sha256 over the kernel sources (3dsift_amd/csrc/*.hip, *.h, include/*.h): stamps measurements that are taken offline
    (profiles/pyramid_traffic_*.json) with the build they belong to"""
    import glob
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "3dsift_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "3dsift_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(root, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


class hook:
    """with capi.hook("list_cap", 64): ...  -- sets a test hook of the library (include/sift3d_hip.h) and restores it"""

    def __init__(self, name, value):
        self.which, self.value = HOOKS[name], int(value)

    def __enter__(self):
        self.prev = lib().sift3d_test_hook(self.which, self.value)
        return self

    def __exit__(self, *exc):
        lib().sift3d_test_hook(self.which, self.prev)
        return False


def copy_bandwidth(nbytes=1 << 30, iters=5, device=0):
    """GB/s of the library's float4 device copy kernel (read + write)"""
    g = C.c_double(0)
    _check(lib().sift3d_debug_copy_bandwidth(nbytes, iters, device, C.byref(g)))
    return g.value


def face_lookup(grad3, route=0, device=0):
    """Check_intersect_faces of k_describe on [n, 3] gradients: (face [n] int32, bary [n, 3]); route 0 predicted+verified, 1 literal scan"""
    g = np.ascontiguousarray(grad3, np.float32).reshape(-1, 3)
    face = np.zeros(len(g), np.int32); bary = np.zeros((len(g), 3), np.float32)
    _check(lib().sift3d_debug_face_lookup(_f(g), len(g), int(route), face.ctypes.data_as(_ip), _f(bary), device))
    return face, bary


def device_count():
    n = C.c_int(0)
    lib().sift3d_device_count(C.byref(n))
    return n.value


class CSIFT3D:
    """Python mirror of CPUSIFT::CSIFT3D over the C-ABI (same method names as the reference class,
    Include/cSIFT3D.h:118-181).  vol is [z, y, x] fp32 (x fastest)."""

    def __init__(self, volume, num_kp_levels=3, sigma_default=1.6, sigma_n_default=1.15, peak_thresh=0.1,
                 max_eig_thres=0.9, corner_thresh=0.4, device=0, device_ptr=None, shape=None):
        L = lib()
        self._h = C.c_void_p()
        p = Params(num_kp_levels, sigma_default, sigma_n_default, peak_thresh, max_eig_thres, corner_thresh)
        self.levels = num_kp_levels
        if device_ptr is not None:
            nz, ny, nx = shape
            self.shape = tuple(shape)
            _check(L.sift3d_create(C.byref(self._h), C.c_void_p(device_ptr), nx, ny, nz, C.byref(p), device, 1))
        else:
            vol = np.ascontiguousarray(volume, dtype=np.float32)
            nz, ny, nx = vol.shape
            self.shape = vol.shape
            _check(L.sift3d_create(C.byref(self._h), vol.ctypes.data_as(C.c_void_p), nx, ny, nz, C.byref(p), device, 0))

    def set_stream(self, stream_handle):
        """every later call enqueues on this hipStream_t (e.g. torch.cuda.Stream().cuda_stream); 0 / None = own stream"""
        _check(lib().sift3d_set_stream(self._h, C.c_void_p(int(stream_handle or 0))))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sift3d_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- reference API names -------------------------------------------------------------------
    def KpSiftAlgorithm(self):
        _check(lib().sift3d_run(self._h))
        return self

    def KpSiftAlgorithmAsync(self, after=None):
        """enqueue the whole pipeline and return (sift3d_run_async); Wait() -- or any accessor -- completes it.  after = another extractor
        with a run in flight on the same GPU: this pipeline starts when that one's orientation stage has ended (sift3d_run_async_after)"""
        if after is not None:
            _check(lib().sift3d_run_async_after(self._h, after._h))
        else:
            _check(lib().sift3d_run_async(self._h))
        return self

    def Wait(self):
        _check(lib().sift3d_wait(self._h))
        return self

    def run_stages(self, upto):
        _check(lib().sift3d_run_stages(self._h, int(upto)))
        return self

    def GetKeypoints(self, with_desc=True, out=None):
        """out = (kp, desc): arrays of the result's size to fill instead of new ones (the C++ caller's vectors)"""
        n = C.c_int(0)
        _check(lib().sift3d_num_keypoints(self._h, C.byref(n)))
        if out is not None:
            kp, desc = out
            assert kp.dtype == KP_DTYPE and kp.shape == (n.value,) and desc.dtype == np.float32 and desc.shape == (n.value, DESC)
            assert kp.flags.c_contiguous and desc.flags.c_contiguous
        else:
            kp = np.zeros(n.value, KP_DTYPE)
            desc = np.zeros((n.value, DESC), np.float32)
        if n.value:
            _check(lib().sift3d_get_keypoints(self._h, kp.ctypes.data, _f(desc) if with_desc else None))
        return kp, desc

    @property
    def m_timer(self):
        t = (C.c_double * 8)()
        _check(lib().sift3d_stage_times(self._h, t))
        keys = ["d_TotalTime", "d_Allocation", "d_BuildGSS", "d_BuildDOG", "d_Detect", "d_AssignOrientation", "d_Extraction", "d_release"]
        return dict(zip(keys, list(t)))

    # --- checking accessors (GET_GSS / GET_DOG / GET_LEVEL) ---------------------------------------
    @property
    def num_octaves(self):
        n = C.c_int(0)
        _check(lib().sift3d_num_octaves(self._h, C.byref(n)))
        return n.value

    def level_info(self, is_dog, idx):
        d = np.zeros(3, np.int32); u = np.zeros(3, np.float32); s = np.zeros(1, np.float32)
        _check(lib().sift3d_level_info(self._h, int(is_dog), idx, d.ctypes.data_as(_ip), _f(u), _f(s)))
        return tuple(int(v) for v in d), tuple(float(v) for v in u), float(s[0])

    def level(self, is_dog, idx):
        (nx, ny, nz), _, _ = self.level_info(is_dog, idx)
        out = np.empty((nz, ny, nx), np.float32)
        _check(lib().sift3d_copy_level(self._h, int(is_dog), idx, _f(out)))
        return out

    def gss(self, octave, i):
        return self.level(0, octave * (self.levels + 3) + i)

    def dog(self, octave, i):
        return self.level(1, octave * (self.levels + 2) + i)

    def input(self):
        out = np.empty(self.shape, np.float32)
        _check(lib().sift3d_copy_input(self._h, _f(out)))
        return out

    def extrema(self):
        n = C.c_int(0)
        _check(lib().sift3d_num_extrema(self._h, C.byref(n)))
        out = np.zeros(n.value, KP_DTYPE)
        if n.value:
            _check(lib().sift3d_get_extrema(self._h, out.ctypes.data))
        return out

    def orientation_codes(self):
        n = C.c_int(0)
        _check(lib().sift3d_num_extrema(self._h, C.byref(n)))
        out = np.zeros(n.value, np.int32)
        if n.value:
            _check(lib().sift3d_get_orientation_codes(self._h, out.ctypes.data_as(_ip)))
        return out

    def device_results(self):
        d = C.c_void_p(); x = C.c_void_p(); n = C.c_int(0)
        _check(lib().sift3d_device_results(self._h, C.byref(d), C.byref(x), C.byref(n)))
        return d.value, x.value, n.value

    def export_device(self, desc_ptr, xyz_ptr=None):
        """D2D copy of the results into caller-owned device buffers (n*768, n*3 floats)"""
        _check(lib().sift3d_export_device(self._h, C.c_void_p(int(desc_ptr)), C.c_void_p(int(xyz_ptr)) if xyz_ptr else None))

    def debug_counters(self):
        """{list_regrows, desc_second_passes, match_exact_rows}: how often the rare paths ran (sift3d_debug_counters)"""
        a = (C.c_int * 4)()
        _check(lib().sift3d_debug_counters(self._h, a))
        return {"list_regrows": a[0], "desc_second_passes": a[1], "match_exact_rows": a[2]}


def _params(kw):
    return Params(kw.get("num_kp_levels", 3), kw.get("sigma_default", 1.6), kw.get("sigma_n_default", 1.15),
                  kw.get("peak_thresh", 0.1), kw.get("max_eig_thres", 0.9), kw.get("corner_thresh", 0.4))


def slab_min_halo(**kw):
    p = _params(kw); h = C.c_int(0)
    _check(lib().sift3d_slab_min_halo(C.byref(p), C.byref(h)))
    return h.value


def slab_admits(nx, ny, nz, first_octave=True, **kw):
    """can slab contexts hold an octave of these GLOBAL dims with these parameters (sift3d_slab_admits)?"""
    p = _params(kw); ok = C.c_int(0)
    lib().sift3d_slab_admits.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    _check(lib().sift3d_slab_admits(C.byref(p), int(nx), int(ny), int(nz), int(bool(first_octave)), C.byref(ok)))
    return bool(ok.value)


def slab_min_halo_partial(**kw):
    """planes per side a slab's level buffers need when the descriptor windows are split along z over the ranks (r05)"""
    p = _params(kw); h = C.c_int(0)
    _check(lib().sift3d_slab_min_halo_partial(C.byref(p), C.byref(h)))
    return h.value


def slab_record_words():
    """int32 words of one keypoint record of the partial-window exchange (opaque to the driver)"""
    n = C.c_int(0)
    _check(lib().sift3d_slab_record_bytes(C.byref(n)))
    assert n.value % 4 == 0
    return n.value // 4


class SeededCSIFT3D(CSIFT3D):
    """Octaves octave_base.. of a volume whose G[octave_base][0] the caller provides (multi-GPU tail, see slab.py).
    shape = (nz, ny, nx) of that level."""

    def __init__(self, shape, octave_base, noct_total, device=0, **kw):
        self._h = C.c_void_p()
        p = _params(kw)
        self.levels = p.num_kp_levels
        self.shape = tuple(shape)
        nz, ny, nx = shape
        _check(lib().sift3d_create_seeded(C.byref(self._h), nx, ny, nz, octave_base, noct_total, C.byref(p), device))

    def seed(self, ptr, on_device=True):
        _check(lib().sift3d_seed_upload(self._h, C.c_void_p(int(ptr)), int(bool(on_device))))

    def seed_host(self, vol):
        vol = np.ascontiguousarray(vol, np.float32)
        _check(lib().sift3d_seed_upload(self._h, vol.ctypes.data_as(C.c_void_p), 0))

    def set_partition(self, rank, world):
        _check(lib().sift3d_set_describe_partition(self._h, rank, world))

    def run_partial_orientation(self):
        _check(lib().sift3d_run_partial_orientation(self._h))

    def export_orientation(self, ptr):
        _check(lib().sift3d_export_orientation_device(self._h, C.c_void_p(int(ptr))))

    def import_orientation(self, ptr):
        _check(lib().sift3d_import_orientation_device(self._h, C.c_void_p(int(ptr))))

    def run_describe(self):
        _check(lib().sift3d_run_describe(self._h))

    def num_extrema(self):
        n = C.c_int(0)
        _check(lib().sift3d_num_extrema(self._h, C.byref(n)))
        return n.value

    def import_descriptors(self, desc_ptr):
        _check(lib().sift3d_import_descriptors_device(self._h, C.c_void_p(int(desc_ptr))))


class SlabCSIFT3D(CSIFT3D):
    """One z-slab of octave 0 (include/sift3d_hip.h, multi-GPU section).  The level buffers live in `arena_ptr`
    (caller-owned device memory of arena_floats(...) floats) so the caller's communication layer can address halo planes."""

    @staticmethod
    def arena_floats(nx, ny, nz, z0, z1, halo, noct_total, octave=0, **kw):
        d = SlabDesc(nx, ny, nz, z0, z1, halo, noct_total, octave); p = _params(kw); n = C.c_size_t(0)
        _check(lib().sift3d_slab_arena_floats(C.byref(d), C.byref(p), C.byref(n)))
        return n.value

    def __init__(self, nx, ny, nz, z0, z1, halo, noct_total, arena_ptr, arena_floats, device=0, octave=0, **kw):
        self._h = C.c_void_p()
        p = _params(kw)
        self.levels = p.num_kp_levels
        self.dims = (nx, ny, nz)
        self.z0, self.z1, self.halo = z0, z1, halo
        self.shape = (z1 - z0 + 2 * halo, ny, nx)
        self.octave = octave
        d = SlabDesc(nx, ny, nz, z0, z1, halo, noct_total, octave)
        _check(lib().sift3d_slab_create(C.byref(self._h), C.byref(d), C.byref(p), device, C.c_void_p(int(arena_ptr)), arena_floats))

    def buffer(self, kind, idx=0):
        """(offset in floats inside the arena, planes held, global z of plane 0); kind 0 input, 1 GSS, 2 DoG"""
        off = C.c_size_t(0); pl = C.c_int(0); zo = C.c_int(0)
        _check(lib().sift3d_slab_buffer(self._h, kind, idx, C.byref(off), C.byref(pl), C.byref(zo)))
        return off.value, pl.value, zo.value

    def held_level(self, is_dog, idx):
        """all planes the buffer holds, [planes, ny, nx]; plane k is global plane (z0 - halo) + k"""
        out = np.empty(self.shape, np.float32)
        _check(lib().sift3d_copy_level(self._h, int(is_dog), idx, _f(out)))
        return out

    def upload(self, planes, zg0, zg1, device_ptr=None):
        if device_ptr is not None:
            _check(lib().sift3d_slab_upload(self._h, C.c_void_p(int(device_ptr)), zg0, zg1, 1))
        else:
            a = np.ascontiguousarray(planes, np.float32)
            assert a.shape == (zg1 - zg0, self.dims[1], self.dims[0])
            _check(lib().sift3d_slab_upload(self._h, a.ctypes.data_as(C.c_void_p), zg0, zg1, 0))

    def input_absmax(self):
        m = C.c_float(0)
        _check(lib().sift3d_slab_input_absmax(self._h, C.byref(m)))
        return m.value

    def input_scale(self, gmax):
        _check(lib().sift3d_slab_input_scale(self._h, C.c_float(gmax)))

    def level_async(self, i):
        _check(lib().sift3d_slab_level(self._h, i))

    def level_hw(self, i):
        n = C.c_int(0)
        _check(lib().sift3d_slab_level_hw(self._h, i, C.byref(n)))
        return n.value

    def halo_planes(self, i):
        n = C.c_int(0)
        _check(lib().sift3d_slab_halo_planes(self._h, i, C.byref(n)))
        return n.value

    def sync(self):
        _check(lib().sift3d_slab_sync(self._h))

    def get_dogmax(self):
        a = np.zeros(8, np.float32)
        _check(lib().sift3d_slab_get_dogmax(self._h, _f(a)))
        return a[: self.levels + 2].copy()

    def set_dogmax(self, a):
        b = np.zeros(8, np.float32); b[: self.levels + 2] = a
        _check(lib().sift3d_slab_set_dogmax(self._h, _f(b)))

    def detect(self):
        _check(lib().sift3d_slab_detect(self._h))

    def describe(self):
        _check(lib().sift3d_slab_describe(self._h))

    # ---- r05: descriptor windows split along z over the ranks (include/sift3d_hip.h) ----
    def set_desc_partial(self, on=True):
        _check(lib().sift3d_slab_set_desc_partial(self._h, int(bool(on))))

    def desc_reach(self):
        n = C.c_int(0)
        _check(lib().sift3d_slab_desc_reach(self._h, C.byref(n)))
        return n.value

    def orient(self):
        """orientation of the owned extrema -> number of accepted keypoints"""
        _check(lib().sift3d_slab_orient(self._h))
        n = C.c_int(0)
        _check(lib().sift3d_num_keypoints(self._h, C.byref(n)))
        return n.value

    def export_records(self, dst_ptr):
        _check(lib().sift3d_slab_export_records(self._h, C.c_void_p(int(dst_ptr))))

    def describe_partial(self, lists):
        """lists: [(records ptr, n, units ptr or None, hist ptr, mass ptr, owner z0, owner z1), ...] -- one launch (include/sift3d_hip.h)"""
        k = len(lists)
        if not k:
            return
        vp = C.c_void_p * k
        ia = C.c_int * k
        recs = vp(*[C.c_void_p(int(t[0])) if t[1] else None for t in lists])
        ns = ia(*[int(t[1]) for t in lists])
        units = vp(*[C.c_void_p(int(t[2])) if t[2] else None for t in lists])
        hist = vp(*[C.c_void_p(int(t[3])) if t[1] else None for t in lists])
        mass = vp(*[C.c_void_p(int(t[4])) if t[1] else None for t in lists])
        z0 = ia(*[int(t[5]) for t in lists]); z1 = ia(*[int(t[6]) for t in lists])
        _check(lib().sift3d_slab_describe_partial(self._h, k, recs, ns, units, hist, mass, z0, z1))

    def describe_finish(self, rec_ptr, n, parts, units_ptr=None, final_round=False, redo_ptr=None, units_next_ptr=None):
        """parts: [(hist ptr, mass ptr), ...] of the n records, ascending rank -> records flagged for the second round"""
        k = C.c_int(0)
        vp = C.c_void_p * max(1, len(parts))
        hs = vp(*[C.c_void_p(int(h)) for h, _ in parts]) if parts else vp()
        ms = vp(*[C.c_void_p(int(m)) for _, m in parts]) if parts else vp()
        _check(lib().sift3d_slab_describe_finish(self._h, C.c_void_p(int(rec_ptr)) if n else None, int(n), len(parts), hs, ms,
                                                 C.c_void_p(int(units_ptr)) if units_ptr else None, int(bool(final_round)),
                                                 C.c_void_p(int(redo_ptr)) if redo_ptr else None,
                                                 C.c_void_p(int(units_next_ptr)) if units_next_ptr else None, C.byref(k)))
        return k.value

    def orient_launch(self):
        _check(lib().sift3d_slab_orient_launch(self._h))

    def orient_count(self):
        n = C.c_int(0)
        _check(lib().sift3d_slab_orient_count(self._h, C.byref(n)))
        return n.value

    def decimate(self, dst_ptr, wait=True):
        fn = lib().sift3d_slab_decimate if wait else lib().sift3d_slab_decimate_async
        _check(fn(self._h, C.c_void_p(int(dst_ptr))))

    def export_dogmax(self, dst_ptr):
        _check(lib().sift3d_slab_export_dogmax_device(self._h, C.c_void_p(int(dst_ptr))))

    def import_dogmax(self, src_ptr):
        _check(lib().sift3d_slab_import_dogmax_device(self._h, C.c_void_p(int(src_ptr))))


class ShardedCSIFT3D:
    """One host volume [z, y, x] sharded as z-slabs by the library's native driver (csrc/sharded.hip): `devices` = one rank per GPU
    over RCCL, or sim_ranks = n ranks simulated on devices[0].  partial_windows: descriptor windows split along z over the ranks
    (sift3d_sharded_create_ex, SIFT3D_SHARDED_PARTIAL_WINDOWS) instead of whole windows on wide halos."""

    def __init__(self, volume, devices=(0,), sim_ranks=0, sharded_octaves=0, partial_windows=None, transport="rccl", ghost_octave0=False, **kw):
        """partial_windows: None = the driver's rule (descriptor windows split along z unless a slab is too thin for that), True = split or
        refuse, False = whole windows on the wide halos.  transport: "rccl" (one rank per device) or "copies" (SIFT3D_SHARDED_COPY_TRANSPORT:
        event + device / peer copies; `devices` may repeat a device -- rank threads sharing one GPU)"""
        vol = np.ascontiguousarray(volume, np.float32)
        assert vol.ndim == 3
        nz, ny, nx = vol.shape
        self._h = C.c_void_p()
        p = _params(kw)
        devs = (C.c_int * len(devices))(*devices)
        flags = 0 if partial_windows is None else (SHARDED_PARTIAL_WINDOWS if partial_windows else SHARDED_WHOLE_WINDOWS)
        assert transport in ("rccl", "copies")
        if transport == "copies":
            flags |= SHARDED_COPY_TRANSPORT
        if ghost_octave0:   # SIFT3D_SHARDED_GHOST_OCTAVE0: octave 0 recomputed on ghost planes instead of exchanged level by level
            flags |= SHARDED_GHOST_OCTAVE0
        _check(lib().sift3d_sharded_create_ex(C.byref(self._h), vol.ctypes.data_as(C.c_void_p), nx, ny, nz, C.byref(p), devs, len(devices),
                                              int(sim_ranks), int(sharded_octaves), flags))

    def KpSiftAlgorithm(self):
        rc = lib().sift3d_sharded_run(self._h)
        if rc:
            raise Sift3dError(f"{lib().sift3d_error_string(rc).decode()}: {lib().sift3d_sharded_error(self._h).decode()}")
        return self

    def GetKeypoints(self):
        n = C.c_int(0)
        _check(lib().sift3d_sharded_num_keypoints(self._h, C.byref(n)))
        kp = np.zeros(n.value, KP_DTYPE); desc = np.zeros((n.value, DESC), np.float32)
        if n.value:
            _check(lib().sift3d_sharded_get_keypoints(self._h, kp.ctypes.data, _f(desc)))
        return kp, desc

    def info(self):
        w = C.c_int(0); s = C.c_int(0); h = C.c_int(0); t = (C.c_double * 2)()
        _check(lib().sift3d_sharded_info(self._h, C.byref(w), C.byref(s), C.byref(h), t))
        pw = C.c_int(0); tr = C.c_int(0); pl = (C.c_int * max(1, w.value))(); sp = (C.c_int * max(1, s.value))()
        _check(lib().sift3d_sharded_plan(self._h, C.byref(pw), C.byref(tr), pl, sp))
        return {"world": w.value, "sharded_octaves": s.value, "halo": h.value, "seconds": t[0], "seconds_incl_merge": t[1],
                "partial_windows": bool(pw.value), "stage_partial": [bool(v) for v in sp][:s.value], "tail_rank": tr.value, "planes": [int(v) for v in pl][:w.value]}

    def traffic(self):
        """bytes every rank receives per step: (plane halos, records + partial histograms of the last run)"""
        w = self.info()["world"]
        a = (C.c_double * w)(); b = (C.c_double * w)()
        lib().sift3d_sharded_traffic.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _check(lib().sift3d_sharded_traffic(self._h, a, b))
        return [float(v) for v in a], [float(v) for v in b]

    def time_rank(self, rank):
        """simulated ranks, after a run: the GPU time (s) of one rank's whole step, re-run alone (sift3d_test_sharded_time_rank)"""
        t = C.c_double(0)
        lib().sift3d_test_sharded_time_rank.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        rc = lib().sift3d_test_sharded_time_rank(self._h, int(rank), C.byref(t))
        if rc:
            raise Sift3dError(f"{lib().sift3d_error_string(rc).decode()}: {lib().sift3d_last_error().decode()}")
        return t.value

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sift3d_sharded_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def CreateCSIFT3D(volume, **kw):
    """CSIFT3DFactory::CreateCSIFT3D (Include/cSIFT3D.h:184-194)."""
    return CSIFT3D(volume, **kw)


def gaussian_smooth(vol, sigma, device=0):
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    nz, ny, nx = vol.shape
    out = np.empty_like(vol)
    _check(lib().sift3d_gaussian_smooth(_f(vol), nx, ny, nz, float(sigma), _f(out), device))
    return out


def downsample(vol, out_shape=None, device=0):
    """DownSample_3D (Include/cSIFT3D.h:210): out(k, m, n) = vol(2k, 2m, 2n); out_shape defaults to the halves (rounded down)."""
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    snz, sny, snx = vol.shape
    nz, ny, nx = out_shape if out_shape is not None else (snz // 2, sny // 2, snx // 2)
    out = np.empty((nz, ny, nx), np.float32)
    _check(lib().sift3d_downsample(_f(vol), snx, sny, snz, _f(out), nx, ny, nz, device))
    return out


def dog_sub(prev, cur, device=0):
    """Sub (Include/cSIFT3D.h:218): (cur - prev) * (-1)."""
    prev = np.ascontiguousarray(prev, dtype=np.float32); cur = np.ascontiguousarray(cur, dtype=np.float32)
    assert prev.shape == cur.shape
    out = np.empty_like(prev)
    _check(lib().sift3d_dog_sub(_f(prev), _f(cur), prev.size, _f(out), device))
    return out


def conv_axis(vol, dim, weight, device=0):
    """GaussianSmooth_3D_Imp (Include/cSIFT3D.h:214): one pass along dim (0 x, 1 y, 2 z) with the caller's taps."""
    vol = np.ascontiguousarray(vol, dtype=np.float32); w = np.ascontiguousarray(weight, dtype=np.float32)
    nz, ny, nx = vol.shape
    out = np.empty_like(vol)
    _check(lib().sift3d_conv_axis(_f(vol), nx, ny, nz, int(dim), _f(w), int(w.size), _f(out), device))
    return out


def orient_keypoint(level, unit, kp, sigma, max_eig_ratio=0.9, corner_thresh=0.4, device=0):
    """Assign_Orientation_Imp (Include/cSIFT3D.h:224) for ONE keypoint record (a KP_DTYPE scalar array of shape (1,), updated in place)
    on a host level [z, y, x]; returns the reference's code."""
    level = np.ascontiguousarray(level, dtype=np.float32)
    nz, ny, nx = level.shape
    assert kp.dtype == KP_DTYPE and kp.shape == (1,) and kp.flags.c_contiguous
    code = C.c_int(0)
    _check(lib().sift3d_orient_keypoint(_f(level), nx, ny, nz, float(unit), kp.ctypes.data_as(C.c_void_p), float(sigma), float(max_eig_ratio),
                                        float(corner_thresh), device, C.byref(code)))
    return code.value


def describe_keypoint(level, unit, kp, device=0):
    """Extract_Descriptor_Imp (Include/cSIFT3D.h:228) for ONE keypoint record (shape (1,), Rotation transposed in place); returns desc[768]."""
    level = np.ascontiguousarray(level, dtype=np.float32)
    nz, ny, nx = level.shape
    assert kp.dtype == KP_DTYPE and kp.shape == (1,) and kp.flags.c_contiguous
    desc = np.zeros(DESC, np.float32)
    _check(lib().sift3d_describe_keypoint(_f(level), nx, ny, nz, float(unit), kp.ctypes.data_as(C.c_void_p), _f(desc), device))
    return desc


class muBruteMatcher:
    """Python mirror of CPUSIFT::muBruteMatcher (Include/cMatcher.h:12-88)."""

    MODES = {"inject": 1, "biject": 2, "enhanced": 3}

    def __init__(self, device=0):
        self.device = device
        self.totalTime = 0.0   # device time of the last call (HIP events on the matcher's stream)
        self.wallTime = 0.0    # host clock around the last call
        self.exact_rows = 0    # rows the near-tie guard re-scored exactly in the last call
        self._last = None
        if device_count() > 0:
            lib().sift3d_match_warmup(int(device))   # (like the C++ shell's constructor; failures surface in the first call)

    def _match(self, ref_desc, ref_xyz, tar_desc, tar_xyz, thresHold, mode, on_device=False, n=None, m=None):
        L = lib()
        if on_device:
            pa, px, pb, py = (C.c_void_p(int(v)) for v in (ref_desc, ref_xyz, tar_desc, tar_xyz))
        else:
            a = np.ascontiguousarray(ref_desc, np.float32); b = np.ascontiguousarray(tar_desc, np.float32)
            ax = np.ascontiguousarray(ref_xyz, np.float32); bx = np.ascontiguousarray(tar_xyz, np.float32)
            n, m = a.shape[0], b.shape[0]
            pa, px, pb, py = (v.ctypes.data_as(C.c_void_p) for v in (a, ax, b, bx))
        nn = max(n, 1)
        gi = np.zeros(nn, np.int32); si = np.zeros(nn, np.int32)
        gd = np.zeros(nn, np.float32); sd = np.zeros(nn, np.float32)
        pairs = np.zeros((nn, 6), np.float32)
        k = C.c_int(0); sec = C.c_double(0)
        _check(L.sift3d_match(pa, px, n, pb, py, m, float(thresHold), int(mode), int(bool(on_device)), self.device,
                              gi.ctypes.data_as(_ip), si.ctypes.data_as(_ip), _f(gd), _f(sd), _f(pairs), C.byref(k), C.byref(sec)))
        self.totalTime = sec.value
        dv = C.c_double(0); wl = C.c_double(0)
        L.sift3d_match_times(C.byref(dv), C.byref(wl))
        self.wallTime = wl.value
        a = (C.c_int * 4)()
        L.sift3d_debug_counters(None, a)
        self.exact_rows = a[2]
        self._last = dict(gIdx=gi[:n], sIdx=si[:n], gDist=gd[:n], sDist=sd[:n], pairs=pairs[:k.value].copy())
        return self._last

    def matchExtractors(self, ref, tar, thresHold=0.85, mode="enhanced"):
        """sift3d_match_handles: the device-resident results of two CSIFT3D objects, wherever they live (peer copy across GPUs)"""
        n = max(len(ref.GetKeypoints(with_desc=False)[0]), 1)
        gi = np.zeros(n, np.int32); si = np.zeros(n, np.int32); gd = np.zeros(n, np.float32); sd = np.zeros(n, np.float32)
        pairs = np.zeros((n, 6), np.float32); k = C.c_int(0); sec = C.c_double(0)
        _check(lib().sift3d_match_handles(ref._h, tar._h, float(thresHold), self.MODES[mode], gi.ctypes.data_as(_ip), si.ctypes.data_as(_ip),
                                          _f(gd), _f(sd), _f(pairs), C.byref(k), C.byref(sec)))
        self.totalTime = sec.value
        nk = len(ref.GetKeypoints(with_desc=False)[0])
        self._last = dict(gIdx=gi[:nk], sIdx=si[:nk], gDist=gd[:nk], sDist=sd[:nk], pairs=pairs[:k.value].copy())
        return self._last

    def injectMatch(self, ref_desc, ref_xyz, tar_desc, tar_xyz, thresHold=0.85, **kw):
        return self._match(ref_desc, ref_xyz, tar_desc, tar_xyz, thresHold, 1, **kw)

    def bijectMatch(self, ref_desc, ref_xyz, tar_desc, tar_xyz, thresHold=0.85, **kw):
        return self._match(ref_desc, ref_xyz, tar_desc, tar_xyz, thresHold, 2, **kw)

    def enhancedMatch(self, ref_desc, ref_xyz, tar_desc, tar_xyz, thresHold=0.85, **kw):
        return self._match(ref_desc, ref_xyz, tar_desc, tar_xyz, thresHold, 3, **kw)

    def getCalculationTime(self):
        return self.totalTime

    def getGlodenIdx(self):
        return self._last["gIdx"]

    def getSilverIdx(self):
        return self._last["sIdx"]

    def getGlodenDistSquare(self):
        return self._last["gDist"]

    def getSilverDistSquare(self):
        return self._last["sDist"]
