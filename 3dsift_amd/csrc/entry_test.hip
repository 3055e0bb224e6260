// entry_test.hip -- test hooks, development switches, rare-path counters and the unit-level debug entry points
// (include/sift3d_hip_test.h; nothing here is part of the drop-in boundary).
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>

#include "ctx_internal.h"

#pragma clang fp contract(off)

namespace s3d {

// test hooks (include/sift3d_hip.h) and development switches (sift3d_internal.h)
static int g_hooks[SIFT3D_HOOK_COUNT] = {0};
int hook(int which) { return which >= 0 && which < SIFT3D_HOOK_COUNT ? g_hooks[which] : 0; }
#ifdef S3D_DEV_SWITCHES
// measurement builds only (scripts/build_variant.sh): the old environment switches
int dev_tune_i(const char *env_name, int dflt) { const char *e = getenv(env_name); return e ? atoi(e) : dflt; }
double dev_tune_d(const char *env_name, double dflt) { const char *e = getenv(env_name); return e ? atof(e) : dflt; }
static const bool g_env_hooks = [] {
	static const char *names[SIFT3D_HOOK_COUNT] = {"S3D_DOG_EAGER", "S3D_GLAST_EAGER", "S3D_DET_SERIAL", "S3D_SEPARABLE", "S3D_DESC_NOCACHE",
	                                               "S3D_MATCH_NODMA", "S3D_ONE_STREAM", "S3D_DESC_MASS_SHIFT", "S3D_LIST_CAP", "S3D_PEER_COPY", "S3D_DESC_NOSPLIT",
	                                               "S3D_MARCH_TILES", "S3D_DESC_EXACT_CELLS", "S3D_LAZY_GENERIC", "S3D_SHARDED_FAIL_RANK"};
	static_assert(sizeof(names) / sizeof(names[0]) == SIFT3D_HOOK_COUNT, "one environment name per hook");
	for (int i = 0; i < SIFT3D_HOOK_COUNT; i++) { const char *e = names[i] ? getenv(names[i]) : nullptr; if (e) g_hooks[i] = atoi(e); }
	return true;
}();
#else
int dev_tune_i(const char *, int dflt) { return dflt; }
double dev_tune_d(const char *, double dflt) { return dflt; }
#endif

}  // namespace s3d

using namespace s3d;

extern "C" int sift3d_test_hook(int which, int value) {
	if (which < 0 || which >= SIFT3D_HOOK_COUNT) return -1;
	const int prev = g_hooks[which];
	g_hooks[which] = value;
	return prev;
}

// best-of-iters bandwidth (GB/s, read + write) of a float4 device-to-device copy of `bytes` bytes: the achievable HBM ceiling
// bench.py reports beside the 8 TB/s spec peak
extern "C" int sift3d_debug_copy_bandwidth(size_t bytes, int iters, int device, double *gbs) {
	if (!gbs || bytes < 4096 || iters < 1) return SIFT3D_ERR_ARG;
	int rc = set_device(device);
	if (rc) return rc;
	const size_t nf = (bytes / 16) * 4;
	float *d = nullptr;
	S3D_HIP(hipMalloc(&d, sizeof(float) * nf * 2));
	hipEvent_t e0 = nullptr, e1 = nullptr;
	hipError_t e = hipMemset(d, 0, sizeof(float) * nf * 2);
	if (e == hipSuccess) e = hipEventCreate(&e0);
	if (e == hipSuccess) e = hipEventCreate(&e1);
	double best = 0.0;
	for (int i = 0; i < iters + 1 && e == hipSuccess; i++) {  // first pass warms up
		(void)hipEventRecord(e0, nullptr);
		launch_copy16(d, d + nf, nf, nullptr);
		(void)hipEventRecord(e1, nullptr);
		e = hipEventSynchronize(e1);
		float ms = 0;
		if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
		if (i > 0 && ms > 0) best = std::max(best, 2.0 * (double)nf * 4.0 / ((double)ms * 1e-3) / 1e9);
	}
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	(void)hipFree(d);
	if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return SIFT3D_ERR_HIP; }
	*gbs = best;
	return SIFT3D_OK;
}

extern "C" int sift3d_debug_counters(sift3d_handle c, int out[4]) {
	if (!out) return SIFT3D_ERR_ARG;
	if (c && c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	out[0] = c ? c->n_regrow : 0;
	out[1] = c ? c->n_desc_redo : 0;
	out[2] = match_redo_rows();
	out[3] = 0;
	return SIFT3D_OK;
}

extern "C" int sift3d_debug_face_lookup(const float *grad3, int n, int route, int *face, float *bary3, int device) {
	if (!grad3 || !face || !bary3 || n < 0) return SIFT3D_ERR_ARG;
	int rc = set_device(device);
	if (rc) return rc;
	if (n == 0) return SIFT3D_OK;
	FaceConst faces[kFaces];
	build_faces(faces);
	FaceSym sym;
	if (!build_facesym(faces, &sym)) { set_last_error("icosahedron symmetry table: no matching face"); return SIFT3D_ERR_STATE; }
	S3D_HIP(upload_faces(faces, &sym));
	float *d = nullptr;
	S3D_HIP(hipMalloc(&d, sizeof(float) * (size_t)n * 7));
	int *d_face = reinterpret_cast<int *>(d + (size_t)n * 6);
	hipError_t e = hipMemcpy(d, grad3, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		launch_face_lookup(d, n, route, d_face, d + (size_t)n * 3, nullptr);
		e = hipDeviceSynchronize();
	}
	if (e == hipSuccess) e = hipMemcpy(bary3, d + (size_t)n * 3, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost);
	if (e == hipSuccess) e = hipMemcpy(face, d_face, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return SIFT3D_ERR_HIP; }
	return SIFT3D_OK;
}

