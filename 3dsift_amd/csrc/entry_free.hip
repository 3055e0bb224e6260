// entry_free.hip -- the reference's free functions (Include/cSIFT3D.h:208-239) on caller-held host data, behind the C-ABI: unit-level parity.
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

using namespace s3d;

extern "C" int sift3d_gaussian_smooth(const float *src, int nx, int ny, int nz, float sigma, float *dst, int device) {
	if (!src || !dst || nx <= 0 || ny <= 0 || nz <= 0) return SIFT3D_ERR_ARG;
	int rc = set_device(device);
	if (rc) return rc;
	Taps t;
	if (!build_taps(sigma, t)) { set_last_error("kernel too wide"); return SIFT3D_ERR_ARG; }
	const size_t n = (size_t)nx * ny * nz;
	float *d = nullptr;
	S3D_HIP(hipMalloc(&d, sizeof(float) * n * 3));
	hipError_t e = hipMemcpy(d, src, sizeof(float) * n, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		launch_conv_axis(0, d, d + n, nx, ny, nz, t, nullptr, nullptr, nullptr, nullptr);
		launch_conv_axis(1, d + n, d + 2 * n, nx, ny, nz, t, nullptr, nullptr, nullptr, nullptr);
		launch_conv_axis(2, d + 2 * n, d, nx, ny, nz, t, nullptr, nullptr, nullptr, nullptr);
		e = hipDeviceSynchronize();
	}
	if (e == hipSuccess) e = hipMemcpy(dst, d, sizeof(float) * n, hipMemcpyDeviceToHost);
	hipFree(d);
	if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return SIFT3D_ERR_HIP; }
	return SIFT3D_OK;
}


// GaussianSmooth_3D_Imp (Src/cSIFT3D.cc:624-788): ONE pass along `dim` with the caller's taps (width odd: the reference reads
// weight[0 .. 2 (width / 2)]), interior and boundary rule of the pipeline's generic pass (k_conv_axis)
extern "C" int sift3d_conv_axis(const float *src, int nx, int ny, int nz, int dim, const float *weight, int width, float *dst, int device) {
	if (!src || !dst || !weight || nx <= 0 || ny <= 0 || nz <= 0 || dim < 0 || dim > 2) return SIFT3D_ERR_ARG;
	if (width < 1 || !(width & 1) || width / 2 > kMaxHW) { set_last_error("sift3d_conv_axis: the kernel width must be odd and at most 2 * 64 + 1"); return SIFT3D_ERR_ARG; }
	int rc = set_device(device);
	if (rc) return rc;
	Taps t;
	t.hw = width / 2;
	for (int i = 0; i < kMaxTaps; i++) t.w[i] = i < width ? weight[i] : 0.f;
	const size_t n = (size_t)nx * ny * nz;
	float *d = nullptr;
	S3D_HIP(hipMalloc(&d, sizeof(float) * n * 2));
	hipError_t e = hipMemcpy(d, src, sizeof(float) * n, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		launch_conv_axis(dim, d, d + n, nx, ny, nz, t, nullptr, nullptr, nullptr, nullptr);
		e = hipDeviceSynchronize();
	}
	if (e == hipSuccess) e = hipMemcpy(dst, d + n, sizeof(float) * n, hipMemcpyDeviceToHost);
	hipFree(d);
	if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return SIFT3D_ERR_HIP; }
	return SIFT3D_OK;
}

// ---- one keypoint on a caller-provided level: Assign_Orientation_Imp / Extract_Descriptor_Imp (Src/cSIFT3D.cc:913-1138, 1152-1381) as
// free functions (Include/cSIFT3D.h:224, 228).  The pipeline's own kernels run on the BOX of the level the window reaches -- the clipped
// window bounds of Src/cSIFT3D.cc:939-955 / 1184-1200 plus the plane either side the central differences read -- with the keypoint's
// coordinates shifted into it: the kernels' own clipping of a window to [1, n - 2] of that box gives the same voxel set as the
// reference's on the whole level, and nothing else of the level is read.  The box lives in level slot 1 of a small one-octave context that
// is kept for the next call (a loop over keypoints, as the reference's callers run, pays for it once); its tables are rebuilt when sigma / scale /
// unit change.  The keypoint must sit on a voxel and the level's unit must be a power of two, as in the pipeline: anything else is refused.
namespace {
struct OneKp {
	sift3d_ctx *c = nullptr;
	int device = -1, octave_base = -1, edge = 0;
	float ori_sigma = -1.f, scale = -1.f;
};
std::mutex g_onekp_mu;
OneKp g_onekp;  // (never destroyed at exit: the HIP runtime may be gone by then)

struct Box { int lo[3], n[3]; };  // first voxel of the box in the level, box dimensions

// window bounds like win_bounds (kernels_orient.hip / kernels_desc.hip), then one voxel either side
bool window_box(const int c[3], const int dims[3], float radius, float unit, Box &b) {
	for (int a = 0; a < 3; a++) {
		if (dims[a] < 3) return false;
		const int s = (int)floorf((float)c[a] - radius / unit), e = (int)ceilf((float)c[a] + radius / unit);
		const int lo = s > 1 ? s : 1, hi = e < dims[a] - 2 ? e : dims[a] - 2;
		if (hi < lo) return false;
		b.lo[a] = lo - 1; b.n[a] = hi - lo + 3;
	}
	return true;
}

int onekp_prepare(OneKp &K, int device, float unit, int edge, float ori_sigma, float scale) {
	int ex = 0;
	const float m = frexpf(unit, &ex);
	if (!(unit >= 1.0f) || m != 0.5f || ex - 1 > 20) { set_last_error("the level's unit must be a power of two >= 1 (2^octave)"); return SIFT3D_ERR_ARG; }
	const int ob = ex - 1;
	if (!K.c || K.device != device || K.octave_base != ob || K.edge < edge) {
		if (K.c) { sift3d_destroy(K.c); K = OneKp(); }
		CreateCfg cfg;
		const int e = std::max(96, (edge + 31) & ~31);
		cfg.nx = cfg.ny = cfg.nz = e; cfg.octave_base = ob; cfg.noct_total = ob + 1; cfg.seeded = true;
		sift3d_ctx *c = nullptr;
		int rc = create_common(&c, cfg, nullptr, device);
		if (rc) return rc;
		K.c = c; K.device = device; K.octave_base = ob; K.edge = e;
	}
	if (K.ori_sigma != ori_sigma || K.scale != scale) {
		sift3d_ctx *c = K.c;
		std::vector<WinLut> luts = blank_luts(c);
		std::vector<float> pool;
		c->desc_lut_lds = true;
		const size_t at = ((size_t)K.octave_base * 8 + 1) * 2;
		(void)append_lut(pool, luts[at], 0, ori_sigma, ori_sigma * 3.0f, unit, scale);
		const float dsig = scale * 7.071067812f;
		if (!append_lut(pool, luts[at + 1], 1, dsig, 2.0f * dsig, unit, scale)) c->desc_lut_lds = false;
		S3D_HIP(hipStreamSynchronize(c->stream));
		int rc = upload_luts(c, luts, pool);
		if (rc) return rc;
		K.ori_sigma = ori_sigma; K.scale = scale;
	}
	return SIFT3D_OK;
}

// the box -> level slot 1, its dimensions -> the level table, the record -> extremum 0
int onekp_load(OneKp &K, const float *level, int nx, int ny, float unit, const Box &b, const DevKp &rec) {
	sift3d_ctx *c = K.c;
	hipStream_t st = c->stream;
	std::vector<float> box((size_t)b.n[0] * b.n[1] * b.n[2]);
	for (int z = 0; z < b.n[2]; z++)
		for (int y = 0; y < b.n[1]; y++)
			memcpy(&box[((size_t)z * b.n[1] + y) * b.n[0]], level + ((size_t)(b.lo[2] + z) * ny + (size_t)(b.lo[1] + y)) * nx + b.lo[0], sizeof(float) * b.n[0]);
	Level &L = c->gss[1];
	const LevelRef ref{L.d, b.n[0], b.n[1], b.n[2], unit, 0};
	const unsigned words[3] = {1u, 0u, 0u};  // extrema, overflow flag, keypoints
	const int code = rec.code;
	S3D_HIP(hipMemcpyAsync(L.d, box.data(), sizeof(float) * box.size(), hipMemcpyHostToDevice, st));
	S3D_HIP(hipMemcpyAsync(c->d_levels + ((size_t)K.octave_base * 8 + 1), &ref, sizeof(ref), hipMemcpyHostToDevice, st));
	S3D_HIP(hipMemcpyAsync(c->d_ext, &rec, sizeof(rec), hipMemcpyHostToDevice, st));
	S3D_HIP(hipMemcpyAsync(c->d_codes, &code, sizeof(int), hipMemcpyHostToDevice, st));
	S3D_HIP(hipMemcpyAsync(c->d_total, words, sizeof(words), hipMemcpyHostToDevice, st));
	S3D_HIP(hipStreamSynchronize(st));  // (the sources are pageable host memory of this frame)
	return SIFT3D_OK;
}

int onekp_check(const float *level, int nx, int ny, int nz, const sift3d_keypoint *kp, int c[3]) {
	if (!level || !kp || nx < 3 || ny < 3 || nz < 3) { set_last_error("bad level / keypoint"); return SIFT3D_ERR_ARG; }
	const float f[3] = {kp->x, kp->y, kp->z};
	const int dims[3] = {nx, ny, nz};
	for (int a = 0; a < 3; a++) {
		c[a] = (int)f[a];
		if ((float)c[a] != f[a] || c[a] < 0 || c[a] >= dims[a]) { set_last_error("the keypoint must sit on a voxel of the level (integral x, y, z inside it)"); return SIFT3D_ERR_ARG; }
	}
	if (!(kp->scale > 0.0f)) { set_last_error("keypoint scale must be positive"); return SIFT3D_ERR_ARG; }
	return SIFT3D_OK;
}
}  // namespace

// kp in: x, y, z (voxel of the level), scale; out: win, eigvalue, eigvector, Rotation (as Assign_Orientation_Imp leaves it: not
// transposed), str_tensor (computed from zero: the reference accumulates into what Initialize_Keypoint zeroed).  *code: the reference's
// return value (1 accepted, -1 weak gradient, -2 eigenvalue ratio / not distinct, -3 corner).
extern "C" int sift3d_orient_keypoint(const float *level, int nx, int ny, int nz, float unit, sift3d_keypoint *kp, float sigma, float max_eig_ratio,
                                      float corner_thresh, int device, int *code) {
	int ctr[3];
	int rc = onekp_check(level, nx, ny, nz, kp, ctr);
	if (rc) return rc;
	if (!code || !(sigma > 0.0f)) { set_last_error("sift3d_orient_keypoint: bad argument"); return SIFT3D_ERR_ARG; }
	if ((rc = set_device(device)) != SIFT3D_OK) return rc;
	const int dims[3] = {nx, ny, nz};
	Box b;
	if (!window_box(ctr, dims, sigma * 3.0f, unit, b)) { set_last_error("the orientation window is empty"); return SIFT3D_ERR_ARG; }
	std::lock_guard<std::mutex> lk(g_onekp_mu);
	OneKp &K = g_onekp;
	if ((rc = onekp_prepare(K, device, unit, std::max(b.n[0], std::max(b.n[1], b.n[2])), sigma, kp->scale)) != SIFT3D_OK) return rc;
	sift3d_ctx *c = K.c;
	DevKp rec;
	memset(&rec, 0, sizeof(rec));
	rec.x = ctr[0] - b.lo[0]; rec.y = ctr[1] - b.lo[1]; rec.z = ctr[2] - b.lo[2];
	rec.octave = K.octave_base; rec.level = 1; rec.scale = kp->scale; rec.slot = -1;
	if ((rc = onekp_load(K, level, nx, ny, unit, b, rec)) != SIFT3D_OK) return rc;
	hipStream_t st = c->stream;
	launch_orient(c->d_ext, c->d_codes, c->d_total, c->ext_cap, c->d_levels, c->d_luts, c->d_lutpool, max_eig_ratio, corner_thresh, 0, 1, c->d_order,
	              c->d_nkp + 3, st);
	S3D_HIP(hipMemcpyAsync(&rec, c->d_ext, sizeof(rec), hipMemcpyDeviceToHost, st));
	S3D_HIP(hipStreamSynchronize(st));
	S3D_HIP(hipGetLastError());
	*code = rec.code;
	for (int i = 0; i < 3; i++) { kp->win[i] = rec.win[i]; kp->eigvalue[i] = rec.eigvalue[i]; }
	for (int i = 0; i < 9; i++) { kp->eigvector[i] = rec.eigvector[i]; kp->Rotation[i] = rec.rot[i]; kp->str_tensor[i] = rec.st[i]; }
	return SIFT3D_OK;
}

// kp in: x, y, z, scale, Rotation (as the orientation stage leaves it), str_tensor (first guess of the histogram's fixed-point unit only);
// out: Rotation TRANSPOSED (Src/cSIFT3D.cc:1214 inverts it in place), desc768 = the normalised descriptor (cc:1350-1358)
extern "C" int sift3d_describe_keypoint(const float *level, int nx, int ny, int nz, float unit, sift3d_keypoint *kp, float *desc768, int device) {
	int ctr[3];
	int rc = onekp_check(level, nx, ny, nz, kp, ctr);
	if (rc) return rc;
	if (!desc768) return SIFT3D_ERR_ARG;
	if ((rc = set_device(device)) != SIFT3D_OK) return rc;
	const int dims[3] = {nx, ny, nz};
	const float dsig = kp->scale * 7.071067812f;
	Box b;
	if (!window_box(ctr, dims, 2.0f * dsig, unit, b)) { set_last_error("the descriptor window is empty"); return SIFT3D_ERR_ARG; }
	std::lock_guard<std::mutex> lk(g_onekp_mu);
	OneKp &K = g_onekp;
	// the orientation table of the pair is ALWAYS the pipeline's for this scale (sigma = 1.5 scale): its weight sum enters the first guess of the
	// fixed-point unit, so a table left behind by an earlier sift3d_orient_keypoint with another sigma would make the descriptor's low bits
	// depend on the call history; onekp_prepare rebuilds the pair when it differs
	if ((rc = onekp_prepare(K, device, unit, std::max(b.n[0], std::max(b.n[1], b.n[2])), 1.5f * kp->scale, kp->scale)) != SIFT3D_OK)
		return rc;
	sift3d_ctx *c = K.c;
	DevKp rec;
	memset(&rec, 0, sizeof(rec));
	rec.x = ctr[0] - b.lo[0]; rec.y = ctr[1] - b.lo[1]; rec.z = ctr[2] - b.lo[2];
	rec.octave = K.octave_base; rec.level = 1; rec.scale = kp->scale; rec.code = 1; rec.slot = -1;
	for (int i = 0; i < 3; i++) { rec.win[i] = kp->win[i]; rec.eigvalue[i] = kp->eigvalue[i]; }
	for (int i = 0; i < 9; i++) { rec.eigvector[i] = kp->eigvector[i]; rec.rot[i] = kp->Rotation[i]; rec.st[i] = kp->str_tensor[i]; }
	if ((rc = onekp_load(K, level, nx, ny, unit, b, rec)) != SIFT3D_OK) return rc;
	hipStream_t st = c->stream;
	launch_slots(c->d_ext, c->d_codes, c->d_total, c->ext_cap, c->d_nkp, c->d_order, c->kp_cap, c->d_slots_part, st);
	launch_describe(c->d_ext, c->d_total, c->ext_cap, c->d_levels, c->d_luts, c->d_lutpool, c->d_desc, c->kp_cap, 0, 1, c->d_order, c->d_nkp, c->d_nkp + 1, st,
	                c->desc_lut_lds, &c->dsplit);
	launch_finalize(c->d_ext, c->d_total, c->ext_cap, 1, c->d_kpout, c->d_xyz, c->kp_cap, st);
	sift3d_keypoint out;
	S3D_HIP(hipMemcpyAsync(desc768, c->d_desc, sizeof(float) * kDesc, hipMemcpyDeviceToHost, st));
	S3D_HIP(hipMemcpyAsync(&out, c->d_kpout, sizeof(out), hipMemcpyDeviceToHost, st));
	S3D_HIP(hipStreamSynchronize(st));
	S3D_HIP(hipGetLastError());
	for (int i = 0; i < 9; i++) kp->Rotation[i] = out.Rotation[i];
	return SIFT3D_OK;
}

extern "C" int sift3d_downsample(const float *src, int snx, int sny, int snz, float *dst, int nx, int ny, int nz, int device) {
	if (!src || !dst || nx <= 0 || ny <= 0 || nz <= 0 || snx <= 0 || sny <= 0 || snz <= 0 || 2 * (nx - 1) >= snx || 2 * (ny - 1) >= sny ||
	    2 * (nz - 1) >= snz)
		return SIFT3D_ERR_ARG;
	int rc = set_device(device);
	if (rc) return rc;
	const size_t ns = (size_t)snx * sny * snz, nd = (size_t)nx * ny * nz;
	float *d = nullptr;
	S3D_HIP(hipMalloc(&d, sizeof(float) * (ns + nd)));
	hipError_t e = hipMemcpy(d, src, sizeof(float) * ns, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		launch_downsample(d, snx, sny, d + ns, nx, ny, nz, nullptr);
		e = hipDeviceSynchronize();
	}
	if (e == hipSuccess) e = hipMemcpy(dst, d + ns, sizeof(float) * nd, hipMemcpyDeviceToHost);
	hipFree(d);
	if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return SIFT3D_ERR_HIP; }
	return SIFT3D_OK;
}

extern "C" int sift3d_dog_sub(const float *prev, const float *cur, size_t n, float *dog, int device) {
	if (!prev || !cur || !dog || n == 0) return SIFT3D_ERR_ARG;
	int rc = set_device(device);
	if (rc) return rc;
	float *d = nullptr;
	S3D_HIP(hipMalloc(&d, sizeof(float) * n * 3));
	hipError_t e = hipMemcpy(d, prev, sizeof(float) * n, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(d + n, cur, sizeof(float) * n, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		launch_dog_from_gss(d + n, d, d + 2 * n, n, nullptr);  // (cur - prev) * (-1)
		e = hipDeviceSynchronize();
	}
	if (e == hipSuccess) e = hipMemcpy(dog, d + 2 * n, sizeof(float) * n, hipMemcpyDeviceToHost);
	hipFree(d);
	if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return SIFT3D_ERR_HIP; }
	return SIFT3D_OK;
}
