// kernels_orient.hip -- dominant-orientation assignment, one 64-lane wave per DoG extremum.
//
// Restates Assign_Orientation + Assign_Orientation_Imp + DistinctEig (reference
// Src/cSIFT3D.cc:427-482, 913-1150): weighted structure tensor and weighted mean gradient over the
// sphere r = 3*(1.5*scale) around the extremum on G[octave][level], symmetric 3x3 eigen-decomposition
// in fp64, eigen-ratio / distinctness / corner rejects, sign alignment, R = [v_max | v_mid | v_max x v_mid].
//
// MI355X mapping (k_orient): one wave per extremum, many short workgroups (window volumes differ 3x between the levels).  Three
// consecutive planes of the window footprint live in the wave's LDS slots (16-byte pieces global -> LDS, the plane after next in
// flight); the lanes walk the lattice points of the window SPHERE of the plane from a per-level list (WinLut::list_off, entries
// requested a plane ahead) -- windows larger than the default take the box scan over LDS tiles or global memory.  The Gaussian
// weight comes from a host-built table indexed by the integer squared offset (bit-identical to the CPU expf, no device exp), the
// nine fp32 sums are reduced across the wave with shuffles; k_orient_finish runs the fp64 Jacobi eigen-solve with one LANE per
// extremum.  Per-voxel terms are bit-identical to the reference; the ORDER of the fp32 additions of this first pass differs
// (lane-strided + butterfly instead of sequential), which moves the tensor by ~1e-6 relative.
//
// r03 -- the window sums of every ACCEPTED keypoint are bit-identical to the reference's.  The descriptor is a discontinuous function
// of the rotation matrix (a window voxel is in or out of the rotated 4x4x4 cube, Src/cSIFT3D.cc:1299-1303): a last-bit difference
// of R flips single voxels, and where such a voxel carries a large gradient (small blobs at octave >= 1) one descriptor element
// moved by up to 1.5e-3 (512^3: 15 of 11 292 keypoints above 8e-4; global RMS 4e-6).  So the parallel sums only SCREEN:
// k_orient_finish flags every extremum that is accepted or within a margin of any threshold, k_orient<true> recomputes the
// flagged ones (21 % of the extrema) with the reference's summation order -- the 9 terms of 64 consecutive lattice points go to
// LDS and nine lanes add them one after the other, exactly like the sequential loop of Src/cSIFT3D.cc:958-998 (points the
// reference skips contribute +0, which leaves an accumulator unchanged) -- and k_orient_finish runs again on them.
#include <float.h>

#include "sift3d_internal.h"

namespace s3d {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = v + __shfl_xor(v, o, 64);
	return v;
}

// window bounds, Src/cSIFT3D.cc:939-955 (rad/u is exact: u is a power of two)
__device__ __forceinline__ void win_bounds(float c, float rad, float u, int n, int &lo, int &hi) {
	int s = (int)floorf(c - __fdiv_rn(rad, u));
	lo = s > 1 ? s : 1;
	int e = (int)ceilf(c + __fdiv_rn(rad, u));
	hi = e < (n - 2) ? e : n - 2;
}

// cyclic Jacobi for a symmetric 3x3 in fp64; columns of V are unit eigenvectors.  The reference
// uses Eigen::EigenSolver<Matrix3d> (Src/cSIFT3D.cc:1027-1029); order and signs are fixed by the
// caller afterwards, so any accurate solver yields the same fp32 values.
__device__ void eig_sym3(const double A[9], double w[3], double V[9]) {
	double a[9];
#pragma unroll
	for (int i = 0; i < 9; i++) { a[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
	for (int sweep = 0; sweep < 64; sweep++) {
		const double off = fabs(a[1]) + fabs(a[2]) + fabs(a[5]);
		const double diag = fabs(a[0]) + fabs(a[4]) + fabs(a[8]);
		if (off <= 1e-300 || off <= 1e-22 * diag) break;
#pragma unroll
		for (int pq = 0; pq < 3; pq++) {
			const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
			const double apq = a[3 * p + q];
			if (apq == 0.0) continue;
			const double theta = (a[4 * q] - a[4 * p]) / (2.0 * apq);
			const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
			const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
			for (int k = 0; k < 3; k++) {
				const double akp = a[3 * k + p], akq = a[3 * k + q];
				a[3 * k + p] = c * akp - s * akq;
				a[3 * k + q] = s * akp + c * akq;
			}
#pragma unroll
			for (int k = 0; k < 3; k++) {
				const double apk = a[3 * p + k], aqk = a[3 * q + k];
				a[3 * p + k] = c * apk - s * aqk;
				a[3 * q + k] = s * apk + c * aqk;
			}
#pragma unroll
			for (int k = 0; k < 3; k++) {
				const double vkp = V[3 * k + p], vkq = V[3 * k + q];
				V[3 * k + p] = c * vkp - s * vkq;
				V[3 * k + q] = s * vkp + c * vkq;
			}
		}
	}
	w[0] = a[0]; w[1] = a[4]; w[2] = a[8];
#pragma unroll
	for (int j = 0; j < 3; j++) {
		const double n = sqrt(V[j] * V[j] + V[3 + j] * V[3 + j] + V[6 + j] * V[6 + j]);
		V[j] /= n; V[3 + j] /= n; V[6 + j] /= n;
	}
}

__device__ __forceinline__ float dot3f(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// everything after the window sums (Src/cSIFT3D.cc:1000-1137), executed by one lane
// *near (may be null): set when the extremum is accepted or any test lies within a margin of its threshold -- the fast sums carry
// a relative error of ~1e-6, the margins are 100x that
__device__ int finish_orientation(DevKp &kp, const float T6[6], const float w3[3], float max_eig_ratio, float corner_thresh, bool *near = nullptr) {
	bool nr = false;
	float T[9] = {T6[0], T6[1], T6[2], T6[1], T6[3], T6[4], T6[2], T6[4], T6[5]};
#pragma unroll
	for (int i = 0; i < 9; i++) kp.st[i] = T[i];
	kp.win[0] = w3[0]; kp.win[1] = w3[1]; kp.win[2] = w3[2];
	const float gg = dot3f(w3, w3);
	nr = fabsf(gg - 1E-10f) <= 1E-13f;
	if (near) *near = nr;
	if (gg < 1E-10f) return -1;  // ori_grad_thresh, Src/cSIFT3D.cc:22

	double A[9], wv[3], V[9];
#pragma unroll
	for (int i = 0; i < 9; i++) A[i] = (double)T[i];
	eig_sym3(A, wv, V);
	float val[3], vec[3][3];
#pragma unroll
	for (int j = 0; j < 3; j++) {
		val[j] = (float)wv[j];
		vec[j][0] = (float)V[j]; vec[j][1] = (float)V[3 + j]; vec[j][2] = (float)V[6 + j];
	}
	// sort ascending by fp32 eigenvalue (Src/cSIFT3D.cc:1050); 3-element insertion network
#define S3D_CSWAP(i, j)                                                                   \
	if (val[j] < val[i]) {                                                                \
		float tv = val[i]; val[i] = val[j]; val[j] = tv;                                  \
		for (int c = 0; c < 3; c++) { float t2 = vec[i][c]; vec[i][c] = vec[j][c]; vec[j][c] = t2; } \
	}
	S3D_CSWAP(0, 1) S3D_CSWAP(1, 2) S3D_CSWAP(0, 1)
#undef S3D_CSWAP
#pragma unroll
	for (int j = 0; j < 3; j++) {
		kp.eigvalue[j] = val[j];
		kp.eigvector[3 * j] = vec[j][0]; kp.eigvector[3 * j + 1] = vec[j][1]; kp.eigvector[3 * j + 2] = vec[j][2];
	}
	const float r01 = fabsf(__fdiv_rn(val[0], val[1])), r12 = fabsf(__fdiv_rn(val[1], val[2]));
	nr = nr || fabsf(r01 - max_eig_ratio) <= 1e-4f || fabsf(r12 - max_eig_ratio) <= 1e-4f || !(r01 == r01) || !(r12 == r12);
	if (near) *near = nr;
	if (r01 > max_eig_ratio || r12 > max_eig_ratio) return -2;
	if ((double)fabsf(val[0] - val[1]) < DBL_EPSILON || (double)fabsf(val[0] - val[2]) < DBL_EPSILON ||
	    (double)fabsf(val[2] - val[1]) < DBL_EPSILON)
		return -2;

	const float d_norm = __fsqrt_rn(dot3f(w3, w3));
	float corner = FLT_MAX;
	for (int i = 2; i > 0; i--) {
		const float d = dot3f(vec[i], w3);
		const float q_norm = __fsqrt_rn(dot3f(vec[i], vec[i]));
		const float cos_ang = __fdiv_rn(d, d_norm * q_norm);
		const float a = fabsf(cos_ang);
		corner = corner < a ? corner : a;
		const float sgn = d > 0.0f ? 1.0f : -1.0f;
		vec[i][0] *= sgn; vec[i][1] *= sgn; vec[i][2] *= sgn;
	}
	nr = nr || fabsf(corner - corner_thresh) <= 1e-4f;
	if (near) *near = nr;
	if (corner < corner_thresh) return -3;
	if (near) *near = true;  // accepted: its sums are recomputed in the reference's order
	const float *v1 = vec[2], *v2 = vec[1];
	float vr[3];
	vr[0] = v1[1] * v2[2] - v1[2] * v2[1];
	vr[1] = v1[2] * v2[0] - v1[0] * v2[2];
	vr[2] = v1[0] * v2[1] - v1[1] * v2[0];
	kp.rot[0] = v1[0]; kp.rot[1] = v2[0]; kp.rot[2] = vr[0];
	kp.rot[3] = v1[1]; kp.rot[4] = v2[1]; kp.rot[5] = vr[1];
	kp.rot[6] = v1[2]; kp.rot[7] = v2[2]; kp.rot[8] = vr[2];
	return 1;
}

// LDS tiles of the orientation window: the default radius (3 * 1.5 * scale <= 11.43 voxels) gives <= 25 voxels per side,
// + 2 for the central differences; larger windows (non-default sigma) take the global-load path
#ifndef S3D_ORI_UN
#define S3D_ORI_UN 2  /* window voxels per lane and pass */
#endif
constexpr int kTileW = 28, kTileH = 27;
constexpr int kTilePc = ((kTileW / 4) * kTileH + 63) / 64;  // 16-byte pieces of a tile plane per lane (3)
typedef float f4o __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte global piece at dword alignment
constexpr int kOriLut = 256;  // squared-offset weight table of the orientation window (r^2 <= 131 by default)

// EXACT = false: every owned extremum, parallel sums (lane-strided + butterfly).  EXACT = true: the extrema of redo_list, sums in
// the reference's order (see the header): a pass hands 64 consecutive window points -- in the reference's (z, y, x) loop order -- to
// the lanes, the nine products of every point go to the wave's term buffer in LDS and lanes 0..8 add the 64 values of "their" sum
// one after the other.  Points the reference skips (outside the sphere / the clipped box) contribute +0.0f: acc + 0 == acc.
constexpr int kTermPitch = 68;  // floats per term row: rows start 4 banks apart, the chain's 16-byte reads do not collide
template <bool EXACT>
__global__ void __launch_bounds__(256) k_orient(DevKp *__restrict__ kps, int *__restrict__ codes, const unsigned *__restrict__ d_count, unsigned cap,
                                                const LevelRef *__restrict__ levels, const WinLut *__restrict__ luts,
                                                const float *__restrict__ lutpool, float max_eig, float corner, int part_rank,
                                                int part_world, const int *__restrict__ redo_list, const unsigned *__restrict__ redo_count) {
	__shared__ __attribute__((aligned(16))) float s_tile[4 * 3 * kTileW * kTileH];  // per wave: planes z-1, z, z+1 of the window footprint
	__shared__ float s_wlut[4 * kOriLut];
	__shared__ __attribute__((aligned(16))) float s_terms[EXACT ? 4 * 9 * kTermPitch : 4];
	const unsigned count = min(d_count[0], cap);
	const int lane = threadIdx.x & 63;
	const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const unsigned nwaves = gridDim.x * (blockDim.x >> 6);
	// partitioned run: this rank orients the extrema k = j * world + rank (dealt densely to the waves); the others get code 0
	const unsigned pw = part_world > 1 ? (unsigned)part_world : 1u, pr = part_world > 1 ? (unsigned)part_rank : 0u;
	if (!EXACT && pw > 1)
		for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x)
			if (k % pw != pr) { kps[k].code = 0; codes[k] = 0; }
	const unsigned owned = EXACT ? min(redo_count[0], cap) : (count > pr ? (count - pr + pw - 1) / pw : 0u);
	float *tb = &s_terms[EXACT ? (threadIdx.x >> 6) * 9 * kTermPitch : 0];
	for (unsigned j = wave; j < owned; j += nwaves) {
		const unsigned k = EXACT ? (unsigned)redo_list[j] : j * pw + pr;
		const int cxi = kps[k].x, cyi = kps[k].y, czi = kps[k].z;
		const int li = kps[k].octave * 8 + kps[k].level;
		const LevelRef L = levels[li];
		const WinLut lut = luts[li * 2 + 0];
		const float *__restrict__ wtab = lutpool + lut.off;
		const float u = L.unit, inv_u = __fdiv_rn(1.0f, u);
		int x0, x1, y0, y1, z0, z1;
		win_bounds((float)cxi, lut.radius, u, L.nx, x0, x1);
		win_bounds((float)cyi, lut.radius, u, L.ny, y0, y1);
		win_bounds((float)czi, lut.radius, u, L.nz, z0, z1);
		const int wx = x1 - x0 + 1, wy = y1 - y0 + 1;
		const int plane = (wx > 0 && wy > 0) ? wx * wy : 0;
		const float inv_wx = 1.0f / (float)(wx > 0 ? wx : 1);
		const size_t sy = (size_t)L.nx, sz = (size_t)L.nx * L.ny;
		const gfloat_p Ld = as_global(L.d);  // global_load instead of flat_load (see sift3d_internal.h)
		float t00 = 0.f, t01 = 0.f, t02 = 0.f, t11 = 0.f, t12 = 0.f, t22 = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
		float chain = 0.f;  // EXACT: lane a < 9 carries sum a (t00 t01 t02 t11 t12 t22 g0 g1 g2)
		// one window point per lane: fast form = nine partial sums per lane; exact form = the 64 points of the pass, which are
		// consecutive in the reference's loop order, are added one after the other (nvalid: wave-uniform bound on the leading
		// points of the pass that can be valid)
		auto emit = [&](bool ok, float c1p, float c1m, float c2p, float c2m, float c3p, float c3m, float ww, int nvalid) {
			float vx = 0.5f * (c1p - c1m);
			float vy = 0.5f * (c2p - c2m);
			float vz = 0.5f * (c3p - c3m);
			vx = vx * inv_u; vy = vy * inv_u; vz = vz * inv_u;
			if (!EXACT) {
				if (ok) {
					// the parallel sums only SCREEN (k_orient_finish flags everything within 100x their error of a threshold and the
					// flagged extrema are redone in the reference's order and arithmetic): fused multiply-adds on pre-weighted components
					const float wx = vx * ww, wy = vy * ww, wz = vz * ww;
					t00 = __fmaf_rn(vx, wx, t00);
					t01 = __fmaf_rn(vx, wy, t01);
					t02 = __fmaf_rn(vx, wz, t02);
					t11 = __fmaf_rn(vy, wy, t11);
					t12 = __fmaf_rn(vy, wz, t12);
					t22 = __fmaf_rn(vz, wz, t22);
					g0 = g0 + wx; g1 = g1 + wy; g2 = g2 + wz;
				}
			} else {
				tb[0 * kTermPitch + lane] = ok ? vx * vx * ww : 0.0f;
				tb[1 * kTermPitch + lane] = ok ? vx * vy * ww : 0.0f;
				tb[2 * kTermPitch + lane] = ok ? vx * vz * ww : 0.0f;
				tb[3 * kTermPitch + lane] = ok ? vy * vy * ww : 0.0f;
				tb[4 * kTermPitch + lane] = ok ? vy * vz * ww : 0.0f;
				tb[5 * kTermPitch + lane] = ok ? vz * vz * ww : 0.0f;
				tb[6 * kTermPitch + lane] = ok ? vx * ww : 0.0f;
				tb[7 * kTermPitch + lane] = ok ? vy * ww : 0.0f;
				tb[8 * kTermPitch + lane] = ok ? vz * ww : 0.0f;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				if (lane < 9) {
					// (eight 16-byte reads in flight, then their 32 dependent adds: the loop is bound by the add chain, not by LDS latency)
					const float4 *row = reinterpret_cast<const float4 *>(tb + lane * kTermPitch);
#pragma unroll
					for (int h = 0; h < 2; h++) {
						if (h * 32 >= nvalid) break;  // wave-uniform
						float4 v[8];
#pragma unroll
						for (int i = 0; i < 8; i++) v[i] = row[h * 8 + i];
#pragma unroll
						for (int i = 0; i < 8; i++) { chain = chain + v[i].x; chain = chain + v[i].y; chain = chain + v[i].z; chain = chain + v[i].w; }
					}
				}
				__builtin_amdgcn_wave_barrier();  // the term rows are free again
			}
		};
		const int tw = wx + 2, th = wy + 2;  // window footprint plus the central-difference border
		if (plane > 0 && z1 >= z0 && tw <= kTileW && th <= kTileH && lut.len <= kOriLut) {
			// ---- LDS path: three consecutive (tw x th) planes of the level live in this wave's LDS slots, so a window
			// voxel costs one global load (plus border) instead of six; plane z+2 is requested into registers while plane
			// z is processed and written to its slot one iteration later.  Same voxel -> lane mapping as the global path below.
			float *tile = &s_tile[(threadIdx.x >> 6) * 3 * kTileW * kTileH];
			float *wl = &s_wlut[(threadIdx.x >> 6) * kOriLut];   // this wave's copy of the window weights
			for (int i = lane; i < lut.len && i < kOriLut; i += 64) wl[i] = wtab[i];
			// The tile planes travel as 16-byte pieces (dword-aligned global_load_dwordx4, ds_write_b128): 3 vector-memory
			// instructions per lane and plane instead of 12 -- their ISSUE is what costs.  The last piece of a row may reach up to
			// 3 floats past column tw-1: inside the level row, or in the next row / plane; only for a window in the far corner
			// of a level these are the first bytes of the NEXT Gaussian level of the arena (levels 1..3 are never the last).
			const int npx = (tw + 3) >> 2;   // pieces per tile row
			const int tnp = npx * th;
			const float inv_npx = 1.0f / (float)npx;
			int toff[kTilePc];   // per-lane tile offsets of the pieces this lane moves (same for every plane)
			int goff[kTilePc];
#pragma unroll
			for (int i = 0; i < kTilePc; i++) {
				const int idx = lane + 64 * i;
				const int ty = (int)(((float)idx + 0.5f) * inv_npx);
				const int px = idx - ty * npx;
				const bool ok = idx < tnp;
				toff[i] = ok ? ty * kTileW + 4 * px : -1;
				goff[i] = ok ? (x0 - 1 + 4 * px) + (int)sy * (y0 - 1 + ty) : (x0 - 1) + (int)sy * (y0 - 1);
			}
			const gfloat_p base = Ld - sz * (size_t)L.zoff;
			f4o pf[kTilePc];
			auto request = [&](int zp) {  // unconditional, clamped loads (planes z0-1 .. z1+1 are inside the level)
				const gfloat_p pl = base + sz * (size_t)zp;
#pragma unroll
				for (int i = 0; i < kTilePc; i++) pf[i] = *reinterpret_cast<const f4o __attribute__((address_space(1))) *>(pl + goff[i]);
			};
			auto deposit = [&](int slot) {
				float *dstp = tile + slot * (kTileW * kTileH);
#pragma unroll
				for (int i = 0; i < kTilePc; i++)
					if (toff[i] >= 0) *reinterpret_cast<float4 *>(dstp + toff[i]) = make_float4(pf[i].x, pf[i].y, pf[i].z, pf[i].w);
			};
			request(z0 - 1); deposit(0);
			request(z0);     deposit(1);
			request(z0 + 1);
			int sm = 0, sc = 1, sp = 2;  // slots of planes z-1, z, z+1
			// lattice list of the window sphere (see the plane loop): up to kEL entries per lane and plane, held in registers
			constexpr int kEL = 8;
			const unsigned *__restrict__ lst = reinterpret_cast<const unsigned *>(lutpool) + (lut.list_off >= 0 ? lut.list_off : 0);
			const int LR = lut.list_R;
			const unsigned *__restrict__ ent = lst + 2 * LR + 2;
			const bool use_list = lut.list_off >= 0 && (int)lst[LR + 1] - (int)lst[LR] <= 64 * kEL;  // the central plane is the largest
			const int ox = cxi - x0 - 128, oy = cyi - y0 - 128;  // entry -> window coordinates
			unsigned En[kEL];
			int cntn = 0;
			auto fetch_entries = [&](int dzp) {
				const int pi = dzp + LR;
				const bool has = use_list && pi >= 0 && pi <= 2 * LR;
				const int e0 = has ? (int)lst[pi] : 0, e1 = has ? (int)lst[pi + 1] : 0;
				cntn = e1 - e0;
#pragma unroll
				for (int i = 0; i < kEL; i++) {
					const int idx = e0 + i * 64 + lane;
					En[i] = ent[idx < e1 ? idx : e0];
				}
			};
			fetch_entries(z0 - czi);
			for (int z = z0; z <= z1; z++) {
#if !defined(S3D_ODIAG) || !(S3D_ODIAG & 2)  // (timing only: no tile traffic)
				deposit(sp);                       // plane z+1 (requested one iteration ago)
				request(min(z + 2, z1 + 1));       // in flight while plane z is processed
#endif
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				const float *pm = tile + sm * (kTileW * kTileH), *pc = tile + sc * (kTileW * kTileH), *pp = tile + sp * (kTileW * kTileH);
				const int dz = z - czi;
				// kUn voxels per pass, all LDS reads (weights included) issued before the first use; inactive voxels read
				// valid addresses and are masked, so the loop body is branch-free and pipelines
				constexpr int kUn = S3D_ORI_UN;
#ifdef S3D_ODIAG
				if (S3D_ODIAG & 1) { } else  // timing only: no window arithmetic
#endif
				if (use_list) {
					// the lattice points of the window sphere in this plane come from a list (WinLut::list_off): 52 % of the box is
					// outside the sphere, and the list also carries n = dx^2 + dy^2 + dz^2.  Windows clipped by the level border skip
					// the entries outside their box.  This plane's entries were requested one plane ago (a load inside the pass loop
					// would put a memory round trip in front of every pass).  The list is in (dy, dx) order: the reference's loop order.
					unsigned E[kEL];
#pragma unroll
					for (int i = 0; i < kEL; i++) E[i] = En[i];
					const int cnt = cntn;
					fetch_entries(dz + 1);
#pragma unroll
					for (int p0 = 0; p0 < kEL; p0 += kUn) {
						if (p0 * 64 >= cnt) break;  // wave-uniform
						float nb[kUn][6], w[kUn];
#pragma unroll
						for (int q = 0; q < kUn; q++) {
							const unsigned e = E[p0 + q];
							const int lx = (int)(e & 255u) + ox, ly = (int)((e >> 8) & 255u) + oy;
							const bool ok = (p0 + q) * 64 + lane < cnt && (unsigned)lx < (unsigned)wx && (unsigned)ly < (unsigned)wy;
							w[q] = ok ? wl[e >> 16] : -1.0f;
							const int o = ok ? __mul24(ly + 1, kTileW) + lx + 1 : kTileW + 1;  // (24-bit multiply: full rate)
							nb[q][0] = pc[o + 1]; nb[q][1] = pc[o - 1]; nb[q][2] = pc[o + kTileW]; nb[q][3] = pc[o - kTileW];
							nb[q][4] = pp[o]; nb[q][5] = pm[o];
						}
#pragma unroll
						for (int q = 0; q < kUn; q++) {
							if (EXACT && (p0 + q) * 64 >= cnt) break;  // wave-uniform: nothing left in this plane
							// w < 0: outside the clipped box / past the end of the plane's list
							emit(!(w[q] < 0.0f), nb[q][0], nb[q][1], nb[q][2], nb[q][3], nb[q][4], nb[q][5], w[q], min(64, cnt - (p0 + q) * 64));
						}
					}
				} else if (EXACT) {
					for (int vb = 0; vb < plane; vb += 64) {  // wave-uniform passes of 64 consecutive box voxels
						const int v = vb + lane;
						const int ly = (int)(((float)v + 0.5f) * inv_wx);
						const int lx = v - ly * wx;
						const int dx = x0 + lx - cxi, dy = y0 + ly - cyi;
						const int n = dx * dx + dy * dy + dz * dz;
						const bool in = v < plane && n < lut.len;
						const float ww = in ? wl[n] : -1.0f;
						const int o = in ? __mul24(ly + 1, kTileW) + lx + 1 : kTileW + 1;
						emit(!(ww < 0.0f), pc[o + 1], pc[o - 1], pc[o + kTileW], pc[o - kTileW], pp[o], pm[o], ww, min(64, plane - vb));
					}
				} else
				for (int v0 = lane; v0 < plane; v0 += 64 * kUn) {
					float nb[kUn][6], w[kUn];
#pragma unroll
					for (int q = 0; q < kUn; q++) {
						const int v = v0 + 64 * q;
						const int ly = (int)(((float)v + 0.5f) * inv_wx);
						const int lx = v - ly * wx;
						const int dx = x0 + lx - cxi, dy = y0 + ly - cyi;
						const int n = dx * dx + dy * dy + dz * dz;
						const bool ok = v < plane && n < lut.len;
						w[q] = ok ? wl[n] : -1.0f;
						const int o = ok ? __mul24(ly + 1, kTileW) + lx + 1 : kTileW + 1;  // (24-bit multiply: full rate)
						nb[q][0] = pc[o + 1]; nb[q][1] = pc[o - 1]; nb[q][2] = pc[o + kTileW]; nb[q][3] = pc[o - kTileW];
						nb[q][4] = pp[o]; nb[q][5] = pm[o];
					}
#pragma unroll
					for (int q = 0; q < kUn; q++)  // w < 0: outside the sphere / past the end of the plane
						emit(!(w[q] < 0.0f), nb[q][0], nb[q][1], nb[q][2], nb[q][3], nb[q][4], nb[q][5], w[q], 64);
				}
				__builtin_amdgcn_wave_barrier();   // every lane is done with plane z-1 before its slot is overwritten
				const int tmp = sm; sm = sc; sc = sp; sp = tmp;
			}
		} else
		for (int z = z0; z <= z1; z++) {
			const int dz = z - czi;
			for (int vb = 0; vb < plane; vb += 64) {  // wave-uniform passes (EXACT needs every lane at the term buffer)
				const int v = vb + lane;
				const int ly = (int)(((float)v + 0.5f) * inv_wx);
				const int lx = v - ly * wx;
				const int x = x0 + lx, y = y0 + ly;
				const int dx = x - cxi, dy = y - cyi;
				const int n = dx * dx + dy * dy + dz * dz;
				const bool in = v < plane && n < lut.len;
				const float w = in ? wtab[n] : -1.0f;  // < 0: outside the sphere
				const bool ok = !(w < 0.0f);
				const gfloat_p c = ok ? Ld + (size_t)x + sy * (size_t)y + sz * (size_t)(z - L.zoff)
				                      : Ld + (size_t)cxi + sy * (size_t)cyi + sz * (size_t)(czi - L.zoff);  // masked lanes read the centre
				emit(ok, c[1], c[-1], c[sy], *(c - sy), c[sz], *(c - sz), w, min(64, plane - vb));
			}
		}
		if (EXACT) {
			if (lane < 9) {
				const int slot = lane < 3 ? lane : (lane == 3 ? 4 : (lane == 4 ? 5 : 8));  // t00 t01 t02 t11 t12 t22 -> st[0 1 2 4 5 8]
				if (lane < 6) kps[k].st[slot] = chain;
				else kps[k].win[lane - 6] = chain;
			}
		} else {
			float T6[6] = {wave_sum(t00), wave_sum(t01), wave_sum(t02), wave_sum(t11), wave_sum(t12), wave_sum(t22)};
			float w3[3] = {wave_sum(g0), wave_sum(g1), wave_sum(g2)};
			if (lane == 0) {
				// window sums only: the eigen decomposition runs in k_orient_finish, one LANE per extremum (here it would
				// occupy one lane of the wave and idle the other 63)
				kps[k].st[0] = T6[0]; kps[k].st[1] = T6[1]; kps[k].st[2] = T6[2]; kps[k].st[4] = T6[3]; kps[k].st[5] = T6[4]; kps[k].st[8] = T6[5];
				kps[k].win[0] = w3[0]; kps[k].win[1] = w3[1]; kps[k].win[2] = w3[2];
			}
		}
	}
}

// second phase of Assign_Orientation_Imp (Src/cSIFT3D.cc:1000-1137): eigen decomposition, rejection tests, rotation
// matrix; one thread per extremum
// pass 0: every owned extremum, from the parallel sums; extrema that are accepted or close to a threshold are appended to
// redo_list.  pass 1: the extrema of redo_list, from the sums in the reference's order.
__global__ void __launch_bounds__(64) k_orient_finish(DevKp *__restrict__ kps, int *__restrict__ codes, const unsigned *__restrict__ d_count,
                                                      unsigned cap, float max_eig, float corner, int part_rank, int part_world, int pass,
                                                      int *__restrict__ redo_list, unsigned *__restrict__ redo_count) {
	const unsigned count = min(d_count[0], cap);
	const unsigned pw = part_world > 1 ? (unsigned)part_world : 1u, pr = part_world > 1 ? (unsigned)part_rank : 0u;
	const unsigned owned = pass ? min(redo_count[0], cap) : (count > pr ? (count - pr + pw - 1) / pw : 0u);
	for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < owned; j += gridDim.x * blockDim.x) {
		const unsigned k = pass ? (unsigned)redo_list[j] : j * pw + pr;
		DevKp kp = kps[k];
		const float T6[6] = {kp.st[0], kp.st[1], kp.st[2], kp.st[4], kp.st[5], kp.st[8]};
		const float w3[3] = {kp.win[0], kp.win[1], kp.win[2]};
		bool near = false;
		kp.code = finish_orientation(kp, T6, w3, max_eig, corner, &near);
		kps[k] = kp;
		codes[k] = kp.code == 1 ? ((kp.level << 4) | 1) : kp.code;  // dense copy for the compaction scan (level in the high bits)
		if (!pass && near) {
			const unsigned pos = atomicAdd(redo_count, 1u);
			if (pos < cap) redo_list[pos] = (int)k;
		}
	}
}

void launch_orient(DevKp *kps, int *codes, const unsigned *d_count, unsigned cap, const LevelRef *d_levels, const WinLut *d_luts,
                   const float *d_lutpool, float max_eig, float corner, int part_rank, int part_world, int *redo_list, unsigned *redo_count,
                   hipStream_t st) {
	// windows differ 3x in volume between the keypoint levels: many short-lived workgroups (1-2 windows per wave) balance better than a
	// resident-sized grid (0.70 vs 0.75 ms at 512^3; S3D_ORI_GRID to measure)
	static const int ori_grid = dev_tune_i("S3D_ORI_GRID", 256 * 32);
	(void)hipMemsetAsync(redo_count, 0, sizeof(unsigned), st);
	hipLaunchKernelGGL(k_orient<false>, dim3(ori_grid), dim3(256), 0, st, kps, codes, d_count, cap, d_levels, d_luts, d_lutpool, max_eig, corner,
	                   part_rank, part_world, (const int *)redo_list, (const unsigned *)redo_count);
	hipLaunchKernelGGL(k_orient_finish, dim3(256 * 8), dim3(64), 0, st, kps, codes, d_count, cap, max_eig, corner, part_rank, part_world, 0,
	                   redo_list, redo_count);
	// accepted and borderline extrema again, with the sums in the reference's order (see the header)
	hipLaunchKernelGGL(k_orient<true>, dim3(ori_grid), dim3(256), 0, st, kps, codes, d_count, cap, d_levels, d_luts, d_lutpool, max_eig, corner,
	                   part_rank, part_world, (const int *)redo_list, (const unsigned *)redo_count);
	hipLaunchKernelGGL(k_orient_finish, dim3(256 * 2), dim3(64), 0, st, kps, codes, d_count, cap, max_eig, corner, part_rank, part_world, 1,
	                   redo_list, redo_count);
}

// Orientation results as kOrientWords int32 words per extremum (code, then the bit patterns of win, eigvalue, eigvector,
// rot, st): rows of extrema this rank did not orient are zero, so an integer all-reduce(SUM) over the ranks of a
// partitioned run restores every row exactly; k_orient_unpack writes the reduced rows back.
static_assert(offsetof(DevKp, st) + sizeof(float) * 9 - offsetof(DevKp, win) == sizeof(float) * (kOrientWords - 1), "DevKp layout");

__global__ void __launch_bounds__(256) k_orient_pack(const DevKp *__restrict__ kps, const unsigned *__restrict__ d_count, unsigned cap,
                                                     int *__restrict__ dst, int part_rank, int part_world) {
	const unsigned count = min(d_count[0], cap);
	const unsigned total = count * (unsigned)kOrientWords;
	for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
		const unsigned k = i / kOrientWords, w = i - k * kOrientWords;
		int v = 0;
		if (part_world <= 1 || (int)(k % (unsigned)part_world) == part_rank)
			v = w == 0 ? kps[k].code : __float_as_int(kps[k].win[w - 1]);  // win..st are contiguous floats
		dst[i] = v;
	}
}

__global__ void __launch_bounds__(256) k_orient_unpack(DevKp *__restrict__ kps, int *__restrict__ codes, const unsigned *__restrict__ d_count,
                                                       unsigned cap, const int *__restrict__ src) {
	const unsigned count = min(d_count[0], cap);
	const unsigned total = count * (unsigned)kOrientWords;
	for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
		const unsigned k = i / kOrientWords, w = i - k * kOrientWords;
		const int v = src[i];
		if (w == 0) { kps[k].code = v; codes[k] = v == 1 ? ((kps[k].level << 4) | 1) : v; }
		else kps[k].win[w - 1] = __int_as_float(v);
	}
}

void launch_orient_pack(const DevKp *kps, const unsigned *d_count, unsigned cap, int *dst, int part_rank, int part_world, hipStream_t st) {
	hipLaunchKernelGGL(k_orient_pack, dim3(1024), dim3(256), 0, st, kps, d_count, cap, dst, part_rank, part_world);
}

void launch_orient_unpack(DevKp *kps, int *codes, const unsigned *d_count, unsigned cap, const int *src, hipStream_t st) {
	hipLaunchKernelGGL(k_orient_unpack, dim3(1024), dim3(256), 0, st, kps, codes, d_count, cap, src);
}

// order-preserving compaction index of the accepted extrema, two launches over 256 waves (r03: one 1024-thread workgroup walked the
// whole list twice, 60 us at 512^3):
//   slot  = exclusive scan of (code == 1) in list order: the keypoint's row in the results (reference order)
//   order = the accepted extrema sorted by keypoint level DESCENDING (stable): the descriptor window volume grows 4x
//           from level 1 to level 3, and handing out the big ones first (longest processing time first) keeps the
//           tail of the descriptor kernel short.  Deterministic (ballot ranks, no atomics), so every rank of a
//           partitioned run derives the same list.
// Wave g of the 256 owns the contiguous range [g * per, (g + 1) * per) and walks it 64 entries at a time (coalesced); k_slots_count
// leaves its totals per key (levels 0..5, key 6 = all accepted) in part[g][], k_slots_write turns them into the wave's start
// positions and writes.  codes[] carries (level << 4) | 1 for accepted extrema, the reject code (< 0) or 0 otherwise.
constexpr int kSlotL = 6;  // levels 0..5 (kMaxKpLevels = 5); key kSlotL = all accepted
constexpr int kSlotBlocks = 64, kSlotWaves = kSlotBlocks * 4;
__device__ __forceinline__ void slot_range(unsigned count, unsigned g, unsigned &lo, unsigned &hi) {
	const unsigned per = ((count + kSlotWaves - 1) / kSlotWaves + 63u) & ~63u;  // per-wave range, multiple of 64
	lo = min(g * per, count); hi = min(lo + per, count);
}
__global__ void __launch_bounds__(256) k_slots_count(const int *__restrict__ codes, const unsigned *__restrict__ d_count, unsigned cap,
                                                     unsigned *__restrict__ part /* [kSlotWaves][kSlotL + 1] */) {
	const unsigned count = min(d_count[0], cap);
	const int lane = threadIdx.x & 63;
	const unsigned g = blockIdx.x * 4 + (threadIdx.x >> 6);
	unsigned lo, hi;
	slot_range(count, g, lo, hi);
	unsigned tot[kSlotL + 1];
#pragma unroll
	for (int l = 0; l <= kSlotL; l++) tot[l] = 0;
	constexpr int kU = 4;  // rows of 64 requested together
	for (unsigned i0 = lo; i0 < hi; i0 += 64 * kU) {
		int c[kU];
#pragma unroll
		for (int q = 0; q < kU; q++) { const unsigned i = i0 + 64 * q + lane; c[q] = i < hi ? codes[i] : 0; }
#pragma unroll
		for (int q = 0; q < kU; q++) {
			const bool acc = c[q] > 0 && (c[q] & 15) == 1;
			const int lv = acc ? min(c[q] >> 4, kSlotL - 1) : -1;
#pragma unroll
			for (int l = 0; l < kSlotL; l++) tot[l] += (unsigned)__popcll(__ballot(lv == l));
			tot[kSlotL] += (unsigned)__popcll(__ballot(acc));
		}
	}
	if (lane == 0)
#pragma unroll
		for (int l = 0; l <= kSlotL; l++) part[g * (kSlotL + 1) + l] = tot[l];
}
__global__ void __launch_bounds__(256) k_slots_write(DevKp *__restrict__ kps, const int *__restrict__ codes, const unsigned *__restrict__ d_count,
                                                     unsigned cap, const unsigned *__restrict__ part, unsigned *__restrict__ d_nkp,
                                                     int *__restrict__ order, unsigned kp_cap) {
	const unsigned count = min(d_count[0], cap);
	const int lane = threadIdx.x & 63;
	const unsigned g = blockIdx.x * 4 + (threadIdx.x >> 6);
	unsigned lo, hi;
	slot_range(count, g, lo, hi);
	const unsigned long long lt = (1ull << lane) - 1ull;
	// per key: the entries of the waves before this one, and of all waves (every wave sums the 256 x 7 table for itself)
	unsigned before[kSlotL + 1], all[kSlotL + 1];
#pragma unroll
	for (int l = 0; l <= kSlotL; l++) {
		unsigned b = 0, a = 0;
#pragma unroll
		for (int q = 0; q < kSlotWaves / 64; q++) {
			const unsigned w = (unsigned)(q * 64 + lane), v = part[w * (kSlotL + 1) + l];
			a += v;
			b += w < g ? v : 0u;
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
		before[l] = b; all[l] = a;
	}
	if (g == 0 && lane == 0) d_nkp[0] = all[kSlotL];
	unsigned run[kSlotL + 1];
	{
		unsigned start = 0;  // higher levels first
#pragma unroll
		for (int l = kSlotL - 1; l >= 0; l--) { run[l] = start + before[l]; start += all[l]; }
		run[kSlotL] = before[kSlotL];
	}
	constexpr int kU = 4;
	for (unsigned i0 = lo; i0 < hi; i0 += 64 * kU) {
		int c[kU];
#pragma unroll
		for (int q = 0; q < kU; q++) { const unsigned i = i0 + 64 * q + lane; c[q] = i < hi ? codes[i] : 0; }
#pragma unroll
		for (int q = 0; q < kU; q++) {
			const unsigned i = i0 + 64 * q + lane;
			const bool acc = c[q] > 0 && (c[q] & 15) == 1;
			const int lv = acc ? min(c[q] >> 4, kSlotL - 1) : -1;
			unsigned pos = 0;
#pragma unroll
			for (int l = 0; l < kSlotL; l++) {
				const unsigned long long m = __ballot(lv == l);
				if (lv == l) pos = run[l] + (unsigned)__popcll(m & lt);
				run[l] += (unsigned)__popcll(m);
			}
			const unsigned long long ma = __ballot(acc);
			if (acc) {
				if (pos < kp_cap) order[pos] = (int)i;
				kps[i].slot = (int)(run[kSlotL] + (unsigned)__popcll(ma & lt));
			} else if (i < hi) kps[i].slot = -1;
			run[kSlotL] += (unsigned)__popcll(ma);
		}
	}
}

size_t slots_scratch_words() { return (size_t)kSlotWaves * (kSlotL + 1); }
void launch_slots(DevKp *kps, const int *codes, const unsigned *d_count, unsigned cap, unsigned *d_nkp, int *order, unsigned kp_cap,
                  unsigned *scratch, hipStream_t st) {
	hipLaunchKernelGGL(k_slots_count, dim3(kSlotBlocks), dim3(256), 0, st, codes, d_count, cap, scratch);
	hipLaunchKernelGGL(k_slots_write, dim3(kSlotBlocks), dim3(256), 0, st, kps, codes, d_count, cap, scratch, d_nkp, order, kp_cap);
}

void preload_orient_kernels() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_slots_count)); }  // (see kernels_march.hip)

}  // namespace s3d
