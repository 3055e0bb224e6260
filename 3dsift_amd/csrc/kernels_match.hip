// kernels_match.hip -- brute-force descriptor matching: dense N x 768 by 768 x M contraction on the
// matrix cores + exact re-scoring.
//
// Restates KP_squareSum + muBruteMatcher::calMatches (reference Src/cMatcher.cc:17-23, 40-79): for
// each ref row i the best and second-best DOT PRODUCT over all target rows j (strict '>', ties keep
// the lower j, both start at FLT_MIN), reported as d = 2 - 2*dot.  The reference accumulates fp32
// products in fp64; an fp32 MFMA chain cannot reproduce that bit-for-bit, so the work is split:
//   k_scores_topk : S = A * B^T with v_mfma_f32_32x32x2_f32 (exact fp32 fma chain), fused running
//                   top-4 per row ordered by (score desc, j asc) -- candidate SELECTION only
//   k_rescore     : the 4 candidates of each row are re-scored exactly like the reference
//                   (fp32 product, fp64 accumulate, k ascending) and the reference's update rule is
//                   replayed over them in ascending j -- bit-identical d1, d2, i1, i2 as long as the
//                   true top-2 are among the fp32 top-4 (score gaps < 1e-6 relative among >= 3 rows
//                   would be needed to break this).
// gfx950 only: 64-lane waves, MFMA C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include <float.h>

#include "sift3d_internal.h"

namespace s3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 32, PITCH = BK + 1, KD = kDesc, TOPK = 4;

__global__ void __launch_bounds__(256) k_scores_topk(const float *__restrict__ A, const int *__restrict__ row_ids, int nrows,
                                                     const float *__restrict__ B, int m, int *__restrict__ cand /*[nrows][TOPK]*/) {
	__shared__ float As[BM * PITCH];
	__shared__ float Bs[BN * PITCH];
	__shared__ float Ss[BM * (BN + 1)];
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const int wr = wid >> 1, wc = wid & 1;  // 2x2 waves, 32x32 each
	const int row0 = blockIdx.x * BM;

	// staging coordinates: 4 threads per row, 8 consecutive floats each
	const int srow = tid >> 2, scol = (tid & 3) * 8;
	int arow = row0 + srow;
	const bool a_ok = arow < nrows;
	const float *Ap = a_ok ? A + (size_t)(row_ids ? row_ids[arow] : arow) * KD : nullptr;

	float best_s[TOPK];
	int best_j[TOPK];
#pragma unroll
	for (int t = 0; t < TOPK; t++) { best_s[t] = -FLT_MAX; best_j[t] = -1; }

	for (int col0 = 0; col0 < m; col0 += BN) {
		f32x16 acc;
#pragma unroll
		for (int r = 0; r < 16; r++) acc[r] = 0.0f;
		const int brow = col0 + srow;
		const float *Bp = brow < m ? B + (size_t)brow * KD : nullptr;
		for (int k0 = 0; k0 < KD; k0 += BK) {
			float4 a0 = make_float4(0, 0, 0, 0), a1 = a0, b0 = a0, b1 = a0;
			if (Ap) { a0 = *reinterpret_cast<const float4 *>(Ap + k0 + scol); a1 = *reinterpret_cast<const float4 *>(Ap + k0 + scol + 4); }
			if (Bp) { b0 = *reinterpret_cast<const float4 *>(Bp + k0 + scol); b1 = *reinterpret_cast<const float4 *>(Bp + k0 + scol + 4); }
			__syncthreads();  // previous chunk fully consumed
			float *as = As + srow * PITCH + scol, *bs = Bs + srow * PITCH + scol;
			as[0] = a0.x; as[1] = a0.y; as[2] = a0.z; as[3] = a0.w; as[4] = a1.x; as[5] = a1.y; as[6] = a1.z; as[7] = a1.w;
			bs[0] = b0.x; bs[1] = b0.y; bs[2] = b0.z; bs[3] = b0.w; bs[4] = b1.x; bs[5] = b1.y; bs[6] = b1.z; bs[7] = b1.w;
			__syncthreads();
			const float *ar = As + (wr * 32 + (lane & 31)) * PITCH + (lane >> 5);
			const float *br = Bs + (wc * 32 + (lane & 31)) * PITCH + (lane >> 5);
#pragma unroll
			for (int kk = 0; kk < BK; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[kk], br[kk], acc, 0, 0, 0);
		}
		// scores -> LDS tile
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
			Ss[(wr * 32 + row) * (BN + 1) + wc * 32 + (lane & 31)] = acc[r];
		}
		__syncthreads();
		// running top-4 of each row, columns in ascending j (strict '>' keeps the lower j on ties)
		if (tid < BM) {
			const int jmax = min(BN, m - col0);
			for (int j = 0; j < jmax; j++) {
				const float s = Ss[tid * (BN + 1) + j];
				if (s > best_s[TOPK - 1]) {
					int pos = TOPK - 1;
#pragma unroll
					for (int t = TOPK - 2; t >= 0; t--)
						if (s > best_s[t]) pos = t;
#pragma unroll
					for (int t = TOPK - 1; t > 0; t--)
						if (t > pos) { best_s[t] = best_s[t - 1]; best_j[t] = best_j[t - 1]; }
#pragma unroll
					for (int t = 0; t < TOPK; t++)
						if (t == pos) { best_s[t] = s; best_j[t] = col0 + j; }
				}
			}
		}
		__syncthreads();
	}
	if (tid < BM && row0 + tid < nrows) {
#pragma unroll
		for (int t = 0; t < TOPK; t++) cand[(size_t)(row0 + tid) * TOPK + t] = best_j[t];
	}
}

// exact re-score + replay of the reference update rule (Src/cMatcher.cc:52-77)
__global__ void __launch_bounds__(256) k_rescore(const float *__restrict__ A, const int *__restrict__ row_ids, int nrows,
                                                 const float *__restrict__ B, const int *__restrict__ cand, float *__restrict__ gd,
                                                 float *__restrict__ sd, int *__restrict__ gi, int *__restrict__ si) {
	// one wave per row: lanes 0..3 each re-score one candidate sequentially (k ascending)
	const int lane = threadIdx.x & 63;
	const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (r >= nrows) return;
	const int row = row_ids ? row_ids[r] : r;
	int j = -1;
	double s = 0.0;
	if (lane < TOPK) {
		j = cand[(size_t)r * TOPK + lane];
		if (j >= 0) {
			const float *a = A + (size_t)row * KD, *b = B + (size_t)j * KD;
			for (int k = 0; k < KD; k++) s += (double)(a[k] * b[k]);
		}
	}
	// gather the 4 (j, s) pairs on lane 0 and replay in ascending j
	int js[TOPK];
	double ss[TOPK];
#pragma unroll
	for (int t = 0; t < TOPK; t++) { js[t] = __shfl(j, t, 64); ss[t] = __shfl(s, t, 64); }
	if (lane == 0) {
#pragma unroll
		for (int a = 1; a < TOPK; a++)
#pragma unroll
			for (int b = TOPK - 1; b >= a; b--) {
				const bool sw = (js[b] >= 0) && (js[b - 1] < 0 || js[b] < js[b - 1]);
				if (sw) { int tj = js[b]; js[b] = js[b - 1]; js[b - 1] = tj; double ts = ss[b]; ss[b] = ss[b - 1]; ss[b - 1] = ts; }
			}
		double d1 = FLT_MIN, d2 = FLT_MIN;
		int i1 = -1, i2 = -1;
#pragma unroll
		for (int t = 0; t < TOPK; t++) {
			if (js[t] < 0) continue;
			const double sij = ss[t];
			if (sij > d1) { d2 = d1; i2 = i1; d1 = sij; i1 = js[t]; }
			else if (sij > d2) { d2 = sij; i2 = js[t]; }
		}
		d2 = 2 - 2 * d2;
		d1 = 2 - 2 * d1;
		gd[row] = (float)d1; sd[row] = (float)d2; gi[row] = i1; si[row] = i2;
	}
}

// rows: optional list of row indices into A (reverse pass over the masked targets only)
int match_rows_device(const float *d_a, const int *d_row_ids, int nrows, const float *d_b, int m, int *d_cand, float *d_gd,
                      float *d_sd, int *d_gi, int *d_si, hipStream_t st) {
	if (nrows <= 0) return SIFT3D_OK;
	hipLaunchKernelGGL(k_scores_topk, dim3((nrows + BM - 1) / BM), dim3(256), 0, st, d_a, d_row_ids, nrows, d_b, m, d_cand);
	hipLaunchKernelGGL(k_rescore, dim3((nrows + 3) / 4), dim3(256), 0, st, d_a, d_row_ids, nrows, d_b, d_cand, d_gd, d_sd, d_gi, d_si);
	return SIFT3D_OK;
}

}  // namespace s3d

// ---------------------------------------------------------------------------------------------
// C-ABI: muBruteMatcher::bijectMatchBase (Src/cMatcher.cc:146-215).  The O(N*M*768) passes run on
// the device; the O(N) bookkeeping (ratio filter, countMatched, toMask, bijectFilter, toCvec) is
// replayed on the host exactly as written in the reference, including its sign-flip quirks.
// ---------------------------------------------------------------------------------------------
#include <algorithm>
#include <vector>

using namespace s3d;

static void ratio_filter(std::vector<int> &gi, const std::vector<float> &gd, const std::vector<float> &sd, double thresHold) {
	const double t2 = thresHold * thresHold;  // filter, Src/cMatcher.cc:81-97
	for (size_t i = 0; i < gi.size(); i++) {
		if (gi[i] < 0) continue;
		if ((double)(gd[i] / sd[i]) >= t2) gi[i] *= -1;
	}
}

extern "C" int sift3d_match(const float *ref_desc, const float *ref_xyz, int n, const float *tar_desc, const float *tar_xyz, int m,
                            double thresHold, int mode, int on_device, int device, int *gIdx, int *sIdx, float *gDist,
                            float *sDist, float *pairs6, int *npairs, double *seconds) {
	if (n < 0 || m < 0 || mode < 1 || mode > 3 || (n > 0 && (!ref_desc || !ref_xyz)) || (m > 0 && (!tar_desc || !tar_xyz))) {
		set_last_error("sift3d_match: bad argument");
		return SIFT3D_ERR_ARG;
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_last_error("no HIP device visible: no CPU fallback"); return SIFT3D_ERR_NO_DEVICE; }
	if (device < 0 || device >= ndev) return SIFT3D_ERR_ARG;
	S3D_HIP(hipSetDevice(device));
	if (npairs) *npairs = 0;
	if (seconds) *seconds = 0;

	std::vector<float> gd(n, 0.f), sd(n, 0.f), gd2(m, 0.f), sd2(m, 0.f);
	std::vector<int> gi(n, -1), si(n, -1), gi2(m, -1), si2(m, -1);
	std::vector<float> hx, hy;  // host copies of coordinates when inputs are device resident
	const float *rx = ref_xyz, *tx = tar_xyz;

	float *d_a = nullptr, *d_b = nullptr, *d_f = nullptr;
	int *d_i = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	int rc = SIFT3D_OK;
#define MCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); rc = SIFT3D_ERR_HIP; goto done; } } while (0)
	{
		const size_t nn = (size_t)std::max(n, 1), mm = (size_t)std::max(m, 1), big = std::max(nn, mm);
		if (on_device) {
			d_a = const_cast<float *>(ref_desc); d_b = const_cast<float *>(tar_desc);
			hx.resize((size_t)n * 3); hy.resize((size_t)m * 3);
			if (n) MCHK(hipMemcpy(hx.data(), ref_xyz, sizeof(float) * 3 * n, hipMemcpyDeviceToHost));
			if (m) MCHK(hipMemcpy(hy.data(), tar_xyz, sizeof(float) * 3 * m, hipMemcpyDeviceToHost));
			rx = hx.data(); tx = hy.data();
		} else {
			MCHK(hipMalloc(&d_a, sizeof(float) * kDesc * nn));
			MCHK(hipMalloc(&d_b, sizeof(float) * kDesc * mm));
			if (n) MCHK(hipMemcpy(d_a, ref_desc, sizeof(float) * kDesc * n, hipMemcpyHostToDevice));
			if (m) MCHK(hipMemcpy(d_b, tar_desc, sizeof(float) * kDesc * m, hipMemcpyHostToDevice));
		}
		MCHK(hipMalloc(&d_f, sizeof(float) * 2 * big));
		MCHK(hipMalloc(&d_i, sizeof(int) * (2 + TOPK + 1) * big));
		float *d_gd = d_f, *d_sd = d_f + big;
		int *d_gi = d_i, *d_si = d_i + big, *d_cand = d_i + 2 * big, *d_rows = d_i + (2 + TOPK) * big;
		MCHK(hipEventCreate(&e0));
		MCHK(hipEventCreate(&e1));
		MCHK(hipEventRecord(e0, nullptr));

		// ---- ref -> tar ----
		if (n > 0 && m > 0) {
			match_rows_device(d_a, nullptr, n, d_b, m, d_cand, d_gd, d_sd, d_gi, d_si, nullptr);
			MCHK(hipMemcpy(gd.data(), d_gd, sizeof(float) * n, hipMemcpyDeviceToHost));
			MCHK(hipMemcpy(sd.data(), d_sd, sizeof(float) * n, hipMemcpyDeviceToHost));
			MCHK(hipMemcpy(gi.data(), d_gi, sizeof(int) * n, hipMemcpyDeviceToHost));
			MCHK(hipMemcpy(si.data(), d_si, sizeof(int) * n, hipMemcpyDeviceToHost));
		} else if (n > 0) {
			// no targets: every dot loop is empty -> d = 2 - 2*FLT_MIN, idx -1
			for (int i = 0; i < n; i++) { gd[i] = sd[i] = (float)(2 - 2 * (double)FLT_MIN); }
		}
		ratio_filter(gi, gd, sd, thresHold);

		if (mode != 1) {
			const int mask_thres = (mode == 2) ? 0 : 1;
			std::vector<int> cnt(m, 0);
			for (int i = 0; i < n; i++) if (gi[i] >= 0) cnt[gi[i]] += 1;               // countMatched :114-120
			std::vector<int> rows;
			for (int j = 0; j < m; j++) { cnt[j] = cnt[j] > mask_thres ? 1 : 0; if (cnt[j]) rows.push_back(j); }  // toMask :122-131
			// ---- tar -> ref over the masked targets only (masked-out rows keep gIdx2 = -1) ----
			if (!rows.empty() && n > 0) {
				MCHK(hipMemcpy(d_rows, rows.data(), sizeof(int) * rows.size(), hipMemcpyHostToDevice));
				match_rows_device(d_b, d_rows, (int)rows.size(), d_a, n, d_cand, d_gd, d_sd, d_gi, d_si, nullptr);
				std::vector<float> tg(m), ts(m);
				std::vector<int> ti(m), tsi(m);
				MCHK(hipMemcpy(tg.data(), d_gd, sizeof(float) * m, hipMemcpyDeviceToHost));
				MCHK(hipMemcpy(ts.data(), d_sd, sizeof(float) * m, hipMemcpyDeviceToHost));
				MCHK(hipMemcpy(ti.data(), d_gi, sizeof(int) * m, hipMemcpyDeviceToHost));
				MCHK(hipMemcpy(tsi.data(), d_si, sizeof(int) * m, hipMemcpyDeviceToHost));
				for (int j : rows) { gd2[j] = tg[j]; sd2[j] = ts[j]; gi2[j] = ti[j]; si2[j] = tsi[j]; }
			}
			ratio_filter(gi2, gd2, sd2, thresHold);
			for (int i = 0; i < n; i++) {                                              // bijectFilter :133-144
				const int j = gi[i];
				if (j < 0 || cnt[j] == 0) continue;
				if (gi2[j] != i) gi[i] *= -1;
			}
		}
		MCHK(hipEventRecord(e1, nullptr));
		MCHK(hipEventSynchronize(e1));
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		if (seconds) *seconds = (double)ms * 1e-3;

		int np = 0;
		for (int i = 0; i < n; i++) {                                                  // toCvec :99-112
			const int j = gi[i];
			if (j < 0) continue;
			if (pairs6) {
				for (int c = 0; c < 3; c++) { pairs6[(size_t)np * 6 + c] = rx[(size_t)i * 3 + c]; pairs6[(size_t)np * 6 + 3 + c] = tx[(size_t)j * 3 + c]; }
			}
			np++;
		}
		if (npairs) *npairs = np;
		for (int i = 0; i < n; i++) {
			if (gIdx) gIdx[i] = gi[i];
			if (sIdx) sIdx[i] = si[i];
			if (gDist) gDist[i] = gd[i];
			if (sDist) sDist[i] = sd[i];
		}
	}
done:
#undef MCHK
	if (e0) hipEventDestroy(e0);
	if (e1) hipEventDestroy(e1);
	if (!on_device) { hipFree(d_a); hipFree(d_b); }
	hipFree(d_f);
	hipFree(d_i);
	return rc;
}
