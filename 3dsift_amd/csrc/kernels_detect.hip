// kernels_detect.hip -- DoG extrema scan with deterministic, order-preserving compaction.
//
// Restates Detect_KeyPoints + IsExtrema_neighbor (reference Src/cSIFT3D.cc:362-425, 884-911):
// for x,y,z in [1, n-2]: (val > thr || val < -thr) and val strictly below (or strictly above) all
// EIGHT neighbours: +-x, +-y, +-z in the current DoG level and the centre voxel of the previous and
// next level.  thr = peak_thresh * max|level| (max from the DoG-producing kernel, fp32 multiply).
//
// The reference emits extrema in (octave, level, z, y, x) scan order and the matcher output order
// depends on it, so instead of an atomic append + sort the scan is made order-preserving.  One launch
// of each kernel covers the THREE keypoint levels of an octave:
//   k_mark : block = (level, z, kRows rows) -- a contiguous range in scan order; lanes run along x (coalesced, no
//            index divisions); voxels above the peak threshold are queued per wave and their eight neighbours gathered 64
//            candidates at a time; each wave stores the 64-bit ballots of its rows, each block its hit count
//   k_lazy_next : the last Gaussian level of an octave is not built; the candidates of the last keypoint level that passed seven
//            tests are parked by k_mark and get the missing neighbour value here (one workgroup per candidate)
//   k_scan : one workgroup turns the block counts into exclusive offsets on top of the running total
//   k_emit : one thread per ballot word; words with hits (rare) write their extrema at offset + rank
// k_mark + k_lazy_next of octaves >= 1 may run on a second stream with their own scratch (launch_detect_mark); k_scan + k_emit
// (launch_detect_emit) run in octave order on one stream.  No host synchronisation anywhere; the running total stays on the device.
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <algorithm>

#include "sift3d_internal.h"

namespace s3d {

#ifndef S3D_DET_ROWS
#define S3D_DET_ROWS 32  /* rows per block: 32 measured best (detect 0.93 -> 0.84 ms at 512^3; 64: 1.01) */
#endif
#ifndef S3D_DET_QUEUE
#define S3D_DET_QUEUE 640  /* 640: the queue is drained after a batch of 8 ballot words; 128: after every word (1.06 vs 0.99 ms detection) */
#endif
// development diagnostics, timing only (wrong results; never set in the product build): 1 no neighbour gathers, 2 nothing is queued
#ifndef S3D_DETDIAG
#define S3D_DETDIAG 0
#endif
#ifndef S3D_DET_LEAN
#define S3D_DET_LEAN 1
#endif
#ifndef S3D_DET_XPRE
#define S3D_DET_XPRE 0  /* generic row loop, 1: the two x neighbours are read with the centre values (two more coalesced loads per word) and tested before a voxel is queued -- no gain on the blob volumes (noise makes two voxels out of three x extrema) */
#endif
constexpr int kRows = S3D_DET_ROWS;   // rows per block
constexpr int kThreads = 256;

// masks layout: word index = ((lvl * nz + z) * ny + y) * wpr + xw, wpr = ceil(nx / 64): scan order
//
// Two-phase per wave: (1) stream the centre values, kBatch ballot words per pass, and push the voxels that pass the peak
// threshold (a few per cent) into a wave-private LDS queue by ballot rank; (2) whenever 64 candidates are queued, all 64
// lanes fetch the 8 neighbours of one candidate each and OR their verdict into the wave's LDS copy of the mask words.
// A per-word `if (candidate) { 8 loads }` issues those 8 load instructions for nearly every word (some lane usually is a
// candidate) and the compiler drains them at each join: one memory round trip per word.
constexpr int kQueue = S3D_DET_QUEUE;  // >= 63 left over + the lanes pushed before the next drain (one word: 64, a whole batch: 512)
__global__ void __launch_bounds__(kThreads) k_mark(DetectLevels L, int nx, int ny, ZRange zr, int nyb, float peak_thresh,
                                                   unsigned long long *__restrict__ masks, unsigned *__restrict__ block_counts,
                                                   unsigned *__restrict__ prov, unsigned *__restrict__ prov_count, unsigned prov_cap,
                                                   unsigned *__restrict__ total) {
	__shared__ unsigned s_cnt[kThreads / 64];
	__shared__ uint2 s_q[kThreads / 64][kQueue];  // (value bits, (row within the wave) << 12 | word << 6 | lane bit): one 8-byte LDS write per candidate
	// [wave][row of the wave][word of the segment]: dynamic, sized by the launcher for min(wpr, 64) words per row -- with the queue it
	// is what limits the workgroups per CU, and the kernel is bound by memory latency x waves in flight
	extern __shared__ unsigned long long s_mask_dyn[];
	const int segw = min((nx + 63) >> 6, 64);
	const int b = blockIdx.x;
	const int nz = zr.zo1 - zr.zo0;                              // planes scanned by this launch (local range [zo0, zo1))
	const int yb = b % nyb, zi = (b / nyb) % nz, lvl = b / (nyb * nz);
	const int z = zr.zo0 + zi;                                   // local plane in the buffers
	const int zg = z + zr.zoff;                                  // global plane (border rule, Src/cSIFT3D.cc:388)
	const float *__restrict__ cur = L.cur[lvl];
	const float *__restrict__ prev = L.prev[lvl];
	const float *__restrict__ next = L.next[lvl];
	const bool lazy_prev = lvl == 0 && L.prev0_hi != nullptr, lazy_next = lvl == L.nextl_slot && L.nextl_hi != nullptr;
	const bool lazy_g = lvl == L.nextl_slot && L.lazy_src != nullptr;  // the level behind `next` does not exist: park the candidates
	const float thr = peak_thresh * __uint_as_float(*L.absmax_bits[lvl]);
	const int wpr = (nx + 63) >> 6;
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
	const bool z_in = (zg >= 1 && zg <= zr.nzg - 2);
	const int y0 = yb * kRows;
	constexpr int kBatch = 8;
	const int nrows = min(kRows, ny - y0);
	const int swid = __builtin_amdgcn_readfirstlane(wid);
	const int r_lo = swid * (kRows / 4), r_hi = min(nrows, (swid + 1) * (kRows / 4));
	uint2 *q = s_q[swid];
	unsigned long long *mloc = s_mask_dyn + (size_t)swid * (kRows / 4) * segw;
	for (int i = lane; i < (kRows / 4) * segw; i += 64) mloc[i] = 0ull;
	const unsigned long long lt = (1ull << lane) - 1ull;
	const size_t plane0 = sz * (size_t)z + sy * (size_t)y0;
	int qn = 0;  // wave-uniform
	const bool lean = S3D_DET_LEAN != 0;  // the lean row loop (below); S3D_DET_LEAN=0 builds keep the generic one for A/B runs
	// evaluate `n` queued candidates starting at entry `first` (n <= 64)
	auto evaluate = [&](int first, int n, int seg0) {
		if (S3D_DETDIAG & 1) {  // no gathers: the queue entry alone decides (keeps the queue traffic alive)
			const uint2 qd = q[first + min(lane, n - 1)];
			if (lane < n && __uint_as_float(qd.x) == 12345.678f) atomicOr(&mloc[0], 1ull);
			return;
		}
		const bool in_q = lane < n;
		const int e = first + (in_q ? lane : 0);
		const uint2 qe = q[e];
		const float v = __uint_as_float(qe.x);
		const unsigned id = qe.y;                        // (row within the wave) << 12 | word << 6 | lane bit
		const int rr = (int)((id >> 12) & 15), xw = (int)((id >> 6) & 63), bit = (int)(id & 63);
		static_assert(kRows / 4 <= 16, "four bits for the row of a wave");
		// (the lean row loop queues by the peak threshold alone: the first and the last voxel of a row are dropped here)
		const int xq = (seg0 + xw) * 64 + bit;
		const bool act = in_q && xq >= 1 && xq <= nx - 2;
		const size_t i = plane0 + sy * (size_t)(r_lo + rr) + (size_t)xq;
		// idle lanes read voxel (1, 1) of the plane: candidates only exist on interior planes of volumes with ny, nx >= 3, so all
		// eight neighbour addresses of that voxel are inside the level
		const size_t ic = act ? i : sz * (size_t)z + sy + 1;
		// elided first / last DoG level: formed from the two Gaussian levels like Sub does (block-uniform choice)
		const float n0 = lazy_prev ? (L.prev0_hi[ic] - L.prev0_lo[ic]) * (-1.0f) : prev[ic];
		// lean row loop: the four in-plane neighbours were tested in registers (bit 16 of the entry: the voxel is a strict MAXIMUM among
		// them, else a strict minimum); only the neighbours in z and in scale are gathered
		const bool pre = lean;  // kernel-uniform
		const bool dmax = (id >> 16) & 1u;
		float n1 = 0.f, n2 = 0.f, n3 = 0.f, n4 = 0.f;
		if (!pre) { n1 = cur[ic - 1]; n2 = cur[ic + 1]; n3 = cur[ic + sy]; n4 = cur[ic - sy]; }
		const float n5 = cur[ic + sz], n6 = cur[ic - sz];
		if (lazy_g) {  // block-uniform
			const bool mn7 = pre ? (!dmax && v < n0 && v < n5 && v < n6) : (v < n0 && v < n1 && v < n2 && v < n3 && v < n4 && v < n5 && v < n6);
			const bool mx7 = pre ? (dmax && v > n0 && v > n5 && v > n6) : (v > n0 && v > n1 && v > n2 && v > n3 && v > n4 && v > n5 && v > n6);
			// one counter increment per wave (a single word takes ~90 atomics per microsecond: one per candidate cost 0.45 ms)
			const bool park = act && (mn7 || mx7);
			const unsigned long long pm = __ballot(park);
			if (pm) {
				unsigned base = 0;
				if (lane == 0) base = atomicAdd(prov_count, (unsigned)__popcll(pm));
				base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
				if (park) {
					const unsigned slot = base + (unsigned)__popcll(pm & lt);
					if (slot < prov_cap) prov[slot] = (unsigned)i | (mx7 ? 0x80000000u : 0u);
					else total[1] = 1u;  // overflow: the host regrows the lists and reruns
				}
			}
			return;
		}
		const float n7 = lazy_next ? (L.nextl_hi[ic] - L.nextl_lo[ic]) * (-1.0f) : next[ic];
		const bool mn = pre ? (!dmax && v < n0 && v < n5 && v < n6 && v < n7) : (v < n0 && v < n1 && v < n2 && v < n3 && v < n4 && v < n5 && v < n6 && v < n7);
		const bool mx = pre ? (dmax && v > n0 && v > n5 && v > n6 && v > n7) : (v > n0 && v > n1 && v > n2 && v > n3 && v > n4 && v > n5 && v > n6 && v > n7);
		if (act && (mn || mx)) atomicOr(&mloc[rr * segw + xw], 1ull << bit);
	};
	unsigned cnt = 0;
	// rows wider than 64 ballot words (nx > 4096) are handled in segments of 64 words: the local mask copy holds one segment
	for (int seg0 = 0; seg0 < wpr; seg0 += 64) {
		const int seg1 = min(wpr, seg0 + 64);
		// r03 -- the lean row loop.  Timing-only builds showed the kernel bound by INSTRUCTION ISSUE, not by
		// memory: with nothing queued and no gather it still took 0.48 of its 0.52 ms at 512^3, ~25 vector + ~26 scalar instructions per
		// 64 voxels (per-lane border predicates, clamped 64-bit addresses, two threshold compares, queue bookkeeping).  Here a word
		// costs three loads (uniform row pointer + lane offset + immediate: the row below, and the row itself shifted by one voxel
		// either way -- L1 hits), seven vector instructions for the tests, the push (two mbcnt, one address, one 8-byte LDS write) and a
		// few scalar ones; border rows / planes are skipped as a whole, the x borders are dropped by evaluate().
		// The FOUR IN-PLANE NEIGHBOURS are tested here, in registers: a wave walks its rows top down and keeps the rows above and
		// below (8 + 2 rows read for 8).  With the threshold alone 10 % of the voxels of the blob volumes were queued and their eight
		// gathers (64 different cache lines per instruction) cost 0.19 of the remaining 0.48 ms; now two candidates out of five
		// survive (noise: a voxel is an extremum of five with probability 2/5) and four neighbours are left to gather.
		if (lean && z_in) {
			// rows of this wave with 1 <= y <= ny-2 (a contiguous range; the rows above and below it exist)
			const int ry_a = max(r_lo, 1 - y0), ry_b = min(r_hi, ny - 1 - y0);
			const int nws = seg1 - seg0;
			const float *__restrict__ pl = cur + sz * (size_t)z + (size_t)seg0 * 64;
			// one word: v against the peak threshold and against its four in-plane neighbours (up / down rows, left / right voxels)
			auto word = [&](float v, float u, float d, float xl, float xr, int wseg, unsigned idrow) {
				const float hi = fmaxf(fmaxf(u, d), fmaxf(xl, xr)), lo = fminf(fminf(u, d), fminf(xl, xr));
				const bool gmax = v > hi;
				const bool c = fabsf(v) > thr && (gmax || v < lo) && !((S3D_DETDIAG & 2) && v != 12345.678f);  // |v| > thr == (v > thr || v < -thr)
				const unsigned long long m = __ballot(c);
				if (c) {
					const unsigned pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)qn));
					q[pos] = make_uint2(__float_as_uint(v), idrow | (gmax ? 0x10000u : 0u) | (unsigned)(wseg << 6) | (unsigned)lane);
				}
				qn += (int)__popcll(m);
			};
			// any row width: the word index of a batch is clamped to the row's last word and the lane offset to its last voxel (the
			// lanes / words beyond repeat it; words beyond the row are not scanned, lanes beyond it are dropped by evaluate())
			const unsigned omax = (unsigned)(nx - 1 - seg0 * 64);
			for (int w0 = 0; w0 < nws && ry_a < ry_b; w0 += kBatch) {
				unsigned off[kBatch];
#pragma unroll
				for (int bb = 0; bb < kBatch; bb++) off[bb] = min((unsigned)(min(w0 + bb, nws - 1) * 64 + lane), omax);
				// r05: the x neighbours of a row come from the LANES beside (DPP wave shifts of the row's own registers) instead of two more
				// loads of the row shifted by one voxel: the kernel's vector-memory pipe was busy for the whole launch (TA_BUSY = kernel time,
				// profiles/r05f_pmc_k_mark.json: 28.6 M load instructions per step, three per word and row) while its VALU idled at 30 %.  Only the
				// batch's two outer voxels -- x - 1 of its first word, x + 1 of its last -- are loaded, by lanes 0 and 1 of one more load per row.
				const int eoff = (lane & 1) ? (int)min((unsigned)((w0 + kBatch) * 64), omax) : w0 * 64 - 1;
				float up[kBatch], mid[kBatch], emid;
				{
					const float *ra = pl + sy * (size_t)(y0 + ry_a - 1), *rb = ra + sy;
#pragma unroll
					for (int bb = 0; bb < kBatch; bb++) { up[bb] = ra[off[bb]]; mid[bb] = rb[off[bb]]; }
					emid = rb[eoff];  // (x - 1 of the row's first voxel: the row above's last element, in bounds; evaluate() drops that voxel)
				}
				for (int ry = ry_a; ry < ry_b; ry++) {
					const float *rd = pl + sy * (size_t)(y0 + ry + 1);
					float dn[kBatch], xl[kBatch], xr[kBatch];
#pragma unroll
					for (int bb = 0; bb < kBatch; bb++) dn[bb] = rd[off[bb]];
					const float edn = rd[eoff];
#pragma unroll
					for (int bb = 0; bb < kBatch; bb++) {
						// lane l <- lane l - 1 (lane 0: the last voxel of the word before) and lane l <- lane l + 1 (lane 63: the first voxel of the word behind)
						const int le = __builtin_amdgcn_readlane(__float_as_int(bb == 0 ? emid : mid[bb > 0 ? bb - 1 : 0]), bb == 0 ? 0 : 63);
						const int re = __builtin_amdgcn_readlane(__float_as_int(bb == kBatch - 1 ? emid : mid[bb < kBatch - 1 ? bb + 1 : 0]), bb == kBatch - 1 ? 1 : 0);
						xl[bb] = __int_as_float(__builtin_amdgcn_update_dpp(le, __float_as_int(mid[bb]), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
						xr[bb] = __int_as_float(__builtin_amdgcn_update_dpp(re, __float_as_int(mid[bb]), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
					}
					const unsigned idrow = (unsigned)((ry - r_lo) << 12);
#pragma unroll
					for (int bb = 0; bb < kBatch; bb++)
						if (w0 + bb < nws) word(mid[bb], up[bb], dn[bb], xl[bb], xr[bb], w0 + bb, idrow);  // wave-uniform
					while (qn >= 64) {  // entries are consumed from the END so the front stays in place
						qn -= 64;
						evaluate(qn, 64, seg0);
					}
#pragma unroll
					for (int bb = 0; bb < kBatch; bb++) { up[bb] = mid[bb]; mid[bb] = dn[bb]; }
					emid = edn;
				}
			}
		}
		for (int ry = r_lo; !lean && ry < r_hi; ry++) {
			const int y = y0 + ry;
			const bool y_in = z_in && y >= 1 && y <= ny - 2;
			const size_t rowbase = sy * (size_t)y + sz * (size_t)z;
			for (int xw0 = seg0; xw0 < seg1; xw0 += kBatch) {
				float val[kBatch], vxm[kBatch], vxp[kBatch];
				bool in[kBatch];
#pragma unroll
				for (int bb = 0; bb < kBatch; bb++) {
					const int x = (xw0 + bb) * 64 + lane;
					in[bb] = y_in && xw0 + bb < seg1 && x >= 1 && x <= nx - 2;
					const float *a = cur + (in[bb] ? rowbase + (size_t)x : sz * (size_t)z + 1);  // unconditional loads, clamped address
					val[bb] = a[0];
					if (S3D_DET_XPRE) { vxm[bb] = a[-1]; vxp[bb] = a[1]; }
				}
#pragma unroll
				for (int bb = 0; bb < kBatch; bb++) {
					const float v = val[bb];
					// r03: the gathers of the queued voxels (8 x 64 different cache lines per 64 candidates, 10 % of the voxels of the blob
					// volumes pass the peak threshold) kept the texture addresser busy for the whole kernel (TA_BUSY = kernel time, 438 M cache
					// accesses per 512^3); a voxel that is not a strict extremum among its two x neighbours cannot pass the full test
					const bool xext = !S3D_DET_XPRE || (((int)(v > vxm[bb]) & (int)(v > vxp[bb])) | ((int)(v < vxm[bb]) & (int)(v < vxp[bb]))) != 0;
					const bool c = in[bb] && (v > thr || v < -thr) && xext && !((S3D_DETDIAG & 2) && v != 12345.678f);
					const unsigned long long m = __ballot(c);
					if (c) {
						const int pos = qn + (int)__popcll(m & lt);
						q[pos] = make_uint2(__float_as_uint(v), (unsigned)(((ry - r_lo) << 12) | ((xw0 + bb - seg0) << 6) | lane));
					}
					qn += (int)__popcll(m);
					if (kQueue < 63 + kBatch * 64 && qn >= 64) {  // small queue: drain after every word
						qn -= 64;
						evaluate(qn, 64, seg0);
					}
				}
				// drain full waves of candidates (entries are consumed from the END so the front stays in place)
				while (qn >= 64) {
					qn -= 64;
					evaluate(qn, 64, seg0);
				}
			}
		}
		if (qn > 0) evaluate(0, qn, seg0);
		qn = 0;
		// masks + count of this wave's rows, then clear the local copy for the next segment
		for (int ry = r_lo; ry < r_hi; ry++) {
			unsigned long long *mrow = masks + ((size_t)(lvl * nz + zi) * ny + (y0 + ry)) * wpr;
			for (int xw = seg0 + lane; xw < seg1; xw += 64) {
				const unsigned long long m = mloc[(ry - r_lo) * segw + xw - seg0];
				mrow[xw] = m;
				cnt += (unsigned)__popcll(m);
				mloc[(ry - r_lo) * segw + xw - seg0] = 0ull;
			}
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
	if (lane == 0) s_cnt[wid] = cnt;
	__syncthreads();
	if (threadIdx.x == 0) block_counts[b] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// Parked candidates of the last keypoint level (DetectLevels::lazy_src): one workgroup evaluates the missing Gaussian level at one
// voxel -- x-blur of every (row, plane) the y and z taps reach, y-blur per plane, z-blur -- with the code of k_conv_axis (interior
// chain where the output is interior along the axis, boundary_term for every tap otherwise), forms the DoG value like Sub and
// finishes the extremum test; hits are ORed into the ballot words and counted into their block, before k_scan runs.
__device__ __forceinline__ void lazy_tap_src(int p, int d, int n, int &lo, int &hi, float &frac, bool interior) {
	if (interior) { lo = hi = p - d; frac = 0.0f; return; }
	const int dim_end = n - 1;
	float c = (float)p - (float)d * 1.0f;
	if (c < 0.0f) c = -1.0f * c;
	else if (c >= (float)dim_end) c = (float)(2 * dim_end) - c - 0.1f;
	lo = (int)c;
	frac = c - (float)lo;
	hi = lo + 1;
	lo = min(max(lo, 0), dim_end);  // same clamp as boundary_term (kernels_pyramid.hip)
	hi = min(max(hi, 0), dim_end);
}
// k_lazy_next<LT>: LT = taps at most.  17 (half width <= 8: the default parameters' last level; 24.6 KB of LDS, six workgroups per CU) and,
// r06, 25 (half width <= 12: the last level of sigma_default up to 2.6 -- it used to be BUILT by three separable passes, the single most
// expensive level of such a run, because the lazy form stopped at 8; 73 KB, two workgroups per CU, a few thousand candidates per volume)
static_assert(kLazySlots / 2 - 1 == 25, "k_lazy_next's wide instantiation covers kLazySlots");
// ---- r05: k_lazy_wave -- ONE WAVE per parked candidate that is interior along all three axes, default half width 8 ----
// (r02-r04: one 256-thread workgroup per candidate, 189 registers, two workgroups per CU, five barriers, y-blur on 17 threads and z-blur
// on one: ~15 us per candidate with 512 candidates in flight -- 1.3 ms of kernel time per 512^3 step over three queues, 0.29 ms of it on
// the detection stage's critical path.)  The 17 x 17 rows of 17 samples the three passes reach travel global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no registers, all 23 instructions of a candidate in flight at once): a row is staged as FIVE 16-byte pieces
// (20 floats: the 17 samples and three more, shifted left where they would cross the end of the row), piece p = 5 * row + q of the
// wave's buffer, lane constants for the 23 offsets.  Then 289 x-chains (lane = row, five rounds), 17 y-chains, one z-chain: the chains
// of k_conv_axis's interior rule, acc = acc + tap[k] * sample(p - (k - hw)) for k = 0 .. 2 hw, bit for bit; a row's result overwrites the
// row's first sample (only the row's own lane reads it).  23.5 KB of LDS per wave, two waves per workgroup, three workgroups per CU.
constexpr int kLwN = 17, kLwPieces = kLwN * kLwN * 5, kLwInstr = (kLwPieces + 63) / 64, kLwFloats = kLwInstr * 256, kLwWaves = 2;
__global__ void __launch_bounds__(64 * kLwWaves) k_lazy_wave(DetectLevels L, Taps t, int nx, int ny, ZRange zr, int nyb,
                                                            const unsigned *__restrict__ prov, const unsigned *__restrict__ prov_count,
                                                            unsigned prov_cap, unsigned long long *__restrict__ masks,
                                                            unsigned *__restrict__ block_counts) {
	__shared__ __attribute__((aligned(1024))) float s_buf[kLwWaves][kLwFloats];
	constexpr int H = 8, N = kLwN;
	const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	float *blk = s_buf[wid];
	const unsigned lds_base = (unsigned)(size_t)blk;  // LDS byte address of the wave's buffer
	const unsigned count = min(*prov_count, prov_cap);
	const int lvl = L.nextl_slot;
	const float *__restrict__ src = L.lazy_src;
	const float *__restrict__ cur = L.cur[lvl];
	const int wpr = (nx + 63) >> 6, nzs = zr.zo1 - zr.zo0, nzg = zr.nzg;
	const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
	// byte offset of this lane's piece of DMA instruction j from the block's first sample (lane constants: the block's shape is fixed)
	unsigned voff[kLwInstr];
#pragma unroll
	for (int j = 0; j < kLwInstr; j++) {
		const int p = min(64 * j + lane, kLwPieces - 1), row = p / 5, q = p - 5 * row, zi = row / N, yi = row - zi * N;
		voff[j] = (unsigned)(((size_t)zi * sz + (size_t)yi * sy + (size_t)(4 * q)) * sizeof(float));
	}
	const unsigned nwaves = gridDim.x * kLwWaves;
	for (unsigned e = blockIdx.x * kLwWaves + (unsigned)wid; e < count; e += nwaves) {  // wave-uniform
		const unsigned ent = prov[e];
		const size_t ic = (size_t)(ent & 0x7FFFFFFFu);
		const bool as_max = (ent >> 31) != 0;
		const int zl = (int)(ic / sz), rem = (int)(ic - (size_t)zl * sz), y = rem / nx, x = rem - y * nx;
		const int zg = zl + zr.zoff;  // global plane
		if (!(x >= H && x <= nx - 2 - H && y >= H && y <= ny - 2 - H && zg >= H && zg <= nzg - 2 - H)) continue;  // k_lazy_next's
		// the 20 floats of a row start at x - 8 - shift: shift = how far x + 11 would reach beyond the row's last sample
		// (never further than the row's first sample: rows shorter than 20 floats let the piece run into the NEXT row instead)
		const int shift = min(max(0, x + 11 - (nx - 1)), x - H);
		const float *base = src + ic - (size_t)H * sz - (size_t)H * sy - (size_t)(H + shift);
#pragma unroll
		for (int j = 0; j < kLwInstr; j++) {
			unsigned keep;
			asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
			             : "=&s"(keep) : "v"(voff[j]), "s"(base), "s"(lds_base + 1024u * (unsigned)j) : "memory");
		}
		const float centre = src[ic], v = cur[ic];
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_wave_barrier();
		// x-blur: tap k reads x - (k - H), i.e. sample 2H - k of the row (+ shift)
#pragma unroll
		for (int rq = 0; rq < (N * N + 63) / 64; rq++) {
			const int row = min(lane + 64 * rq, N * N - 1);
			const float *r = blk + row * 20 + shift;
			float acc = 0.0f;
#pragma unroll
			for (int k = 0; k < N; k++) acc = acc + t.w[k] * r[2 * H - k];
			if (lane + 64 * rq < N * N) blk[row * 20] = acc;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		{  // y-blur per plane: lanes 0 .. 16 (the others repeat plane 16's chain and store nothing)
			const int pl = min(lane, N - 1);
			const float *c = blk + pl * N * 20;
			float acc = 0.0f;
#pragma unroll
			for (int k = 0; k < N; k++) acc = acc + t.w[k] * c[(2 * H - k) * 20];
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			if (lane < N) blk[pl * N * 20] = acc;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		{
			float acc = 0.0f;
#pragma unroll
			for (int k = 0; k < N; k++) acc = acc + t.w[k] * blk[(2 * H - k) * N * 20];
			const float n7 = (acc - centre) * (-1.0f);  // Sub, Src/cSIFT3D.cc:875
			if (lane == 0 && (as_max ? (v > n7) : (v < n7))) {
				const int zi = zl - zr.zo0;
				atomicOr(&masks[((size_t)(lvl * nzs + zi) * ny + y) * wpr + (x >> 6)], 1ull << (x & 63));
				atomicAdd(&block_counts[(lvl * nzs + zi) * nyb + y / kRows], 1u);
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();  // the buffer is rewritten by the next candidate's DMA
	}
}

// the other parked candidates (next to a border of the level, or half widths other than 8): one WORKGROUP per candidate
template <int kLT>
__global__ void __launch_bounds__(256) k_lazy_next(DetectLevels L, Taps t, int nx, int ny, ZRange zr, int nyb,
                                                   const unsigned *__restrict__ prov, const unsigned *__restrict__ prov_count,
                                                   unsigned prov_cap, unsigned long long *__restrict__ masks,
                                                   unsigned *__restrict__ block_counts, int skip_interior) {
	// the block of real samples the three passes reach (at most 18^3 voxels, rows contiguous in memory) is staged in LDS with
	// coalesced loads, then x-blur per (row, plane), y-blur per plane, z-blur
	constexpr int kLR = kLT + 1;              // real samples an axis reads: one contiguous range of at most 2 hw + 2 indices
	constexpr int kLP = kLR + 1;              // x pitch of the staged block (odd: lanes = rows read conflict-free)
	__shared__ float s_blk[kLR * kLR * kLP];
	__shared__ float s_x[kLR * kLR];
	__shared__ float s_y[kLR];
	__shared__ float s_w[kMaxTaps];
	__shared__ int s_lo[3][kLT], s_hi[3][kLT];
	__shared__ float s_fr[3][kLT];
	__shared__ int s_rng[6];  // xmin, cx, ymin, cy, zmin, cz
	const int hw = t.hw, nt = 2 * hw + 1, tid = threadIdx.x;
	for (int i = tid; i < nt; i += 256) s_w[i] = t.w[i];
	const unsigned count = min(*prov_count, prov_cap);
	const int lvl = L.nextl_slot;
	const float *__restrict__ src = L.lazy_src;
	const float *__restrict__ cur = L.cur[lvl];
	const int wpr = (nx + 63) >> 6, nzs = zr.zo1 - zr.zo0, nzg = zr.nzg;
	const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
	// The workgroup's share of the parked list is read 256 entries at a time and the entries that are this kernel's --
	// all of them, or with skip_interior only the ones next to a border -- are compacted into an LDS list: with k_lazy_wave beside it nine
	// entries out of ten are not, and skipping them one load at a time cost a memory round trip each (r05: 103 us on the critical path
	// of the detection stage for a tenth of the candidates).
	__shared__ unsigned s_list[256];
	__shared__ unsigned s_nlist;
	// (the share is STRIDED -- entry e belongs to workgroup e % gridDim -- because the list is in scan order: the candidates next to the
	// z borders sit at its two ends, and contiguous shares left them to a few workgroups: detection 0.77 -> 0.98 ms)
	for (unsigned e0 = blockIdx.x; e0 < count; e0 += 256 * gridDim.x) {
	__syncthreads();  // (the list of the previous round has been consumed)
	if (tid == 0) s_nlist = 0u;
	__syncthreads();
	if (e0 + (unsigned)tid * gridDim.x < count) {
		const unsigned ent = prov[e0 + (unsigned)tid * gridDim.x];
		const size_t ic = (size_t)(ent & 0x7FFFFFFFu);
		const int zl = (int)(ic / sz), rem = (int)(ic - (size_t)zl * sz), y = rem / nx, x = rem - y * nx;
		const int zg = zl + zr.zoff;
		const bool inner = x >= hw && x <= nx - 2 - hw && y >= hw && y <= ny - 2 - hw && zg >= hw && zg <= nzg - 2 - hw;
		if (!(skip_interior && inner)) s_list[atomicAdd(&s_nlist, 1u)] = ent;
	}
	__syncthreads();
	const unsigned nlist = s_nlist;
	for (unsigned li = 0; li < nlist; li++) {
		const unsigned ent = s_list[li];
		const size_t ic = (size_t)(ent & 0x7FFFFFFFu);
		const bool as_max = (ent >> 31) != 0;
		const int zl = (int)(ic / sz), rem = (int)(ic - (size_t)zl * sz), y = rem / nx, x = rem - y * nx;
		const int zg = zl + zr.zoff;  // global plane
		const bool ix = x >= hw && x <= nx - 2 - hw, iy = y >= hw && y <= ny - 2 - hw, iz = zg >= hw && zg <= nzg - 2 - hw;
		__syncthreads();  // the previous candidate is finished with the LDS arrays
		if (tid < 3 * nt) {  // tap tables of the three axes (boundary_term's source samples; interior: lo = hi = p - d, frac 0)
			const int ax = tid / nt, k = tid - ax * nt;
			int lo, hi; float fr;
			if (ax == 0) lazy_tap_src(x, k - hw, nx, lo, hi, fr, ix);
			else if (ax == 1) lazy_tap_src(y, k - hw, ny, lo, hi, fr, iy);
			else lazy_tap_src(zg, k - hw, nzg, lo, hi, fr, iz);
			s_lo[ax][k] = lo; s_hi[ax][k] = hi; s_fr[ax][k] = fr;
		}
		__syncthreads();
		if (tid < 3) {  // contiguous range of real samples per axis
			int mn = 1 << 30, mxv = -1;
			for (int k = 0; k < nt; k++) { mn = min(mn, s_lo[tid][k]); mxv = max(mxv, s_hi[tid][k]); }
			s_rng[2 * tid] = mn; s_rng[2 * tid + 1] = min(mxv - mn + 1, kLR);
		}
		__syncthreads();
		const int xmin = s_rng[0], cx = s_rng[1], ymin = s_rng[2], cy = s_rng[3], zmin = s_rng[4], cz = s_rng[5];
		// ---- stage the block: x fastest, so a wave's loads run along rows ----
		for (int i0 = tid; i0 < cx * cy * cz; i0 += 256 * 8) {
			float v[8];
#pragma unroll
			for (int q = 0; q < 8; q++) {
				const int i = min(i0 + 256 * q, cx * cy * cz - 1);
				const int xi = i % cx, r = i / cx, yi = r % cy, zi2 = r / cy;
				v[q] = src[sz * (size_t)(zmin + zi2 - zr.zoff) + sy * (size_t)(ymin + yi) + (size_t)(xmin + xi)];
			}
#pragma unroll
			for (int q = 0; q < 8; q++) {
				const int i = i0 + 256 * q;
				if (i < cx * cy * cz) {
					const int xi = i % cx, r = i / cx, yi = r % cy, zi2 = r / cy;
					s_blk[(zi2 * kLR + yi) * kLP + xi] = v[q];
				}
			}
		}
		__syncthreads();
		// ---- x-blur per (row, plane) ----
		for (int s = tid; s < cy * cz; s += 256) {
			const int zi2 = s / cy, yi = s - zi2 * cy;
			const float *row = &s_blk[(zi2 * kLR + yi) * kLP] - xmin;
			float acc = 0.0f;
			if (ix) {
				for (int k = 0; k < nt; k++) acc = acc + s_w[k] * row[s_lo[0][k]];
			} else {
				for (int k = 0; k < nt; k++) {
					const float f = s_fr[0][k];
					acc = acc + s_w[k] * ((1.0f - f) * row[s_lo[0][k]] + f * row[s_hi[0][k]]);
				}
			}
			s_x[zi2 * kLR + yi] = acc;
		}
		__syncthreads();
		// ---- y-blur per plane ----
		if (tid < cz) {
			const float *col = &s_x[tid * kLR] - ymin;
			float acc = 0.0f;
			if (iy) {
				for (int k = 0; k < nt; k++) acc = acc + s_w[k] * col[s_lo[1][k]];
			} else {
				for (int k = 0; k < nt; k++) {
					const float f = s_fr[1][k];
					acc = acc + s_w[k] * ((1.0f - f) * col[s_lo[1][k]] + f * col[s_hi[1][k]]);
				}
			}
			s_y[tid] = acc;
		}
		__syncthreads();
		if (tid == 0) {
			const float *pl = s_y - zmin;
			float acc = 0.0f;
			if (iz) {
				for (int k = 0; k < nt; k++) acc = acc + s_w[k] * pl[s_lo[2][k]];
			} else {
				for (int k = 0; k < nt; k++) {
					const float f = s_fr[2][k];
					acc = acc + s_w[k] * ((1.0f - f) * pl[s_lo[2][k]] + f * pl[s_hi[2][k]]);
				}
			}
			const float n7 = (acc - src[ic]) * (-1.0f);  // Sub, Src/cSIFT3D.cc:875
			const float v = cur[ic];
			if (as_max ? (v > n7) : (v < n7)) {
				const int zi = zl - zr.zo0;
				atomicOr(&masks[((size_t)(lvl * nzs + zi) * ny + y) * wpr + (x >> 6)], 1ull << (x & 63));
				atomicAdd(&block_counts[(lvl * nzs + zi) * nyb + y / kRows], 1u);
			}
		}
	}
	}
}

// exclusive scan of block_counts[0..nblocks) by ONE workgroup; offsets start at the running total
// total[0], which is then advanced.
// exclusive scan of the block counts of one octave on top of `base` (one 1024-thread workgroup); returns base + the octave's total
__device__ __forceinline__ unsigned scan_octave(const unsigned *__restrict__ block_counts, unsigned *__restrict__ block_offsets, unsigned nblocks,
                                                unsigned base, unsigned *s_wave) {
	const unsigned t = threadIdx.x;
	// a thread owns `chunk` consecutive counts (a multiple of four: 16-byte loads and stores where the arrays' alignment allows --
	// r03: 24 dependent 4-byte loads per thread at a stride of 96 bytes across the lanes made this launch 33 us at 512^3)
	const unsigned chunk = (((nblocks + 1023u) / 1024u) + 3u) & ~3u;
	const unsigned lo = min(t * chunk, nblocks), hi = min(lo + chunk, nblocks);
	const bool vec = (((size_t)block_counts | (size_t)block_offsets) & 15) == 0 && hi - lo == chunk && chunk <= 32;
	unsigned cnt[32];
	unsigned sum = 0;
	if (vec) {
#pragma unroll
		for (unsigned q = 0; q < 8; q++)
			if (q * 4 < chunk) {
				const uint4 c4 = *reinterpret_cast<const uint4 *>(block_counts + lo + 4 * q);
				cnt[4 * q] = c4.x; cnt[4 * q + 1] = c4.y; cnt[4 * q + 2] = c4.z; cnt[4 * q + 3] = c4.w;
				sum += c4.x + c4.y + c4.z + c4.w;
			}
	} else {
		for (unsigned i = lo; i < hi; i++) sum += block_counts[i];
	}
	unsigned v = sum;
	const int lane = t & 63, wid = t >> 6;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		unsigned u = __shfl_up(v, o, 64);
		if (lane >= o) v += u;
	}
	if (lane == 63) s_wave[wid] = v;
	__syncthreads();
	if (t == 0) {
		unsigned a = 0;
		for (int w = 0; w < 16; w++) { unsigned c = s_wave[w]; s_wave[w] = a; a += c; }
	}
	__syncthreads();
	const unsigned incl = v + s_wave[wid];
	unsigned run = base + (incl - sum);
	if (vec) {
#pragma unroll
		for (unsigned q = 0; q < 8; q++)
			if (q * 4 < chunk) {
				uint4 o4;
				o4.x = run; run += cnt[4 * q]; o4.y = run; run += cnt[4 * q + 1]; o4.z = run; run += cnt[4 * q + 2]; o4.w = run; run += cnt[4 * q + 3];
				*reinterpret_cast<uint4 *>(block_offsets + lo + 4 * q) = o4;
			}
	} else {
		for (unsigned i = lo; i < hi; i++) { block_offsets[i] = run; run += block_counts[i]; }
	}
	__syncthreads();  // every thread has read its slot of s_wave
	if (t == 1023) s_wave[0] = base + incl;
	__syncthreads();
	const unsigned nbase = s_wave[0];
	__syncthreads();
	return nbase;
}

__global__ void __launch_bounds__(1024) k_scan(const unsigned *__restrict__ block_counts, unsigned *__restrict__ block_offsets,
                                               unsigned nblocks, unsigned *__restrict__ total) {
	__shared__ unsigned s_wave[16];
	const unsigned nb = scan_octave(block_counts, block_offsets, nblocks, total[0], s_wave);
	if (threadIdx.x == 0) total[0] = nb;
}

// r03: the octaves behind the first one in ONE scan launch and ONE emit launch (their 2 x 6 launches of ~5 us each sat, one after
// the other, between the extremum masks and the orientation stage)
struct EmitOct {
	const unsigned long long *masks;
	const unsigned *counts;
	unsigned *offsets;
	int nx, ny, nyb, octave;
	ZRange zr;
	unsigned nblocks, block0;
	int level_id[kMaxKpLevels];
	float scale[kMaxKpLevels];
};
struct EmitMulti { EmitOct o[8]; int n; };
__global__ void __launch_bounds__(1024) k_scan_multi(EmitMulti M, unsigned *__restrict__ total) {
	__shared__ unsigned s_wave[16];
	unsigned base = total[0];
	__syncthreads();
	for (int k = 0; k < M.n; k++) base = scan_octave(M.o[k].counts, M.o[k].offsets, M.o[k].nblocks, base, s_wave);
	if (threadIdx.x == 0) total[0] = base;
}

// one thread per ballot word of a block (16 rows x wpr words, same block decomposition as k_mark)
__device__ __forceinline__ void emit_block(int b, const unsigned long long *__restrict__ masks, const unsigned *__restrict__ block_offsets, int nx,
                                           int ny, const ZRange &zr, int nyb, int octave, const int *level_id, const float *scale,
                                           DevKp *__restrict__ out, unsigned cap, unsigned *__restrict__ total, unsigned *s_wave) {
	const int nz = zr.zo1 - zr.zo0;
	const int yb = b % nyb, zi = (b / nyb) % nz, lvl = b / (nyb * nz);
	const int z = zr.zo0 + zi + zr.zoff;  // GLOBAL plane: keypoint coordinates are global
	const int wpr = (nx + 63) >> 6;
	const int y0 = yb * kRows;
	const int nwords = min(kRows, ny - y0) * wpr;
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	unsigned running = block_offsets[b];
	// words are consumed in scan order, 256 at a time; the rank of a word = hits of all earlier words
	for (int w0 = 0; w0 < nwords; w0 += kThreads) {
		const int wi = w0 + threadIdx.x;
		unsigned long long m = 0;
		int y = 0, xw = 0;
		if (wi < nwords) {
			const int ry = wi / wpr;
			xw = wi - ry * wpr;
			y = y0 + ry;
			// k_mark dealt the words to waves round-robin; the layout in memory is scan order
			m = masks[((size_t)(lvl * nz + zi) * ny + y) * wpr + xw];
		}
		const unsigned c = (unsigned)__popcll(m);
		unsigned v = c;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			unsigned u = __shfl_up(v, o, 64);
			if (lane >= o) v += u;
		}
		__syncthreads();
		if (lane == 63) s_wave[wid] = v;
		__syncthreads();
		unsigned before = v - c;
		for (int w = 0; w < wid; w++) before += s_wave[w];
		const unsigned all = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
		unsigned pos = running + before;
		while (m) {
			const int bit = __ffsll((long long)m) - 1;
			m &= m - 1;
			if (pos < cap) {
				DevKp k;
				k.x = xw * 64 + bit; k.y = y; k.z = z;
				k.octave = octave; k.level = level_id[lvl]; k.scale = scale[lvl]; k.code = 0; k.slot = -1;
#pragma unroll
				for (int j = 0; j < 3; j++) { k.win[j] = 0.f; k.eigvalue[j] = 0.f; }
#pragma unroll
				for (int j = 0; j < 9; j++) { k.eigvector[j] = 0.f; k.rot[j] = 0.f; k.st[j] = 0.f; }
				out[pos] = k;
			} else {
				total[1] = 1u;  // overflow flag; the host regrows and reruns
			}
			pos++;
		}
		running += all;
	}
}

__global__ void __launch_bounds__(kThreads) k_emit(const unsigned long long *__restrict__ masks,
                                                   const unsigned *__restrict__ block_offsets, int nx, int ny, ZRange zr, int nyb,
                                                   int octave, DetectLevels L, DevKp *__restrict__ out, unsigned cap,
                                                   unsigned *__restrict__ total) {
	__shared__ unsigned s_wave[kThreads / 64];
	emit_block((int)blockIdx.x, masks, block_offsets, nx, ny, zr, nyb, octave, L.level_id, L.scale, out, cap, total, s_wave);
}

__global__ void __launch_bounds__(kThreads) k_emit_multi(EmitMulti M, DevKp *__restrict__ out, unsigned cap, unsigned *__restrict__ total) {
	__shared__ unsigned s_wave[kThreads / 64];
	int k = 0;
	while (k + 1 < M.n && blockIdx.x >= M.o[k + 1].block0) k++;  // block-uniform
	const EmitOct &E = M.o[k];
	emit_block((int)(blockIdx.x - E.block0), E.masks, E.offsets, E.nx, E.ny, E.zr, E.nyb, E.octave, E.level_id, E.scale, out, cap, total, s_wave);
}

// first half of an octave's detection: the ballot masks of the strict extrema (k_mark) and the parked candidates of the lazy level
// (k_lazy_next).  Octaves with their own scratch (DetectBufs) can run this half concurrently.
void launch_detect_mark(const DetectLevels &L, int nlevels, int nx, int ny, const ZRange &zr, float peak_thresh, int octave,
                        const DetectBufs &b, hipStream_t st, const Taps *lazy_taps) {
	const int nyb = (ny + kRows - 1) / kRows;
	const int nzl = zr.zo1 - zr.zo0;
	if (nzl <= 0) return;
	const unsigned nblocks = (unsigned)(nlevels * nzl * nyb);
	if (nblocks == 0) return;
	const bool lazy = L.lazy_src != nullptr && lazy_taps != nullptr && b.prov != nullptr;
	if (lazy) (void)hipMemsetAsync(b.prov_count, 0, sizeof(unsigned), st);
	const size_t mask_lds = sizeof(unsigned long long) * (kThreads / 64) * (kRows / 4) * (size_t)std::min((nx + 63) >> 6, 64);
	hipLaunchKernelGGL(k_mark, dim3(nblocks), dim3(kThreads), mask_lds, st, L, nx, ny, zr, nyb, peak_thresh, b.masks, b.block_counts, b.prov,
	                   b.prov_count, b.prov_cap, b.total);
	if (lazy) {
		// interior candidates of the default half width: one wave each (k_lazy_wave: 768 workgroups of two waves are resident); the rest
		// (next to a border, other half widths): one workgroup each
		// (the wave form keeps its 23 piece offsets as 32-bit byte offsets from the block's first sample: 17 planes must span < 4 GB)
		const bool wave_form = lazy_taps->hw == 8 && !hook(SIFT3D_HOOK_LAZY_GENERIC) && (size_t)nx * (size_t)ny * 17u * sizeof(float) < ((size_t)1 << 32);
		if (wave_form)
			hipLaunchKernelGGL(k_lazy_wave, dim3(256 * 3), dim3(64 * kLwWaves), 0, st, L, *lazy_taps, nx, ny, zr, nyb, b.prov, b.prov_count, b.prov_cap,
			                   b.masks, b.block_counts);
		if (2 * lazy_taps->hw + 1 <= 17)
			hipLaunchKernelGGL(k_lazy_next<17>, dim3(wave_form ? 1536 : 1024), dim3(256), 0, st, L, *lazy_taps, nx, ny, zr, nyb, b.prov, b.prov_count, b.prov_cap,
			                   b.masks, b.block_counts, wave_form ? 1 : 0);
		else
			hipLaunchKernelGGL(k_lazy_next<25>, dim3(512), dim3(256), 0, st, L, *lazy_taps, nx, ny, zr, nyb, b.prov, b.prov_count, b.prov_cap,
			                   b.masks, b.block_counts, 0);
	}
}

// second half: ordered compaction into the extrema list (appends after everything emitted before: octaves in order, one stream)
void launch_detect_emit(const DetectLevels &L, int nlevels, int nx, int ny, const ZRange &zr, int octave, const DetectBufs &b, DevKp *out,
                        unsigned cap, hipStream_t st) {
	const int nyb = (ny + kRows - 1) / kRows;
	const int nzl = zr.zo1 - zr.zo0;
	if (nzl <= 0) return;
	const unsigned nblocks = (unsigned)(nlevels * nzl * nyb);
	if (nblocks == 0) return;
	hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, b.block_counts, b.block_offsets, nblocks, b.total);
	hipLaunchKernelGGL(k_emit, dim3(nblocks), dim3(kThreads), 0, st, b.masks, b.block_offsets, nx, ny, zr, nyb, octave, L, out, cap,
	                   b.total);
}

// second half for several octaves at once (in the order given = octave order): one scan launch, one emit launch
void launch_detect_emit_multi(const DetectEmitItem *items, int n, DevKp *out, unsigned cap, unsigned *total, hipStream_t st) {
	EmitMulti M;
	memset(&M, 0, sizeof(M));
	unsigned blocks = 0;
	for (int k = 0; k < n && M.n < 8; k++) {
		const DetectEmitItem &it = items[k];
		const int nyb = (it.ny + kRows - 1) / kRows, nzl = it.zr.zo1 - it.zr.zo0;
		if (nzl <= 0) continue;
		const unsigned nblocks = (unsigned)(it.nlevels * nzl * nyb);
		if (nblocks == 0) continue;
		EmitOct &E = M.o[M.n++];
		E.masks = it.b->masks; E.counts = it.b->block_counts; E.offsets = it.b->block_offsets;
		E.nx = it.nx; E.ny = it.ny; E.nyb = nyb; E.octave = it.octave; E.zr = it.zr; E.nblocks = nblocks; E.block0 = blocks;
		for (int l = 0; l < kMaxKpLevels; l++) { E.level_id[l] = it.L->level_id[l]; E.scale[l] = it.L->scale[l]; }
		blocks += nblocks;
	}
	if (M.n == 0) return;
	hipLaunchKernelGGL(k_scan_multi, dim3(1), dim3(1024), 0, st, M, total);
	hipLaunchKernelGGL(k_emit_multi, dim3(blocks), dim3(kThreads), 0, st, M, out, cap, total);
}

void launch_detect_octave(const DetectLevels &L, int nlevels, int nx, int ny, const ZRange &zr, float peak_thresh, int octave,
                          const DetectBufs &b, DevKp *out, unsigned cap, hipStream_t st, const Taps *lazy_taps) {
	launch_detect_mark(L, nlevels, nx, ny, zr, peak_thresh, octave, b, st, lazy_taps);
	launch_detect_emit(L, nlevels, nx, ny, zr, octave, b, out, cap, st);
}

void preload_detect_kernels() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_scan)); }  // (see kernels_march.hip)

}  // namespace s3d
