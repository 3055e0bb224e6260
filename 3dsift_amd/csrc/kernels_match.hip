// kernels_match.hip -- brute-force descriptor matching: dense N x 768 by 768 x M contraction on the
// matrix cores + exact re-scoring.
//
// Restates KP_squareSum + muBruteMatcher::calMatches (reference Src/cMatcher.cc:17-23, 40-79): for
// each ref row i the best and second-best DOT PRODUCT over all target rows j (strict '>', ties keep
// the lower j, both start at FLT_MIN), reported as d = 2 - 2*dot.  The reference accumulates fp32
// products in fp64; an fp32 MFMA chain cannot reproduce that bit-for-bit, so the work is split:
//   k_scores_topk2: S = A * B^T with v_mfma_f32_32x32x2_f32 (exact fp32 fma chain), fused running
//                   top-K per row ordered by (score desc, j asc) -- candidate SELECTION only.  (r03: the second form of the kernel,
//                   A straight into registers / B by LDS-DMA / three workgroups per CU; k_scores_top4<DMA> is the r02 form, kept for
//                   A/B builds -DS3D_MATCH_V2=0, k_scores_top4<false> the register-staged one for matrices of 4 GB and more)
//   k_rescore     : the 4 candidates of each row are re-scored exactly like the reference
//                   (fp32 product, fp64 accumulate, k ascending) and the reference's update rule is
//                   replayed over them in ascending j -- bit-identical d1, d2, i1, i2 as long as the
//                   true top-2 are among the fp32 top-4
//   guard (r02)   : that condition is CHECKED per row: every column outside the list has an fp32 score <= the list's 4th
//                   score s4, and an fp32 chain of 768 products is within E = 768 * 2^-23 * |a| * max|b| of the exact value, so
//                   the list is complete whenever s4 + E < exact second-best score.  Rows that fail the test (near-duplicate
//                   targets) are re-done by k_exact_rows: exact scores of ALL columns, top-2 by (score desc, column asc) --
//                   which is what the reference's strict-'>' scan in ascending j yields.
// gfx950 only: 64-lane waves, MFMA C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "sift3d_internal.h"

namespace s3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef S3D_MATCH_TOPK
#define S3D_MATCH_TOPK 6  /* candidates kept per row by the fp32 selection (r02, two similar 512^3 volumes, one full pass: 4 -> 3.6 ms because ~10 % of the rows fail the guard and are re-done exactly, 6 -> 2.6 ms, 8 -> 2.9 ms) */
#endif
constexpr int KD = kDesc, TOPK = S3D_MATCH_TOPK;
// score kernel tiling: a 256-thread workgroup owns 128 rows x 128 columns per tile; wave w owns rows 32w..32w+31 across all
// 128 columns (four 32x32 accumulators), so every row's running top-4 belongs to exactly one wave and lives in registers
constexpr int BM = 128, BN = 128, BK = 32, PITCH = BK + 4;  // +16 B: the 16-B reads of 8 consecutive rows hit distinct banks
constexpr int kMaxSplits = 16;
constexpr int SP = BN + 1;  // pitch of the score tile in LDS (row-per-lane scans: conflict-free)

struct Cand { float s; int j; };  // partial top-4 entry (score, column); j < 0 = empty

// insert (s, j) into a list ordered by (score desc, column asc): same order the reference's strict '>' scan in ascending j
// produces, independent of the order in which candidates arrive
__device__ __forceinline__ void top4_insert(float (&bs)[TOPK], int (&bj)[TOPK], float s, int j) {
	int pos = TOPK;
#pragma unroll
	for (int t = TOPK - 1; t >= 0; t--)
		if (bj[t] < 0 || s > bs[t] || (s == bs[t] && j < bj[t])) pos = t;
#pragma unroll
	for (int t = TOPK - 1; t > 0; t--)
		if (t > pos) { bs[t] = bs[t - 1]; bj[t] = bj[t - 1]; }
#pragma unroll
	for (int t = 0; t < TOPK; t++)
		if (t == pos) { bs[t] = s; bj[t] = j; }
}

// the same insertion for a candidate that is known to beat the list's last entry (or finds it empty): it takes the last slot and
// bubbles up with five branch-free compare-exchanges -- about half the instructions of the general form.  The selection scan of
// k_scores_top4 calls it for every step in which ANY lane of the wave has a candidate: a workgroup restarts its lists for its
// column split, so with 64 lists per wave that is three steps out of four (0.41 ms of the 2.4 ms pass with the general insert).
__device__ __forceinline__ void topk_bubble(float (&bs)[TOPK], int (&bj)[TOPK], float s, int j) {
	bs[TOPK - 1] = s; bj[TOPK - 1] = j;
#pragma unroll
	for (int t = TOPK - 1; t > 0; t--) {
		const bool up = bj[t - 1] < 0 || bs[t] > bs[t - 1] || (bs[t] == bs[t - 1] && bj[t] < bj[t - 1]);
		const float s0 = bs[t - 1], s1 = bs[t];
		const int j0 = bj[t - 1], j1 = bj[t];
		bs[t - 1] = up ? s1 : s0; bj[t - 1] = up ? j1 : j0;
		bs[t] = up ? s0 : s1; bj[t] = up ? j0 : j1;
	}
}

// S = A * B^T on v_mfma_f32_32x32x2_f32 with a fused running top-4 per row.  grid = (row blocks, column splits); each
// workgroup streams the tiles of its column range, double-buffered through LDS.  An MFMA step multiplies two k indices:
// lanes 0-31 feed k = s, lanes 32-63 feed k = s + 16 of the 32-wide chunk, so a lane's operands for four consecutive steps
// are four consecutive floats of a plain row-major LDS tile (one ds_read_b128).  The summation order over k is free here:
// the scores only SELECT candidates, k_rescore recomputes them exactly.
// LDS-DMA: lane l's 16 bytes at base + voff land at lds_dst + 16*l (global_load_lds_dwordx4).  M0 carries the wave-uniform LDS byte
// address (restored).  Counted by vmcnt like any load.
__device__ __forceinline__ void x_dma16(const float *base, unsigned voff, unsigned lds_dst) {
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

// DMA = true (every byte offset into A and B fits 32 bits): the 32-wide k chunks travel global -> LDS by LDS-DMA instead of through
// registers (no staging registers, no ds_write, no VALU).  A wave's instruction fills 8 rows x 128 B contiguously, so the rows are
// NOT padded; instead the 16-byte piece g of tile row r is stored at slot g ^ ((r >> 1) & 7) -- the lane that owns slot c of row r
// simply requests piece c ^ ((r >> 1) & 7) -- which makes the ds_read_b128 fragment reads of 16 consecutive rows hit the 16
// distinct 4-bank groups (also in the lane groups the hardware serves a b128 read in).  Rows / columns past the end request row 0:
// their scores are never looked at.
template <bool DMA>
__global__ void __launch_bounds__(256, 2) k_scores_top4(const float *__restrict__ A, const int *__restrict__ row_ids, int nrows,
                                                        const float *__restrict__ B, int m, int slots,
                                                        Cand *__restrict__ part /*[nrows][slots][TOPK]*/) {
	// [A buf0 | A buf1 | B buf0 | B buf1]; after the MFMAs of a tile the same memory holds the tile's 128 x 128 scores
	__shared__ __attribute__((aligned(16))) float smem[4 * BM * PITCH];
	float(*As)[BM * PITCH] = reinterpret_cast<float(*)[BM * PITCH]>(smem);
	float(*Bs)[BN * PITCH] = reinterpret_cast<float(*)[BN * PITCH]>(smem + 2 * BM * PITCH);
	static_assert(4 * 32 * SP <= 4 * BM * PITCH, "score tile must fit the staging buffers");
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const int li = lane & 31, lh = lane >> 5;
	const int ntiles = (m + BN - 1) / BN;
	// The (row block, column tile) pairs, row-block major, are dealt to the workgroups in equal contiguous shares (one residency
	// round; a share may end in the middle of a row block and go on in the next one): with whole column splits per row block
	// 89 x 89 tiles gave 445 workgroups of 18 tiles for 512 slots -- now 512 of 15-16.  A row block is covered by at most
	// `slots` workgroups; each leaves one partial list per row (k_merge_top4 joins them, unused slots stay empty).
	const long long total = (long long)((nrows + BM - 1) / BM) * ntiles;
	const long long L0 = (long long)blockIdx.x * total / gridDim.x, L1 = (long long)(blockIdx.x + 1) * total / gridDim.x;
	for (long long L = L0; L < L1;) {
	const int rbi = (int)(L / ntiles);
	const int tile_lo = (int)(L - (long long)rbi * ntiles), tile_hi = (int)min((long long)ntiles, tile_lo + (L1 - L));
	L += tile_hi - tile_lo;
	// index of this workgroup among those that cover row block rbi: the first one is the largest w with L0(w) <= rbi * ntiles
	const long long X = (long long)rbi * ntiles;
	const int wfirst = (int)(((X + 1) * gridDim.x + total - 1) / total) - 1;
	const int slot = (int)blockIdx.x - wfirst;
	const int row0 = rbi * BM;

	// staging: 8 consecutive lanes load one 128-B row piece (32 floats) as float4s; 32 rows per pass, 4 passes
	const int srow = tid >> 3, spiece = (tid & 7) * 4;
	const float *Ap[4];
#pragma unroll
	for (int p = 0; p < 4; p++) {
		const int r = row0 + srow + 32 * p;
		Ap[p] = r < nrows ? A + (size_t)(row_ids ? row_ids[r] : r) * KD + spiece : nullptr;
	}
	// DMA staging: byte offset of this lane's piece of tile row srow + 32 p (k = 0) and the LDS address of the wave's 8-row slab
	const unsigned swz = (unsigned)((tid & 7) ^ ((srow >> 1) & 7)) * 16u;  // (row + 32 p) >> 1 has the same low three bits
	unsigned aoff[4];
#pragma unroll
	for (int p = 0; p < 4; p++) {
		const int r = row0 + srow + 32 * p;
		aoff[p] = (unsigned)(r < nrows ? (row_ids ? row_ids[r] : r) : (row_ids ? row_ids[0] : 0)) * (unsigned)(KD * 4) + swz;
	}
	const unsigned lds_a = (unsigned)(unsigned long long)&As[0][0], lds_b = (unsigned)(unsigned long long)&Bs[0][0];
	const unsigned slab = (unsigned)__builtin_amdgcn_readfirstlane(wid) * (8u * 32u * 4u);  // this wave's rows 8 wid .. 8 wid + 7 of a 32-row pass
	const unsigned fswz = (unsigned)((li >> 1) & 7);  // fragment reads: tile rows wid*32 + li and li + 32 nb share (row >> 1) & 7
	float bs[TOPK];
	int bj[TOPK];
#pragma unroll
	for (int t = 0; t < TOPK; t++) { bs[t] = -FLT_MAX; bj[t] = -1; }
	// selection: lane (li, lh) scans columns lh*64 .. lh*64+63 of row li of this wave's band, in ascending column order
	const bool my_row_ok = (row0 + wid * 32 + li) < nrows;

	for (int tile = tile_lo; tile < tile_hi; tile++) {
		const int col0 = tile * BN;
		const float *Bp[4];
#pragma unroll
		for (int p = 0; p < 4; p++) {
			const int c = col0 + srow + 32 * p;
			Bp[p] = c < m ? B + (size_t)c * KD + spiece : nullptr;
		}
		f32x16 acc[4];
#pragma unroll
		for (int nb = 0; nb < 4; nb++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[nb][r] = 0.0f;
		f32x4 pa[4], pb[4];
		auto fetch = [&](int k0) {
#pragma unroll
			for (int p = 0; p < 4; p++) {
				pa[p] = Ap[p] ? *reinterpret_cast<const f32x4 *>(Ap[p] + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
				pb[p] = Bp[p] ? *reinterpret_cast<const f32x4 *>(Bp[p] + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
			}
		};
		auto stash = [&](int buf) {
#pragma unroll
			for (int p = 0; p < 4; p++) {
				*reinterpret_cast<f32x4 *>(&As[buf][(srow + 32 * p) * PITCH + spiece]) = pa[p];
				*reinterpret_cast<f32x4 *>(&Bs[buf][(srow + 32 * p) * PITCH + spiece]) = pb[p];
			}
		};
		unsigned boff[4];
#pragma unroll
		for (int p = 0; p < 4; p++) {
			const int c = col0 + srow + 32 * p;
			boff[p] = (unsigned)(c < m ? c : 0) * (unsigned)(KD * 4) + swz;
		}
		auto dma = [&](int buf, int k0) {  // chunk k0 .. k0 + 31 of both tiles -> buffer buf
#pragma unroll
			for (int p = 0; p < 4; p++) {
				x_dma16(A, aoff[p] + (unsigned)(k0 * 4), lds_a + (unsigned)(buf * BM * PITCH * 4) + (unsigned)(p * 32 * 32 * 4) + slab);
				x_dma16(B, boff[p] + (unsigned)(k0 * 4), lds_b + (unsigned)(buf * BN * PITCH * 4) + (unsigned)(p * 32 * 32 * 4) + slab);
			}
		};
		constexpr int NCH = KD / BK;
		if (DMA) {
			__syncthreads();  // every wave is done with both buffers (and the score tile) of the previous tile
			dma(0, 0);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__syncthreads();
		} else {
			fetch(0);
			__syncthreads();  // every wave is done with both buffers of the previous tile
			stash(0);
			__syncthreads();
		}
		for (int ch = 0; ch < NCH; ch++) {
			const int buf = ch & 1;
			if (ch + 1 < NCH) {  // in flight during the MFMAs of this chunk (the other buffer was last read in chunk ch-1, a barrier ago)
				if (DMA) dma(buf ^ 1, (ch + 1) * BK);
				else fetch((ch + 1) * BK);
			}
			constexpr int RP = DMA ? 32 : PITCH;  // row pitch of the staged tiles
			const float *ar = &As[buf][(wid * 32 + li) * RP];
			const float *br = &Bs[buf][li * RP];
#pragma unroll
			for (int q = 0; q < 4; q++) {
				const int po = DMA ? (int)((((unsigned)lh << 2 | (unsigned)q) ^ fswz) * 4u) : lh * 16 + 4 * q;  // piece lh*4 + q of the row
				const f32x4 a4 = *reinterpret_cast<const f32x4 *>(ar + po);
				f32x4 b4[4];
#pragma unroll
				for (int nb = 0; nb < 4; nb++) b4[nb] = *reinterpret_cast<const f32x4 *>(br + nb * 32 * RP + po);
#pragma unroll
				for (int e = 0; e < 4; e++)
#pragma unroll
					for (int nb = 0; nb < 4; nb++) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[nb][e], acc[nb], 0, 0, 0);
			}
			if (ch + 1 < NCH) {
				if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				else stash(buf ^ 1);
				__syncthreads();
			}
		}
		// ---- fused selection.  The staging buffers are dead now: every wave dumps its 32 x 128 score band there (C layout
		// of a 32x32 block: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) and each lane scans half a
		// row in ascending column order with the reference's strict '>' rule; one compare per score, inserts are rare.
		__syncthreads();  // all waves finished reading As / Bs
		float *S = smem + wid * 32 * SP;
#pragma unroll
		for (int nb = 0; nb < 4; nb++)
#pragma unroll
			for (int r = 0; r < 16; r++) S[((r & 3) + 8 * (r >> 2) + 4 * lh) * SP + nb * 32 + li] = acc[nb][r];
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		if (my_row_ok) {
			const float *srow_p = S + li * SP + lh * 64;
			const int cbase = col0 + lh * 64;
			const int ncol = min(64, m - cbase);
#pragma unroll 4
			for (int jj = 0; jj < ncol; jj++) {
				const float sv = srow_p[jj];
				if (sv > bs[TOPK - 1] || bj[TOPK - 1] < 0) topk_bubble(bs, bj, sv, cbase + jj);
			}
		}
	}
	// the two half-row lists of a row -> one (lanes 32..63 hand theirs to lanes 0..31)
#pragma unroll
	for (int t = 0; t < TOPK; t++) {
		const float os = __shfl(bs[t], li + 32, 64);
		const int oj = __shfl(bj[t], li + 32, 64);
		if (lane < 32 && oj >= 0) top4_insert(bs, bj, os, oj);
	}
	if (lane < 32 && my_row_ok && slot < slots) {
		Cand *o = part + ((size_t)(row0 + wid * 32 + lane) * slots + slot) * TOPK;
#pragma unroll
		for (int t = 0; t < TOPK; t++) o[t] = Cand{bs[t], bj[t]};
	}
	}  // next row block of this workgroup's share
}

// ---- r03: the score kernel, second form (S3D_MATCH_V2; the DMA form above stays for A/B builds, the register-staged one for >= 4 GB) ----
// What is different:
//   * A never touches the LDS.  A wave's 32 rows are read by nobody else: lane (li, lh) loads the 16 floats k0 + 16 lh .. + 15 of
//     its row straight into registers (two lanes use one 128-byte line completely).  Piece q of the NEXT chunk is requested into
//     the same four registers right behind the sixteen MFMAs that consumed it, by an untracked load; each MFMA group waits for its
//     piece with a COUNTED s_waitcnt (vmcnt retires in order; the eight younger operations -- three A pieces, the four DMA
//     instructions of the chunk, one refreshed piece -- stay in flight).  Only the B chunks travel by LDS-DMA: 33 KB instead of
//     73 KB of LDS per workgroup -> THREE workgroups per CU, half the LDS-DMA traffic.
//   * every chunk issues exactly four DMA instructions: the last chunk of a tile requests the first chunk of the NEXT tile into the
//     free stage, so it lands during the selection and a tile costs one barrier more than its 24 chunks.
//   * the selection does not run the insert in lock step.  Each lane compares its 16 scores of a quarter band with its list's last
//     entry and keeps a bit mask; only the marked scores are re-read and inserted (bubble), so a wave iterates max-over-lanes
//     popcount times instead of once per column in which ANY lane has a candidate (three out of four before).  The score band of a
//     wave is dumped one 32-column block at a time into the stage the prefetch does not use.
//   * the work is dealt in HALF tiles (128 rows x 64 columns): 7 921 tiles over 768 resident workgroups are 10.3 each and the
//     launch lasts 11 (+ 6.7 %); 15 842 halves are 20.6 each (21: + 1.8 %).  A share's first / last tile may be an upper / lower
//     half: the same code with two of the four accumulators.
// Lists are per (row, lane half): lane (li, lh) scans the columns 16 lh .. 16 lh + 15 of each 32-column block, in ascending column
// order over the tiles, with the reference's strict '>' rule; the halves are merged at the end (order independent).
#ifndef S3D_MATCH_V2
#define S3D_MATCH_V2 1
#endif
#ifndef S3D_MATCH_V2_OCC
#define S3D_MATCH_V2_OCC 3
#endif
constexpr int SP2 = 33;  // pitch of a 32-column score block
// development diagnostics, timing only (wrong results by construction; never set in the product build): 1 no selection, 2 no B DMA,
// 4 no A refresh loads, 8 no barriers inside the chunk loop
#ifndef S3D_XDIAG
#define S3D_XDIAG 0
#endif
template <int N>
__device__ __forceinline__ void x_wait_piece(f32x4 &piece) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(piece) : "n"(N) : "memory"); }

__global__ void __launch_bounds__(256, S3D_MATCH_V2_OCC) k_scores_topk2(const float *__restrict__ A, const int *__restrict__ row_ids, int nrows,
                                                                      const float *__restrict__ B, int m, int slots,
                                                                      Cand *__restrict__ part /*[nrows][slots][TOPK]*/) {
	constexpr int kStage = BN * BK;  // floats per B stage; stage 1 is followed by the tail of the score blocks (4 waves x 32 x 33 floats)
	__shared__ __attribute__((aligned(1024))) float smem[kStage + 4 * 32 * SP2];
	const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int li = lane & 31, lh = lane >> 5;
	const int ntiles = (m + BN - 1) / BN, nunits = 2 * ntiles;  // units: half tiles
	const long long total = (long long)((nrows + BM - 1) / BM) * nunits;
	const long long L0 = (long long)blockIdx.x * total / gridDim.x, L1 = (long long)(blockIdx.x + 1) * total / gridDim.x;
	const int srow = tid >> 3;
	const unsigned swz = (unsigned)((tid & 7) ^ ((srow >> 1) & 7)) * 16u;
	const unsigned lds_b = (unsigned)(unsigned long long)&smem[0];
	const unsigned slab = (unsigned)wid * (8u * 32u * 4u);
	const unsigned fswz = (unsigned)((li >> 1) & 7);
	constexpr int NCH = KD / BK;
	for (long long L = L0; L < L1;) {
		const int rbi = (int)(L / nunits);
		const int u_lo = (int)(L - (long long)rbi * nunits), u_hi = (int)min((long long)nunits, u_lo + (L1 - L));
		L += u_hi - u_lo;
		const int tile_lo = u_lo >> 1, tile_hi = (u_hi + 1) >> 1;
		// index of this workgroup among those that cover row block rbi: the first one is the largest w with L0(w) <= rbi * nunits
		const long long X = (long long)rbi * nunits;
		const int wfirst = (int)(((X + 1) * gridDim.x + total - 1) / total) - 1;
		const int slot = (int)blockIdx.x - wfirst;
		const int row0 = rbi * BM;
		const int my_row = row0 + wid * 32 + li;
		const bool my_row_ok = my_row < nrows;
		const int arow_id = my_row_ok ? (row_ids ? row_ids[my_row] : my_row) : (row_ids ? row_ids[0] : 0);
		const unsigned arow_off = (unsigned)arow_id * (unsigned)(KD * 4) + (unsigned)(64 * lh);  // byte offset (both matrices < 4 GB)
		float bs[TOPK];
		int bj[TOPK];
#pragma unroll
		for (int t = 0; t < TOPK; t++) { bs[t] = -FLT_MAX; bj[t] = -1; }

		auto dma = [&](int buf, int col0, int k0) {  // chunk k0 .. k0 + 31 of the 128 B rows from col0 on -> stage buf (always four instructions)
#pragma unroll
			for (int p = 0; p < 4; p++) {
				const int c = col0 + srow + 32 * p;
				const unsigned off = (unsigned)(c < m ? c : 0) * (unsigned)(KD * 4) + swz + (unsigned)(k0 * 4);
				if (!(S3D_XDIAG & 2)) x_dma16(B, off, lds_b + (unsigned)(buf * kStage * 4) + (unsigned)(p * 32 * 32 * 4) + slab);
			}
		};
		f32x4 a[4];
		__syncthreads();  // every wave is done with the score blocks of the previous share (they alias stage 1)
		dma(0, tile_lo * BN, 0);
#pragma unroll
		for (int q = 0; q < 4; q++) a[q] = *reinterpret_cast<const f32x4 *>(A + (size_t)arow_id * KD + 16 * lh + 4 * q);
		asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::"memory");

		// one tile: NB0 .. NB1 - 1 are the 32-column blocks this share owns of it
		auto tile_body = [&](int tile, auto nb0_c, auto nb1_c) {
			constexpr int NB0 = decltype(nb0_c)::value, NB1 = decltype(nb1_c)::value;
			const int col0 = tile * BN;
			f32x16 acc[4];
#pragma unroll
			for (int nb = NB0; nb < NB1; nb++)
#pragma unroll
				for (int r = 0; r < 16; r++) acc[nb][r] = 0.0f;
			__syncthreads();  // chunk 0 has landed in stage 0 (every wave waited for its DMA); the score blocks are free
#pragma unroll 1
			for (int ch = 0; ch < NCH; ch++) {
				const int buf = ch & 1;
				const bool more = ch + 1 < NCH;
				// next chunk of this tile, or the first chunk of the next tile (the last tile of the share: its own, never read)
				dma(buf ^ 1, more ? col0 : (tile + 1 < tile_hi ? col0 + BN : col0), more ? (ch + 1) * BK : 0);
				const float *br = smem + buf * kStage + li * 32;
				const unsigned anext = arow_off + (more ? (unsigned)((ch + 1) * BK * 4) : 0u);  // (the last chunk re-requests chunk 0: the next tile's)
				f32x4 bq[2][4];
#pragma unroll
				for (int nb = NB0; nb < NB1; nb++) bq[0][nb] = *reinterpret_cast<const f32x4 *>(br + nb * 32 * 32 + (int)((((unsigned)lh << 2) ^ fswz) * 4u));
#pragma unroll
				for (int q = 0; q < 4; q++) {
					if (q < 3) {  // the fragments of the next group are in flight during this group's MFMAs
						const int po = (int)((((unsigned)lh << 2 | (unsigned)(q + 1)) ^ fswz) * 4u);
#pragma unroll
						for (int nb = NB0; nb < NB1; nb++) bq[(q + 1) & 1][nb] = *reinterpret_cast<const f32x4 *>(br + nb * 32 * 32 + po);
						__builtin_amdgcn_sched_barrier(0);  // (hipcc sinks these reads behind the last MFMAs of the group otherwise)
					}
					// in flight behind piece q: the other three pieces (or their refreshes) and the four DMA instructions of this chunk
					if (!(S3D_XDIAG & 4)) x_wait_piece<7>(a[q]);
#pragma unroll
					for (int e = 0; e < 4; e++)
#pragma unroll
						for (int nb = NB0; nb < NB1; nb++) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][e], bq[q & 1][nb][e], acc[nb], 0, 0, 0);
					if (!(S3D_XDIAG & 4)) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(a[q]) : "v"(anext + (unsigned)(q * 16)), "s"(A) : "memory");  // ("+v": the same registers in every chunk, no copy at the back-edge while the load is in flight)
				}
				// the four DMA instructions are the oldest of the eight operations in flight
				asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
				if (!(S3D_XDIAG & 8) || !more) __syncthreads();  // (after the last chunk: stage 1 is free for the score blocks)
			}
			if (S3D_XDIAG & 1) {  // keep the accumulators alive
				float sum = 0.f;
#pragma unroll
				for (int nb = NB0; nb < NB1; nb++)
#pragma unroll
					for (int r = 0; r < 16; r++) sum += acc[nb][r];
				if (sum == 12345.678f) bs[0] = sum;
			}
			float *S = smem + kStage + wid * 32 * SP2;
#pragma unroll
			for (int nb = (S3D_XDIAG & 1) ? NB1 : NB0; nb < NB1; nb++) {
#pragma unroll
				for (int r = 0; r < 16; r++) S[((r & 3) + 8 * (r >> 2) + 4 * lh) * SP2 + li] = acc[nb][r];
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				const float *sp = S + li * SP2 + 16 * lh;
				const int cbase = col0 + 32 * nb + 16 * lh;
				const int ncol = my_row_ok ? max(0, min(16, m - cbase)) : 0;
				const float thr = bs[TOPK - 1];  // -FLT_MAX while the list is short: every finite score passes
				unsigned mask = 0;
#pragma unroll
				for (int jj = 0; jj < 16; jj++) mask |= (sp[jj] > thr ? 1u : 0u) << jj;
				mask &= (1u << ncol) - 1u;
				while (__any(mask != 0)) {
					if (mask) {
						const int jj = __builtin_ctz(mask);
						mask &= mask - 1;
						const float sv = sp[jj];
						if (sv > bs[TOPK - 1] || bj[TOPK - 1] < 0) topk_bubble(bs, bj, sv, cbase + jj);
					}
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();  // the block is rewritten by the next one
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			}
		};
		using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;
		for (int tile = tile_lo; tile < tile_hi; tile++) {
			const bool from_half = 2 * tile < u_lo, to_half = 2 * tile + 2 > u_hi;  // (never both: a share of one half tile has u_hi = u_lo + 1)
			if (from_half) tile_body(tile, I2{}, I4{});
			else if (to_half) tile_body(tile, I0{}, I2{});
			else tile_body(tile, I0{}, I4{});
		}
		asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::"memory");  // the last refreshes and the idle prefetch
#pragma unroll
		for (int t = 0; t < TOPK; t++) {
			const float os = __shfl(bs[t], li + 32, 64);
			const int oj = __shfl(bj[t], li + 32, 64);
			if (lane < 32 && oj >= 0) top4_insert(bs, bj, os, oj);
		}
		if (lane < 32 && my_row_ok && slot < slots) {
			Cand *o = part + ((size_t)my_row * slots + slot) * TOPK;
#pragma unroll
			for (int t = 0; t < TOPK; t++) o[t] = Cand{bs[t], bj[t]};
		}
	}  // next row block of this workgroup's share
}

// merge the per-split partial lists of a row into its global top-K (score desc, column asc).  Eight lanes per row: lane s takes the
// slots s and s + 8, then three butterfly rounds join the eight lists (r03: one thread per row walked up to 16 x K entries one
// after the other, 50-70 us per pass).  The order (score desc, column asc) does not depend on the order of the inserts.
__global__ void __launch_bounds__(256) k_merge_top4(const Cand *__restrict__ part, int nrows, int splits, int *__restrict__ cand,
                                                    float *__restrict__ s4 /* K-th fp32 score of the row, -FLT_MAX if the list is short */,
                                                    int nunits, int nwg, int *__restrict__ redo_count) {
	const int gid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
	if (gid == 0) *redo_count = 0;  // (k_rescore, the next launch, counts the rows that fail the guard)
	const int r = gid >> 3, sub = gid & 7;
	const bool ok = r < nrows;
	// slots the score kernel wrote for this row's block: one per workgroup whose share [L0, L1) of the nunits units per row block
	// meets the block (the same arithmetic as in the score kernels), so the partial lists need no initialisation
	int used = 0;
	if (ok) {
		const long long total = (long long)((nrows + BM - 1) / BM) * nunits, X = (long long)(r / BM) * nunits, Y = X + nunits;
		const int wfirst = (int)(((X + 1) * nwg + total - 1) / total) - 1, wlast = (int)((Y * nwg + total - 1) / total) - 1;
		used = min(splits, wlast - wfirst + 1);
	}
	float bs[TOPK];
	int bj[TOPK];
#pragma unroll
	for (int t = 0; t < TOPK; t++) { bs[t] = -FLT_MAX; bj[t] = -1; }
	if (ok)
		for (int s = sub; s < used; s += 8)
#pragma unroll
			for (int t = 0; t < TOPK; t++) {
				const Cand c = part[((size_t)r * splits + s) * TOPK + t];
				if (c.j >= 0) top4_insert(bs, bj, c.s, c.j);
			}
#pragma unroll
	for (int o = 4; o > 0; o >>= 1) {
		float os[TOPK];
		int oj[TOPK];
#pragma unroll
		for (int t = 0; t < TOPK; t++) { os[t] = __shfl(bs[t], lane ^ o, 64); oj[t] = __shfl(bj[t], lane ^ o, 64); }
#pragma unroll
		for (int t = 0; t < TOPK; t++)
			if (oj[t] >= 0) top4_insert(bs, bj, os[t], oj[t]);
	}
	if (ok && sub == 0) {
#pragma unroll
		for (int t = 0; t < TOPK; t++) cand[(size_t)r * TOPK + t] = bj[t];
		s4[r] = bj[TOPK - 1] >= 0 ? bs[TOPK - 1] : -FLT_MAX;
	}
}

// squared norms of the rows of both matrices in ONE launch (one wave per row, four rows per wave, 16-byte loads), and their maxima
// (bits of a non-negative float): one atomic per workgroup (r03: one per row -- 11 000 atomics on one address were 0.13 ms of a pass)
__global__ void __launch_bounds__(256) k_row_norm2(const float *__restrict__ X0, int n0, float *__restrict__ o0, const float *__restrict__ X1,
                                                   int n1, float *__restrict__ o1, unsigned *__restrict__ n2max /* [2] */) {
	__shared__ float s_m[4];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int nb0 = (n0 + 15) / 16;
	const bool second = (int)blockIdx.x >= nb0;
	const float *X = second ? X1 : X0;
	float *out = second ? o1 : o0;
	const int nrows = second ? n1 : n0, blk = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x;
	float mx = 0.f;
	for (int i = 0; i < 4; i++) {
		const int r = (blk * 4 + wv) * 4 + i;
		if (r >= nrows) break;  // wave-uniform
		const f32x4 *x = reinterpret_cast<const f32x4 *>(X + (size_t)r * KD);
		float a = 0.f;
#pragma unroll
		for (int k = 0; k < KD / 256; k++) {
			const f32x4 v = x[k * 64 + lane];
			a = a + v.x * v.x; a = a + v.y * v.y; a = a + v.z * v.z; a = a + v.w * v.w;
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) a = a + __shfl_xor(a, o, 64);
		a = a * 1.0001f;  // fp32 summation error of 768 non-negative terms
		if (lane == 0) out[r] = a;
		mx = fmaxf(mx, a);
	}
	if (lane == 0) s_m[wv] = mx;
	__syncthreads();
	if (threadIdx.x == 0) {
		const float m4 = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
		if (m4 > 0.f) atomicMax(n2max + (second ? 1 : 0), __float_as_uint(m4));
	}
}

// exact re-score + replay of the reference update rule (Src/cMatcher.cc:52-77).  A candidate's exact score is the reference's chain
// s += (double)(a[k] * b[k]), k ascending.  A wave holds 64 / K rows (K candidates each: 60 chains for K = 6) and walks the 768
// products in pieces of 64: the PRODUCTS are formed cooperatively (sixteen lanes per chain, 16-byte loads along the rows -- a lane
// per chain reading its own two rows touched 60 cache lines per load instruction and the pass took 0.07 ms, one row per wave as in
// r02 0.16 ms), parked in the wave's LDS block, and lane c then adds the 128 products of chain c in order.  The first lane of a row
// gathers its K (column, score) pairs and replays them in ascending column order.
constexpr int kRowsPerWave = 64 / TOPK, kChains = kRowsPerWave * TOPK, kRsPiece = 64, kRsPitch = kRsPiece + 4, kRsWaves = 2;
static_assert(kChains % 4 == 0, "k_rescore forms the products of four chains per step");
__global__ void __launch_bounds__(64 * kRsWaves) k_rescore(const float *__restrict__ A, const int *__restrict__ row_ids, int nrows,
                                                 const float *__restrict__ B, const int *__restrict__ cand, float *__restrict__ gd,
                                                 float *__restrict__ sd, int *__restrict__ gi, int *__restrict__ si,
                                                 const float *__restrict__ s4, const float *__restrict__ a_n2,
                                                 const unsigned *__restrict__ b_n2max, int *__restrict__ redo /* [0] count, then r */) {
	__shared__ __attribute__((aligned(16))) float s_prod[kRsWaves][kChains * kRsPitch];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int rr = lane / TOPK, t0 = lane - rr * TOPK;
	const int r = (blockIdx.x * kRsWaves + wv) * kRowsPerWave + rr;
	const bool ok = rr < kRowsPerWave && r < nrows;
	const int row = ok ? (row_ids ? row_ids[r] : r) : 0;
	const int j = ok ? cand[(size_t)r * TOPK + t0] : -1;
	float *P = s_prod[wv];
	const int quarter = lane >> 4, l4 = lane & 15;
	double s = 0.0;
#pragma unroll 1
	for (int k0 = 0; k0 < KD; k0 += kRsPiece) {
		// products of chains 4 i + quarter, piece k0 .. k0 + 63: lane l4 takes the four floats 4 l4 .. 4 l4 + 3
#pragma unroll 5
		for (int i = 0; i < kChains / 4; i++) {
			const int c = 4 * i + quarter;
			const int crow = __shfl(row, c, 64), cj = __shfl(j, c, 64);
			const f32x4 av = *reinterpret_cast<const f32x4 *>(A + (size_t)crow * KD + k0 + 4 * l4);
			const f32x4 bv = *reinterpret_cast<const f32x4 *>(B + (size_t)(cj >= 0 ? cj : 0) * KD + k0 + 4 * l4);
			*reinterpret_cast<f32x4 *>(P + c * kRsPitch + 4 * l4) = f32x4{av.x * bv.x, av.y * bv.y, av.z * bv.z, av.w * bv.w};
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		if (lane < kChains) {
			const f32x4 *pc = reinterpret_cast<const f32x4 *>(P + lane * kRsPitch);
#pragma unroll 8
			for (int q = 0; q < kRsPiece / 4; q++) {
				const f32x4 v = pc[q];
				s += (double)v.x; s += (double)v.y; s += (double)v.z; s += (double)v.w;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();  // the block is rewritten by the next piece
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
	if (j < 0) s = 0.0;
	// gather the K (j, s) pairs of a row on its first lane and replay in ascending j
	int js[TOPK];
	double ss[TOPK];
	const int base = lane - t0;
#pragma unroll
	for (int t = 0; t < TOPK; t++) { js[t] = __shfl(j, (base + t) & 63, 64); ss[t] = __shfl(s, (base + t) & 63, 64); }
	if (ok && t0 == 0) {
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
		// guard: can a column outside the candidate list reach the second place?  (see the header)
		const float thr4 = s4[r];
		if (thr4 > -FLT_MAX) {
			const float e = 9.16e-5f /* 768 * 2^-23, rounded up */ * (__fsqrt_rn(a_n2[row]) * 1.000001f) * (__fsqrt_rn(__uint_as_float(*b_n2max)) * 1.000001f);
			if (!((double)thr4 + (double)e < d2)) redo[1 + atomicAdd(&redo[0], 1)] = r;
		}
		d2 = 2 - 2 * d2;
		d1 = 2 - 2 * d1;
		gd[row] = (float)d1; sd[row] = (float)d2; gi[row] = i1; si[row] = i2;
	}
}

// Rows whose candidate list may be incomplete: exact scores (fp32 product, fp64 accumulate, k ascending -- Src/cMatcher.cc:57-61)
// of ALL columns, best and second best by (score desc, column asc), both starting at FLT_MIN like the reference.
__global__ void __launch_bounds__(256) k_exact_rows(const float *__restrict__ A, const int *__restrict__ row_ids, const float *__restrict__ B,
                                                    int m, const int *__restrict__ redo, float *__restrict__ gd, float *__restrict__ sd,
                                                    int *__restrict__ gi, int *__restrict__ si) {
	__shared__ double s_d[256][2];
	__shared__ int s_i[256][2];
	__shared__ float s_a[KD];
	const int n = redo[0], tid = threadIdx.x;
	for (int it = blockIdx.x; it < n; it += gridDim.x) {
		const int r = redo[1 + it];
		const int row = row_ids ? row_ids[r] : r;
		__syncthreads();
		for (int k = tid; k < KD; k += 256) s_a[k] = A[(size_t)row * KD + k];
		__syncthreads();
		double d1 = FLT_MIN, d2 = FLT_MIN;
		int i1 = -1, i2 = -1;
		for (int j = tid; j < m; j += 256) {  // ascending j per thread
			const float *b = B + (size_t)j * KD;
			double sc = 0.0;
			for (int k = 0; k < KD; k++) sc += (double)(s_a[k] * b[k]);
			if (sc > d1) { d2 = d1; i2 = i1; d1 = sc; i1 = j; }
			else if (sc > d2) { d2 = sc; i2 = j; }
		}
		s_d[tid][0] = d1; s_d[tid][1] = d2; s_i[tid][0] = i1; s_i[tid][1] = i2;
		__syncthreads();
		if (tid == 0) {
			// merge the 512 partial entries by (score desc, column asc); entries with index -1 are the FLT_MIN start values
			double b1 = FLT_MIN, b2 = FLT_MIN;
			int j1 = -1, j2 = -1;
			auto better = [](double s, int j, double bs, int bj) { return bj < 0 ? true : (s > bs || (s == bs && j < bj)); };
			for (int t = 0; t < 256; t++)
				for (int q = 0; q < 2; q++) {
					const double sc = s_d[t][q];
					const int j = s_i[t][q];
					if (j < 0) continue;
					if (better(sc, j, b1, j1)) { b2 = b1; j2 = j1; b1 = sc; j1 = j; }
					else if (better(sc, j, b2, j2)) { b2 = sc; j2 = j; }
				}
			gd[row] = (float)(2 - 2 * b1); sd[row] = (float)(2 - 2 * b2); gi[row] = j1; si[row] = j2;
		}
	}
}

// rows: optional list of row indices into A (reverse pass over the masked targets only)
int match_rows_device(const float *d_a, const int *d_row_ids, int nrows, const float *d_b, int m, int *d_cand, void *d_part,
                      float *d_gd, float *d_sd, int *d_gi, int *d_si, const MatchGuard &g, hipStream_t st) {
	if (nrows <= 0) return SIFT3D_OK;
	// as many workgroups as fit ONE residency round (two per CU), each with an equal share of the (row block, column tile) pairs;
	// at most kMaxSplits - 1 workgroups per row block (plus the one that straddles its start)
	const int rb = (nrows + BM - 1) / BM, ntiles = (m + BN - 1) / BN;
	const bool v2 = S3D_MATCH_V2 && g.small_offsets;
	const int per_cu = v2 ? S3D_MATCH_V2_OCC : 2;
	const int nunits = v2 ? 2 * ntiles : ntiles;  // the second form deals half tiles
	const long long total = (long long)rb * nunits;
	const int nwg = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(per_cu * 256, total), (long long)(kMaxSplits - 1) * rb));
	// workgroups that can touch one row block: those starting inside it plus the one running into it
	const int slots = std::min(kMaxSplits, (int)(((long long)nunits * nwg + total - 1) / total) + 1);
	// (the DMA form addresses A and B with 32-bit byte offsets; row_ids index the caller's whole A, whose size is not known here:
	// the caller says whether both matrices stay below 4 GB)
	if (v2)
		hipLaunchKernelGGL(k_scores_topk2, dim3(nwg), dim3(256), 0, st, d_a, d_row_ids, nrows, d_b, m, slots, (Cand *)d_part);
	else if (g.small_offsets)
		hipLaunchKernelGGL(k_scores_top4<true>, dim3(nwg), dim3(256), 0, st, d_a, d_row_ids, nrows, d_b, m, slots, (Cand *)d_part);
	else
		hipLaunchKernelGGL(k_scores_top4<false>, dim3(nwg), dim3(256), 0, st, d_a, d_row_ids, nrows, d_b, m, slots, (Cand *)d_part);
	hipLaunchKernelGGL(k_merge_top4, dim3((nrows + 31) / 32), dim3(256), 0, st, (const Cand *)d_part, nrows, slots, d_cand, g.s4, nunits, nwg, g.redo);
	hipLaunchKernelGGL(k_rescore, dim3((nrows + kRsWaves * kRowsPerWave - 1) / (kRsWaves * kRowsPerWave)), dim3(64 * kRsWaves), 0, st, d_a, d_row_ids, nrows, d_b, d_cand, d_gd, d_sd, d_gi, d_si,
	                   g.s4, g.a_n2, g.b_n2max, g.redo);
	hipLaunchKernelGGL(k_exact_rows, dim3(std::min(nrows, 1024)), dim3(256), 0, st, d_a, d_row_ids, d_b, m, g.redo, d_gd, d_sd, d_gi, d_si);
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

// Per-device matcher state, created on first use and reused by every later call (r03: a call used to do 3-5 hipMalloc / hipFree and
// ran on the null stream): a grow-only device scratch, a pinned host block for the results, the matcher's own non-blocking stream
// and its timing events.  One call at a time per device (the mutex is held for the whole call).
#include <chrono>
#include <mutex>

namespace {
struct MatchState {
	std::mutex mu;
	bool ready = false;
	hipStream_t stream = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr, e_in = nullptr;
	char *d_scratch = nullptr; size_t d_bytes = 0;   // device: everything but the descriptor matrices
	float *d_ab = nullptr; size_t ab_floats = 0;     // device copies of host-resident descriptor matrices
	char *h_pin = nullptr; size_t h_bytes = 0;       // pinned host: results of a pass (+ the coordinates of device-resident inputs)
	int last_redo = 0;
};
constexpr int kMaxDev = 64;
MatchState g_match[kMaxDev];
thread_local double t_match_dev = 0.0, t_match_wall = 0.0;
thread_local int t_match_redo_rows = 0;  // rows re-scored exactly by the calling thread's last call (sift3d_debug_counters)

int ensure(MatchState &S, size_t d_bytes, size_t ab_floats, size_t h_bytes) {
	if (!S.ready) {  // (a failed creation leaves the objects made so far in place: the next call goes on from there)
		if (!S.stream) S3D_HIP(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
		if (!S.e0) S3D_HIP(hipEventCreate(&S.e0));
		if (!S.e1) S3D_HIP(hipEventCreate(&S.e1));
		if (!S.e_in) S3D_HIP(hipEventCreateWithFlags(&S.e_in, hipEventDisableTiming));
		S.ready = true;
	}
	auto grow = [](size_t want) { return want + want / 4 + 4096; };  // head room: sets of similar size do not reallocate
	if (d_bytes > S.d_bytes) {
		S3D_HIP(hipStreamSynchronize(S.stream));
		if (S.d_scratch) (void)hipFree(S.d_scratch);
		S.d_scratch = nullptr; S.d_bytes = 0;
		S3D_HIP(hipMalloc(&S.d_scratch, grow(d_bytes)));
		S.d_bytes = grow(d_bytes);
	}
	if (ab_floats > S.ab_floats) {
		S3D_HIP(hipStreamSynchronize(S.stream));
		if (S.d_ab) (void)hipFree(S.d_ab);
		S.d_ab = nullptr; S.ab_floats = 0;
		S3D_HIP(hipMalloc(&S.d_ab, sizeof(float) * grow(ab_floats)));
		S.ab_floats = grow(ab_floats);
	}
	if (h_bytes > S.h_bytes) {
		if (S.h_pin) (void)hipHostFree(S.h_pin);
		S.h_pin = nullptr; S.h_bytes = 0;
		S3D_HIP(hipHostMalloc(&S.h_pin, grow(h_bytes), hipHostMallocDefault));
		S.h_bytes = grow(h_bytes);
	}
	return SIFT3D_OK;
}
size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
}  // namespace

namespace s3d { int match_redo_rows() { return t_match_redo_rows; } }

// First use of the matcher on a device, ahead of time (muBruteMatcher's constructor, sift3d_create): the translation unit's code object is
// loaded (HIP loads it at the first launch of one of its kernels: the first pass of a process took 2.48 ms instead of 1.70), the
// stream and the events exist.  Never fails loudly: a device that cannot be prepared fails in sift3d_match.
namespace s3d {
void preload_match_kernels() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_scores_topk2)); }
}
extern "C" int sift3d_match_warmup(int device) {
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_last_error("no HIP device visible: no CPU fallback"); return SIFT3D_ERR_NO_DEVICE; }
	if (device < 0 || device >= ndev || device >= kMaxDev) return SIFT3D_ERR_ARG;
	S3D_HIP(hipSetDevice(device));
	s3d::preload_match_kernels();
	MatchState &S = g_match[device];
	std::lock_guard<std::mutex> lock(S.mu);
	return ensure(S, 0, 0, 0);  // (stream and events; the scratch is sized by the first call)
}

extern "C" int sift3d_match_times(double *device_seconds, double *wall_seconds) {
	if (device_seconds) *device_seconds = t_match_dev;
	if (wall_seconds) *wall_seconds = t_match_wall;
	return SIFT3D_OK;
}

extern "C" int sift3d_match(const float *ref_desc, const float *ref_xyz, int n, const float *tar_desc, const float *tar_xyz, int m,
                            double thresHold, int mode, int on_device, int device, int *gIdx, int *sIdx, float *gDist,
                            float *sDist, float *pairs6, int *npairs, double *seconds) {
	const auto wall0 = std::chrono::steady_clock::now();
	if (n < 0 || m < 0 || mode < 1 || mode > 3 || (n > 0 && (!ref_desc || !ref_xyz)) || (m > 0 && (!tar_desc || !tar_xyz))) {
		set_last_error("sift3d_match: bad argument");
		return SIFT3D_ERR_ARG;
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_last_error("no HIP device visible: no CPU fallback"); return SIFT3D_ERR_NO_DEVICE; }
	if (device < 0 || device >= ndev || device >= kMaxDev) return SIFT3D_ERR_ARG;
	S3D_HIP(hipSetDevice(device));
	if (npairs) *npairs = 0;
	if (seconds) *seconds = 0;
	t_match_dev = t_match_wall = 0.0;

	MatchState &S = g_match[device];
	std::lock_guard<std::mutex> lock(S.mu);
	const size_t nn = (size_t)std::max(n, 1), mm = (size_t)std::max(m, 1), big = std::max(nn, mm);
	// device scratch layout (256-B aligned pieces): results [gd | sd | gi | si] of a pass (ONE D2H copy), then the working arrays
	const size_t o_res = 0, res_bytes = al256(4 * 4 * big);
	const size_t o_cand = o_res + res_bytes, o_rows = o_cand + al256(4 * (size_t)TOPK * big), o_redo = o_rows + al256(4 * big);
	const size_t o_s4 = o_redo + al256(4 * (big + 1)), o_an2 = o_s4 + al256(4 * big), o_bn2 = o_an2 + al256(4 * nn);
	const size_t o_nmax = o_bn2 + al256(4 * mm), o_part = o_nmax + 256, d_bytes = o_part + al256(sizeof(Cand) * TOPK * kMaxSplits * big);
	const size_t h_res = 0, h_xyz = res_bytes + 256, h_bytes = h_xyz + (on_device ? 4 * 3 * (nn + mm) : 0);
	int rc = ensure(S, d_bytes, on_device ? 0 : (size_t)kDesc * (nn + mm), h_bytes);
	if (rc) return rc;
	hipStream_t st = S.stream;

	std::vector<float> gd(n, 0.f), sd(n, 0.f), gd2(m, 0.f), sd2(m, 0.f);
	std::vector<int> gi(n, -1), si(n, -1), gi2(m, -1), si2(m, -1);
	const float *rx = ref_xyz, *tx = tar_xyz;
	int redo_rows = 0;
#define MCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); (void)hipStreamSynchronize(st); return SIFT3D_ERR_HIP; } } while (0)
	float *d_a, *d_b;
	if (on_device) {
		// device-resident inputs: work the caller queued on the legacy default stream (torch's default stream is that one) is
		// ordered in front of the matcher; inputs produced on other streams must be complete when the call is made (sift3d_run is)
		MCHK(hipEventRecord(S.e_in, nullptr));
		MCHK(hipStreamWaitEvent(st, S.e_in, 0));
		d_a = const_cast<float *>(ref_desc); d_b = const_cast<float *>(tar_desc);
		float *hx = reinterpret_cast<float *>(S.h_pin + h_xyz), *hy = hx + 3 * nn;
		if (n) MCHK(hipMemcpyAsync(hx, ref_xyz, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, st));
		if (m) MCHK(hipMemcpyAsync(hy, tar_xyz, sizeof(float) * 3 * m, hipMemcpyDeviceToHost, st));
		rx = hx; tx = hy;  // read after the first synchronisation below
	} else {
		d_a = S.d_ab; d_b = S.d_ab + (size_t)kDesc * nn;
		if (n) MCHK(hipMemcpyAsync(d_a, ref_desc, sizeof(float) * kDesc * n, hipMemcpyHostToDevice, st));
		if (m) MCHK(hipMemcpyAsync(d_b, tar_desc, sizeof(float) * kDesc * m, hipMemcpyHostToDevice, st));
	}
	{
		char *D = S.d_scratch;
		float *d_gd = reinterpret_cast<float *>(D + o_res), *d_sd = d_gd + big;
		int *d_gi = reinterpret_cast<int *>(d_sd + big), *d_si = d_gi + big;
		int *d_cand = reinterpret_cast<int *>(D + o_cand), *d_rows = reinterpret_cast<int *>(D + o_rows), *d_redo = reinterpret_cast<int *>(D + o_redo);
		float *d_s4 = reinterpret_cast<float *>(D + o_s4), *d_an2 = reinterpret_cast<float *>(D + o_an2), *d_bn2 = reinterpret_cast<float *>(D + o_bn2);
		unsigned *d_nmax = reinterpret_cast<unsigned *>(D + o_nmax);
		void *d_part = D + o_part;
		float *h_gd = reinterpret_cast<float *>(S.h_pin + h_res), *h_sd = h_gd + big;
		int *h_gi = reinterpret_cast<int *>(h_sd + big), *h_si = h_gi + big;
		int *h_redo = reinterpret_cast<int *>(S.h_pin + res_bytes);

		MCHK(hipEventRecord(S.e0, st));
		MCHK(hipMemsetAsync(d_nmax, 0, 2 * sizeof(unsigned), st));
		if (n + m > 0) hipLaunchKernelGGL(k_row_norm2, dim3((n + 15) / 16 + (m + 15) / 16), dim3(256), 0, st, d_a, n, d_an2, d_b, m, d_bn2, d_nmax);
		// matrices of 4 GB and more take the register-staged form (SIFT3D_HOOK_MATCH_NODMA forces it on any size, for the tests)
		const bool small = !hook(SIFT3D_HOOK_MATCH_NODMA) && (size_t)std::max(n, m) * KD * sizeof(float) < ((size_t)1 << 32);
		const MatchGuard g_fwd{d_s4, d_an2, d_nmax + 1, d_redo, small}, g_rev{d_s4, d_bn2, d_nmax, d_redo, small};

		// ---- ref -> tar ----
		if (n > 0 && m > 0) {
			match_rows_device(d_a, nullptr, n, d_b, m, d_cand, d_part, d_gd, d_sd, d_gi, d_si, g_fwd, st);
			MCHK(hipMemcpyAsync(h_gd, d_gd, 4 * 4 * big, hipMemcpyDeviceToHost, st));
			MCHK(hipMemcpyAsync(h_redo, d_redo, sizeof(int), hipMemcpyDeviceToHost, st));
			MCHK(hipEventRecord(S.e1, st));  // device time ends with the last device operation of the call (recorded again behind a reverse pass)
			MCHK(hipStreamSynchronize(st));
			memcpy(gd.data(), h_gd, sizeof(float) * n); memcpy(sd.data(), h_sd, sizeof(float) * n);
			memcpy(gi.data(), h_gi, sizeof(int) * n); memcpy(si.data(), h_si, sizeof(int) * n);
			redo_rows += h_redo[0];
		} else {
			MCHK(hipEventRecord(S.e1, st));
			MCHK(hipStreamSynchronize(st));
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
				MCHK(hipMemcpyAsync(d_rows, rows.data(), sizeof(int) * rows.size(), hipMemcpyHostToDevice, st));
				match_rows_device(d_b, d_rows, (int)rows.size(), d_a, n, d_cand, d_part, d_gd, d_sd, d_gi, d_si, g_rev, st);
				MCHK(hipMemcpyAsync(h_gd, d_gd, 4 * 4 * big, hipMemcpyDeviceToHost, st));
				MCHK(hipMemcpyAsync(h_redo, d_redo, sizeof(int), hipMemcpyDeviceToHost, st));
				MCHK(hipEventRecord(S.e1, st));
				MCHK(hipStreamSynchronize(st));
				for (int j : rows) { gd2[j] = h_gd[j]; sd2[j] = h_sd[j]; gi2[j] = h_gi[j]; si2[j] = h_si[j]; }
				redo_rows += h_redo[0];
			}
			ratio_filter(gi2, gd2, sd2, thresHold);
			for (int i = 0; i < n; i++) {                                              // bijectFilter :133-144
				const int j = gi[i];
				if (j < 0 || cnt[j] == 0) continue;
				if (gi2[j] != i) gi[i] *= -1;
			}
		}
		MCHK(hipEventSynchronize(S.e1));
		float ms = 0;
		hipEventElapsedTime(&ms, S.e0, S.e1);
		t_match_dev = (double)ms * 1e-3;
		if (seconds) *seconds = t_match_dev;

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
#undef MCHK
	t_match_redo_rows = redo_rows;
	t_match_wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
	return SIFT3D_OK;
}
