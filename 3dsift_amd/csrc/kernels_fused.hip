// kernels_fused.hip -- the pyramid hot kernel: ONE pass over HBM per level.
//
//   G[i] = gauss_z(gauss_y(gauss_x(G[i-1])))   and   DoG[i-1] = (G[i] - G[i-1]) * (-1)   and   max|DoG[i-1]|
//
// replaces, per level, the reference's 3 convolution sweeps + 4 transposes + serial boundary sweep
// (GaussianSmooth_3D, Src/cSIFT3D.cc:535-622, 624-847), the separate Sub sweep (:849-882) and the
// serial im_max_abs sweep (Src/cUtil.cc:587-605).  Algorithmic HBM traffic: read G[i-1] once, write
// G[i] and DoG[i-1] once = 12 B/voxel.
//
// Structure (2.5-D streaming, 256 threads = 4 waves per workgroup):
//   * a workgroup owns a TX x TY = 32 x 32 column of the volume and marches along z over a chunk
//   * per plane q (software pipelined, two barriers per plane):
//       regs -> LDS tile of plane q (loaded one step earlier), then the global loads of plane q+1 and of
//       the DoG centre values of plane p = q-HW are ISSUED and stay in flight during the step
//       x-blur  LDS tile -> LDS   (each thread slides a register window over 8 outputs of one row)
//       y-blur  LDS -> registers  (each thread: 4 consecutive y of one x, lanes along x)
//       the xy-blurred value enters a per-thread REGISTER RING of 2*HW+2 planes, kept in plane order (shifted by one
//       register per plane), so every ring index is static
//       z-blur of plane p straight from the ring, DoG against G[i-1](p), coalesced stores, running max
//   * block -> tile mapping is XCD aware (neighbouring tiles share their halo in one XCD's L2)
//   * the grid is sized to whole residency rounds (CUs x workgroups/CU) to avoid a tail
//
// Parity: every output is the literal sum acc = acc + tap[d+hw]*term for d = -hw..+hw with separate
// IEEE multiply and add (no FMA: file built with -ffp-contract=off), interior term = src[p-d],
// boundary term = (1-frac)*src[lo] + frac*src[hi] with the reference's fp32 coordinate rule
// (Src/cSIFT3D.cc:736-764), i.e. bit-identical to the CPU path.
//
// Boundaries without a slow path: for n >= 2*hw+2 the reference rule is EXACTLY the plain tap chain over an
// extended line E:  E[-k] = src[k] (c<0 -> -c, frac = 0 so the term is tap*src[k]),  and
// E[dim_end+k] = (1-f_k)*src[m-1] + f_k*src[m], m = dim_end-k, k = 0..hw, where f_k is the fp32 fraction the
// reference obtains from `2*dim_end - c - 0.1f` (host table EdgeFrac; the term tap*((1-f)*lo + f*hi) evaluates
// the bracket first, so precomputing E[] with the same two multiplies and one add is bit-identical; interior
// outputs never reach index dim_end, boundary outputs always see it through the lerp).  Edge tiles form E[] where its
// sources already sit in registers (volumes that are whole tiles wide and high, see `fast` in the body) or patch their LDS
// halo columns (x) / rows (y), and run the same blur code as interior tiles.  In z the few
// boundary planes per volume use wave-uniform ring selects.  Levels with a dimension < 2*hw+2 use the generic
// separable kernels of kernels_pyramid.hip.
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#ifndef S3D_BG_K_DEFAULT
#define S3D_BG_K_DEFAULT 0.0
#endif

#include "sift3d_internal.h"

namespace s3d {

#if defined(S3D_EXP) && S3D_EXP == 20
// in-kernel stamps (development): cycles per phase of a workgroup-plane, summed per wave, for a few workgroups
__device__ unsigned long long g_stamp[64][4][10];
#define S3D_STAMP(i) { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[i] += t_ - st_last; st_last = t_; }
#else
#define S3D_STAMP(i)
#endif
// development diagnostics, timing only (never set in the product build; results are wrong by construction).  Mask values:
// 1 no tile loads, 2 no stores (kept alive behind a runtime-false test), 4 no x-blur, 8 no y-blur, 16 no z-blur, 32 no
// barriers, 64 every tile treated as interior, 256 / 512 / 1024 no x / top / bottom edge handling (fast form)
#ifndef S3D_DIAG
#define S3D_DIAG 0
#endif

// Kernel-argument forms of Taps / EdgeFrac sized for the fused instantiations (hw <= 8): the general structs hold 65 taps
// for the separable path, and a larger argument block costs this kernel 1-2 % (scalar loads / SGPR pressure).
constexpr int kFusedMaxHW = 8;
struct FTaps { float w[2 * kFusedMaxHW + 1]; };
struct FEdge { float f[3][kFusedMaxHW + 1]; };

typedef float f2 __attribute__((ext_vector_type(2)));  // x-pair of a thread's 16-byte piece (register ring element)
typedef float f4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence over ALL memory: hipcc puts
// s_waitcnt vmcnt(0) in front of s_barrier whenever global stores are pending, which made every plane wait for the stores
// it had just issued (in-kernel stamps: 1 600 cycles per plane).  The global stores of this kernel are never read by the
// workgroup, so waiting for the LDS queue is sufficient.
__device__ __forceinline__ void lds_barrier() {
#if defined(S3D_DIAG) && (S3D_DIAG & 32)
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

__device__ __forceinline__ void store_f4_untracked(float *p, float4 v) {
	typedef float f4v __attribute__((ext_vector_type(4)));
	const f4v d = {v.x, v.y, v.z, v.w};
	asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 2" ::"v"(p), "v"(d));  // nt / sc0 sc1 policies: measured, no difference
}

// LDS-DMA (gfx950 global_load_lds_dwordx4): each lane's 16 bytes at `base + voff` go straight to LDS at lds_dst + 16*lane --
// no destination registers, no ds_write pass.  M0 carries the wave-uniform LDS byte address and is restored (compiler-reserved).
// Not tracked by hipcc: completion is awaited with explicit counted s_waitcnt vmcnt (see the plane loop).
__device__ __forceinline__ void glds16(const float *base, unsigned voff, unsigned lds_dst) {
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ float absmax_step_f(float m, float v) {
	const float a = fabsf(v);
	return (a > m) ? a : m;
}

template <int HW>
struct FusedCfg {
#ifndef S3D_TX
#define S3D_TX 32
#endif
#ifndef S3D_TY
#define S3D_TY (1024 / S3D_TX)
#endif
	static constexpr int TX = S3D_TX, TY = S3D_TY, NT = TX * TY / 4;  // 4 outputs (consecutive x, one 16-B piece) per thread
	static constexpr int NW = NT / 64;                              // waves per workgroup
	// DoG centre values G[i-1](p), p = q - HW: the raw tile of plane q holds them, so each thread parks its 16-B piece in a
	// private LDS ring of HW+1 planes instead of re-reading G[i-1] from memory (3.1 GB of 16.7 GB per 512^3 pyramid).  The
	// ring does not fit beside three resident workgroups at hw 8; that level keeps the global re-read.
#ifndef S3D_CRING_MASK
#define S3D_CRING_MASK 0x7C  /* bit HW set: DoG centre ring in LDS for that half width (hw 2..6) */
#endif
	static constexpr bool CRING = ((S3D_CRING_MASK >> HW) & 1) != 0;
	static constexpr bool CRING_DMA = HW <= 4;  // beside the double-buffered DMA tile the ring only fits three workgroups per CU up to hw 4
	static constexpr int CR = HW + 1;
	static constexpr int XP = TX + 16;                             // xb row pitch: 16-B column reads of 8 consecutive rows use every bank once per 4 rows
	static constexpr int SEGS = TX / 8;                            // x-blur items per row (8 outputs each)
	static constexpr int HXL = ((HW + 1 + 3) / 4) * 4;  // low-side x halo: the right-boundary rule reaches p-hw-1
	static constexpr int HXH = ((HW + 3) / 4) * 4;
	static constexpr int W = HXL + TX + HXH;            // tile row width (floats, multiple of 4)
	static constexpr int W4 = W / 4;
	static constexpr int PITCH = W + 4;                 // +16 B: rows shift by 4 banks
	static constexpr int ROWS = TY + 2 * HW + 1;        // row 0 = low halo HW+1 (only bottom tiles need it), then HW + TY + HW
	static constexpr int RING = 2 * HW + 2;             // planes p-hw-1 .. p+hw
	static constexpr int WSTART = (HXL - HW) & ~3;      // 16-B aligned start of a thread's x window
	static constexpr int WOFF = HXL - HW - WSTART;      // window index of input x-hw for output j=0
	static constexpr int WN4 = (WOFF + 8 + 2 * HW + 3) / 4;
	static constexpr int NLD = (ROWS * W4 + NT - 1) / NT;  // float4 tile loads per thread and plane
	// workgroups per CU the register budget is set for; 3 at hw 8 costs a dozen spilled VGPRs and is still ahead of 2
	// (packed v_pk_mul_f32 / v_pk_add_f32 forms of the three blurs were measured bit-exact and no faster -- packed fp32 runs at
	// the scalar flop rate on gfx950 -- and live in scripts/experiments/kernels_fused_r01e_all_variants.hip.txt)
#ifndef S3D_OCC_LO
#define S3D_OCC_LO 3  /* r02: 3 workgroups per CU also at hw <= 4 (4.09 -> 3.99 ms per 512^3 pyramid: these levels are bandwidth-bound and
                         fewer resident tiles keep more of the shared halo lines in the XCD's L2) */
#endif
#ifndef S3D_OCC_HI
#define S3D_OCC_HI 3
#endif
#ifndef S3D_OCC_8
#define S3D_OCC_8 S3D_OCC_HI
#endif
#ifndef S3D_OCC_5
#define S3D_OCC_5 S3D_OCC_HI
#endif
	static constexpr int OCC = HW <= 4 ? S3D_OCC_LO : (HW >= 8 ? S3D_OCC_8 : (HW == 5 ? S3D_OCC_5 : S3D_OCC_HI));
};

// reference boundary coordinate rule for output position p, tap offset d, axis length n
__device__ __forceinline__ void boundary_src(int p, int d, int n, int &lo, int &hi, float &frac) {
	const int dim_end = n - 1;
	float c = (float)p - (float)d * 1.0f;
	if (c < 0.0f) c = -1.0f * c;
	else if (c >= (float)dim_end) c = (float)(2 * dim_end) - c - 0.1f;
	lo = (int)c;
	frac = c - (float)lo;
	hi = lo + 1;
	lo = min(max(lo, 0), dim_end);  // the reference reads out of bounds when n <= 9 and hw == 8; clamp
	hi = min(max(hi, 0), dim_end);
}

// ring insert at register K + interior z-blur; the caller keeps the ring in plane order (K = RING-1 is the newest plane), so
// every ring index is static
template <int HW, int K>
__device__ __forceinline__ void zblur_static(f2 (&ring)[2][2 * HW + 2], const f2 (&v)[2], const FTaps &t, bool interior,
                                             f2 (&out)[2]) {
	constexpr int RING = 2 * HW + 2;
#pragma unroll
	for (int j = 0; j < 2; j++) ring[j][K] = v[j];
	if (S3D_DIAG & 16) {
		out[0] = ring[0][(K - HW + 2 * RING) % RING]; out[1] = ring[1][(K - HW + 2 * RING) % RING];
	} else if (interior) {
#pragma unroll
		for (int j = 0; j < 2; j++) {
			float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
			for (int d = -HW; d <= HW; d++) {  // plane p-d, p = q-HW
				a0 = a0 + t.w[d + HW] * ring[j][(K - HW - d + 2 * RING) % RING].x;
				a1 = a1 + t.w[d + HW] * ring[j][(K - HW - d + 2 * RING) % RING].y;
			}
			out[j] = f2{a0, a1};
		}
	}
}

// VEC: nx % 4 == 0, so every 16-B tile piece is either fully inside the volume or fully outside and all
// tile loads are branch-free (clamped address + select) -- essential: a branch around a load makes hipcc wait
// for it at the join, which serialises the prefetch.  !VEC (odd widths, small volumes) keeps guarded scalar loads.
// DMA (with VEC): the tile of plane q+1 travels global -> LDS by LDS-DMA into the other half of a double-buffered, lane-linear
// tile (row pitch W, piece i of the plane at float offset 4*i) while plane q is processed: no prefetch registers, no deposit
// pass, and the wave never waits for the load it has just issued.
template <int HW, bool DOG, bool VEC, bool DMA, bool CR_ON>
__device__ __forceinline__ void fused_level_body(const float *__restrict__ src, float *__restrict__ dst, float *__restrict__ dog,
                                                 unsigned *__restrict__ dogmax, int nx, int ny, const ZRange zr, const FTaps &t,
                                                 const FEdge &ef, int ntx, int nty, int cz, float *in_t, float *xb,
                                                 float *s_red, float *s_ef, float4 *cring) {
	using C = FusedCfg<HW>;
	static_assert(!DMA || VEC, "LDS-DMA tiles need 16-byte pieces");
	constexpr int IP = DMA ? C::W : C::PITCH;            // row pitch of the LDS tile
	constexpr int BUF = C::NLD * C::NT * 4;              // floats per DMA tile buffer (whole wave-instructions: 1 KiB each)
	// the boundary fractions are read through LDS: indexed dynamically out of the kernel-argument struct they would be global
	// loads whose s_waitcnt vmcnt(0) makes EDGE tiles drain the tile prefetch every plane (and the slowest workgroup sets the
	// kernel time)
	if (threadIdx.x < 2 * (kFusedMaxHW + 1)) s_ef[threadIdx.x] = ef.f[threadIdx.x / (kFusedMaxHW + 1)][threadIdx.x % (kFusedMaxHW + 1)];
	__syncthreads();

	// ---- XCD-aware, bijective block -> (chunk, tile) mapping: blocks b, b+8, ... share an XCD ----
	const int nblocks = gridDim.x;
	int lb;
	{
		const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
		const int per = nblocks >> 3, rem = nblocks & 7;
		lb = xcd * per + min(xcd, rem) + idx;
	}
	const int tile_x = lb % ntx;
	const int tile_y = (lb / ntx) % nty;
	const int chunk = lb / (ntx * nty);
	const int x0 = tile_x * C::TX, y0 = tile_y * C::TY;
	const int nz = zr.nz, zoff = zr.zoff, nzg = zr.nzg;  // local planes, global z of local plane 0, global planes
	const int zc0 = zr.zo0 + chunk * cz, zc1 = min(zr.zo1, zc0 + cz);

	// output mapping: thread (xq, ty) owns the 16-B piece x = 4*xq .. 4*xq+3 of tile row ty: one dwordx4 load (DoG centre) and
	// two dwordx4 stores per plane instead of 12 scalar VMEM instructions (VMEM issue, not bytes, was the cost)
	const int tid = threadIdx.x, xq = tid % (C::TX / 4), ty = tid / (C::TX / 4), wlane = tid & 63, wid = tid >> 6;
	// Edge tiles, fast form (volumes whose width and height are multiples of the tile; S3D_EDGE_FAST): the halo extension is
	// produced where its source values already sit in registers, so edge tiles run the same two barriers per plane as interior
	// tiles (a launch is one residency round: it lasts as long as its slowest tile).
	//   left   E[-k] = src[k] and right E[xend+k] = (1-f_k) src[xend-k-1] + f_k src[xend-k]: formed in the register window of the
	//          x-blur items that read them (first / last segment of a row; static window indices because xend = x0 + 31); the
	//          LDS tile keeps the raw values, so the DoG centre ring needs no care
	//   top    E[-k] = xb[k]: the x-blur item of row k stores its result a second time in row -k
	//   bottom E[yend+k] = (1-f_k) xb[yend-k-1] + f_k xb[yend-k]: the fourth wave (it holds no x-blur item) blurs rows
	//          yend-hw-1 .. yend again, one row per lane group, and combines neighbouring rows through a wave shuffle; it also
	//          takes the extra low halo row the bottom tiles need
	// Other shapes keep the patch loops below (two to three extra barriers per plane on edge tiles).
#ifndef S3D_EDGE_FAST
#define S3D_EDGE_FAST 1
#endif
	constexpr bool kFastOK = S3D_EDGE_FAST && VEC && !DMA && C::NT == 256 && C::TX == 32 && C::TY == 32 && !(S3D_DIAG & 64);
	const bool fast = kFastOK && (nx % C::TX == 0) && (ny % C::TY == 0);
	const bool left_f = fast && !(S3D_DIAG & 256) && x0 == 0, right_f = fast && !(S3D_DIAG & 256) && x0 + C::TX == nx;
	const bool top_f = fast && !(S3D_DIAG & 512) && y0 == 0, bottom_f = fast && !(S3D_DIAG & 1024) && y0 + C::TY == ny;
	const bool edge_x = !fast && !(S3D_DIAG & 64) && ((x0 < HW) || (x0 + C::TX - 1 > nx - 2 - HW));
	const bool edge_y = !fast && !(S3D_DIAG & 64) && ((y0 < HW) || (y0 + C::TY - 1 > ny - 2 - HW));
	const bool load_row0 = !(S3D_DIAG & 64) && (y0 + C::TY - 1 > ny - 2 - HW);  // tile holds right-boundary y outputs (they reach y-hw-1)
	const bool need_row0 = load_row0 && !fast;                                 // ... blurred by the loop below (fast: by wave 3)
	const int sy = nx, sz = nx * ny;                         // levels are < 2^31 voxels (checked at create)
	const bool full_tile = (x0 + C::TX <= nx) && (y0 + C::TY <= ny);  // every output of the tile is inside the volume

	// per-thread load items (independent of the plane): plane-relative global offset or -1, LDS offset
	int ld_goff[C::NLD], ld_lds[C::NLD];
#pragma unroll
	for (int i = 0; i < C::NLD; i++) {
		const int item = tid + i * C::NT;
		const int r = item / C::W4, c4 = item - r * C::W4;
		const int gy = y0 - HW - 1 + r, gx = x0 - C::HXL + 4 * c4;
		bool ok = item < C::ROWS * C::W4 && gy >= 0 && gy < ny && (r > 0 || load_row0);
		if (VEC) ok = ok && gx >= 0 && gx + 3 < nx;
		ld_goff[i] = ok ? gy * sy + gx : (DMA ? 0 : -1);  // DMA: clamped to a valid piece (its LDS image is never used: halo patches
		                                                   // overwrite what edge outputs read, everything else feeds discarded outputs)
		// items past the end of the tile write a dump slot behind it: the LDS writes stay branch-free, so the compiler can wait for
		// exactly the load it needs (vmcnt(k)) instead of draining everything at the join of an exec-masked block
		ld_lds[i] = item < C::ROWS * C::W4 ? r * C::PITCH + 4 * c4 : C::ROWS * C::PITCH;
	}
	const unsigned lds_tile = (unsigned)(unsigned long long)in_t;  // LDS byte address (low 32 bits of the flat address)
	const unsigned lds_wave = lds_tile + (unsigned)__builtin_amdgcn_readfirstlane(wid) * 1024u;
	auto issue_plane_dma = [&](int q) {  // plane q -> buffer q & 1
		const float *plane = src + (size_t)sz * (size_t)min(max(q, 0), nz - 1);
		const unsigned dstb = lds_wave + (unsigned)(q & 1) * (unsigned)(BUF * 4);
#pragma unroll
		for (int i = 0; i < C::NLD; i++) glds16(plane, (unsigned)ld_goff[i] * 4u, dstb + (unsigned)(i * C::NT * 16));
	};
	float4 pf[C::NLD];  // prefetched tile pieces of the NEXT plane
#pragma unroll
	for (int i = 0; i < C::NLD; i++) pf[i] = make_float4(0.f, 0.f, 0.f, 0.f);

	auto issue_plane_loads = [&](int q) {
		if (S3D_DIAG & 1) return;
		if (DMA) { issue_plane_dma(q); return; }
		// no branch around the loads (a join would make the compiler drain them): planes outside [0, nz) are
		// clamped to a valid plane and simply never used
		if (!VEC && (q < 0 || q >= nz)) return;
		const float *plane = src + (size_t)sz * (size_t)min(max(q, 0), nz - 1);
#pragma unroll
		for (int i = 0; i < C::NLD; i++) {
			float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
			if (VEC) {
				const int g = ld_goff[i];
				// unconditional and always in bounds; pieces outside the volume are zeroed when the registers are
				// written to LDS one step later (a select here would make the compiler wait for the load right away)
				val = *reinterpret_cast<const float4 *>(plane + (g < 0 ? 0 : g));
			} else if (ld_goff[i] >= 0) {
				{
					const int item = tid + i * C::NT;
					const int r = item / C::W4, c4 = item - r * C::W4;
					const int gx = x0 - C::HXL + 4 * c4;
					const float *row = plane + (ld_goff[i] - gx);
					if (gx >= 0 && gx < nx) val.x = row[gx];
					if (gx + 1 >= 0 && gx + 1 < nx) val.y = row[gx + 1];
					if (gx + 2 >= 0 && gx + 2 < nx) val.z = row[gx + 2];
					if (gx + 3 >= 0 && gx + 3 < nx) val.w = row[gx + 3];
				}
			}
			pf[i] = val;
		}
	};

	f2 ring[2][C::RING];  // x-pairs (0,1) and (2,3) of the thread's piece
#pragma unroll
	for (int j = 0; j < 2; j++)
#pragma unroll
		for (int k = 0; k < C::RING; k++) ring[j][k] = f2{0.f, 0.f};
	float mx = 0.0f;

	const int gx_out = x0 + 4 * xq;
	const int out_off = gx_out + (y0 + ty) * sy;  // plane-relative offset of this thread's first output
	const bool row_ok = y0 + ty < ny;

#if defined(S3D_EXP) && S3D_EXP == 20
	unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter();
#endif
	const int q_begin = zc0 - HW - 1, q_end = zc1 - 1 + HW;  // inclusive
	// DoG centre values G[i-1](p): requested one whole plane ahead (at the END of the previous iteration, after its
	// stores).  vmcnt retires in order, so requesting them before the tile prefetch of the same plane -- or consuming them in
	// the plane that requested them -- makes the DoG wait for the prefetch as well and exposes the HBM latency every plane
	// (in-kernel stamps, r01c: 1 800 of 7 400 cycles per plane at hw 8 sat in "DoG + stores").
	float cen[4] = {0.f, 0.f, 0.f, 0.f};
	// branch-free (a branch around a load makes hipcc drain it at the join): addresses of outputs outside the volume are
	// clamped to the tile's first valid element, their values are never used
	int cen_off[4];
	bool cen_vec = false;
	{
		const int safe = min(x0, nx - 1) + min(y0, ny - 1) * sy;
		if (VEC) {
			cen_vec = true;
			cen_off[0] = (row_ok && gx_out + 3 < nx) ? out_off : (safe & ~3);
			cen_off[1] = cen_off[2] = cen_off[3] = 0;
		} else {
#pragma unroll
			for (int j = 0; j < 4; j++) cen_off[j] = (row_ok && gx_out + j < nx) ? out_off + j : safe;
		}
	}
	auto request_centres = [&](int pn) {
		if (!DOG) return;
		const float *cp = src + (size_t)sz * (size_t)min(max(pn, 0), nz - 1);
		if (VEC) {
			const float4 c4 = *reinterpret_cast<const float4 *>(cp + cen_off[0]);
			cen[0] = c4.x; cen[1] = c4.y; cen[2] = c4.z; cen[3] = c4.w;
		} else {
#pragma unroll
			for (int j = 0; j < 4; j++) cen[j] = cp[cen_off[j]];
		}
	};
	(void)cen_vec;
	auto plane_valid = [&](int qq) { return (qq >= 0 && qq < nz) && (qq + zoff >= 0) && (qq + zoff < nzg); };
	auto deposit_tile = [&](int qq) {  // prefetched registers of plane qq -> LDS tile
		if (DMA) return;
		// unconditional: a plane outside the volume goes to the dump slot.  If the registers were not consumed on every path,
		// hipcc would make the NEXT request wait (write-after-write) -- with vmcnt(0), i.e. also for the stores just issued.
		const bool valid = plane_valid(qq);
#pragma unroll
		for (int i = 0; i < C::NLD; i++) {
			float4 w4 = pf[i];
			if (VEC) {  // selects, not branches
				const bool outside = ld_goff[i] < 0;
				w4.x = outside ? 0.f : w4.x; w4.y = outside ? 0.f : w4.y; w4.z = outside ? 0.f : w4.z; w4.w = outside ? 0.f : w4.w;
			}
			*reinterpret_cast<float4 *>(&in_t[valid ? ld_lds[i] : C::ROWS * C::PITCH]) = w4;
		}
	};
	// Schedule of the global memory operations (hipcc waits with vmcnt(0), i.e. for EVERYTHING outstanding, wherever a
	// prefetched register is consumed in this loop): the tile of plane q+1 is deposited in LDS AFTER the z-blur of plane q
	// (in_t is free since barrier B), immediately followed by the DoG and the stores, and only then are the tile of plane q+2
	// and the DoG centres of plane p+1 requested.  At the single wait per plane every outstanding load and store is one whole
	// plane old; nothing young is ever waited for.
	issue_plane_loads(q_begin);
	deposit_tile(q_begin);
	if (DMA) wait_vmcnt<0>();  // the first plane: one exposed latency per workgroup
	// DMA bookkeeping: VMEM operations this wave issued AFTER the DMA of the plane it is about to read (vmcnt retires in order):
	// the asm stores of the previous plane (0 in the ramp, 1 or 2 on the fast path), this plane's DoG centre request and the DMA
	// of the next plane.  2 = the previous plane stored through compiler-tracked stores (partial tiles): wait for everything.
	int prev_stores = 0;
	int cslot_w = 0;  // ring slot of plane q (planes enter one per iteration, so slot = iteration index mod CR)
	for (int q = q_begin; q <= q_end; q++) {
		const bool have_plane = (q >= 0 && q < nz) && (q + zoff >= 0) && (q + zoff < nzg);
		const int p = q - HW;
		const bool emit = (p >= zc0 && p < zc1);
		// DoG centre values of THIS plane: the first memory operation of the iteration, consumed after the z-blur together with
		// the tile prefetch (one wait, everything it covers is most of a plane old).  Not carried across the back-edge: hipcc
		// copies a loop-carried load result there and waits for it.
		if (!CR_ON) request_centres(p);
		issue_plane_loads(q + 1);  // tile of the next plane: consumed (deposited in LDS) after this plane's z-blur
		S3D_STAMP(8)  // issue of the centre + tile loads
		float *const tin = DMA ? in_t + (q & 1) * BUF : in_t;  // tile of plane q
		f2 v[2] = {f2{0.f, 0.f}, f2{0.f, 0.f}};
		if (have_plane) {
			if (DMA) {
				constexpr int kAfter = C::NLD + ((DOG && !CR_ON) ? 1 : 0), kSt = DOG ? 2 : 1;
				if (prev_stores == 1) wait_vmcnt<kAfter + kSt>();
				else if (prev_stores == 0) wait_vmcnt<kAfter>();
				else wait_vmcnt<0>();
			}
			lds_barrier();  // barrier A: tile visible (and every thread is done with the previous xb)
			S3D_STAMP(2)  // wait at barrier A
			if (CR_ON) {  // this thread's piece of the raw plane q -> its private ring slot
				float4 c4v = *reinterpret_cast<const float4 *>(&tin[(ty + HW + 1) * IP + C::HXL + 4 * xq]);
				cring[(cslot_w)*C::NT + tid] = c4v;
			}
			if (edge_x) {
				// the right-edge extension below overwrites column nx-1 of the tile with E[dim_end]: every thread must have parked
				// its RAW centre piece first (the left-edge mirror only writes halo columns)
				if (CR_ON && x0 + C::TX - 1 > nx - 2 - HW) lds_barrier();
				// x extension columns of the LDS tile (see header), every row of the tile
				const int xend = nx - 1;
				const int nleft = (x0 < HW) ? HW : 0;                       // x0 < HW  =>  x0 == 0
				const int nright = (x0 + C::TX - 1 > nx - 2 - HW) ? (HW + 1) : 0;
				const int ne = nleft + nright;
				// item / ne without an integer division (ne <= 17 is uniform, item < ROWS*ne < 3855: the 16-bit magic is exact)
				const unsigned ne_magic = 65536u / (unsigned)ne + 1u;
				for (int item = tid; item < C::ROWS * ne; item += C::NT) {
					const int r = (int)(((unsigned)item * ne_magic) >> 16), e = item - r * ne;
					float *trow = &tin[r * IP] + (C::HXL - x0);  // trow[gx] addresses volume column gx
					if (e < nleft) {
						const int k = e + 1;
						trow[-k] = trow[k];
					} else {
						const int k = e - nleft, m = xend - k;
						if (xend + k - x0 < C::TX + C::HXH) {
							const float f = s_ef[k];                 // x fractions
							const float a = trow[m - 1], b = trow[m];
							trow[xend + k] = (1.0f - f) * a + f * b;
						}
					}
				}
				lds_barrier();
			}
			// ---------------- x-blur: in_t rows 1..ROWS-1 -> xb (exactly one item per thread) ----------------
			const int xitem0 = tid;
			// one item (8 outputs of one tile row) per thread; the fourth wave holds none and serves the bottom tiles (see `fast`)
			int r = 1 + xitem0 / C::SEGS, seg = xitem0 % C::SEGS, xmode = 0;  // xmode 1: bottom helper row, 2: extra low halo row
			bool xact = !(S3D_DIAG & 4) && xitem0 < (C::ROWS - 1) * C::SEGS;
			if (bottom_f && tid >= 192) {  // wave-uniform
				const int l = tid - 192;
				if (l < (HW + 2) * 4) { xmode = 1; r = C::TY - 1 + (l >> 2); seg = l & 3; xact = true; }        // rows yend-hw-1 .. yend
				else if (l < (HW + 3) * 4) { xmode = 2; r = 0; seg = l & 3; xact = true; }                  // row y0-hw-1
			}
			const int gy = y0 - HW - 1 + r;
			xact = xact && gy >= 0 && gy < ny && !(bottom_f && xmode == 0 && gy == ny - 1);  // row yend of a bottom tile receives E[yend]
			float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
			{
				if (xact) {
					const float *trow = &tin[r * IP];
					float win[C::WN4 * 4];
#pragma unroll
					for (int k = 0; k < C::WN4; k++) {
						const float4 f = *reinterpret_cast<const float4 *>(trow + C::WSTART + seg * 8 + 4 * k);
						win[4 * k] = f.x; win[4 * k + 1] = f.y; win[4 * k + 2] = f.z; win[4 * k + 3] = f.w;
					}
					// window index of tile column c for segment s: C::WOFF + HW + c - 8*s
					if (left_f && seg == 0) {
#pragma unroll
						for (int k = 1; k <= HW; k++) win[C::WOFF + HW - k] = win[C::WOFF + HW + k];
					}
					if (right_f && seg == C::SEGS - 1) {
						float e[HW + 1];
#pragma unroll
						for (int k = 0; k <= HW; k++) {
							const float f = s_ef[k];  // x fractions
							e[k] = (1.0f - f) * win[C::WOFF + HW + 6 - k] + f * win[C::WOFF + HW + 7 - k];
						}
#pragma unroll
						for (int k = 0; k <= HW; k++) win[C::WOFF + HW + 7 + k] = e[k];
					}
					if (HW == 8 && right_f && seg == C::SEGS - 2) {  // the window of segment 2 ends on column xend = E[xend]
						const float f = s_ef[0];
						win[C::WOFF + HW + 15] = (1.0f - f) * win[C::WOFF + HW + 14] + f * win[C::WOFF + HW + 15];
					}
#pragma unroll
					for (int j = 0; j < 8; j++) {
						float acc = 0.0f;
#pragma unroll
						for (int d = -HW; d <= HW; d++) acc = acc + t.w[d + HW] * win[C::WOFF + j + HW - d];
						o[j] = acc;
					}
				}
			}
				if (xact && xmode != 1) {
					float4 *xo = reinterpret_cast<float4 *>(&xb[r * C::XP + seg * 8]);
					xo[0] = make_float4(o[0], o[1], o[2], o[3]);
					xo[1] = make_float4(o[4], o[5], o[6], o[7]);
					if (top_f && gy >= 1 && gy <= HW) {  // E[-k] = row k: second copy in the mirror row (those rows hold no item)
						float4 *xm = reinterpret_cast<float4 *>(&xb[(HW + 1 - gy) * C::XP + seg * 8]);
						xm[0] = make_float4(o[0], o[1], o[2], o[3]);
						xm[1] = make_float4(o[4], o[5], o[6], o[7]);
					}
				}
				if (bottom_f && tid >= 192) {  // wave-uniform: every lane of the fourth wave takes part in the shuffles
					float a[8];
#pragma unroll
					for (int j = 0; j < 8; j++) a[j] = __shfl_up(o[j], 4, 64);  // the row below (same segment)
					const int jr = (tid - 192) >> 2;                            // this lane holds xb row yend-hw-1+jr
					if (xmode == 1 && jr >= 1) {
						const int k = HW + 1 - jr;                                  // E[yend+k] = (1-f_k) xb[yend-k-1] + f_k xb[yend-k]
						const float f = s_ef[kFusedMaxHW + 1 + k];                  // y fractions
						float4 *xe = reinterpret_cast<float4 *>(&xb[(C::TY + HW + k) * C::XP + seg * 8]);
						xe[0] = make_float4((1.0f - f) * a[0] + f * o[0], (1.0f - f) * a[1] + f * o[1], (1.0f - f) * a[2] + f * o[2],
						                    (1.0f - f) * a[3] + f * o[3]);
						xe[1] = make_float4((1.0f - f) * a[4] + f * o[4], (1.0f - f) * a[5] + f * o[5], (1.0f - f) * a[6] + f * o[6],
						                    (1.0f - f) * a[7] + f * o[7]);
					}
				}
			if (need_row0) {
				// bottom tiles only: the extra low halo row (one wave's worth of work, plain loop)
				const int gy = y0 - HW - 1;
				if (gy >= 0 && tid < C::TX) {
					const float *trow = &tin[0];
					float acc = 0.0f;
#pragma unroll
					for (int d = -HW; d <= HW; d++) acc = acc + t.w[d + HW] * trow[C::HXL + tid - d];  // unrolled: the LDS reads go out together
					xb[tid] = acc;
				}
			}
			if (edge_y) {
				// y extension rows of xb (see header): mirror above row 0, lerp rows from dim_end on
				lds_barrier();
				const int yend = ny - 1;
				const int ntop = (y0 < HW) ? HW : 0;                       // y0 < HW  =>  y0 == 0
				const int nbot = need_row0 ? (HW + 1) : 0;
				for (int item = tid; item < (ntop + nbot) * C::TX; item += C::NT) {
					const int e = item / C::TX, xx = item % C::TX;
					if (e < ntop) {
						const int k = e + 1;                                // E[-k] = row k
						xb[(HW + 1 - k - y0) * C::XP + xx] = xb[(HW + 1 + k - y0) * C::XP + xx];
					} else {
						const int k = e - ntop, m = yend - k;               // E[yend+k] = (1-f)*row[m-1] + f*row[m]
						const int rd = yend + k - (y0 - HW - 1);
						if (rd < C::ROWS) {
							const float f = s_ef[kFusedMaxHW + 1 + k];    // y fractions
							const float a = xb[(m - 1 - (y0 - HW - 1)) * C::XP + xx], b = xb[(m - (y0 - HW - 1)) * C::XP + xx];
							xb[rd * C::XP + xx] = (1.0f - f) * a + f * b;
						}
					}
				}
			}
			S3D_STAMP(3)  // x-blur
			lds_barrier();  // barrier B: xb visible, in_t free for the next plane
			S3D_STAMP(4)  // wait at barrier B
			// ---------------- y-blur: xb -> registers (the 4 x-neighbours of row ty: four independent chains) ----------------
			if (S3D_DIAG & 8) {
				const f4 c0 = *reinterpret_cast<const f4 *>(&xb[(ty + 1 + HW) * C::XP + 4 * xq]);
				v[0] = c0.xy; v[1] = c0.zw;
			} else {
				// rows are consumed in tap order from LDS in groups of kG, the next group requested before the current one is used
				// (a full register window of 2*HW+1 float4 would cost 68 VGPRs at hw 8; one row at a time exposes the LDS latency
				// 2*HW+1 times)
				constexpr int kG = 4, NT_ = 2 * HW + 1;
				const float *ycol = &xb[(ty + 1) * C::XP + 4 * xq];
				// step s = d + HW = 0 .. 2*HW in the reference's order; it reads row (2*HW - s)
				f4 cur[kG], nxt[kG];
#pragma unroll
				for (int i = 0; i < kG; i++) cur[i] = *reinterpret_cast<const f4 *>(ycol + (i < NT_ ? 2 * HW - i : 0) * C::XP);
#pragma unroll
				for (int g0 = 0; g0 < NT_; g0 += kG) {
#pragma unroll
					for (int i = 0; i < kG; i++)
						if (g0 + kG + i < NT_) nxt[i] = *reinterpret_cast<const f4 *>(ycol + (2 * HW - (g0 + kG + i)) * C::XP);
#pragma unroll
					for (int i = 0; i < kG; i++) {
						const int st = g0 + i;
						if (st < NT_) {
							const float tap = t.w[st];
							v[0].x = v[0].x + tap * cur[i].x; v[0].y = v[0].y + tap * cur[i].y;
							v[1].x = v[1].x + tap * cur[i].z; v[1].y = v[1].y + tap * cur[i].w;
						}
					}
#pragma unroll
					for (int i = 0; i < kG; i++) cur[i] = nxt[i];
				}
			}
		}
		S3D_STAMP(5)  // y-blur
		// ---------------- ring insert (slot q mod RING) + z-blur of plane p = q - HW ----------------
		const int pg = p + zoff;  // global plane
		const bool z_interior = emit && (pg >= HW) && (pg <= nzg - 2 - HW);
		f2 outp[2] = {f2{0.f, 0.f}, f2{0.f, 0.f}};
		// ring kept in plane order (oldest first) and shifted by one register per plane: RING moves per pair.  (Addressing a
		// ring that stays in place through an 18-way switch with 18 copies of the z-blur cost 4 %: instruction-cache footprint.)
#pragma unroll
		for (int k = 0; k + 1 < C::RING; k++) { ring[0][k] = ring[0][k + 1]; ring[1][k] = ring[1][k + 1]; }
		zblur_static<HW, C::RING - 1>(ring, v, t, z_interior, outp);
		S3D_STAMP(6)  // z-blur
		deposit_tile(q + 1);
		S3D_STAMP(0)  // wait for the prefetch + tile registers -> LDS
		float out[4] = {outp[0].x, outp[0].y, outp[1].x, outp[1].y};
		if (emit) {
			if (!z_interior) {
				// wave-uniform tap sources; plane s sits in slot s mod RING
				float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
				for (int d = -HW; d <= HW; d++) {
					int lo, hi;
					float frac;
					boundary_src(pg, d, nzg, lo, hi, frac);  // global rule; the ring is keyed by LOCAL plane index
					// plane s sits RING-1-(q-s) registers from the start of the ring
					const int slo = min(max(C::RING - 1 - (q - (lo - zoff)), 0), C::RING - 1), shi = min(max(C::RING - 1 - (q - (hi - zoff)), 0), C::RING - 1);
					const float tap = t.w[d + HW];
#pragma unroll
					for (int j = 0; j < 4; j++) {
						float a = (j & 1) ? ring[j >> 1][0].y : ring[j >> 1][0].x, b = a;
#pragma unroll
						for (int k = 1; k < C::RING; k++) {
							const float rk = (j & 1) ? ring[j >> 1][k].y : ring[j >> 1][k].x;
							a = (slo == k) ? rk : a;
							b = (shi == k) ? rk : b;
						}
						acc[j] = acc[j] + tap * ((1.0f - frac) * a + frac * b);
					}
				}
#pragma unroll
				for (int j = 0; j < 4; j++) out[j] = acc[j];
			}
		}
		if (CR_ON) {
			// plane p = q - HW sits HW slots behind the one written in this iteration: (cslot_w + 1) mod CR
			const float4 c4 = cring[(cslot_w + 1 == C::CR ? 0 : cslot_w + 1) * C::NT + tid];
			cen[0] = c4.x; cen[1] = c4.y; cen[2] = c4.z; cen[3] = c4.w;
		}
		float dg[4];
#pragma unroll
		for (int j = 0; j < 4; j++) dg[j] = (out[j] - cen[j]) * (-1.0f);
		// loads BEFORE the stores: a load that re-uses a register a pending store still reads makes hipcc wait for the
		// store to complete (vmcnt(0)); this way nothing is written that an older memory operation reads
		S3D_STAMP(7)  // DoG + issue of the prefetch + DoG centre loads
		prev_stores = emit ? ((full_tile && VEC && !(S3D_DIAG & 2)) ? 1 : 2) : 0;
		if (emit) {
			const size_t base = (size_t)sz * (size_t)p + (size_t)out_off;
			if ((S3D_DIAG & 2) && !(out[0] == 12345.678f && dg[1] == 3.25f)) {
			} else if (full_tile && VEC) {
				// The hot-path stores are issued through inline asm so that hipcc does not track them: it would otherwise make
				// the next loads that re-use the stores' data registers wait for the stores to COMPLETE (s_waitcnt vmcnt).  The
				// hardware only needs the data registers to be read, which the wait states below cover; nothing in this kernel
				// reads dst / dog back.
				store_f4_untracked(dst + base, make_float4(out[0], out[1], out[2], out[3]));
				if (DOG) {
					store_f4_untracked(dog + base, make_float4(dg[0], dg[1], dg[2], dg[3]));
#pragma unroll
					for (int j = 0; j < 4; j++) mx = absmax_step_f(mx, dg[j]);
				}
			} else if (row_ok) {
#pragma unroll
				for (int j = 0; j < 4; j++) {
					if (gx_out + j < nx) {
						dst[base + j] = out[j];
						if (DOG) { dog[base + j] = dg[j]; mx = absmax_step_f(mx, dg[j]); }
					}
				}
			}
		}
		S3D_STAMP(1)  // stores
		cslot_w = cslot_w + 1 == C::CR ? 0 : cslot_w + 1;
	}
#if defined(S3D_EXP) && S3D_EXP == 20
	if (HW == S3D_STAMP_HW && blockIdx.x < 64 && wlane == 0)
		for (int i = 0; i < 10; i++) g_stamp[blockIdx.x][wid][i] = st_acc[i];
#endif
	if (DOG) {
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
		__syncthreads();
		if (wlane == 0) s_red[wid] = mx;
		__syncthreads();
		if (tid == 0) {
			float r = s_red[0];
#pragma unroll
			for (int w = 1; w < C::NW; w++) r = fmaxf(r, s_red[w]);
			if (r > 0.0f) atomicMax(dogmax, __float_as_uint(r));
		}
	}
}

#ifndef S3D_DMA_MASK
#define S3D_DMA_MASK 0  /* bit HW set: LDS-DMA tile path for that half width (widths that are multiples of 4).  Measured r02: bit-exact,
                          frees the 12 prefetch registers (no spills at hw 8) and the deposit pass, hw 6 / 8 levels 5-8 % faster in
                          isolation, hw 3 8 % slower, whole pyramid unchanged (4.05 vs 4.03 ms) -> off */
#endif
template <int HW, bool DOG>
__global__ void __launch_bounds__(FusedCfg<HW>::NT, (FusedCfg<HW>::OCC * 4 + FusedCfg<HW>::NW - 1) / FusedCfg<HW>::NW) k_fused_level(const float *__restrict__ src, float *__restrict__ dst,
                                                                       float *__restrict__ dog, unsigned *__restrict__ dogmax,
                                                                       int nx, int ny, ZRange zr, FTaps t, FEdge ef, int ntx, int nty,
                                                                       int cz) {
	using C = FusedCfg<HW>;
	constexpr bool DMA = ((S3D_DMA_MASK >> HW) & 1) != 0;
	constexpr int kTile = C::ROWS * C::PITCH + 4, kTileDma = 2 * C::NLD * C::NT * 4;
	constexpr bool CR = DOG && (DMA ? C::CRING_DMA : C::CRING);  // DoG centre ring in LDS (both bodies of this kernel follow it)
	__shared__ __attribute__((aligned(16))) float in_t[(DMA && kTileDma > kTile) ? kTileDma : kTile];  // + dump slot, see ld_lds
	__shared__ __attribute__((aligned(16))) float xb[C::ROWS * C::XP];
	__shared__ float s_red[C::NW];
	__shared__ float s_ef[2 * (kFusedMaxHW + 1)];
	__shared__ __attribute__((aligned(16))) float4 cring[CR ? C::CR * C::NT : 1];
	if ((nx & 3) == 0) fused_level_body<HW, DOG, true, DMA, CR>(src, dst, dog, dogmax, nx, ny, zr, t, ef, ntx, nty, cz, in_t, xb, s_red, s_ef, cring);
	else fused_level_body<HW, DOG, false, false, CR>(src, dst, dog, dogmax, nx, ny, zr, t, ef, ntx, nty, cz, in_t, xb, s_red, s_ef, cring);
}

// fp32 fractions of the reference's right-boundary rule for an axis of length n (see file header):
// position dim_end+k maps to c' = 2*dim_end - c - 0.1f, lo = (int)c', frac = c' - lo   (Src/cSIFT3D.cc:751-760)
static void edge_fractions(int n, int hw, float *f) {
	const int dim_end = n - 1;
	for (int k = 0; k <= hw; k++) {
		const float c = (float)(dim_end + k);
		const float cc = (float)(2 * dim_end) - c - 0.1f;
		const int lo = (int)cc;
		f[k] = cc - (float)lo;
	}
}

template <int HW>
static void launch_hw(const float *src, float *dst, float *dog, unsigned *dogmax, int nx, int ny, const ZRange &zr, const Taps &tg,
                      hipStream_t st, int plan_slots) {
	static_assert(HW <= kFusedMaxHW, "argument structs");
	FTaps t;
	for (int i = 0; i < 2 * kFusedMaxHW + 1; i++) t.w[i] = i < 2 * HW + 1 ? tg.w[i] : 0.0f;
	const int nz = zr.zo1 - zr.zo0;  // planes to produce
	if (nz <= 0) return;
	using C = FusedCfg<HW>;
	const int ntx = (nx + C::TX - 1) / C::TX, nty = (ny + C::TY - 1) / C::TY;
	const int ntiles = ntx * nty;
	// z chunking: pick the chunk count that minimises (residency rounds) x (planes marched per workgroup);
	// every chunk pays a ramp of 2*HW+1 planes, every partially filled round leaves CUs idle
	// workgroup slots the chunking is planned for: the whole machine by default; the caller passes fewer when other octaves'
	// launches are meant to run beside this one (a single-round launch holds every slot it takes for its whole duration)
	const int full_slots = 256 * ((C::OCC * 4 + C::NW - 1) / C::NW);  // OCC counts 4-wave units per CU
	const int slots = plan_slots > 0 ? std::min(plan_slots, full_slots) : full_slots;
	const int ramp = 2 * HW + 1;
	int best_cz = nz;
	double best_cost = 1e300;
	// Levels with few tiles (octaves >= 1) run beside the big octave-0 launches on their own streams: what they cost the
	// machine is their total work (workgroup-planes), not their latency, so their chunks are kept >= bg_k ramps long
	static const double bg_k = dev_tune_d("S3D_BG_K", S3D_BG_K_DEFAULT);
	const bool background = ntiles * 4 <= slots;
	for (int n = 1; n <= nz && n <= 64; n++) {
		const int czn = (nz + n - 1) / n;
		if (background && n > 1 && czn < bg_k * ramp) break;
		const int nch = (nz + czn - 1) / czn;
		const long wgs = (long)ntiles * nch;
		const long rounds = (wgs + slots - 1) / slots;
		const double cost = (double)rounds * (czn + ramp);
		if (cost < best_cost - 1e-9) { best_cost = cost; best_cz = czn; }
	}
	const int cz = best_cz;
	const int nchunks = (nz + cz - 1) / cz;
	FEdge ef;
	memset(&ef, 0, sizeof(ef));
	edge_fractions(nx, HW, ef.f[0]);
	edge_fractions(ny, HW, ef.f[1]);
	edge_fractions(zr.nzg, HW, ef.f[2]);
	dim3 grid((unsigned)(ntiles * nchunks)), block(C::NT);
#if defined(S3D_EXP) && S3D_EXP == 20
	if (HW == S3D_STAMP_HW && nx >= 512 && dog) {
		hipLaunchKernelGGL((k_fused_level<HW, true>), grid, block, 0, st, src, dst, dog, dogmax, nx, ny, zr, t, ef, ntx, nty, cz);
		hipStreamSynchronize(st);
		static unsigned long long h[64][4][10];
		hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h));
		const char *nm[10] = {"wait+tile->LDS", "stores", "wait bar A", "x-blur", "wait bar B", "y-blur", "z-blur", "DoG", "issue loads", "-"};
		const int planes = cz + 2 * HW + 1;
		for (int w = 0; w < 4; w++) {
			fprintf(stderr, "STAMP hw %d wave %d (cycles per plane, mean of 64 WGs, %d planes):", HW, w, planes);
			double tot = 0;
			for (int i = 0; i < 9; i++) { double a = 0; for (int b = 0; b < 64; b++) a += (double)h[b][w][i]; a /= 64.0 * planes; tot += a; fprintf(stderr, " %s %.0f", nm[i], a); }
			fprintf(stderr, " | total %.0f\n", tot);
		}
		return;
	}
#endif
	if (dog) hipLaunchKernelGGL((k_fused_level<HW, true>), grid, block, 0, st, src, dst, dog, dogmax, nx, ny, zr, t, ef, ntx, nty, cz);
	else hipLaunchKernelGGL((k_fused_level<HW, false>), grid, block, 0, st, src, dst, dog, dogmax, nx, ny, zr, t, ef, ntx, nty, cz);
}

// returns false when no fused instantiation exists for this half width (caller uses the generic
// separable kernels of kernels_pyramid.hip instead -- still the HIP path)
bool launch_fused_level(const float *src, float *dst, float *dog, unsigned *dogmax, int nx, int ny, const ZRange &zr,
                        const Taps &t, hipStream_t st, int plan_slots, int prio) {
	// the halo-extension form of the boundary rule needs n >= 2*hw+2 along x and y (see header)
	if (nx < 2 * t.hw + 2 || ny < 2 * t.hw + 2) return false;
	if (launch_march_level(src, dst, dog, dogmax, nx, ny, zr, t, st, plan_slots, prio)) return true;  // tile-aligned shapes (kernels_march.hip)
	switch (t.hw) {
	case 2: launch_hw<2>(src, dst, dog, dogmax, nx, ny, zr, t, st, plan_slots); return true;
	case 3: launch_hw<3>(src, dst, dog, dogmax, nx, ny, zr, t, st, plan_slots); return true;
	case 4: launch_hw<4>(src, dst, dog, dogmax, nx, ny, zr, t, st, plan_slots); return true;
	case 5: launch_hw<5>(src, dst, dog, dogmax, nx, ny, zr, t, st, plan_slots); return true;
	case 6: launch_hw<6>(src, dst, dog, dogmax, nx, ny, zr, t, st, plan_slots); return true;
	case 8: launch_hw<8>(src, dst, dog, dogmax, nx, ny, zr, t, st, plan_slots); return true;
	default: return false;
	}
}

}  // namespace s3d
