// kernels_march.hip -- the pyramid hot kernel (levels of at least 32 x 32 voxels per plane): ONE pass over HBM per level,
// one workgroup barrier per plane, no register ring.
//
//   G[i] = gauss_z(gauss_y(gauss_x(G[i-1])))   and   DoG[i-1] = (G[i] - G[i-1]) * (-1)   and   max|DoG[i-1]|
//
// replaces, per level, GaussianSmooth_3D + Im_permute + Sub + im_max_abs (Src/cSIFT3D.cc:535-882, Src/cUtil.cc:587-605) exactly
// (r01's kernels_fused.hip did the same with a register ring and is gone since r03); the arithmetic contract: every output is the
// literal chain acc = acc + tap[d+hw]*term for d = -hw..+hw with separate IEEE multiply and add (-ffp-contract=off), boundary terms
// through the extended line E[] (E[-k] = src[k]; E[dim_end+k] = (1-f_k) src[dim_end-k-1] + f_k src[dim_end-k], see "Boundaries" below).
//
// Boundaries (Src/cSIFT3D.cc:722-788).  Along an axis of length n (dim_end = n - 1) output p in [hw, n-2-hw] is the plain tap chain; the
// others take, for every tap, the coordinate c = p - d mirrored at 0 (c < 0 -> -c) or mapped by c' = 2 dim_end - c - 0.1f (c >= dim_end)
// and the term (1 - frac) src[lo] + frac src[lo + 1], lo = (int)c', frac = c' - lo.  For c = dim_end + k (k = 0..hw) that is
// lo = dim_end - k - 1 and ONE fp32 fraction f_k per k (computed on the host like the reference does): the taps of every output read
// an EXTENDED line E[] whose entries beyond the ends are those lerps / mirrored values -- so edge tiles run the interior code on a
// patched window, and the z ends become a feed order (below).
//
// Shapes (r03).  Tiles are 32 x 32; a level whose width or height is not a multiple of 32 gets ONE shifted tile column / row: the last
// tile starts at nx - 32 (ny - 32) and overlaps its neighbour, which stores only the pieces in front of it (columns < nx - 32 rounded
// up to a whole 16-byte piece: the two or three columns both write carry identical values -- same taps, same operands, interior
// rule on both sides).  Every tile is a full tile: no masked lanes, no partial-tile code.  A shifted tile's global addresses are
// only dword aligned; LDS-DMA and 16-byte stores take that (scripts/microbench/glds_unaligned.hip).
//
// What is different (r02; the old kernel issued 230 lane-instructions per voxel at hw 8 against 102 of blur arithmetic, had 41-52 %
// LDS bank-conflict cycles and two barriers per plane):
//   * the z-blur is a SCATTER into 2*hw running sums instead of a gather from a ring of 2*hw+2 planes.  The reference adds the terms
//     of output p in the order src[p+hw], src[p+hw-1], ... src[p-hw] (d = -hw..+hw), so a workgroup that marches DOWN in z sees the
//     terms of every pending output in exactly that order: with v the xy-blurred value of the plane just processed,
//         out   = A[2hw-1] + tap[2hw]*v        (output p = plane + hw is complete)
//         A[s]  = A[s-1]   + tap[s]  *v        s = 2hw-1 .. 1
//         A[0]  = 0        + tap[0]  *v
//     -- the shift of the pending sums is the destination register of the add: no ring, no moves, static register indices.
//   * z boundaries are a FEED ORDER, not a slow path: the chunk that ends at the top of the volume first walks up the planes
//     dim_end-hw-1 .. dim_end, feeding the lerps E[dim_end+hw] .. E[dim_end] of consecutive xy-blurred planes, then down through the
//     real planes; the chunk that ends at plane 0 goes on with planes 1 .. hw again (E[-k] = plane k).
//   * the tile of the next plane travels global -> LDS by LDS-DMA (global_load_lds_dwordx4) into the other half of a double buffer;
//     the x-blurred tile is double buffered too and the y/z work runs one plane behind the x-blur, so a plane costs ONE barrier.
//   * LDS layouts are conflict-free for the 16-byte accesses of gfx950 (ds_read_b128 is served in the lane groups {0-3,12-15,20-27},
//     {4-11,16-19,28-31}, +32; 64 banks): tile row pitch = W4 pieces with piece c of row r stored at piece c ^ (r & 1), x-blur work items
//     dealt to lanes so that the four rows of a lane group differ in (r & 1) and in bit 3 of r * W4; x-blurred tile pitch 32 floats.
//   * global stores and DMA loads take an SGPR base + 32-bit VGPR offset (no 64-bit vector address arithmetic in the loop).
#include <string.h>
#include <stdlib.h>
#include <algorithm>

#include "sift3d_internal.h"

namespace s3d {

constexpr int kMarchMaxHW = 8;
struct MTaps { float w[2 * kMarchMaxHW + 1]; };
struct MEdge { float f[3][kMarchMaxHW + 1]; };  // x, y, z fractions of the right-boundary rule
typedef float mf4 __attribute__((ext_vector_type(4)));
typedef float mf2 __attribute__((ext_vector_type(2)));

// TX_ = 32: the r02 geometry (32 x 32 tiles, four waves).  TX_ = 64 (r04): 64 x 32 tiles, eight waves, for the big levels: the level
// time is set by the memory skeleton of the march (scripts/microbench/tile_march.hip: tile DMA + stores + one barrier per plane, no
// arithmetic, runs within 10 % of the kernel), and that skeleton is 9-15 % faster with 64-wide tiles (x halo 1.25x instead of 1.5x,
// row segments of 288-320 bytes instead of 160-192).
template <int HW, int TX_ = 32>
struct MCfg {
	static constexpr int TX = TX_, TY = 32, NT = TX * 8, NW = NT / 64;
	static constexpr int SEG = TX / 8;                     // x-blur items (8 outputs each) per tile row
	static constexpr int RPW = 64 / SEG;                   // tile rows one wave of x-blur items covers
	static constexpr int HX = ((HW + 3) / 4) * 4;          // x halo per side (floats, whole 16-byte pieces)
	static constexpr int W = TX + 2 * HX, W4 = W / 4;      // tile row: 12 pieces (hw >= 5) or 10 (hw <= 4); 20 / 18 at TX 64
	static constexpr int ROWS = TY + 2 * HW;               // row r <-> y = y0 - HW + r
	static constexpr int NITEMS = ROWS * W4;               // 16-byte pieces per plane
	static constexpr int NWI = (NITEMS + 63) / 64;         // DMA wave-instructions per plane (1 KiB each)
	static constexpr int NDMA = (NWI + NW - 1) / NW;       // ... per wave (wave w takes instructions w, w+NW, ...)
	static constexpr int TILE_F = NWI * 256;               // floats per tile buffer
	static constexpr int XP = TX, XB_F = ROWS * XP;        // x-blurred tile
	static constexpr bool XSW = TX == 64;                  // x-blurred tile: piece p of row r stored at p ^ (r & 1) (a row is a whole bank sweep)
	static constexpr int WOFF = HX - HW;                   // window index of input x-hw of output 0 (the window starts at piece 2*seg)
	static constexpr int WN4 = (WOFF + 8 + 2 * HW + 3) / 4;
	static constexpr int CRN = HW + 1;                     // DoG centre ring: plane p is needed HW steps after it was loaded
	static constexpr int NR = (ROWS + RPW - 1) / RPW;      // waves' worth of x-blur items; role NR serves the bottom tiles
	static_assert(W4 % 2 == 0 && 2 * (SEG - 1) + WN4 <= W4, "tile geometry");
	static_assert(NR < NW, "one wave without x-blur items serves the bottom tiles");
	static_assert((HW + 2) * SEG <= 64, "the bottom rows are one wave's worth of items");
	static_assert(RPW % 2 == 0, "the swizzle bit of a lane's row does not depend on its role");
};

// development diagnostics, timing only (never set in the product build; results are wrong by construction): 1 no tile DMA, 2 no stores,
// 4 no wait for the DMA, 8 no barriers, 16 / 32 / 64 no x / y / z blur arithmetic (the LDS traffic stays), 128 the y-blur reads ONE row,
// 256 the x-blur reads ONE window piece, 512 no centre-ring traffic
#ifndef S3D_MDIAG
#define S3D_MDIAG 0
#endif
__device__ __forceinline__ void m_barrier() {
	if (S3D_MDIAG & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <int N>
__device__ __forceinline__ void m_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// The output stores are non-temporal (streamed: they should not displace the tile rows neighbouring workgroups re-read from the L2;
// 2.35 -> 2.32 ms at 512^3); the tile DMA loads are plain (nt LOADS: 3.48 ms -- the L2 does absorb most of the halo re-reads).
// LDS-DMA: lane l's 16 bytes at base + voff land at lds_dst + 16*l.  M0 carries the wave-uniform LDS byte address (restored).
__device__ __forceinline__ void m_dma16(const float *base, unsigned voff, unsigned lds_dst) {
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
// untracked store (hipcc would make later loads that re-use the data registers wait for the store to COMPLETE); nothing reads
// dst / dog back in this kernel.  The wait states cover the hardware's read of the four data registers.
__device__ __forceinline__ void m_store16(float *base, unsigned voff, mf4 d) {
	asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 2" ::"v"(voff), "v"(d), "s"(base));
}

__device__ __forceinline__ float m_absmax(float m, float v) {
	const float a = fabsf(v);
	return (a > m) ? a : m;
}

// Workgroups per CU a 32 x 32 launch is planned for (residency rounds): three, for every half width.  The kernels without the DoG centre
// ring get the register budget of FOUR (128): the fourth slot is where the small octaves' workgroups run beside a big launch.
// (Measured and closed, r03 / r04: four to six workgroups per CU for the light levels, 64 x 32 tiles at three per CU: no gain.)
constexpr int kMarchOcc = 3;
template <int HW, bool CR>
constexpr int march_occ() { return kMarchOcc; }
template <int HW, bool CR>
constexpr int march_lb() { return CR ? kMarchOcc : 4; }  // __launch_bounds__ second argument
constexpr int kMarchYG = 4;          // y-blur rows requested per group (6 spills at hw 8 under the 128-register budget)
constexpr int kMarchCrMaxHW = 5;     // DoG centre ring in LDS up to this half width (three workgroups per CU still fit); wider levels re-read the centre plane
constexpr int kMarchBgRinglessTiles = 16;  // launches planned beside another octave with at most this many tiles per plane take the ring-less form (LDS fit)
constexpr int kMarchWideMinTiles = 100;    // 64 x 32 tiles per plane from which a level takes them (512 x 512: 128; 256 x 256 would march 16 chunks of 16 planes + ramp)

// KR (TX 64): the newest KR planes of the DoG centre ring live in REGISTERS and move on to an LDS ring of HW + 1 - KR planes (two
// workgroups of eight waves per CU leave 80 KB each: tile + x-blurred tile + a whole ring of 64 x 32 planes do not fit at hw >= 4)
// (the second __launch_bounds__ argument is waves per SIMD: two workgroups of eight waves = 4)
template <int HW, bool DOG, bool CR, int TX = 32, int KR = 0>
__global__ void __launch_bounds__(TX * 8, (TX == 64 ? 4 : march_lb<HW, CR>())) k_march_level(const float *__restrict__ src, float *__restrict__ dst, float *__restrict__ dog,
                                                        unsigned *__restrict__ dogmax, int nx, int ny, ZRange zr, MTaps t, MEdge ef, int ntx,
                                                        int nty, int cz, int prio, float *__restrict__ half, int hnx, int hny, int hnz) {
	using C = MCfg<HW, TX>;
	constexpr bool ZSYM = HW <= 6;  // (r04: the z-scatter forms each product tap * v once for the two pending sums that take it: symmetric taps, bit-identical)
	constexpr int KL = C::CRN - KR;  // planes of the centre ring in LDS
	static_assert(KR == 0 || (CR && KL >= 1), "register part of the centre ring");
	__shared__ __attribute__((aligned(1024))) float tile[2 * C::TILE_F];
	__shared__ __attribute__((aligned(16))) float xb[2 * C::XB_F];
	__shared__ __attribute__((aligned(16))) mf4 cring[CR ? KL * C::NT : 1];
	// DOG without the centre ring (hw >= 6): the centre piece of the plane the NEXT step completes travels by LDS-DMA as well (a
	// compiler-tracked global load would be waited for with vmcnt(0), i.e. together with the tile DMA issued right after it)
	__shared__ __attribute__((aligned(1024))) mf4 cenb[(DOG && !CR) ? 2 * C::NT : 1];
	__shared__ float s_ef[3 * (kMarchMaxHW + 1)];
	__shared__ float s_red[C::NW];

	// wave priority of the launch (issue arbitration between resident waves): the small octaves' launches sit on the critical chain
	// octave -> octave while the big launches of octave 0 fill the machine; they run at a raised priority
	if (prio >= 2) __builtin_amdgcn_s_setprio(3);
	else if (prio == 1) __builtin_amdgcn_s_setprio(1);
	const int tid = threadIdx.x, lane = tid & 63;
	const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
	if (tid < 3 * (kMarchMaxHW + 1)) s_ef[tid] = ef.f[tid / (kMarchMaxHW + 1)][tid % (kMarchMaxHW + 1)];

	// ---- XCD-aware, bijective block -> (chunk, tile): blocks b, b+8, ... share an XCD and get neighbouring tiles ----
	int lb;
	{
		const int nblocks = gridDim.x, b = blockIdx.x, xcd = b & 7, idx = b >> 3;
		const int per = nblocks >> 3, rem = nblocks & 7;
		lb = xcd * per + min(xcd, rem) + idx;
	}
	const int tile_x = lb % ntx, tile_y = (lb / ntx) % nty, chunk = lb / (ntx * nty);
	// the last tile column / row is shifted back to end on the level's edge (see "Shapes" in the header); the tile in front of it keeps
	// the pieces / rows below that start
	const int x0 = min(tile_x * C::TX, nx - C::TX), y0 = min(tile_y * C::TY, ny - C::TY);
	const int xlim = tile_x + 1 < ntx ? min((tile_x + 1) * C::TX, (nx - C::TX + 3) & ~3) : nx;
	const int ylim = tile_y + 1 < nty ? min((tile_y + 1) * C::TY, ny - C::TY) : ny;
	const int nz = zr.nz, zoff = zr.zoff, dim_end = zr.nzg - 1;
	const int zc0 = zr.zo0 + chunk * cz, zc1 = min(zr.zo1, zc0 + cz);
	const int sy = nx, sz = nx * ny;
	const bool left_f = x0 == 0, right_f = x0 + C::TX == nx, top_f = y0 == 0, bottom_f = y0 + C::TY == ny;

	// ---- DMA items of this lane: LDS piece i of the tile buffer holds (row r = i / W4, piece (i % W4) ^ (r & 1)) ----
	unsigned goffb[C::NDMA];
#pragma unroll
	for (int i = 0; i < C::NDMA; i++) {
		const int item = (wid + C::NW * i) * 64 + lane;
		const int r = item / C::W4, cs = item - r * C::W4, c = cs ^ (r & 1);
		const int gy = y0 - HW + r, gx = x0 - C::HX + 4 * c;
		const bool ok = item < C::NITEMS && gy >= 0 && gy < ny && gx >= 0 && gx + 3 < nx;
		// pieces outside the volume: any valid address (their LDS image is replaced in registers by the edge rules or never read)
		goffb[i] = (unsigned)(ok ? gy * sy + gx : y0 * sy + x0) * 4u;
	}
	const unsigned lds_tile = (unsigned)(unsigned long long)tile + (unsigned)wid * 1024u;
	const unsigned lds_cen = (unsigned)(unsigned long long)cenb + (unsigned)wid * 1024u;

	// ---- x-blur item of this lane: 8 outputs of one tile row; rows dealt so that every ds_read_b128 lane group is conflict-free ----
	// The x-blur holds 3 waves' worth of items (ROWS <= 48 rows x 4 segments); the wave without items is the one with role 3, and the
	// roles rotate with the step (role = (wave + step) & 3): co-resident workgroups place their waves on the SIMDs in a fixed pattern
	// (hwid_probe: three workgroups put their fourth waves on three different SIMDs, the remaining SIMD carries three item waves), so
	// a fixed role assignment leaves one SIMD with 13 % more vector work than the average.
	int rs, xseg_r;
	{
		const int l5 = lane & 31;
		int g, jj;  // lane group (of 16) and index inside it, in the hardware's service order
		if (l5 < 4) { g = 0; jj = l5; }
		else if (l5 < 12) { g = 1; jj = l5 - 4; }
		else if (l5 < 16) { g = 0; jj = l5 - 8; }
		else if (l5 < 20) { g = 1; jj = l5 - 8; }
		else if (l5 < 28) { g = 0; jj = l5 - 12; }
		else { g = 1; jj = l5 - 16; }
		g += (lane >> 5) * 2;
		const int q = jj >> 2;
		if (TX == 64) {
			// a service group = two rows of different parity x eight segments: piece 2 seg + k of the even row and (2 seg + k) ^ 1 of the odd
			// row cover different banks for every k (row pitch 18 or 20 pieces = 8 or 16 banks mod 64)
			rs = 2 * g + (jj >> 3);
			xseg_r = jj & 7;
		} else {
			// W4 = 12: four consecutive rows (r*12 mod 16 = 0,12,8,4; odd rows swizzled);  W4 = 10: rows {r, r+4, r+1, r+5}
			rs = C::W4 == 12 ? 4 * g + q : ((g & 1) * 2 + (g >> 1) * 8 + (q & 1) * 4 + (q >> 1));
			xseg_r = jj & 3;
		}
	}
	// per-role activity / mirror bits of the regular items (role r: tile row 16*r + rs)
	int amask = 0, mmask = 0;
#pragma unroll
	for (int r = 0; r < C::NR; r++) {
		const int row = C::RPW * r + rs, gy = y0 - HW + row;
		const bool act = row < C::ROWS && gy >= 0 && gy < ny && !(bottom_f && gy == ny - 1);  // row yend of a bottom tile receives E[yend]
		amask |= (act ? 1 : 0) << r;
		mmask |= ((act && top_f && gy >= 1 && gy <= HW) ? 1 : 0) << r;  // E[-k] = row k: second copy in the mirror row
	}
	const int sw_r = rs & 1;  // RPW*role is even: the swizzle bit does not depend on the role
	const int xbe_r = (rs * C::W4 + 2 * xseg_r + sw_r) * 4, xbo_r = (rs * C::W4 + 2 * xseg_r - sw_r) * 4;  // float offsets of even / odd window pieces
	// (XSW: the two pieces of an item trade places in odd rows; the mirror row 2 HW - row has the row's parity)
	const int xout_r = rs * C::XP + xseg_r * 8 + ((C::XSW && sw_r) ? 4 : 0);
	const int xhi_r = (C::XSW && sw_r) ? -4 : 4;  // float offset of the second piece of an item
	// bottom tiles: the wave with role 3 re-blurs rows yend-hw-1 .. yend (one row per 4 lanes) for the bottom extension E[yend+k]
	const int row_h = C::TY - 2 + lane / C::SEG, xseg_h = lane % C::SEG, sw_h = row_h & 1;
	const bool act_h = bottom_f && lane < (HW + 2) * C::SEG;
	const int xbe_h = (row_h * C::W4 + 2 * xseg_h + sw_h) * 4, xbo_h = (row_h * C::W4 + 2 * xseg_h - sw_h) * 4;

	// ---- y/z work: thread (xq, ty) owns the 16-byte piece x = x0 + 4*xq .. +3 of row y0 + ty ----
	const int xq = tid & (C::TX / 4 - 1), ty = tid / (C::TX / 4);
	// column offsets of this thread's piece in the x-blurred rows of ty's parity / the other parity (equal without the swizzle)
	const int ycol = ty * C::XP + 4 * (C::XSW ? (xq ^ (ty & 1)) : xq), ycol_o = ty * C::XP + 4 * (C::XSW ? (xq ^ (ty & 1) ^ 1) : xq);
	const unsigned out_voff = (unsigned)((y0 + ty) * sy + x0 + 4 * xq) * 4u;
	const bool own = x0 + 4 * xq < xlim && y0 + ty < ylim;  // this thread's piece belongs to this tile (always, unless the next tile is shifted)
	const int park_off = ((ty + HW) * C::W4 + ((C::HX / 4 + xq) ^ ((ty + HW) & 1))) * 4;  // raw centre piece in the tile

	float A[2 * HW][4];
#pragma unroll
	for (int s = 0; s < 2 * HW; s++)
#pragma unroll
		for (int c = 0; c < 4; c++) A[s][c] = 0.0f;
	float prevx[4] = {0.f, 0.f, 0.f, 0.f};
	float mx = 0.0f;
	mf4 creg[KR > 0 ? KR : 1];
#pragma unroll
	for (int i = 0; i < (KR > 0 ? KR : 1); i++) creg[i] = mf4{0.f, 0.f, 0.f, 0.f};

	// feed order (global E index e, descending): e_top = highest term of the chunk's top output
	const int e_top = zc1 - 1 + zoff + HW, e_bot = zc0 + zoff - HW;
	const int e_start = e_top + (e_top >= dim_end ? 1 : 0);  // +1: priming step (loads plane dim_end-hw-1, feeds nothing that is kept)
	const int nsteps = e_start - e_bot + 1;
	auto plane_ptr = [&](int e) {
		const int L = e < 0 ? -e : (e > dim_end ? 2 * dim_end - e : e);
		return src + (size_t)sz * (size_t)min(max(L - zoff, 0), nz - 1);
	};
	auto issue_dma = [&](int jn) {
		const float *pl = plane_ptr(e_start - jn);
		const unsigned dstb = lds_tile + (unsigned)(jn & 1) * (unsigned)(C::TILE_F * 4);
#pragma unroll
		for (int i = 0; i < C::NDMA; i++)
			if (!(S3D_MDIAG & 1) && wid + C::NW * i < C::NWI) m_dma16(pl, goffb[i], dstb + (unsigned)(i * C::NW * 1024));
	};
	if (zc0 >= zc1) return;  // uniform (never for a planned grid)
	issue_dma(0);
	m_wait_vmcnt<0>();
	__syncthreads();  // also publishes s_ef

#pragma unroll 1
	for (int j = 0; j <= nsteps; j++) {
		const int buf = j & 1;
		// DoG centre values of the output this step completes, without a centre ring: requested before anything else of the step
		const int e_out = e_start - (j - 1);            // feed consumed by the y/z stage of this step
		const int p_loc = e_out + HW - zoff;            // output plane it completes (local)
		const bool emit = j >= 1 && p_loc >= zc0 && p_loc < zc1 && own;
		mf4 cen = {0.f, 0.f, 0.f, 0.f};
		if (DOG && !CR) {
			if (j >= 1) cen = cenb[(j & 1) * C::NT + tid];  // landed before the previous step's barrier (own wave's DMA, own lanes)
			const int pn = min(max(p_loc - 1, 0), nz - 1);  // the output plane of step j + 1
			if (!(S3D_MDIAG & 1)) m_dma16(src + (size_t)sz * (size_t)pn, out_voff, lds_cen + (unsigned)(((j + 1) & 1) * C::NT * 16));
		}
		if (j + 1 < nsteps) issue_dma(j + 1);

		constexpr int KG = kMarchYG, NTAP = 2 * HW + 1;
		constexpr bool YPRE = HW <= 6;  // the first group of y-blur rows is requested in front of the x-blur (their latency hides behind its arithmetic; hw 8 has no registers left)
		mf4 ypre[KG];
		if (YPRE) {  // (step 0 reads rows nobody has written: never used)
			const float *yc0 = xb + (buf ^ 1) * C::XB_F;
#pragma unroll
			for (int i = 0; i < KG; i++) if (!((S3D_MDIAG & 128) && i > 0)) ypre[i] = *reinterpret_cast<const mf4 *>(yc0 + ((i < NTAP && (i & 1)) ? ycol_o : ycol) + (i < NTAP ? 2 * HW - i : 0) * C::XP);
			__builtin_amdgcn_sched_barrier(0);
		}
		// ---------------- x-blur of feed j: tile[buf] -> xb[buf] ----------------
		if (j < nsteps) {
			const int role = (wid + j) & (C::NW - 1);  // wave-uniform: the x-blur roles rotate over the waves with the step
			const bool xhelp = role == C::NR;
			const bool xact = xhelp ? act_h : (((amask >> role) & 1) != 0);
			const bool xmirror = !xhelp && (((mmask >> role) & 1) != 0);
			const int xseg = xhelp ? xseg_h : xseg_r;
			const int xbase_e = xhelp ? xbe_h : xbe_r + role * (C::RPW * C::W4 * 4), xbase_o = xhelp ? xbo_h : xbo_r + role * (C::RPW * C::W4 * 4);
			const int xout = xout_r + role * (C::RPW * C::XP);
			// the mirror row of tile row `row` of a top tile (y0 = 0: gy = row - HW) is row 2 HW - row, same column: formed from xout where it is
			// stored (r05: as a lane constant of its own it was the one register too many of the hw-5 / hw-6 wide kernels -- spilled, and
			// reloaded inside the plane loop behind an s_waitcnt vmcnt(0) that also drained the tile DMA)
			static_assert((C::XP & (C::XP - 1)) == 0 && (C::SEG - 1) * 8 + 4 < C::XP, "row index = offset / XP");
			const int xout_m = xout + 2 * C::XP * HW - 2 * (xout & ~(C::XP - 1));
			float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
			if (xact) {
				const float *tb = tile + buf * C::TILE_F;
				float win[C::WN4 * 4];
#pragma unroll
				for (int k = 0; k < C::WN4; k++) {
					if ((S3D_MDIAG & 256) && k > 0) { win[4 * k] = win[0]; win[4 * k + 1] = win[1]; win[4 * k + 2] = win[2]; win[4 * k + 3] = win[3]; continue; }
					const mf4 f = *reinterpret_cast<const mf4 *>(tb + ((k & 1) ? xbase_o : xbase_e) + 4 * k);
					win[4 * k] = f.x; win[4 * k + 1] = f.y; win[4 * k + 2] = f.z; win[4 * k + 3] = f.w;
				}
				// window index of tile column c (0 = x0) for segment s: WOFF + HW + c - 8*s
				if (left_f && xseg == 0) {
#pragma unroll
					for (int k = 1; k <= HW; k++) win[C::WOFF + HW - k] = win[C::WOFF + HW + k];
				}
				if (right_f && xseg == C::SEG - 1) {
					float e[HW + 1];
#pragma unroll
					for (int k = 0; k <= HW; k++) {
						const float f = s_ef[k];
						e[k] = (1.0f - f) * win[C::WOFF + HW + 6 - k] + f * win[C::WOFF + HW + 7 - k];
					}
#pragma unroll
					for (int k = 0; k <= HW; k++) win[C::WOFF + HW + 7 + k] = e[k];
				}
				if (HW == 8 && right_f && xseg == C::SEG - 2) {  // the window of segment 2 ends on column xend = E[xend]
					const float f = s_ef[0];
					win[C::WOFF + HW + 15] = (1.0f - f) * win[C::WOFF + HW + 14] + f * win[C::WOFF + HW + 15];
				}
				// tap-major: the eight outputs are eight independent chains (one dependent chain issues at half the rate)
#pragma unroll
				for (int d = -HW; d <= HW; d++) {
					if ((S3D_MDIAG & 16) && d != 0) continue;
#pragma unroll
					for (int jo = 0; jo < 8; jo++) o[jo] = o[jo] + t.w[d + HW] * win[C::WOFF + jo + HW - d];
				}
				if (!xhelp) {
					float *xo = xb + buf * C::XB_F;
					*reinterpret_cast<mf4 *>(xo + xout) = mf4{o[0], o[1], o[2], o[3]};
					*reinterpret_cast<mf4 *>(xo + xout + xhi_r) = mf4{o[4], o[5], o[6], o[7]};
					if (xmirror) {
						*reinterpret_cast<mf4 *>(xo + xout_m) = mf4{o[0], o[1], o[2], o[3]};
						*reinterpret_cast<mf4 *>(xo + xout_m + xhi_r) = mf4{o[4], o[5], o[6], o[7]};
					}
				}
			}
			if (bottom_f && xhelp) {  // wave-uniform: every lane takes part in the shuffles
				float a[8];
#pragma unroll
				for (int jo = 0; jo < 8; jo++) a[jo] = __shfl_up(o[jo], C::SEG, 64);  // the row below (same segment)
				const int jr = lane / C::SEG;                                   // this lane holds x-blurred row yend-hw-1+jr
				if (xact && jr >= 1) {
					const int k = HW + 1 - jr;                                  // E[yend+k] = (1-f_k) xb[yend-k-1] + f_k xb[yend-k]
					const float f = s_ef[kMarchMaxHW + 1 + k];
					const int erow = C::TY - 1 + HW + k;
					const bool esw = C::XSW && (erow & 1);
					float *xe = xb + buf * C::XB_F + erow * C::XP + xseg * 8 + (esw ? 4 : 0);
					*reinterpret_cast<mf4 *>(xe) = mf4{(1.0f - f) * a[0] + f * o[0], (1.0f - f) * a[1] + f * o[1], (1.0f - f) * a[2] + f * o[2],
					                                   (1.0f - f) * a[3] + f * o[3]};
					*reinterpret_cast<mf4 *>(xe + (esw ? -4 : 4)) = mf4{(1.0f - f) * a[4] + f * o[4], (1.0f - f) * a[5] + f * o[5], (1.0f - f) * a[6] + f * o[6],
					                                       (1.0f - f) * a[7] + f * o[7]};
				}
			}
		}

		// ---------------- y-blur, feed and z-scatter of feed j-1: xb[buf ^ 1] ----------------
		int nst = 0;
		if (j >= 1) {
			const float *yc = xb + (buf ^ 1) * C::XB_F;
			float v[4] = {0.f, 0.f, 0.f, 0.f};
			{
				// step s = d + HW reads row ty + 2*HW - s; rows are requested a group ahead of their use (hipcc keeps only two reads
				// in flight on its own and exposes the LDS latency nine times per plane)
				mf4 cur[KG], nxt[KG];
#pragma unroll
				for (int i = 0; i < KG; i++) cur[i] = YPRE ? ypre[i] : *reinterpret_cast<const mf4 *>(yc + ((i < NTAP && (i & 1)) ? ycol_o : ycol) + (i < NTAP ? 2 * HW - i : 0) * C::XP);
#pragma unroll
				for (int g0 = 0; g0 < NTAP; g0 += KG) {
#pragma unroll
					for (int i = 0; i < KG; i++)
						if (g0 + KG + i < NTAP && !(S3D_MDIAG & 128)) nxt[i] = *reinterpret_cast<const mf4 *>(yc + (((g0 + KG + i) & 1) ? ycol_o : ycol) + (2 * HW - (g0 + KG + i)) * C::XP);
#pragma unroll
					for (int i = 0; i < KG; i++)
						if (g0 + i < NTAP && !((S3D_MDIAG & 32) && (g0 + i) % 4 != 0)) {
							const float tap = t.w[g0 + i];
							v[0] = v[0] + tap * cur[i].x; v[1] = v[1] + tap * cur[i].y; v[2] = v[2] + tap * cur[i].z; v[3] = v[3] + tap * cur[i].w;
						}
#pragma unroll
					for (int i = 0; i < KG; i++) cur[i] = nxt[i];
				}
			}
			if (e_out >= dim_end) {  // wave-uniform, top chunk only: E[dim_end+k] = (1-f_k) X[dim_end-k-1] + f_k X[dim_end-k]
				const float f = s_ef[2 * (kMarchMaxHW + 1) + min(e_out - dim_end, HW)];
#pragma unroll
				for (int c = 0; c < 4; c++) {
					const float xn = v[c];
					v[c] = (1.0f - f) * prevx[c] + f * xn;
					prevx[c] = xn;
				}
			}
			float out[4];
			if (ZSYM) {
				// r04: the taps are symmetric bit for bit (checked on the host: t.w[s] == t.w[2 HW - s]), so the product of v with tap s is
				// also the product with tap 2 HW - s: HW + 1 multiplies per value instead of 2 HW + 1, the adds and their order unchanged
				float m[HW + 1][4];
#pragma unroll
				for (int k = 0; k <= HW; k++)
#pragma unroll
					for (int c = 0; c < 4; c++) m[k][c] = t.w[k] * v[c];
#pragma unroll
				for (int c = 0; c < 4; c++) out[c] = A[2 * HW - 1][c] + m[0][c];
#pragma unroll
				for (int s = 2 * HW - 1; s >= 1; s--) {
#pragma unroll
					for (int c = 0; c < 4; c++) A[s][c] = A[s - 1][c] + m[s <= HW ? s : 2 * HW - s][c];
				}
#pragma unroll
				for (int c = 0; c < 4; c++) A[0][c] = 0.0f + m[0][c];
			} else {
#pragma unroll
			for (int c = 0; c < 4; c++) out[c] = A[2 * HW - 1][c] + t.w[2 * HW] * v[c];
#pragma unroll
			for (int s = 2 * HW - 1; s >= 1; s--) {
				if ((S3D_MDIAG & 64) && s != HW) continue;
#pragma unroll
				for (int c = 0; c < 4; c++) A[s][c] = A[s - 1][c] + t.w[s] * v[c];
			}
#pragma unroll
			for (int c = 0; c < 4; c++) A[0][c] = 0.0f + t.w[0] * v[c];
			}

			if (DOG && CR && !(S3D_MDIAG & 512)) cen = cring[(j % KL) * C::NT + tid];  // the centre piece of HW+1 steps ago (read before this step's park below)
			// stores this WAVE issues in this step (the counted wait below): a store instruction is issued when any lane of the wave
			// owns its piece -- in the tile in front of a shifted one some lanes, or whole waves (rows), do not
			nst = __any(emit && !((S3D_MDIAG & 2) && out[0] != 12345.678f)) ? (DOG ? 2 : 1) : 0;
			// r03: the seed level of the next octave -- DownSample_3D (Src/cSIFT3D.cc:506-533): every second voxel of every second row of
			// every second plane -- leaves with the level's own store (`half` = level 0 of the next octave, whole volumes with nx % 4 ==
			// 0 only): the decimation launch sat on the stage's critical chain octave -> octave and read the level again
			if (half != nullptr && (p_loc & 1) == 0) {  // wave-uniform
				const int hy = (y0 + ty) >> 1, hx = (x0 + 4 * xq) >> 1, hz = p_loc >> 1;
				const bool hs = emit && ((y0 + ty) & 1) == 0 && hy < hny && hx < hnx && hz < hnz;
				if (__any(hs)) {
					nst += 1;
					if (hs) {
						const unsigned hoff = (unsigned)((hz * hny + hy) * hnx + hx) * 4u;
						if (hx + 1 < hnx) asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" ::"v"(hoff), "v"(mf2{out[0], out[2]}), "s"(half));
						else asm volatile("global_store_dword %0, %1, %2\n\ts_nop 1" ::"v"(hoff), "v"(out[0]), "s"(half));
					}
				}
			}
			if (emit && !((S3D_MDIAG & 2) && out[0] != 12345.678f)) {
				float *gb = dst + (size_t)sz * (size_t)p_loc;
				m_store16(gb, out_voff, mf4{out[0], out[1], out[2], out[3]});
				if (DOG) {
					float dg[4];
					dg[0] = (out[0] - cen.x) * (-1.0f); dg[1] = (out[1] - cen.y) * (-1.0f);
					dg[2] = (out[2] - cen.z) * (-1.0f); dg[3] = (out[3] - cen.w) * (-1.0f);
					m_store16(dog + (size_t)sz * (size_t)p_loc, out_voff, mf4{dg[0], dg[1], dg[2], dg[3]});
#pragma unroll
					for (int c = 0; c < 4; c++) mx = m_absmax(mx, dg[c]);
				}
			}
		}
		if (DOG && CR && j < nsteps && !(S3D_MDIAG & 512)) {
			const mf4 fresh = *reinterpret_cast<const mf4 *>(tile + buf * C::TILE_F + park_off);
			if (KR == 0) cring[(j % KL) * C::NT + tid] = fresh;
			else {  // the piece that has been KR steps in registers moves to the LDS ring; the ring slot was read above
				cring[(j % KL) * C::NT + tid] = creg[KR - 1];
#pragma unroll
				for (int i = KR - 1; i >= 1; i--) creg[i] = creg[i - 1];
				creg[0] = fresh;
			}
		}

		// the DMA of the next tile was issued before this step's stores (vmcnt retires in order)
		if (S3D_MDIAG & 4) {}
		else if (nst == 3) m_wait_vmcnt<3>();
		else if (nst == 2) m_wait_vmcnt<2>();
		else if (nst == 1) m_wait_vmcnt<1>();
		else m_wait_vmcnt<0>();
		m_barrier();
	}

	if (DOG) {
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
		if (lane == 0) s_red[wid] = mx;
		__syncthreads();
		if (tid == 0) {
			float r = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
			if (C::NW == 8) r = fmaxf(r, fmaxf(fmaxf(s_red[4], s_red[5]), fmaxf(s_red[6], s_red[7])));
			if (r > 0.0f) atomicMax(dogmax, __float_as_uint(r));
		}
	}
}

static void march_edge_fractions(int n, int hw, float *f) {  // Src/cSIFT3D.cc:751-760, see "Boundaries" in the header
	const int dim_end = n - 1;
	for (int k = 0; k <= hw; k++) {
		const float c = (float)(dim_end + k);
		const float cc = (float)(2 * dim_end) - c - 0.1f;
		const int lo = (int)cc;
		f[k] = cc - (float)lo;
	}
}

template <int HW>
static void launch_march_hw(const float *src, float *dst, float *dog, unsigned *dogmax, int nx, int ny, const ZRange &zr, const Taps &tg,
                            hipStream_t st, int plan_slots, int prio, const MarchHalf &hf) {
	using C = MCfg<HW>;
	MTaps t;
	for (int i = 0; i < 2 * kMarchMaxHW + 1; i++) t.w[i] = i < 2 * HW + 1 ? tg.w[i] : 0.0f;
	const int nzo = zr.zo1 - zr.zo0;
	if (nzo <= 0) return;
	const int ntx = (nx + C::TX - 1) / C::TX, nty = (ny + C::TY - 1) / C::TY, ntiles = ntx * nty;
	// z chunking.  A CU retires workgroup-planes at about the same rate with three or four resident workgroups (measured: the fourth
	// buys nothing at 512^3), so a residency round of R workgroups costs (planes marched) x max(1, R / 768); what the fourth slot does
	// buy is ONE round instead of two when a level has more tiles than 768 (1024^2 levels: 1024 tiles -- the z-slabs of configs[3]).
	// Kernels with the DoG centre ring in LDS fit three per CU; where a fourth slot saves a round the ring-less instantiation runs.
	constexpr bool kHasCR = HW <= kMarchCrMaxHW;
	const int ramp = 2 * HW + 1;
	const int sat = (dog && kHasCR) ? 3 : march_occ<HW, false>();  // workgroups per CU at which a CU's plane rate saturates
	auto plan = [&](int cap, int &cz_out) {
		if (plan_slots > 0) cap = std::min(cap, plan_slots);
		double best = 1e300;
		cz_out = nzo;
		for (int n = 1; n <= nzo && n <= 64; n++) {
			const int czn = (nzo + n - 1) / n, nch = (nzo + czn - 1) / czn;
			long left = (long)ntiles * nch;
			double cost = 0.0;
			while (left > 0) {
				const long r = std::min<long>(left, cap);
				cost += (double)(czn + ramp) * std::max(1.0, (double)r / (256.0 * sat));
				left -= r;
			}
			if (cost < best - 1e-9) { best = cost; cz_out = czn; }
		}
		return best;
	};
	// (kernels without the ring: planned for march_occ workgroups per CU; one more slot may be used where it saves a round)
	constexpr int kOcc = march_occ<HW, false>();
	int cz3 = nzo, cz4 = nzo;
	const double cost3 = plan(256 * (dog && kHasCR ? 3 : kOcc), cz3), cost4 = plan(256 * (dog && kHasCR ? 4 : kOcc + 1), cz4);
	// dog + ring: three per CU unless four without the ring are clearly ahead; everything else may use the extra slot
	// r04: the small octaves (at most 16 tiles per plane) run BESIDE octave 0's widest level (2 x 38 KB of LDS per CU) and octave 1's
	// (38 KB): a ring kernel's 45 / 52 KB does not fit next to them, and the head of octave 2 -- on the stage's critical chain -- waited
	// 130 us for octave 1's last level to drain (profiles/r04e_timeline.txt); the ring-less form (33 / 35 KB) fits, and re-reading the
	// centre plane of a 128 x 128 level costs nothing
	const bool small_bg = plan_slots > 0 && ntiles <= kMarchBgRinglessTiles;
	const bool use_cr = dog && kHasCR && !small_bg && !(cost4 * 1.12 < cost3);
	const int cz = (dog && kHasCR) ? (use_cr ? cz3 : cz4) : (cost4 < cost3 ? cz4 : cz3);
	const int nchunks = (nzo + cz - 1) / cz;
	MEdge ef;
	memset(&ef, 0, sizeof(ef));
	march_edge_fractions(nx, HW, ef.f[0]);
	march_edge_fractions(ny, HW, ef.f[1]);
	march_edge_fractions(zr.nzg, HW, ef.f[2]);
	const dim3 grid((unsigned)(ntiles * nchunks)), block(C::NT);
	// DoG centre ring in LDS where three workgroups per CU still fit (hw <= 5); otherwise the centre piece travels by LDS-DMA
	// (late r04, measured and closed: octave 0's widest level with a hybrid centre ring -- 3 planes in registers + 4 in LDS instead of
	// re-reading its centre plane, 0.54 GB of the pyramid's 8.9 GB of traffic -- bit-identical, stage 1.5 % slower on the bench volume)
	if (dog && use_cr) hipLaunchKernelGGL((k_march_level<HW, true, kHasCR>), grid, block, 0, st, src, dst, dog, dogmax, nx, ny, zr, t, ef, ntx, nty, cz, prio, hf.d, hf.nx, hf.ny, hf.nz);
	else if (dog) hipLaunchKernelGGL((k_march_level<HW, true, false>), grid, block, 0, st, src, dst, dog, dogmax, nx, ny, zr, t, ef, ntx, nty, cz, prio, hf.d, hf.nx, hf.ny, hf.nz);
	else hipLaunchKernelGGL((k_march_level<HW, false, false>), grid, block, 0, st, src, dst, dog, dogmax, nx, ny, zr, t, ef, ntx, nty, cz, prio, hf.d, hf.nx, hf.ny, hf.nz);
}

// r04: the 64 x 32 tiles of the big levels (MCfg<HW, 64>): eight waves, two workgroups per CU, planned for 512 slots.
static bool march_wide_ok(int nx, int ny, int nzo, int hw, int plan_slots) {
	const int mode = hook(SIFT3D_HOOK_MARCH_TILES);  // 0 product rule, 1 wherever the geometry allows (parity tests on small volumes), 2 never
	// (a planned launch shares the machine with another octave: 32 x 32 tiles, three per CU.  r06, measured again with the wide form for octave 0's
	// widest level beside the chain of the smaller octaves -- S3D_WIDE_TAIL=1 in a -DS3D_DEV_SWITCHES build: see docs/experiments.md)
	static const int wide_tail = dev_tune_i("S3D_WIDE_TAIL", 0);
	if (mode == 2 || (plan_slots > 0 && !(wide_tail && plan_slots >= 512)) || hw < 2 || hw > 6) return false;
	if (!(nx == 64 || nx >= 64 + ((hw + 3) / 4) * 4)) return false;  // the shifted last tile column starts at nx - 64: 0 or beyond the left halo (march_applicable)
	const int ntiles = ((nx + 63) / 64) * ((ny + 31) / 32);
	// (short columns: four chunks of a few planes + the ramp cost what three chunks of the 32 x 32 form do -- 512 x 512 x 32: 0.384 vs 0.376 ms)
	return mode == 1 || (ntiles >= kMarchWideMinTiles && nzo / std::max(1, 512 / ntiles) >= 24);  // planes per chunk
}
template <int HW>
static void launch_march_wide(const float *src, float *dst, float *dog, unsigned *dogmax, int nx, int ny, const ZRange &zr, const Taps &tg,
                              hipStream_t st, int prio, const MarchHalf &hf) {
	using C = MCfg<HW, 64>;
	MTaps t;
	for (int i = 0; i < 2 * kMarchMaxHW + 1; i++) t.w[i] = i < 2 * HW + 1 ? tg.w[i] : 0.0f;
	const int nzo = zr.zo1 - zr.zo0;
	if (nzo <= 0) return;
	const int ntx = (nx + C::TX - 1) / C::TX, nty = (ny + C::TY - 1) / C::TY, ntiles = ntx * nty;
	// z chunking as in launch_march_hw: residency rounds of at most 512 workgroups, a round costs its planes + the ramp
	const int ramp = 2 * HW + 1, cap = 512;
	int cz = nzo;
	{
		double best = 1e300;
		for (int n = 1; n <= nzo && n <= 64; n++) {
			const int czn = (nzo + n - 1) / n, nch = (nzo + czn - 1) / czn;
			long left = (long)ntiles * nch;
			double cost = 0.0;
			while (left > 0) {
				const long r = std::min<long>(left, cap);
				cost += (double)(czn + ramp) * std::max(1.0, (double)r / (double)cap);
				left -= r;
			}
			if (cost < best - 1e-9) { best = cost; cz = czn; }
		}
	}
	const int nchunks = (nzo + cz - 1) / cz;
	MEdge ef;
	memset(&ef, 0, sizeof(ef));
	march_edge_fractions(nx, HW, ef.f[0]);
	march_edge_fractions(ny, HW, ef.f[1]);
	march_edge_fractions(zr.nzg, HW, ef.f[2]);
	const dim3 grid((unsigned)(ntiles * nchunks)), block(C::NT);
	// DoG centres: a whole LDS ring up to hw 3, the newest 1 (hw 4) / 3 (hw 5) planes in registers, by one more LDS-DMA at hw 6
	constexpr int kKR = HW == 4 ? 1 : (HW == 5 ? 3 : 0);
	if (!dog) hipLaunchKernelGGL((k_march_level<HW, false, false, 64, 0>), grid, block, 0, st, src, dst, dog, dogmax, nx, ny, zr, t, ef, ntx, nty, cz, prio, hf.d, hf.nx, hf.ny, hf.nz);
	else if (HW <= 5) hipLaunchKernelGGL((k_march_level<HW, true, (HW <= 5), 64, kKR>), grid, block, 0, st, src, dst, dog, dogmax, nx, ny, zr, t, ef, ntx, nty, cz, prio, hf.d, hf.nx, hf.ny, hf.nz);
	else hipLaunchKernelGGL((k_march_level<HW, true, false, 64, 0>), grid, block, 0, st, src, dst, dog, dogmax, nx, ny, zr, t, ef, ntx, nty, cz, prio, hf.d, hf.nx, hf.ny, hf.nz);
}

// false => not applicable (a plane smaller than one tile, a shifted last tile that would reach into the mirrored left / top zone, a level
// too thin for the extended-line boundary form, a half width without an instantiation): the caller takes the generic separable
// kernels of kernels_pyramid.hip
bool march_half_ok(int nx, int ny, const ZRange &zr) { return (nx & 3) == 0 && zr.zoff == 0 && zr.nz == zr.nzg; }  // whole volumes, pieces on even x

// would launch_march_level take a level of nx x ny planes, nzg of them, with this kernel?
bool march_applicable(int nx, int ny, int nzg, const Taps &t) {
	// the shifted last tile starts at n - 32: 0, or beyond the mirror zone [0, hw) -- and along x beyond the whole left halo of HX =
	// 4 or 8 columns: the tile travels in 16-byte pieces from x0 - HX, and a piece that straddles column 0 is not loaded at all (late
	// r04, found by test_wide_tiles_match_the_32x32_form_on_many_shapes: widths 34, 35 at hw 2 / 3 and 37 .. 39 at hw 5 / 6 read junk
	// for their first columns since r03; no test or bench shape had such a level)
	const int hx = ((t.hw + 3) / 4) * 4;
	if (!(nx == 32 || nx >= 32 + hx) || !(ny == 32 || ny >= 32 + t.hw) || nzg < 2 * t.hw + 2) return false;
	if (!(t.hw >= 2 && t.hw <= kMarchMaxHW)) return false;  // (r06: hw 7 -- sigma_default 1.65 .. 1.75 -- has its instantiation too)
	// the symmetric form of the z-scatter needs what GaussianSmooth_3D's generator gives: tap[hw + d] == tap[hw - d] bit for bit
	for (int d = 1; d <= t.hw; d++) if (memcmp(&t.w[t.hw + d], &t.w[t.hw - d], sizeof(float)) != 0) return false;
	return true;
}

bool launch_march_level(const float *src, float *dst, float *dog, unsigned *dogmax, int nx, int ny, const ZRange &zr, const Taps &t,
                        hipStream_t st, int plan_slots, int prio, const MarchHalf *half) {
	MarchHalf hf;
	if (half && half->d && march_half_ok(nx, ny, zr)) hf = *half;
	else if (half && half->d) return false;  // (the caller asks first: march_half_ok)
	if (!march_applicable(nx, ny, zr.nzg, t)) return false;
	if (march_wide_ok(nx, ny, zr.zo1 - zr.zo0, t.hw, plan_slots)) {
		switch (t.hw) {
		case 2: launch_march_wide<2>(src, dst, dog, dogmax, nx, ny, zr, t, st, prio, hf); return true;
		case 3: launch_march_wide<3>(src, dst, dog, dogmax, nx, ny, zr, t, st, prio, hf); return true;
		case 4: launch_march_wide<4>(src, dst, dog, dogmax, nx, ny, zr, t, st, prio, hf); return true;
		case 5: launch_march_wide<5>(src, dst, dog, dogmax, nx, ny, zr, t, st, prio, hf); return true;
		case 6: launch_march_wide<6>(src, dst, dog, dogmax, nx, ny, zr, t, st, prio, hf); return true;
		default: break;
		}
	}
	switch (t.hw) {
	case 2: launch_march_hw<2>(src, dst, dog, dogmax, nx, ny, zr, t, st, plan_slots, prio, hf); return true;
	case 3: launch_march_hw<3>(src, dst, dog, dogmax, nx, ny, zr, t, st, plan_slots, prio, hf); return true;
	case 4: launch_march_hw<4>(src, dst, dog, dogmax, nx, ny, zr, t, st, plan_slots, prio, hf); return true;
	case 5: launch_march_hw<5>(src, dst, dog, dogmax, nx, ny, zr, t, st, plan_slots, prio, hf); return true;
	case 6: launch_march_hw<6>(src, dst, dog, dogmax, nx, ny, zr, t, st, plan_slots, prio, hf); return true;
	case 7: launch_march_hw<7>(src, dst, dog, dogmax, nx, ny, zr, t, st, plan_slots, prio, hf); return true;
	case 8: launch_march_hw<8>(src, dst, dog, dogmax, nx, ny, zr, t, st, plan_slots, prio, hf); return true;
	default: return false;
	}
}

// code-object preload (context.hip, first create on a device): HIP loads a translation unit's kernels at the first launch of one of them,
// which used to add ~1 ms to the first KpSiftAlgorithm of a process
void preload_march_kernels() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_march_level<2, false, false>)); }

}  // namespace s3d
