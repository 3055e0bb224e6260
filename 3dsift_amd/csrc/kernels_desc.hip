// kernels_desc.hip -- icosahedral gradient-histogram descriptor, one workgroup per keypoint.
//
// Restates Extract_Description + Extract_Descriptor_Imp + Trilinear_interpolation_over_desc_debug +
// Check_intersect_faces + cart2bary + normailize_desc (reference Src/cSIFT3D.cc:484-502, 1152-1381,
// 1450-1573, 1592-1656).
//
// Per window voxel (sphere r = 2*7.0711*scale, box up to 73^3): rotate the offset into the
// keypoint frame -> 4x4x4 bin coordinates; central-difference gradient * Gaussian weight, rotated;
// first icosahedron face (mesh order) hit by the gradient ray (Moller-Trumbore with the
// reference's eps tolerances) -> barycentric weights; trilinear scatter of |g| into 8 cells x 3
// face vertices of the 768-bin histogram.  Then L2-normalise, clamp at 0.2*128/768, normalise.
//
// MI355X mapping (256 threads per keypoint, keypoints handed out through a global counter, large windows first):
//   * a lane owns two adjacent (x, y) COLUMNS of the window and marches along z over their in-sphere, cube-clipped chord; eight lanes
//     (2 column pairs x 4 rows) form a unit that marches in lock step, the units of a window are sorted by the length of their z
//     range and dealt to the four waves longest first ("Sorted units" below)
//   * per step a lane loads one 16-byte row piece (x-1 .. x+2); the rows y-1 / y+1 come from the lanes beside it (DPP row shifts),
//     only the outer rows of a unit from memory; the centre column is carried in registers (z-1, z, z+1).  The loads travel through
//     a RING of two planes in flight (r05, "A ring of planes in flight" below): untracked loads into fixed registers, counted waits
//   * the Gaussian weight / in-sphere test come from the host-built table indexed by the integer
//     squared offset, staged in LDS (bit-identical to the CPU expf; no device exp on the path)
//   * lane compaction: chords are ragged and many voxels are inactive, so each wave pushes its ACTIVE voxels
//     (bin coordinates + weighted gradient) into a small LDS queue by ballot rank and runs the heavy part (rotation, face
//     test, trilinear weights, 24 atomics) only on full 64-lane batches; all loops are wave-uniform
//   * face lookup by the symmetry of the icosahedron (face_lookup below): |g| falls into the canonical octant face or one of its
//     three neighbours, the sign bits pick the mesh face; accepted when all three barycentrics clear a 6e-6 margin (then no other
//     face can pass the reference's -1.19e-6 test, so "first passing face in mesh order" is this face); otherwise the literal
//     20-face ordered scan runs
//   * the 24 products of a voxel go to an LDS histogram kept in 32-BIT FIXED POINT and are added with ds_add_u32: on gfx950
//     an LDS float atomic (ds_add_f32) costs ~190 cycles per wave instruction, ds_add_u32 4.3 + 3.8 per extra lane on the same
//     address, ds_add_u64 6.4 + 7.5 (scripts/microbench/lds_atomics.hip), and integer sums are order independent, so descriptors
//     are bitwise reproducible whatever the schedule.  The unit 2^-k is chosen PER KEYPOINT from the gradient mass of its window
//     (estimate first, exact bound and one redo if the estimate was too small; WinLut::fix_scale is the coarsest unit ever used):
//     a contribution is rounded to < 1e-6 of the mass.  The fp32 product mag*w*bary is formed like the reference, scaled by the
//     power of two and converted with one v_cvt_rpi_i32_f32.  Four replicas and a per-lane cell order spread the adds over the
//     banks ("Bank spreading" below); the replicas are summed once per keypoint
// Every per-voxel contribution equals the reference's up to that rounding; the ORDER of the additions is free (integers).
// Measured differences to the oracle: 1e-6 .. 4e-6 RMS -- tolerance 1e-4 RMS (BASELINE.json).
#include <float.h>

#include <stdio.h>

#include <algorithm>

#include "sift3d_internal.h"

namespace s3d {

__constant__ FaceConst c_faces[kFaces];
__constant__ FaceSym c_sym;

// __constant__ symbols live per device: called by every create on its own device (set by the caller)
hipError_t upload_faces(const FaceConst *faces, const FaceSym *sym) {
	hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_faces), faces, sizeof(FaceConst) * kFaces);
	if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(c_sym), sym, sizeof(FaceSym));
	return e;
}

__device__ __forceinline__ void win_bounds_d(float c, float rad, float u, int n, int &lo, int &hi) {
	int s = (int)floorf(c - __fdiv_rn(rad, u));
	lo = s > 1 ? s : 1;
	int e = (int)ceilf(c + __fdiv_rn(rad, u));
	hi = e < (n - 2) ? e : n - 2;
}

constexpr float kBaryEps = (float)(FLT_EPSILON * 1E1);  // Src/cSIFT3D.cc:23

// Histogram bins are 32-bit two's-complement fixed point in units of 1 / WinLut::fix_scale (see the header).
// The neighbour across an edge is the mirror image of the face (every edge of the icosahedron lies in a symmetry plane), so a weight of
// +m on this side is -m on the other: no other face can pass the reference's >= -1.19e-6 test when m exceeds 1.19e-6 plus the rounding
// of both evaluations (measured: the two routes differ by <= 3e-7).  6e-6 leaves a factor of three; the ordered scan behind it costs
// ~900 instructions per wave that has one such lane (2e-5: 0.17 ms of 3.4 at 512^3; tests/test_gpu_parity.py samples the zone densely).
constexpr float kFastMargin = 6.0e-6f;
// Four waves per SIMD (r03): a fourth workgroup per CU (4 x 39 KB of LDS) hides more of the march's load latency: 3.68 -> 3.32 ms at
// 512^3.  r05: 120 registers + the ring's 12 fixed ones, no scratch (profiles/r05_kernel_resources.txt; scripts/kernel_resources.py
// --check fails the build of the evidence set when a VGPR spills)
#define S3D_DESC_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
constexpr float kClipMargin = 0.25f;  // widening of the cube clip of a column's z range, voxels
typedef unsigned bin_t;
typedef int sbin_t;
constexpr int kRep = 4;  // histogram replicas (bin-major: address = bin * kRep + replica of the lane)
// Bank spreading (r02).  The voxels that leave the march side by side mostly share the cell AND the face, so every one of the 24 adds
// sent all lanes to the SAME bin -- R replicas = R banks, 64/R lanes queued on each (an LDS atomic costs ~1.9 cycles per lane on the
// busiest bank; measured 16 cycles per ds_add with 8 replicas).  Now lane l walks the 8 cells of its voxel in the order d ^ r,
// r = l & 7, with the replica (l >> 3) & 3, and the bin layout gives the three cell strides the residues 1, 2, 4 modulo 8: the eight
// lanes of a group (one unit of the march: one cell, as a rule) hit 8 different banks of their replica at every step, and groups with
// different replicas never meet.  (r = (l >> 2) & 7, replica l & 3 -- neighbours on different replicas, lanes four apart on
// different orders -- let voxels of different units, i.e. different cells, collide: +38 % conflict cycles, +0.15 ms.)
// vertex-major: idx = 72 v + ix + 18 iy + 4 iz.  The cell strides have the residues 1, 2, 4 modulo 8 (and ix + 4 iz < 16 <= 18), the
// vertex stride is 0 modulo 8: the bank of an add depends on the cell and the replica only, not on the face of the voxel
// (864 bins = 13.8 KB with 4 replicas; the cell-major layout 17 ix + 74 iy + 300 iz + v: 1200 bins, 19.2 KB, 24 % more conflict cycles).
constexpr int kSX = 1, kSY = 18, kSZ = 4, kSV = 72, kBins = 12 * kSV;
// (r05) the 24 adds of a voxel are straight-line code: a cell outside the 4x4x4 block gets a ZERO weight -- an integer add of 0 -- instead
// of an exec-mask branch per cell; a cell coordinate of 4 addresses up to bin 4 + 18 * 4 + 4 * 4 + 72 * 11 = 884, so the array is
// padded for those adds of 0 to stay inside it
constexpr int kBinsAlloc = kBins + 24;
__device__ __forceinline__ int bin_index(int j) {  // descriptor element j = (ix + 4 iy + 16 iz) * 12 + v  ->  histogram index
	const int c = j / 12, v = j - c * 12;
	return (c & 3) * kSX + ((c >> 2) & 3) * kSY + (c >> 4) * kSZ + v * kSV;
}

// literal Check_intersect_faces: first face in mesh order that passes (wave-uniform loop, constant memory)
__device__ __forceinline__ int intersect_scan(float gx, float gy, float gz, float &b0, float &b1, float &b2) {
	int found = -1;
	for (int f = 0; f < kFaces; f++) {
		const FaceConst &F = c_faces[f];
		const float px = gy * F.e2[2] - gz * F.e2[1];
		const float py = gz * F.e2[0] - gx * F.e2[2];
		const float pz = gx * F.e2[1] - gy * F.e2[0];
		const float det = F.e1[0] * px + F.e1[1] * py + F.e1[2] * pz;
		bool ok = found < 0 && !(fabsf(det) < kBaryEps);
		const float det_inv = __fdiv_rn(1.0f, det);  // == (float)(1.0 / (double)det), see face_test
		const float y = det_inv * (px * F.t[0] + py * F.t[1] + pz * F.t[2]);
		const float z = det_inv * (gx * F.q[0] + gy * F.q[1] + gz * F.q[2]);
		const float x = 1.0f - y - z;
		const float k = det_inv * F.qe2;
		ok = ok && !(x < -kBaryEps || y < -kBaryEps || z < -kBaryEps || k < 0.0f);
		if (ok) { found = f; b0 = x; b1 = y; b2 = z; }
		if (__all(found >= 0)) break;
	}
	return found;
}

// lookup tables -> LDS; 256 threads, caller syncs.  s_fidx[f * 4 + j]: BYTE offset (inside a replica-interleaved histogram) of the first
// bin of vertex idx[j] of face f (slow path); s_sym[key]: the same for the three roles of the symmetric lookup, plus
// face | slot0 << 8 | slot1 << 10 | slot2 << 12 (see FaceSym)
__device__ __forceinline__ void stage_face_tables(int tid, int *s_fidx, int4 *s_sym) {
	constexpr int kVB = kSV * kRep * (int)sizeof(bin_t);  // bytes between the first bins of consecutive vertices
	for (int i = tid; i < kFaces * 4; i += 256) s_fidx[i] = (i & 3) < 3 ? c_faces[i >> 2].idx[i & 3] * kVB : 0;
	if (tid < 32)
		s_sym[tid] = make_int4(c_sym.vert[tid][0] * kVB, c_sym.vert[tid][1] * kVB, c_sym.vert[tid][2] * kVB,
		                       c_sym.face[tid] | c_sym.slot[tid][0] << 8 | c_sym.slot[tid][1] << 10 | c_sym.slot[tid][2] << 12);
}

// Check_intersect_faces + cart2bary for one gradient (Src/cSIFT3D.cc:1542-1573, 1592-1637) by the symmetry of the mesh (r03; before:
// four dot products to predict a face, then the reference's Moller-Trumbore arithmetic for that face from a 13-constant LDS gather).
// With a = |g| componentwise the ray hits the canonical octant face A = (0,1,phi), B = (1,phi,0), C = (phi,0,1) or, across one of
// its edges, the half of a neighbouring face that reaches into the octant; the signs of g then select the mesh face (s_sym).
//   octant face:   a = lA A + lB B + lC C  with  lA = az + ay/phi^2 - ax/phi  (and cyclically): inside iff all three >= 0
//   lX < 0 (the smallest): the neighbour across the edge opposite X.  After the cyclic permutation (u, v, w) of a that maps it onto
//   {A, A' = (0,-1,phi), C}:  weights  t + v,  t - v,  2 u / phi  with  t = w / phi - u / phi^2  (v >= 0: the larger one belongs to the
//   vertex on g's side of the straddled axis)
// The barycentrics are the weights over their sum: the same quantities the reference's b = (1 - y - z, y, z) expresses, to rounding
// (|difference| <~ 5e-7).  They are accepted when all three clear kFastMargin: no other face can then pass the reference's
// >= -1.19e-6 test, so "first passing face in mesh order" is this face; otherwise (direction within the margin of an edge or a
// vertex) the literal ordered 20-face scan decides.  Wave-uniform control flow.
// Out: f, the three weights and the BYTE offsets of the first bins of their vertices; *packed = f | slot_r << (8 + 2 r): position of
// weight r in the reference's bary[] (debug entry).
constexpr float kInvPhi = 0.6180339888f, kInvPhi2 = 0.3819660113f;
__device__ __forceinline__ int face_lookup(bool valid, float rx, float ry, float rz, const int *s_fidx, const int4 *s_sym, float &b0,
                                           float &b1, float &b2, int &o0, int &o1, int &o2, int *packed = nullptr, bool scan_only = false) {
	int f = -1, pk = 0;
	bool slow = false;
	o0 = o1 = o2 = 0;
	if (valid) {
		const float ax = fabsf(rx), ay = fabsf(ry), az = fabsf(rz);
		const float lA = __fmaf_rn(-kInvPhi, ax, __fmaf_rn(kInvPhi2, ay, az));
		const float lB = __fmaf_rn(-kInvPhi, az, __fmaf_rn(kInvPhi2, ax, ay));
		const float lC = __fmaf_rn(-kInvPhi, ay, __fmaf_rn(kInvPhi2, az, ax));
		const float mn = fminf(fminf(lA, lB), lC);
		const bool oct = !(mn < 0.0f);
		const bool isA = lA == mn, isB = !isA && lB == mn;
		const float u = isA ? ay : (isB ? ax : az), v = isA ? az : (isB ? ay : ax), w = isA ? ax : (isB ? az : ay);
		const float t = __fmaf_rn(kInvPhi, w, -kInvPhi2 * u);
		const float l0 = oct ? lA : t + v, l1 = oct ? lB : t - v, l2 = oct ? lC : (2.0f * kInvPhi) * u;
		const float inv = __builtin_amdgcn_rcpf(l0 + l1 + l2);  // 1 ulp
		b0 = l0 * inv; b1 = l1 * inv; b2 = l2 * inv;
		const int type = oct ? 0 : (isA ? 1 : (isB ? 2 : 3));
		const int bits = (rx < 0.f ? 1 : 0) | (ry < 0.f ? 2 : 0) | (rz < 0.f ? 4 : 0);
		const int4 e = s_sym[type * 8 + bits];
		o0 = e.x; o1 = e.y; o2 = e.z; pk = e.w; f = e.w & 31;
		slow = scan_only || !(fminf(fminf(b0, b1), b2) >= kFastMargin);
	}
	if (__any(slow)) {
		if (slow) {
			f = intersect_scan(rx, ry, rz, b0, b1, b2);
			const int ff = f < 0 ? 0 : f;
			o0 = s_fidx[ff * 4]; o1 = s_fidx[ff * 4 + 1]; o2 = s_fidx[ff * 4 + 2];
			pk = ff | 0 << 8 | 1 << 10 | 2 << 12;
		}
	}
	if (packed) *packed = pk;
	return f;
}

// heavy part of one ACTIVE voxel (inside sphere and cube, |g|^2 >= eps): face lookup, trilinear weights,
// 24 fixed-point adds.  Runs on compacted full waves (see the queue in k_describe).
// fp32 -> int32, round to nearest (ties up): ONE instruction (__float2int_rn is v_rndne_f32 + v_cvt_i32_f32)
__device__ __forceinline__ int cvt_rpi(float x) {
	int r;
	asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
	return r;
}

// R = the transposed rotation (rows R0..R8), fix_scale = 2^k of the histogram's fixed point.  The queue holds the WEIGHTED, NOT YET
// ROTATED gradient of voxels that passed a slightly relaxed magnitude test; the rotation and the reference's exact test
// (Src/cSIFT3D.cc:1323-1325, 1468) run here, on the compacted voxels only.
// r04: the queue carries the cell coordinates MINUS 1.5, formed by the march as one fused multiply-add per axis from
// per-column constants (a few 1e-7 off the reference's five roundings per axis; the march keeps every voxel with max |c| < 2 + band).
// The descriptor is a discontinuous function of the cell coordinates b = c + 1.5 in two places only: the faces of the 4x4x4 cube
// (!(b <= -0.5 || b >= 3.5), Src/cSIFT3D.cc:1299-1303) and b = 0 on every axis (the cell index truncates toward zero while the
// fraction uses floor: at b = -eps the weights eps and 1 - eps trade places).  A batch with a lane inside a band of kCellBand around
// one of them recovers the integer voxel offsets (R^T applied to the coordinates, rounded: errors of 1e-4 voxel) and repeats the
// reference's arithmetic for those lanes: the decisions are the reference's, everywhere else the weights move by ~1e-7.
struct CellGeom { float desc_hw, bin_fctr, inv_bin, u, inv_u, band; };  // band: kCellBand, or huge (hook desc_exact_cells: every voxel takes the exact path)
constexpr float kCellBand = 1.0e-4f;
__device__ __forceinline__ float accumulate_voxel(bool valid, float bx, float by, float bz, float gx, float gy, float gz,
                                                 float R0, float R1, float R2, float R3, float R4, float R5, float R6, float R7, float R8,
                                                 float fix_scale, const int *s_fidx, const int4 *s_sym, bin_t *hist_rep,
                                                 int spread /* lane constant: bits 0..2 = r */, const CellGeom &cg) {
#if defined(S3D_DDIAG) && (S3D_DDIAG & 2)  // timing only: march and queue without the heavy part
	return 0.0f;
#endif
	{
		const float m = fmaxf(fmaxf(fabsf(bx), fabsf(by)), fabsf(bz));  // (bx, by, bz hold c = b - 1.5 here)
		bx = bx + 1.5f; by = by + 1.5f; bz = bz + 1.5f;
		const float z3 = fminf(fminf(fabsf(bx), fabsf(by)), fabsf(bz));
		const bool amb = valid && (!(m < 2.0f - cg.band) || z3 < cg.band);
		if (__any(amb)) {  // wave-uniform, a few per cent of the batches
			const float px = (bx + 0.5f) * cg.inv_bin - cg.desc_hw, py = (by + 0.5f) * cg.inv_bin - cg.desc_hw, pz = (bz + 0.5f) * cg.inv_bin - cg.desc_hw;
			const float vxd = rintf((R0 * px + R3 * py + R6 * pz) * cg.inv_u) * cg.u;  // (dx, dy, dz) * unit, exact
			const float vyd = rintf((R1 * px + R4 * py + R7 * pz) * cg.inv_u) * cg.u;
			const float vzd = rintf((R2 * px + R5 * py + R8 * pz) * cg.inv_u) * cg.u;
			// the reference's order: (R0 vx + R1 vy) + R2 vz, + desc_hw, * bin_fctr, - 0.5 (Src/cSIFT3D.cc:1286-1297)
			float ex = R0 * vxd + R1 * vyd, ey = R3 * vxd + R4 * vyd, ez = R6 * vxd + R7 * vyd;
			ex = ex + R2 * vzd; ey = ey + R5 * vzd; ez = ez + R8 * vzd;
			ex = (ex + cg.desc_hw) * cg.bin_fctr; ey = (ey + cg.desc_hw) * cg.bin_fctr; ez = (ez + cg.desc_hw) * cg.bin_fctr;
			ex = ex - 0.5f; ey = ey - 0.5f; ez = ez - 0.5f;
			// per LANE: a voxel's coordinates must not depend on which voxels share its batch (the batches form differently from run to run)
			if (amb) {
				bx = ex; by = ey; bz = ez;
				valid = fminf(fminf(bx, by), bz) > -0.5f && fmaxf(fmaxf(bx, by), bz) < 3.5f;
			}
		}
	}
	// The rotated gradient is formed exactly like the reference's (separate multiplies and adds, left to right): the face lookup
	// below is a DISCONTINUOUS function of its direction -- Initialize_geometry swaps the coordinates of the first two vertices of
	// a face whose normal points inwards but not their bin indices (Src/cUtil.cc:164-171), so across an edge between such a face
	// and a regular one the weights of two bins trade places -- and a last-bit difference (fused multiply-adds were tried) moved
	// single strong voxels across: one descriptor element off by 1e-3 in a handful of the 11 292 keypoints of the 512^3 volume
	const float rx = R0 * gx + R1 * gy + R2 * gz;
	const float ry = R3 * gx + R4 * gy + R5 * gz;
	const float rz = R6 * gx + R7 * gy + R8 * gz;
	const float g2 = rx * rx + ry * ry + rz * rz;
	valid = valid && !(g2 < kBaryEps);
	float b0 = 0.f, b1 = 0.f, b2 = 0.f;
	int o0, o1, o2;
	const int f = face_lookup(valid, rx, ry, rz, s_fidx, s_sym, b0, b1, b2, o0, o1, o2);
	if (!valid || f < 0) return 0.0f;
	const float mag = __fsqrt_rn(g2);
	const float fx = bx - floorf(bx), fy = by - floorf(by), fz = bz - floorf(bz);
	const int ix = (int)bx, iy = (int)by, iz = (int)bz;  // truncation toward zero, like the reference: 0..3
	// Trilinear weights (Src/cSIFT3D.cc:1510-1512 forms them as double products rounded to fp32; fp32 products differ
	// from that by <= 1.5 ulp, far inside the descriptor tolerance)
	const float wx0 = 1.0f - fx, wy0 = 1.0f - fy, wz0 = 1.0f - fz;
	const float ms = mag * fix_scale;                               // exact (power of two)
	const float m0 = ms * b0, m1 = ms * b1, m2 = ms * b2;
	// cells ix+ddx etc. are >= 0 by construction; only the upper bound can fail.  The upper cell of an axis whose base cell is 3 lies
	// outside the block (Src/cSIFT3D.cc:1493-1497 skips it): its weight becomes 0, every product of the cell is then +-0, cvt_rpi gives
	// the integer 0 and the add changes nothing -- straight-line code instead of an exec-mask branch per cell (r05: -40 scalar
	// instructions and eight branches per batch; 3.31 -> 3.27 ms, same integers)
	const float fxe = ix < 3 ? fx : 0.0f, fye = iy < 3 ? fy : 0.0f, fze = iz < 3 ? fz : 0.0f;
	// step d of this lane is the cell offset d ^ r: weights and cell strides swap roles per axis where the bit of r is set
	const bool qx = spread & 1, qy = spread & 2, qz = spread & 4;
	const float ax[2] = {qx ? fxe : wx0, qx ? wx0 : fxe}, ay[2] = {qy ? fye : wy0, qy ? wy0 : fye}, az[2] = {qz ? fze : wz0, qz ? wz0 : fze};
	const int stx = qx ? -kSX * kRep : kSX * kRep, sty = qy ? -kSY * kRep : kSY * kRep, stz = qz ? -kSZ * kRep : kSZ * kRep;
	// (24-bit multiplies: full rate; v_mul_lo_u32 issues at a quarter of it)
	const int base = __mul24(ix + (qx ? 1 : 0), kSX * kRep) + __mul24(iy + (qy ? 1 : 0), kSY * kRep) + __mul24(iz + (qz ? 1 : 0), kSZ * kRep);
	const float pxy[4] = {ax[0] * ay[0], ax[0] * ay[1], ax[1] * ay[0], ax[1] * ay[1]};  // index ddx*2 + ddy
	char *hb = reinterpret_cast<char *>(hist_rep + base);
	char *h0 = hb + o0, *h1 = hb + o1, *h2 = hb + o2;
#pragma unroll
	for (int d = 0; d < 8; d++) {
		const int ddx = d >> 2, ddy = (d >> 1) & 1, ddz = d & 1;
		const float wgt = pxy[ddx * 2 + ddy] * az[ddz];
		const float p0 = wgt * m0, p1 = wgt * m1, p2 = wgt * m2;
		const int off = ((ddx ? stx : 0) + (ddy ? sty : 0) + (ddz ? stz : 0)) * (int)sizeof(bin_t);  // bytes: a lane constant, hoisted
#if defined(S3D_DDIAG) && (S3D_DDIAG & 1)  // timing only: the 24 adds are computed but not sent to the LDS
		asm volatile("" ::"v"(h0 + off), "v"(cvt_rpi(p0)));
		asm volatile("" ::"v"(h1 + off), "v"(cvt_rpi(p1)));
		asm volatile("" ::"v"(h2 + off), "v"(cvt_rpi(p2)));
#else
		atomicAdd(reinterpret_cast<bin_t *>(h0 + off), (bin_t)(sbin_t)cvt_rpi(p0));
		atomicAdd(reinterpret_cast<bin_t *>(h1 + off), (bin_t)(sbin_t)cvt_rpi(p1));
		atomicAdd(reinterpret_cast<bin_t *>(h2 + off), (bin_t)(sbin_t)cvt_rpi(p2));
#endif
	}
	return mag;
}

// timing-only builds (wrong results, same control flow; never set in the product build): -DS3D_DDIAG=1 no histogram adds, 2 no heavy part
// Sorted units.  A lane marches two adjacent columns (x, x+1).  A unit is a block of kPX such pairs by kSH rows on
// kPX * kSH consecutive lanes.  The units of a window -- in chunks of kPairCap pairs, row-major -- are sorted by the length of
// their z range and dealt to the waves 64 lanes at a time, longest first: the lanes of a wave finish together.  (A fixed 16 x 8
// tiling of the circular footprint leaves 35-45 % of the lane-steps idle: rim tiles march their longest chord with most lanes
// outside the sphere.  Measured at 512^3, k_describe: tiles 4.92 ms, sorted single pairs 4.71, 1x4 units 4.39, 2x4 units 4.26,
// 2x2 / 4x2 4.32, 4x4 4.46, 1x8 4.50.)
// runs with fewer keypoints than this take eight waves per keypoint; measured crossover between 1100 (0.49 vs 0.60 ms) and 1850 keypoints (0.75 vs 0.71)
constexpr unsigned kWideBelow = 1400;
// r04: keypoint counts below which a window is split over 8 / 4 workgroups (DescSplit; 1024 workgroups are resident; two parts never
// paid off against the eight-wave variant)
constexpr unsigned kSplit8Below = 320, kSplit4Below = 700;
// The lanes of a unit march the union of their z ranges in lock step, so the y neighbours of a pair are
// the centre values the lanes kPX below / above hold in registers (DPP row shifts); only the first / last row of a unit loads its
// outer row from memory: two vector-memory instructions per step, the second with a quarter of the lanes, instead of three.  The
// march is bound by the cache lines its loads touch (timing-only builds, S3D_DDIAG): wider units share the lines of a row.
constexpr int kPairCap = 1024, kLenBins = 128, kSH = 4, kPX = 2;
constexpr int kUL = kPX * kSH;  // lanes of a unit: kPX pairs wide, kSH rows high, row-major on consecutive lanes
static_assert(kSH >= 2 && kPX >= 1 && kUL <= 16 && (kSH & (kSH - 1)) == 0 && (kPX & (kPX - 1)) == 0, "a unit is a power of two of lanes within a DPP row");
typedef float f4g __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2g __attribute__((ext_vector_type(2), aligned(4)));
// values of the lanes N below / above in the 16-lane row.  Inline asm and volatile: hipcc sinks __builtin_amdgcn_update_dpp into the
// branch it makes of a following select, where the source lanes are masked off and the DPP read returns 0.  (s_nop: a VALU write of
// the source needs two wait states before a DPP read, and the hazard pass does not look into inline asm; r04: the four moves of a step
// behind ONE pair of wait states.)  bound_ctrl (r05): a lane whose source lies outside its row reads 0 instead of keeping its
// destination -- no initialised destinations, four v_mov fewer per step; those lanes are the first / last row of a unit or of the
// 16-lane row and take their outer row from memory anyway.
template <int N>
__device__ __forceinline__ void dpp_neighbours(float a, float b, f2g &below, f2g &above) {
	float r0, r1, r2, r3;
	asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %4 row_shr:%6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %1, %5 row_shr:%6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
	             "v_mov_b32_dpp %2, %4 row_shl:%6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %3, %5 row_shl:%6 row_mask:0xf bank_mask:0xf bound_ctrl:1"
	             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a), "v"(b), "n"(N));
	below = f2g{r0, r1}; above = f2g{r2, r3};
}
// ---- r05: A ring of planes in flight ----
// hipcc copies a loop-carried load result at the loop's back-edge and waits for it there (s_waitcnt vmcnt(0)): the row piece the r02
// form requested at the top of a march step was waited for at the bottom of the SAME step.  Now the loads are inline asm the compiler
// does not track, into a ring of two slots (a slot = the 16-byte row piece of a plane + the 8-byte outer row of the plane below it),
// the step loop is unrolled by two so that each slot has a fixed name, and a step waits with a COUNTED vmcnt for its slot only: vmcnt
// counts in order and every step issues exactly two loads, so a slot is complete when at most two loads are outstanding.
// The slots live in FIXED physical registers (operand constraints "{v[a:b]}"), and ONE asm statement waits for a slot, copies it out
// component by component and requests the next plane into the same registers: with wait, copy and reload as separate statements hipcc
// copied the slot into other registers IN FRONT of the wait (it believes a load's result is there when the statement ends) -- reading
// registers whose load was still in flight.  The only reader of a slot is inside the block that waits for it; between two turns the
// slot is a live value the compiler has no reason to touch.  scripts/check_desc_ring.py verifies on the generated code that no
// instruction outside these blocks reads or writes the ring's registers and that the kernel has no scratch (scratch traffic counts in
// vmcnt too); tests/test_cabi_cpu.py runs it.
// Edge mask: the lanes that own an outer row (first / last row of a unit); never empty (lane 0 of a wave is a first row), so a step
// always issues its two loads.  s_and_saveexec writes SCC: declared (a ring of three was first built without the clobber and its
// loop-entry compare, scheduled in front of the prologue's loads, read a clobbered SCC -- zero descriptors, no fault).
// Measured at 512^3 (scripts/ab_full.py, same hash): two planes in flight 3.31 -> 3.21 ms, three 3.22; with the march at raised wave
// priority (s_setprio around the heavy part) 3.19 -> 3.15.
typedef const float __attribute__((address_space(1))) *gcf_p;
#define S3D_RING_LOAD(P, E, PREG, EREG, prow, pedge, emask)                                                                              \
	do {                                                                                                                                  \
		unsigned long long sv_;                                                                                                           \
		asm volatile("s_and_saveexec_b64 %[sv], %[em]\n\tglobal_load_dwordx2 %[e], %[ea], off\n\ts_mov_b64 exec, %[sv]\n\t"                   \
		             "global_load_dwordx4 %[p], %[pa], off offset:-4"                                                                      \
		             : [p] "={" PREG "}"(P), [e] "={" EREG "}"(E), [sv] "=&s"(sv_)                                                         \
		             : [pa] "v"(prow), [ea] "v"(pedge), [em] "s"(emask)                                                                   \
		             : "memory", "scc");  /* s_and_saveexec writes SCC */                                                                   \
	} while (0)
#define S3D_RING_TURN(N, P, E, PREG, EREG, P0, P1, P2, P3, E0, E1, rout, eout, prow, pedge, emask)                                        \
	do {                                                                                                                                  \
		unsigned long long sv_;                                                                                                           \
		asm volatile("s_waitcnt vmcnt(" #N ")\n\t"                                                                                        \
		             "v_mov_b32 %[r0], " P0 "\n\tv_mov_b32 %[r1], " P1 "\n\tv_mov_b32 %[r2], " P2 "\n\tv_mov_b32 %[r3], " P3 "\n\t"            \
		             "v_mov_b32 %[o0], " E0 "\n\tv_mov_b32 %[o1], " E1 "\n\t"                                                            \
		             "s_and_saveexec_b64 %[sv], %[em]\n\tglobal_load_dwordx2 %[e], %[ea], off\n\ts_mov_b64 exec, %[sv]\n\t"                   \
		             "global_load_dwordx4 %[p], %[pa], off offset:-4"                                                                      \
		             : [p] "+{" PREG "}"(P), [e] "+{" EREG "}"(E), [r0] "=&v"(rout.x), [r1] "=&v"(rout.y), [r2] "=&v"(rout.z), [r3] "=&v"(rout.w), \
		               [o0] "=&v"(eout.x), [o1] "=&v"(eout.y), [sv] "=&s"(sv_)                                                             \
		             : [pa] "v"(prow), [ea] "v"(pedge), [em] "s"(emask)                                                                   \
		             : "memory", "scc");  /* s_and_saveexec writes SCC */                                                                   \
	} while (0)
constexpr int kQCap = 128;  // per-wave queue capacity (entries); a push adds <= 64, a pop removes exactly 64
// r04: a queue entry is a 16-byte piece (bx, by, bz, gx) + an 8-byte piece (gy, gz) in two arrays instead of six 4-byte arrays: a push
// is 2 LDS writes instead of 6, a pop 2 reads instead of 6 (consecutive ranks -> consecutive pieces: conflict-free) -- 8 fewer
// instructions per march step, 4 fewer per batch, of a kernel that is bound by instruction issue
typedef float qf4 __attribute__((ext_vector_type(4)));
typedef float qf2 __attribute__((ext_vector_type(2)));

// Fixed-point unit of a keypoint's 32-bit histogram (see k_describe): the largest power of two <= 2^31 / mass, clamped to [the unit that is
// provable for any data of this window size, 2^29]
__device__ __forceinline__ float pick_scale_d(float mass, float provable) {
	const float q = __fdiv_rn(2147483648.0f * 0.98f, fmaxf(mass, 1e-30f));
	const float p2 = __uint_as_float(__float_as_uint(q) & 0xFF800000u);
	return fminf(fmaxf(p2, provable), 536870912.0f);
}
// ... of the FIRST pass: from an estimate of the gradient mass (rms gradient of the orientation window, from the structure tensor, times
// the descriptor window's weight sum, with 4x head room) -- a function of the keypoint's record and the level's tables alone, so every
// GPU that marches a part of the window (PARTIAL) uses the same unit.  dev_flags bits 8..: the hook SIFT3D_HOOK_DESC_MASS_SHIFT (estimate / 2^s)
// r05: ... times the share of the window's bounding box that lies inside the level: a window cut off by the volume's border collects that
// much less (scripts/soak_random.py: a radius-55 window in a 64 x 128 x 80 volume got a unit 16x coarser than its mass allowed and lost
// 2.9e-5 RMS to rounding).  Whole windows: the factor is exactly 1.
// (wx, wy, wz: the window's clipped box, Src/cSIFT3D.cc:1184-1200, of the WHOLE window -- the same on every rank that marches a part)
__device__ __forceinline__ float first_pass_unit(const DevKp &kp, const WinLut &lut_o, const WinLut &lut, int wx, int wy, int wz, int dev_flags) {
	const float st_tr = fmaxf(kp.st[0] + kp.st[4] + kp.st[8], 0.0f);
	const float side = (float)(2 * (int)__fsqrt_rn((float)max(lut.nin, 0)) + 1);  // lattice points across the sphere
	const float inside = fminf(__fdiv_rn((float)max(wx, 0) * (float)max(wy, 0) * (float)max(wz, 0), side * side * side), 1.0f);
	const float m_est = __fsqrt_rn(__fdiv_rn(st_tr, lut_o.wsum)) * lut.wsum * inside;
	return pick_scale_d(m_est * 4.0f * __uint_as_float((unsigned)(127 - ((dev_flags >> 8) & 63)) << 23), lut.fix_scale);
}
// does a pass whose unit was fix_scale have to be repeated with the exact unit?  mass = the window's gradient mass * 1.001.  Every bin
// (and replica) sum is <= mass * fix_scale + half a unit per contribution (< 2^20 contributions): overflow.  And a first guess far ABOVE
// the mass (a sharp structure inside the orientation window, a flat descriptor window: the zero background of CT / MR volumes) leaves a
// unit that much coarser than necessary: below 1/64 of the range the keypoint is redone as well, whenever the exact bound gives a finer
// unit (rounding noise per bin stays < 1e-5 of the descriptor norm).
// r05 (scripts/soak_random.py, sigma_default 2.47: a window of radius 55 voxels, 7e5 lattice points in its sphere): what counts is the
// size of the AVERAGE contribution in units -- products below half a unit round to zero, and the products of a voxel (trilinear x
// barycentric weights) crowd towards zero, so coarse units lose a share of every bin that normalisation only cancels where it is the
// same in all bins: 13 M contributions of 2.5 units left 2.9e-5 RMS in that keypoint (5e-7 with the exact unit).  A window with more
// lattice points in its sphere than 2^18 (the default parameters' largest: 2.1e5) must therefore use a share of the range larger by
// that ratio, at most 1/4 -- the average contribution stays what it is for the default windows at 1/64.
__device__ __forceinline__ bool unit_fails(float mass, float fix_scale, float provable, int nin) {
	const bool overflow = !(mass * fix_scale + 1048576.0f < 2147483648.0f);
	const float nsphere = 4.18879f * (float)nin * __fsqrt_rn((float)nin);  // lattice points within radius sqrt(nin)
	const float share = fminf(fmaxf(nsphere * (1.0f / 262144.0f), 1.0f) * (1.0f / 64.0f), 0.25f);
	const bool coarse = mass * fix_scale < 2147483648.0f * share && pick_scale_d(mass, provable) > fix_scale;
	return overflow || coarse;
}
// normalise -> clamp -> normalise (Src/cSIFT3D.cc:1350-1358, 1639-1656) of the 768 values the first 256 threads of a workgroup hold
// three each (elements te, te + 256, te + 512), and the store of the descriptor row.  Every thread of the workgroup calls it (barriers).
__device__ __forceinline__ void normalise_store(float v0, float v1, float v2, int te, int lane, int wid, float *red /*[>= 4]*/, float *out) {
	const float trunc_thresh = (float)(0.2 * 128 / kDesc);
	for (int pass = 0; pass < 2; pass++) {
		float s = v0 * v0 + v1 * v1 + v2 * v2;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) s = s + __shfl_xor(s, o, 64);
		__syncthreads();
		if (lane == 0) red[wid] = s;
		__syncthreads();
		float norm = (red[0] + red[1]) + (red[2] + red[3]);  // (waves 0..3 hold the 768 elements in every variant)
		norm = (float)((double)__fsqrt_rn(norm) + DBL_EPSILON);
		const float inv = (float)(1.0 / (double)norm);
		v0 = v0 * inv; v1 = v1 * inv; v2 = v2 * inv;
		if (pass == 0) {
			v0 = v0 < trunc_thresh ? v0 : trunc_thresh;
			v1 = v1 < trunc_thresh ? v1 : trunc_thresh;
			v2 = v2 < trunc_thresh ? v2 : trunc_thresh;
		}
	}
	if (te < 256) { out[te] = v0; out[te + 256] = v1; out[te + 512] = v2; }
}

// LUT_LDS: the window's weight table is staged in LDS (default parameters: 1293 entries).  Larger windows (sigma_default well
// above 1.6) read the table from global memory instead (L2-resident, a few KB): slower, but no size limit.
// NT threads per workgroup (= per keypoint): 256, or 512 for runs with few keypoints (r03: a keypoint's window is marched by ONE
// workgroup, 0.1-0.4 ms for the large windows -- with fewer keypoints than a few per resident workgroup that latency is the
// kernel's time: 0.53 ms for the 286 keypoints of a 128^3 volume, 0.45 ms for the 80 a rank gets of a sharded run's replicated tail;
// eight waves halve it).  Both variants are launched; the one whose range [nkp_min, nkp_max) does not hold the keypoint count
// returns at once (the count is only known on the device).  Results are bit-identical: the histograms are integer sums, and the
// final normalisation always runs on the first 256 threads in the same order.
// PARTIAL (r05, the z-slab sharding of one volume over several GPUs): the launch marches, for each of kp_cap RECORDS -- the keypoints of
// this rank and of the z-neighbours whose windows reach into it, in up to kDescSegs lists (DescPartial::seg) -- this rank's PART of the
// descriptor window and leaves the integer histogram (768 sums in descriptor order) and the part's gradient mass in the list's
// hist / mass; nothing is normalised.  The planes of a window are partitioned over the ranks: the OWNER of a keypoint takes the planes
// its level buffer holds, [o0 - H, o1 + H) with H = the halo of the keypoint's level minus the plane of the central difference (the halo
// is there for the orientation windows anyway), every other rank the planes it owns outside that range.  Every contribution is the
// integer the single-volume run adds, so the parts of all ranks sum to that run's histogram bit for bit (k_describe_finish).
template <bool LUT_LDS, int NT, bool PARTIAL = false>
__global__ void __launch_bounds__(NT) S3D_DESC_ATTR k_describe(const DevKp *__restrict__ kps, const unsigned *__restrict__ d_count, unsigned cap,
                                                  const LevelRef *__restrict__ levels, const WinLut *__restrict__ luts,
                                                  const float *__restrict__ lutpool, float *__restrict__ d_desc, unsigned kp_cap,
                                                  int part_rank, int part_world, const int *__restrict__ order,
                                                  const unsigned *__restrict__ d_nkp, unsigned *__restrict__ d_work, int dev_flags, unsigned nkp_min,
                                                  unsigned nkp_max, DescSplit sp, DescPartial pp) {
	constexpr int NW = NT / 64;
	static_assert(NT == 256 || NT == 512, "k_describe: 4 or 8 waves per keypoint");
	static_assert(!PARTIAL || NT == 256, "partial windows: four waves per record");
	if (!PARTIAL) {
		// [nkp_min, nkp_max): the keypoint counts (of this handle's share) the EIGHT-wave variant takes; the four-wave variant takes the rest
		const unsigned n_all = min(d_nkp[0], kp_cap);
		const unsigned w = part_world > 1 ? (unsigned)part_world : 1u, r = part_world > 1 ? (unsigned)part_rank : 0u;
		const unsigned n_own = n_all > r ? (n_all - r + w - 1) / w : 0u;
		const bool wide_range = n_own >= nkp_min && n_own < nkp_max;
		if (wide_range != (NT == 512)) return;  // the other variant's run
	}
	// d_work[0] = the work counter, d_work[1] = keypoints that took the second pass (sift3d_debug_counters)
	// dev_flags (test hooks): bit 0 = recompute the chords (SIFT3D_HOOK_DESC_NOCACHE), bit 1 = SIFT3D_HOOK_DESC_EXACT_CELLS, bits 8.. = s of SIFT3D_HOOK_DESC_MASS_SHIFT
	__shared__ unsigned s_item, s_tile, s_last;
	__shared__ bin_t hist[kBinsAlloc * kRep];  // [bin][replica], two's-complement fixed point, units of 1 / lut.fix_scale
	__shared__ float s_lut[LUT_LDS ? kMaxDescLut : 1];
	__shared__ __attribute__((aligned(16))) float s_q[NW][6][kQCap];  // per-wave queue of active voxels: bx,by,bz,rx,ry,rz
	__shared__ int4 s_sym[32];
	__shared__ int s_fidx[kFaces * 4];
	__shared__ float red[NW];
	__shared__ unsigned short s_units[kPairCap];  // the non-empty column pairs of the chunk, longest z range first
	__shared__ unsigned s_chord[kPairCap];        // z ranges of a pair's two columns: (za0, zb0, za1, zb1) - z0, one byte each
	__shared__ unsigned s_cnt[kLenBins];          // counting sort: pairs per length, then the running start of each length
	__shared__ unsigned s_nnz;
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

	stage_face_tables(tid, s_fidx, s_sym);
	int cur_lut = -1;
	qf4 *qa = reinterpret_cast<qf4 *>(&s_q[wid][0][0]);  // [kQCap] x (bx, by, bz, gx) in the first four rows' storage
	qf2 *qb = reinterpret_cast<qf2 *>(&s_q[wid][4][0]);  // [kQCap] x (gy, gz) in the last two
	// eight consecutive lanes (voxels that left the march side by side: one unit, mostly one cell) take the eight cell orders and share a
	// replica; the next eight use the next replica.  Lanes from different units -- different cells -- then never meet on a bank.
	bin_t *hist_rep = &hist[(lane >> 3) % kRep];
	const int spread = lane & 7;

	// The accepted keypoints (slot -> extremum list from k_slots) are handed out one at a time through a global counter:
	// window sizes differ 4x between keypoint levels, so a static deal leaves a long tail.  A partitioned run (multi-GPU
	// split of replicated octaves) takes the slots rank, rank + world, ... and zeroes the rows of the other ranks.
	(void)d_count; (void)cap;
	const unsigned nkp = PARTIAL ? kp_cap : min(d_nkp[0], kp_cap);
	const unsigned pw = (!PARTIAL && part_world > 1) ? (unsigned)part_world : 1u, pr = (!PARTIAL && part_world > 1) ? (unsigned)part_rank : 0u;
	if (pw > 1)
		for (unsigned pos = blockIdx.x; pos < nkp; pos += gridDim.x)
			if (pos % pw != pr) {
				const size_t row = (size_t)kps[order[pos]].slot;
				for (int i = tid; i < kDesc; i += NT) d_desc[row * kDesc + i] = 0.0f;
			}
	const unsigned nown = nkp > pr ? (nkp - pr + pw - 1) / pw : 0u;
	// r04, few keypoints: a window split over S0 workgroups (see DescSplit).  S0 is a function of the keypoint count alone, the same
	// in every workgroup.
	int S0 = 1;
	if (!PARTIAL && NT == 256 && sp.gacc != nullptr && nown <= sp.cap) S0 = nown < kSplit8Below ? 8 : (nown < kSplit4Below ? 4 : 1);
	const unsigned nitems = nown * (unsigned)S0;
	for (;;) {
		__syncthreads();
		if (tid == 0) s_item = atomicAdd(d_work, 1u);
		__syncthreads();
		const unsigned item = s_item;
		if (item >= nitems) break;  // block-uniform
		const unsigned kpos = item / (unsigned)S0;  // position in the processing order
		const int part = (int)(item % (unsigned)S0);
		int S = S0;  // parts of THIS pass over the window (1 when the finisher repeats a split window alone with the exact unit)
		// PARTIAL: the list the record belongs to (block-uniform scalar walk over at most kDescSegs lists) and its place in it
		int sg = 0;
		if (PARTIAL)
			while (sg + 1 < pp.nseg && kpos >= pp.seg[sg + 1].first) sg++;
		const DevKp *__restrict__ kpb = PARTIAL ? pp.seg[sg].recs : kps;
		const unsigned k = PARTIAL ? kpos - pp.seg[sg].first : (unsigned)order[kpos * pw + pr];  // processing order: big windows first (k_slots)
		int *const p_hist = PARTIAL ? pp.seg[sg].hist + (size_t)k * kDesc : nullptr;
		float *const p_mass = PARTIAL ? pp.seg[sg].mass + k : nullptr;
		const int slot = kpb[k].slot;                        // row of the keypoint in the results (reference order)
		const int cxi = kpb[k].x, cyi = kpb[k].y, czi = kpb[k].z;
		const int li = __builtin_amdgcn_readfirstlane(kpb[k].octave * 8 + kpb[k].level);  // (block-uniform: a scalar, like the loop-carried cur_lut it is compared with)
		const float scale = kpb[k].scale;
		const LevelRef L = levels[li];
		const WinLut lut = luts[li * 2 + 1];
		// R <- R^T (Transpose_Matrix, Src/cSIFT3D.cc:1214)
		const float R0 = kpb[k].rot[0], R1 = kpb[k].rot[3], R2 = kpb[k].rot[6];
		const float R3 = kpb[k].rot[1], R4 = kpb[k].rot[4], R5 = kpb[k].rot[7];
		const float R6 = kpb[k].rot[2], R7 = kpb[k].rot[5], R8 = kpb[k].rot[8];
		// window constants, Src/cSIFT3D.cc:1155-1159
		const float sigma = scale * 7.071067812f;
		const float win_radius = 2.0f * sigma;
		const float desc_hw = (float)((double)win_radius / sqrt(2.0));
		const float desc_width = 2.0f * desc_hw;
		const float bin_fctr = __fdiv_rn(4.0f, desc_width);
		const float u = L.unit;
		// Fixed-point unit of the 32-bit histogram, per keypoint: every bin sum is bounded by the gradient mass M = sum of |g| w over
		// the window, so a unit 2^-k with M 2^k < 2^31 cannot overflow.  M is not known yet: the first pass uses an estimate (rms
		// gradient of the orientation window, from the structure tensor, times the descriptor window's weight sum, with 4x head
		// room), sums the true M on the way and, should the estimate have been too small, the keypoint is redone once with the
		// exact bound.  WinLut::fix_scale (provable for ANY data of this window size) is the coarsest unit ever used.
		int x0, x1, y0, y1, z0, z1;
		win_bounds_d((float)cxi, win_radius, u, L.nx, x0, x1);
		win_bounds_d((float)cyi, win_radius, u, L.ny, y0, y1);
		win_bounds_d((float)czi, win_radius, u, L.nz, z0, z1);
		float fix_scale = first_pass_unit(kpb[k], luts[li * 2], lut, x1 - x0 + 1, y1 - y0 + 1, z1 - z0 + 1, dev_flags);
		if (PARTIAL && pp.seg[sg].units != nullptr && pp.seg[sg].units[k] > 0.0f) fix_scale = pp.seg[sg].units[k];  // second round: the exact unit, from the owner
		if (PARTIAL) {
			// this rank's z part of the window (the chords below are clipped to it; the planes z0 - 1 and z1 + 1 the gradient reads lie in
			// the buffer's halo).  Nothing of the window here: zeros, and on to the next record (block-uniform).
			const int o0 = pp.seg[sg].o0, o1 = pp.seg[sg].o1, H = pp.H[kpb[k].level & 7];
			int c0, c1;  // [c0, c1)
			if (o0 == pp.zc0 && o1 == pp.zc1) { c0 = o0 - H; c1 = o1 + H; }                 // the owner's part
			else if (pp.zc1 <= o0) { c0 = pp.zc0; c1 = min(pp.zc1, o0 - H); }               // a rank below the owner
			else { c0 = max(pp.zc0, o1 + H); c1 = pp.zc1; }                                  // a rank above
			z0 = max(z0, c0); z1 = min(z1, c1 - 1);
			if (z1 < z0) {
				for (int e = tid; e < kDesc; e += NT) p_hist[e] = 0;
				if (tid == 0) *p_mass = 0.0f;
				continue;
			}
		}
		const int wx = x1 - x0 + 1, wy = y1 - y0 + 1;
		const int ncol = (wx > 0 && wy > 0) ? wx * wy : 0;
		const int sy = L.nx, sz = L.nx * L.ny;  // levels are < 2^31 voxels
		const int nin = lut.nin;                 // largest integer squared offset inside the sphere
		const gfloat_p Ld = as_global(L.d);  // global_load instead of flat_load: in-order vmcnt, loads stay in flight
		const gfloat_p centre = Ld + (size_t)cxi + (size_t)sy * (size_t)cyi + (size_t)sz * (size_t)(czi - L.zoff);  // always valid

		bool finished = false;
		for (int attempt = 0;; attempt++) {  // block-uniform; a second pass only when the first unit was too fine
		float msum = 0.0f;  // this lane's share of the gradient mass
		__syncthreads();  // previous keypoint / pass finished with hist / s_lut
		for (int i = tid; i < kBins * kRep; i += NT) hist[i] = 0;
		if (tid == 0) s_tile = 0u;
		if (LUT_LDS && cur_lut != li) {
			for (int i = tid; i < lut.len && i < kMaxDescLut; i += NT) s_lut[i] = lutpool[lut.off + i];
			cur_lut = li;
		}
		const gfloat_p lut_g = as_global(lutpool) + lut.off;
		__syncthreads();

		int qhead = 0, qcount = 0;  // wave-uniform (every lane executes every push / pop below)
		// All control flow from here to the drain is wave-uniform: lanes without work are predicated, never branched
		// away, because the queue bookkeeping must see every ballot.
		// Two adjacent columns (x, x+1) per lane: the stencil of both comes from one 16-byte row piece (x-1 .. x+2) that also carries
		// the centre values, requested two planes ahead, plus the rows y-1 and y+1 (from the lanes beside this one, or an 8-byte load
		// for the outer rows of a unit; see "Sorted units" above).  (One column per lane: 4 vector-memory instructions per voxel, 7.2 ms.)
		// chord of the lane's two columns (xa, xa + 1) of window row ly: in-sphere range clipped to the rotated 4x4x4 cube
		const float rr3[3] = {R2 * u, R5 * u, R8 * u};  // z step of the three rotated coordinates (per keypoint)
		const float qxk = rr3[0] * bin_fctr, qyk = rr3[1] * bin_fctr, qzk = rr3[2] * bin_fctr;  // ... of the three cell coordinates
		const CellGeom cg = {desc_hw, bin_fctr, desc_width * 0.25f, u, __frcp_rn(u), (dev_flags & 2) ? 1.0e30f : kCellBand};
		const float rr3_inv[3] = {__frcp_rn(rr3[0]), __frcp_rn(rr3[1]), __frcp_rn(rr3[2])};
		auto setup_pair = [&](int lxa, int ly, bool lane_ok, int (&rr)[2], int (&za)[2], int (&zb)[2], float (&px)[2], float (&py)[2],
		                      float (&pz)[2], bool (&colok)[2]) {
			const int dy = y0 + ly - cyi;
			const float vyd = (float)dy * u;
#pragma unroll
			for (int k = 0; k < 2; k++) {
				const int dx = x0 + lxa + k - cxi;
				rr[k] = dx * dx + dy * dy;
				colok[k] = lane_ok && lxa + k < wx && ly < wy && rr[k] <= nin;
				const float vxd = (float)dx * u;
				// partial rotations: (R0*vx + R1*vy) is evaluated first in the reference's left-to-right sums
				px[k] = R0 * vxd + R1 * vyd; py[k] = R3 * vxd + R4 * vyd; pz[k] = R6 * vxd + R7 * vyd;
				za[k] = 0; zb[k] = -1;
				if (colok[k]) {
					// in-sphere chord: dz^2 <= nin - rr
					int h = (int)__fsqrt_rn((float)(nin - rr[k]));  // within 1 of the integer square root (nin < 2^23): one correction each way
					h += (h + 1) * (h + 1) <= nin - rr[k] ? 1 : 0;
					h -= h * h > nin - rr[k] ? 1 : 0;
					za[k] = max(z0, czi - h); zb[k] = min(z1, czi + h);
					// clip the z range to the rotated 4x4x4 cube (iteration-count optimisation only, widened by kClipMargin voxels:
					// the reference's exact fp32 test still runs on every visited voxel)
					float lo = (float)(za[k] - czi), hi = (float)(zb[k] - czi);
					const float pr[3] = {px[k], py[k], pz[k]};
#pragma unroll
					for (int r = 0; r < 3; r++) {
						if (fabsf(rr3[r]) > 1e-6f * desc_hw) {
							const float inv = rr3_inv[r];
							const float t0 = (-desc_hw - pr[r]) * inv, t1 = (desc_hw - pr[r]) * inv;
							lo = fmaxf(lo, fminf(t0, t1) - kClipMargin);
							hi = fminf(hi, fmaxf(t0, t1) + kClipMargin);
						} else if (fabsf(pr[r]) > desc_hw * 1.001f + 1.0f) {
							hi = lo - 1.0f;  // this row never enters the cube
						}
					}
					if (lo <= hi) { za[k] = max(za[k], czi + (int)floorf(lo)); zb[k] = min(zb[k], czi + (int)ceilf(hi)); }
					else zb[k] = za[k] - 1;
				}
				if (!(colok[k] && zb[k] >= za[k])) { colok[k] = false; za[k] = 1 << 28; zb[k] = -(1 << 28); }  // empty column
			}
		};
		const int nux = ((wx + 1) / 2 + kPX - 1) / kPX, nuy = (wy + kSH - 1) / kSH;  // units per unit row, unit rows
		// (z ranges beyond a byte: only with windows far larger than the default parameters')
		const bool chord_cached = z1 - z0 < 255 && !(dev_flags & 1);
		constexpr int kChunkUnits = kPairCap / kUL;
		const int nunits = ncol > 0 ? nux * nuy : 0;
		for (int u0 = 0; u0 < nunits; u0 += kChunkUnits) {  // chunks of units in row-major order
		const int nch = min(nunits - u0, kChunkUnits);
		if (u0 > 0) __syncthreads();                         // previous chunk's march is done with s_units / s_tile
		for (int i = tid; i < kLenBins; i += NT) s_cnt[i] = 0u;
		if (tid == 0) s_tile = 0u;
		__syncthreads();
		int tc = tid;  // (opaque: the first pair slot's unit / position constants are not values to carry through the keypoint loop and the march)
		asm volatile("" : "+v"(tc));
		for (int ps = tc; ps < nch * kUL; ps += NT) {  // pair slot = unit * kUL + position in the unit: kUL consecutive lanes per unit
			const int uu = ps / kUL, spos = ps % kUL;
			const int uyi = (u0 + uu) / nux, uxi = (u0 + uu) - uyi * nux;
			int rr[2], za[2], zb[2];
			float px[2], py[2], pz[2];
			bool colok[2];
			setup_pair((uxi * kPX + spos % kPX) * 2, uyi * kSH + spos / kPX, true, rr, za, zb, px, py, pz, colok);
			int lo = min(za[0], za[1]), hi = max(zb[0], zb[1]);  // empty columns: (2^28, -2^28)
#pragma unroll
			for (int o = kUL / 2; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
			const int len = hi >= lo ? hi - lo + 1 : 0;  // z steps of the strip
			// an empty column is stored as the range (255, 0)
			s_chord[ps] = chord_cached ? (colok[0] ? (unsigned)(za[0] - z0) | (unsigned)(zb[0] - z0) << 8 : 255u) |
			                                 (colok[1] ? (unsigned)(za[1] - z0) | (unsigned)(zb[1] - z0) << 8 : 255u) << 16
			                           : (unsigned)len;
			// (a split window: part p of S marches the units u with u % S == p -- by the unit's index, not by its place in the sorted
			// order, which differs between workgroups where lengths tie)
			if (spos == 0 && len > 0 && (S == 1 || (u0 + uu) % S == part)) atomicAdd(&s_cnt[min(len, kLenBins - 1)], 1u);  // key 0 = empty
		}
		__syncthreads();
		if (wid == 0) {  // running start of every key, longest first
			const unsigned ca = s_cnt[kLenBins - 1 - 2 * lane], cb = s_cnt[kLenBins - 2 - 2 * lane];
			unsigned incl = ca + cb;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const unsigned t = __shfl_up(incl, o, 64);
				if (lane >= o) incl += t;
			}
			s_cnt[kLenBins - 1 - 2 * lane] = incl - ca - cb;
			s_cnt[kLenBins - 2 - 2 * lane] = incl - cb;
			if (lane == 63) s_nnz = incl - cb;  // key 0 comes last: everything before it is non-empty
		}
		__syncthreads();
		for (int uu = tc; uu < nch; uu += NT) {
			int len;
			if (chord_cached) {
				int lo = 255, hi = 0;
#pragma unroll
				for (int r = 0; r < kUL; r++) {
					const unsigned ch = s_chord[uu * kUL + r];
					// (255, 0) never wins a min / max against a real range
					lo = min(lo, (int)min(ch & 255, (ch >> 16) & 255)); hi = max(hi, (int)max((ch >> 8) & 255, ch >> 24));
				}
				len = hi - lo + 1;  // all empty: < 0
			} else {
				len = (int)s_chord[uu * kUL];
			}
			if (len > 0 && (S == 1 || (u0 + uu) % S == part)) s_units[atomicAdd(&s_cnt[min(len, kLenBins - 1)], 1u)] = (unsigned short)uu;
		}
		__syncthreads();
		const int nnz = (int)s_nnz, ntiles = (nnz * kUL + 63) / 64;
		// groups of units (or tiles) are handed to the four waves through an LDS counter: they differ several-fold in work, so
		// a static deal leaves waves idle at the barrier that closes the keypoint
		for (;;) {
			int tile = 0;
			if (lane == 0) tile = (int)atomicAdd(&s_tile, 1u);
			tile = __builtin_amdgcn_readfirstlane(tile);  // wave-uniform
			if (tile >= ntiles) break;
			const int uidx = tile * (64 / kUL) + lane / kUL, spos = lane % kUL;
			const bool lane_ok = uidx < nnz;
			const int uu = s_units[lane_ok ? uidx : 0];
			const int uyi = (u0 + uu) / nux, uxi = (u0 + uu) - uyi * nux;
			const int lxa = (uxi * kPX + spos % kPX) * 2, ly = uyi * kSH + spos / kPX;
			const int xa = x0 + lxa, y = y0 + ly;
			int rr[2], za[2], zb[2];
			float px[2], py[2], pz[2];
			bool colok[2];
			if (chord_cached) {  // block-uniform
				const unsigned ch = s_chord[uu * kUL + spos];
				const int dy = y - cyi;
				const float vyd = (float)dy * u;
#pragma unroll
				for (int k = 0; k < 2; k++) {
					const int dx = xa + k - cxi;
					rr[k] = dx * dx + dy * dy;
					const float vxd = (float)dx * u;
					px[k] = R0 * vxd + R1 * vyd; py[k] = R3 * vxd + R4 * vyd; pz[k] = R6 * vxd + R7 * vyd;
					const int a = (ch >> (16 * k)) & 255, b = (ch >> (16 * k + 8)) & 255;
					colok[k] = lane_ok && b >= a;
					za[k] = colok[k] ? z0 + a : 1 << 28; zb[k] = colok[k] ? z0 + b : -(1 << 28);
				}
			} else
				setup_pair(lxa, ly, lane_ok, rr, za, zb, px, py, pz, colok);
			// every lane of a strip marches the strip's z range: it holds the y neighbours of the lanes beside it, also where its own
			// columns are outside the sphere.  Row wy (one past the window) is such a provider; rows beyond it are nobody's neighbour.
			int zA = min(za[0], za[1]), zB = max(zb[0], zb[1]);
#pragma unroll
			for (int o = kUL / 2; o > 0; o >>= 1) { zA = min(zA, __shfl_xor(zA, o, 64)); zB = max(zB, __shfl_xor(zB, o, 64)); }
			const int zlen = (lane_ok && zB >= zA && ly <= wy && lxa < wx) ? zB - zA + 1 : 0;
			const bool top = spos / kPX == 0, bot = spos / kPX == kSH - 1;
			// the strip's first lane loads row y-1, its last lane row y+1 (when its own row is inside the window: y+1 <= ny-1)
			const ptrdiff_t e_off = top ? -(ptrdiff_t)sy : (bot && ly < wy ? (ptrdiff_t)sy : (ptrdiff_t)0);
			int maxlen = zlen;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
			maxlen = __builtin_amdgcn_readfirstlane(maxlen);
			if (maxlen == 0) continue;  // wave-uniform
			// lanes without a column march on the keypoint's own column (always in bounds) and are masked.  Column a of a lane
			// satisfies 1 <= x <= nx-2 whenever one of its columns is valid, so x-1 .. x+2 stays inside the level (x+2 = nx is the
			// first element of the next row, and y <= ny-2; the provider row of a strip may be y = ny-1: its x+2 can then be the
			// first element of the next plane -- of the next LEVEL of the arena for a window in the far corner; keypoint levels are
			// never the last level of the arena).
			gfloat_p c = zlen > 0 ? Ld + (size_t)xa + (size_t)sy * (size_t)y + (size_t)sz * (size_t)(zA - L.zoff) : centre;
			// cell coordinate - 1.5 of column k at plane z: c = cx0[k] + (z - czi) * qx (one fused multiply-add per axis and voxel)
			float cx0[2], cy0[2], cz0[2];
			unsigned zlo[2], zspan[2];
#pragma unroll
			for (int k = 0; k < 2; k++) {
				cx0[k] = (px[k] + desc_hw) * bin_fctr - 2.0f; cy0[k] = (py[k] + desc_hw) * bin_fctr - 2.0f; cz0[k] = (pz[k] + desc_hw) * bin_fctr - 2.0f;
				zlo[k] = (unsigned)(za[k] - zA); zspan[k] = (unsigned)(zb[k] - za[k]);  // empty column (2^28, -2^28): never inside
			}
			// one march step on the planes held in registers: rowC = the row piece of plane z (x-1, a, b, x+2), rowN = of plane z+1 (its centres),
			// cmv = the centres of plane z-1, edC = the unit's outer row of plane z (first / last row of the unit only); dz = z - czi
			auto step_body = [&](const int step, const int dz, const f4g &rowC, const f4g &rowN, const f2g &cmv, const f2g &edC) {
				// y neighbours of plane z: the centre values of the lanes beside this one (same unit, same plane)
				f2g dn, up;
				dpp_neighbours<kPX>(rowC.y, rowC.z, dn, up);
				const f2g ymC = top ? edC : dn, ypC = bot ? edC : up;
				const int dz2 = __mul24(dz, dz);  // |dz| < 2^11 (full-rate 24-bit multiply; v_mul_lo_u32 issues at quarter rate)
				const float dzf = (float)dz;
				float bxk[2], byk[2], bzk[2], rxk[2], ryk[2], rzk[2];
				bool actk[2];
#pragma unroll
				for (int k = 0; k < 2; k++) {
					// (z - za <= zb - za as unsigned: inside the column's chord, which lies inside the unit's range: step < zlen is implied)
					const bool in = (unsigned)((unsigned)step - zlo[k]) <= zspan[k];
					const float bx = __fmaf_rn(dzf, qxk, cx0[k]), by = __fmaf_rn(dzf, qyk, cy0[k]), bz = __fmaf_rn(dzf, qzk, cz0[k]);  // b - 1.5
					// inside the cube or within the band the heavy part decides exactly (accumulate_voxel)
					const bool act = ((int)in & (int)(fmaxf(fmaxf(fabsf(bx), fabsf(by)), fabsf(bz)) < 2.0f + kCellBand)) != 0;
					const float w = LUT_LDS ? s_lut[in ? rr[k] + dz2 : 0] : lut_g[in ? rr[k] + dz2 : 0];
					const float nxm = k ? rowC.y : rowC.x, nxp = k ? rowC.w : rowC.z;
					const float nym = k ? ymC.y : ymC.x, nyp = k ? ypC.y : ypC.x;
					const float cp = k ? rowN.z : rowN.y, cm = k ? cmv.y : cmv.x;
					float gx = nxp - nxm;
					float gy = nyp - nym;
					float gz = cp - cm;
					// w carries the reference's 0.5 and 1/u (exact power-of-two scalings, WinLut).  The rotation of the gradient and the
					// exact |R g|^2 >= eps test run on the compacted voxels; here a voxel is only dropped when it fails by a margin
					gx = gx * w; gy = gy * w; gz = gz * w;
					const float g2 = gx * gx + gy * gy + gz * gz;
					actk[k] = ((int)act & (int)!(g2 < kBaryEps * 0.99f)) != 0;
					bxk[k] = bx; byk[k] = by; bzk[k] = bz; rxk[k] = gx; ryk[k] = gy; rzk[k] = gz;
				}
#pragma unroll
				for (int k = 0; k < 2; k++) {
					// ---- push the active lanes into the wave's queue (compaction by ballot rank) ----
					const unsigned long long m = __ballot(actk[k]);
					if (m) {
						if (actk[k]) {
							const int pos = (qhead + qcount + (int)__popcll(m & ((1ull << lane) - 1ull))) & (kQCap - 1);
							qa[pos] = qf4{bxk[k], byk[k], bzk[k], rxk[k]}; qb[pos] = qf2{ryk[k], rzk[k]};
						}
						qcount += (int)__popcll(m);
					}
					// ---- a full wave of active voxels is ready: run the heavy part on all 64 lanes ----
					if (qcount >= 64) {
						const int pos = (qhead + lane) & (kQCap - 1);
						// the march runs at raised wave priority, the heavy part at the base priority (r05: a wave that is about to request its
						// next planes goes first on the SIMD; 3.19 -> 3.15 ms, the other way round 3.26)
						__builtin_amdgcn_s_setprio(0);
						const qf4 ea = qa[pos]; const qf2 eb = qb[pos];
						msum += accumulate_voxel(true, ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, R0, R1, R2, R3, R4, R5, R6, R7, R8, fix_scale, s_fidx, s_sym, hist_rep, spread, cg);
						__builtin_amdgcn_s_setprio(2);
						qhead = (qhead + 64) & (kQCap - 1);
						qcount -= 64;
					}
				}
			};
			{
				// The ring ("A ring of planes in flight" above).  Rotate at the TOP of a step: before step s the registers hold plane s-1 in rowC
				// -- of which only the centres are used -- and plane s in rowN; slot s % 2 holds plane s+1 and the outer row of plane s.
				// Every load of the loop and of its prologue is untracked: a tracked one pending at the loop's head makes hipcc wait with
				// vmcnt(0) inside the loop, which would drain the ring at every step.
				const unsigned long long emask = __ballot(top || bot);
				f4g rowC, rowN = f4g{0.f, 0.f, 0.f, 0.f};
				f2g cmv = f2g{0.f, 0.f}, edC = f2g{0.f, 0.f}, cm0 = f2g{0.f, 0.f};
				asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off offset:-4" : "+v"(cm0), "+v"(rowN) : "v"(c - sz), "v"(c) : "memory");  // centres of plane zA - 1, plane zA
				f4g P0, P1;  // (first written by the prologue's loads; an outer-row slot is only defined in the lanes that own an outer row)
				f2g E0, E1;
				gcf_p pf = c;  // plane of the youngest row piece requested: min(step + 2, zlen) planes above zA (zlen: the plane zB + 1, the last one read)
				{
					gcf_p pe = pf + e_off;
					pf += 1 <= zlen ? sz : 0;
					S3D_RING_LOAD(P0, E0, "v[112:115]", "v[116:117]", pf, pe, emask);
					pe = pf + e_off;
					pf += 2 <= zlen ? sz : 0;
					S3D_RING_LOAD(P1, E1, "v[120:123]", "v[118:119]", pf, pe, emask);
				}
				asm volatile("s_waitcnt vmcnt(4)" : "+v"(cm0), "+v"(rowN));  // the two prologue loads; the ring's four stay in flight
				rowC = f4g{0.f, cm0.x, cm0.y, 0.f};
				const int dz0 = zA - czi;
				// a turn: the slot (plane step + 1, outer row of plane step) is copied out, then plane step + 3 and the outer row of plane step + 2 are
				// requested into it (clamped to the planes of the unit's range; a clamped outer-row address belongs to a step beyond the range, whose
				// value nobody uses)
#define S3D_RING_STEP(STEP, P, E, PREG, EREG, P0_, P1_, P2_, P3_, E0_, E1_)                                             \
				{                                                                                                       \
					const int st_ = (STEP);                                                                             \
					cmv = f2g{rowC.y, rowC.z}; rowC = rowN;                                                             \
					const gcf_p pe_ = pf + e_off; /* (outer row of the plane BELOW the one requested next) */          \
					pf += st_ + 3 <= zlen ? sz : 0;                                                                     \
					S3D_RING_TURN(2, P, E, PREG, EREG, P0_, P1_, P2_, P3_, E0_, E1_, rowN, edC, pf, pe_, emask);        \
					step_body(st_, dz0 + st_, rowC, rowN, cmv, edC);                                                   \
				}
				__builtin_amdgcn_s_setprio(2);
				for (int step = 0; step < maxlen; step += 2) {
					S3D_RING_STEP(step, P0, E0, "v[112:115]", "v[116:117]", "v112", "v113", "v114", "v115", "v116", "v117")
					if (step + 1 < maxlen) S3D_RING_STEP(step + 1, P1, E1, "v[120:123]", "v[118:119]", "v120", "v121", "v122", "v123", "v118", "v119")
				}
#undef S3D_RING_STEP
				// nothing may still be in flight into the ring when its registers are given to somebody else
				asm volatile("s_waitcnt vmcnt(0)" : "+{v[112:115]}"(P0), "+{v[120:123]}"(P1), "+{v[116:117]}"(E0), "+{v[118:119]}"(E1));
				__builtin_amdgcn_s_setprio(0);
			}
		}
		}  // chunk of units
		if (qcount > 0) {  // drain (wave-uniform)
			const int pos = (qhead + lane) & (kQCap - 1);
			const bool valid = lane < qcount;
			const qf4 ea = qa[pos]; const qf2 eb = qb[pos];
			msum += accumulate_voxel(valid, ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, R0, R1, R2, R3, R4, R5, R6, R7, R8, fix_scale, s_fidx, s_sym, hist_rep, spread, cg);
		}
		// gradient mass of the window (block sum; fp32 sums of non-negative terms, 1e-4 relative at worst: covered by the margins)
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) msum = msum + __shfl_xor(msum, o, 64);
		__syncthreads();
		{
			int tm = tid;  // (opaque: the address of this wave's slot is not a value to carry through the march)
			asm volatile("" : "+v"(tm));
			if ((tm & 63) == 0) red[tm >> 6] = msum;
		}
		__syncthreads();
		float mass_sum = (red[0] + red[1]) + (red[2] + red[3]);
		if (NW == 8) mass_sum = mass_sum + ((red[4] + red[5]) + (red[6] + red[7]));
		if (PARTIAL) {
			// (red[] was read by every thread above; the LDS histogram is complete: the barriers of the mass reduction lie behind every wave's drain)
			int tp = tid;  // (opaque: keeps the bin indices out of the registers that live through the march, like `te` below)
			asm volatile("" : "+v"(tp));
#pragma unroll
			for (int j = 0; j < 3; j++) {
				const int e = tp + 256 * j;
				int a = 0;
#pragma unroll
				for (int r = 0; r < kRep; r++) a += (int)(sbin_t)hist[bin_index(e) * kRep + (r + tp) % kRep];
				p_hist[e] = a;
			}
			if (tid == 0) *p_mass = mass_sum;
			break;  // (leaves the attempt loop with finished == false: the owner decides about a second round)
		}
		// (r05) the thread index of the epilogue is opaque: its lane constants (bin indices, result addresses: ~20 registers) were hoisted
		// out of the KEYPOINT loop and lived through the march, which is where the kernel's registers are short (9 spilled VGPRs, 128 used)
		int te = tid;
		asm volatile("" : "+v"(te));
		if (S > 1) {
			// this part's integer histogram (replicas summed) and its share of the gradient mass go to the keypoint's accumulators in
			// global memory; the part that arrives last carries on as the keypoint's finisher
#pragma unroll
			for (int j = 0; j < 3; j++) {
				const int e = te + 256 * j;
				int a = 0;
#pragma unroll
				for (int r = 0; r < kRep; r++) a += (int)(sbin_t)hist[bin_index(e) * kRep + (r + te) % kRep];
				if (a != 0) atomicAdd(&sp.gacc[(size_t)kpos * kDesc + e], a);
			}
			if (tid == 0) __hip_atomic_store(&sp.gmass[kpos * 8 + part], mass_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			// arrival: the payload is device-scope atomics and one device-scope store (performed at the memory side, not in this XCD's L2);
			// every wave waits for its own operations to be acknowledged, then ONE lane counts the part in with an acquire-release
			// read-modify-write at agent scope (the relaxed counter relied on that payload never sitting in a cache; the
			// one-lane form is formally ordered and costs a few microseconds per part.  __threadfence() by all 256 threads here and once
			// more in the finisher cost 40-100 us per part: the split ran 2.3x slower than no split.)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__syncthreads();
			if (tid == 0) s_last = __hip_atomic_fetch_add(&sp.gdone[kpos], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
			__syncthreads();
			if (s_last != (unsigned)(S - 1)) break;  // block-uniform: another part finishes the keypoint (leaves the attempt loop; see below)
			mass_sum = 0.0f;
			for (int q = 0; q < S; q++) mass_sum = mass_sum + __hip_atomic_load(&sp.gmass[kpos * 8 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // part order: deterministic
		}
		const float mass = mass_sum * 1.001f;
		if (attempt == 1 || !unit_fails(mass, fix_scale, lut.fix_scale, lut.nin)) {
			finished = true;
			break;
		}
		if (S > 1) {
			// a split window whose first unit failed: its finisher repeats the window ALONE with the exact unit, like an unsplit run's
			// second pass (inline: a separate launch for these few keypoints ended the stage 0.2 ms later); the accumulators are cleared
			// like those of a finished window
			for (int e = tid; e < kDesc; e += NT) sp.gacc[(size_t)kpos * kDesc + e] = 0;
			if (tid == 0) sp.gdone[kpos] = 0u;
			S = 1;
		}
		if (tid == 0) atomicAdd(d_work + 1, 1u);
		fix_scale = pick_scale_d(mass, lut.fix_scale);  // exact bound: this pass cannot overflow
		}
		if (!finished) continue;  // block-uniform: not this workgroup's keypoint to finish (a split part, or sent to the redo list)
		int te = tid;  // (opaque, see above)
		asm volatile("" : "+v"(te));
		const double fix_inv = 1.0 / (double)fix_scale;
		__syncthreads();

		float v0 = 0.f, v1 = 0.f, v2 = 0.f;
		if (NT == 256) {
			long long a0 = 0, a1 = 0, a2 = 0;
			if (S > 1) {  // the finisher of a split window: the sums of all parts (the same integers an unsplit run holds in its LDS histogram)
				int *ga = sp.gacc + (size_t)kpos * kDesc;
				a0 = (long long)__hip_atomic_load(&ga[te], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				a1 = (long long)__hip_atomic_load(&ga[te + 256], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				a2 = (long long)__hip_atomic_load(&ga[te + 512], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				ga[te] = 0; ga[te + 256] = 0; ga[te + 512] = 0;  // clean for the next run
				if (tid == 0) sp.gdone[kpos] = 0u;
			} else {
#pragma unroll
			for (int r = 0; r < kRep; r++) {
				const int rr2 = (r + te) % kRep;  // stagger the replica order across lanes
				a0 += (long long)(sbin_t)hist[bin_index(te) * kRep + rr2];
				a1 += (long long)(sbin_t)hist[bin_index(te + 256) * kRep + rr2];
				a2 += (long long)(sbin_t)hist[bin_index(te + 512) * kRep + rr2];
			}
			}
			v0 = (float)((double)a0 * fix_inv); v1 = (float)((double)a1 * fix_inv); v2 = (float)((double)a2 * fix_inv);
		} else {
			// every thread converts its bins (exact integer sums), the first 256 threads then normalise exactly like the 256-thread variant
			float *vbuf = &s_q[0][0][0];  // (the queues are idle: every wave has drained)
			for (int e = te; e < kDesc; e += NT) {
				long long a = 0;
#pragma unroll
				for (int r = 0; r < kRep; r++) a += (long long)(sbin_t)hist[bin_index(e) * kRep + (r + e) % kRep];
				vbuf[e] = (float)((double)a * fix_inv);
			}
			__syncthreads();
			if (te < 256) { v0 = vbuf[te]; v1 = vbuf[te + 256]; v2 = vbuf[te + 512]; }
		}
		normalise_store(v0, v1, v2, te, lane, wid, red, d_desc + (size_t)slot * kDesc);

	}
}

// sift3d_debug_face_lookup: the face lookup of k_describe on caller-provided gradients (unit parity against golden g7)
__global__ void __launch_bounds__(256) k_face_lookup(const float *__restrict__ g3, int n, int route, int *__restrict__ face, float *__restrict__ bary3) {
	__shared__ int4 s_sym[32];
	__shared__ int s_fidx[kFaces * 4];
	stage_face_tables(threadIdx.x, s_fidx, s_sym);
	__syncthreads();
	for (int base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {  // wave-uniform trip count (face_lookup votes)
		const int i = base + threadIdx.x;
		const bool valid = i < n;
		const float gx = valid ? g3[3 * i] : 1.f, gy = valid ? g3[3 * i + 1] : 0.f, gz = valid ? g3[3 * i + 2] : 0.f;
		float b[3] = {0.f, 0.f, 0.f};
		int o0, o1, o2, pk = 0;
		// |g|^2 < eps rejects first, like Check_intersect_faces (Src/cSIFT3D.cc:1546) and accumulate_voxel
		const float g2 = gx * gx + gy * gy + gz * gz;
		const int f = face_lookup(valid && !(g2 < kBaryEps), gx, gy, gz, s_fidx, s_sym, b[0], b[1], b[2], o0, o1, o2, &pk, route != 0);
		if (valid) {
			face[i] = f;
			// weights back into the reference's order bary[0..2] (= v0, v1, v2 of the face after the winding fix)
#pragma unroll
			for (int r = 0; r < 3; r++) bary3[3 * i + ((pk >> (8 + 2 * r)) & 3)] = f < 0 ? 0.0f : b[r];
		}
	}
}
void launch_face_lookup(const float *d_g3, int n, int route, int *d_face, float *d_bary3, hipStream_t st) {
	hipLaunchKernelGGL(k_face_lookup, dim3(std::max(1, std::min((n + 255) / 256, 1024))), dim3(256), 0, st, d_g3, n, route, d_face, d_bary3);
}

void launch_describe(const DevKp *kps, const unsigned *d_count, unsigned cap, const LevelRef *d_levels, const WinLut *d_luts,
                     const float *d_lutpool, float *d_desc, unsigned kp_cap, int part_rank, int part_world, const int *order,
                     const unsigned *d_nkp, unsigned *d_work, hipStream_t st, bool lut_in_lds, const DescSplit *split) {
	(void)hipMemsetAsync(d_work, 0, 2 * sizeof(unsigned), st);
	DescSplit sp{};
	if (split && split->gacc && lut_in_lds && !hook(SIFT3D_HOOK_DESC_NOSPLIT)) sp = *split;
	static const int desc_grid = dev_tune_i("S3D_DESC_GRID", 256 * 8);  // persistent workgroups (work counter)
	const int dev_flags = (hook(SIFT3D_HOOK_DESC_NOCACHE) ? 1 : 0) | (hook(SIFT3D_HOOK_DESC_EXACT_CELLS) ? 2 : 0) | (hook(SIFT3D_HOOK_DESC_MASS_SHIFT) & 63) << 8;
	// few keypoints: eight waves per keypoint (see k_describe); the count lives on the device, so both variants are launched
	// keypoint counts [wide_lo, wide_hi) take the eight-wave variant of r03; below wide_lo a window is split over 8 / 4 four-wave
	// workgroups (r04; scripts/split_probe.py: 0.19 / 0.22 / 0.28 / 0.32 ms for 40 / 151 / 286 / 437 keypoints against 0.25 / 0.32 / 0.38 /
	// 0.41 with eight waves; from ~700 keypoints the eight-wave variant is ahead: 0.49 vs 0.51-0.58 ms at 1096)
	const unsigned wide_hi = kWideBelow, wide_lo = sp.gacc ? std::min(kSplit4Below, wide_hi) : 0u;
	if (lut_in_lds) {
		hipLaunchKernelGGL((k_describe<true, 256>), dim3(desc_grid), dim3(256), 0, st, kps, d_count, cap, d_levels, d_luts, d_lutpool, d_desc, kp_cap,
		                   part_rank, part_world, order, d_nkp, d_work, dev_flags, wide_lo, wide_hi, sp, DescPartial{});
		if (wide_hi > wide_lo)
			hipLaunchKernelGGL((k_describe<true, 512>), dim3(256 * 2), dim3(512), 0, st, kps, d_count, cap, d_levels, d_luts, d_lutpool, d_desc, kp_cap,
			                   part_rank, part_world, order, d_nkp, d_work, dev_flags, wide_lo, wide_hi, DescSplit{}, DescPartial{});
	} else {
		hipLaunchKernelGGL((k_describe<false, 256>), dim3(256 * 8), dim3(256), 0, st, kps, d_count, cap, d_levels, d_luts, d_lutpool, d_desc, kp_cap,
		                   part_rank, part_world, order, d_nkp, d_work, dev_flags, 0u, 0u, DescSplit{}, DescPartial{});
	}
}

// ---- r05: descriptor windows split along z over the GPUs of a node (DescPartial; SURVEY 8e, no reference counterpart) ----
// The owner of a keypoint adds the parts' integer histograms (its own and its z-neighbours': plain int32 sums) and their gradient masses
// in the order the parts are given (ascending rank), and finishes: the unit the parts used either stands -- normalise, clamp, normalise,
// store the row (the same arithmetic as k_describe's epilogue: normalise_store) -- or fails the test of k_describe's first pass, in which
// case the record is flagged for a second round with the exact unit (units_next) unless this already is that round.
struct FinishParts {
	int nparts = 0;
	const int *hist[kDescSegs] = {};
	const float *mass[kDescSegs] = {};
};
__global__ void __launch_bounds__(256) k_describe_finish(const DevKp *__restrict__ recs, unsigned n, const LevelRef *__restrict__ levels, const WinLut *__restrict__ luts, FinishParts fp,
                                                          const float *__restrict__ units, int final_round, int dev_flags,
                                                          float *__restrict__ d_desc, int *__restrict__ redo, float *__restrict__ units_next,
                                                          unsigned *__restrict__ d_counters) {
	__shared__ float red[4];
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	for (unsigned k = blockIdx.x; k < n; k += gridDim.x) {  // block-uniform
		const int li = recs[k].octave * 8 + recs[k].level;
		const WinLut lut = luts[li * 2 + 1];
		const LevelRef L = levels[li];
		int wx0, wx1, wy0, wy1, wz0, wz1;
		win_bounds_d((float)recs[k].x, lut.radius, L.unit, L.nx, wx0, wx1);
		win_bounds_d((float)recs[k].y, lut.radius, L.unit, L.ny, wy0, wy1);
		win_bounds_d((float)recs[k].z, lut.radius, L.unit, L.nz, wz0, wz1);
		float fix_scale = first_pass_unit(recs[k], luts[li * 2], lut, wx1 - wx0 + 1, wy1 - wy0 + 1, wz1 - wz0 + 1, dev_flags);
		if (units != nullptr && units[k] > 0.0f) fix_scale = units[k];
		float msum = 0.0f;
		for (int p = 0; p < fp.nparts; p++) msum = msum + fp.mass[p][k];
		const float m = msum * 1.001f;
		if (!final_round && unit_fails(m, fix_scale, lut.fix_scale, lut.nin)) {
			if (tid == 0) { redo[k] = 1; units_next[k] = pick_scale_d(m, lut.fix_scale); atomicAdd(d_counters, 1u); }
			continue;
		}
		if (tid == 0 && redo != nullptr) redo[k] = 0;
		const double fix_inv = 1.0 / (double)fix_scale;
		long long a0 = 0, a1 = 0, a2 = 0;
		for (int p = 0; p < fp.nparts; p++) {
			const int *h = fp.hist[p] + (size_t)k * kDesc;
			a0 += (long long)h[tid]; a1 += (long long)h[tid + 256]; a2 += (long long)h[tid + 512];
		}
		const float v0 = (float)((double)a0 * fix_inv), v1 = (float)((double)a1 * fix_inv), v2 = (float)((double)a2 * fix_inv);
		normalise_store(v0, v1, v2, tid, lane, wid, red, d_desc + (size_t)recs[k].slot * kDesc);
		__syncthreads();  // red[] is reused by the next record
	}
}

// records of the accepted keypoints in processing order: dst[pos] = ext[order[pos]] (sift3d_slab_export_records)
__global__ void __launch_bounds__(256) k_export_records(const DevKp *__restrict__ ext, const int *__restrict__ order, unsigned n, DevKp *__restrict__ dst) {
	constexpr unsigned W = sizeof(DevKp) / 4;
	static_assert(sizeof(DevKp) % 4 == 0, "records are copied word by word");
	const unsigned total = n * W;
	for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
		const unsigned pos = i / W, w = i - pos * W;
		reinterpret_cast<unsigned *>(dst)[i] = reinterpret_cast<const unsigned *>(ext + order[pos])[w];
	}
}
void launch_export_records(const DevKp *ext, const int *order, unsigned n, DevKp *dst, hipStream_t st) {
	if (n == 0) return;
	const unsigned total = n * (unsigned)(sizeof(DevKp) / 4);
	hipLaunchKernelGGL(k_export_records, dim3((int)std::min((total + 255u) / 256u, 2048u)), dim3(256), 0, st, ext, order, n, dst);
}

void launch_describe_partial(const LevelRef *d_levels, const WinLut *d_luts, const float *d_lutpool, const DescPartial &pp, unsigned *d_work,
                             hipStream_t st, bool lut_in_lds) {
	unsigned n = 0;
	for (int i = 0; i < pp.nseg; i++) n += pp.seg[i].n;
	if (n == 0) return;
	(void)hipMemsetAsync(d_work, 0, 2 * sizeof(unsigned), st);
	const int dev_flags = (hook(SIFT3D_HOOK_DESC_NOCACHE) ? 1 : 0) | (hook(SIFT3D_HOOK_DESC_EXACT_CELLS) ? 2 : 0) | (hook(SIFT3D_HOOK_DESC_MASS_SHIFT) & 63) << 8;
	const int grid = (int)std::min(n, 256u * 8u);
	if (lut_in_lds)
		hipLaunchKernelGGL((k_describe<true, 256, true>), dim3(grid), dim3(256), 0, st, (const DevKp *)nullptr, (const unsigned *)nullptr, 0u, d_levels, d_luts,
		                   d_lutpool, (float *)nullptr, n, 0, 1, (const int *)nullptr, (const unsigned *)nullptr, d_work, dev_flags, 0u, 0u, DescSplit{}, pp);
	else
		hipLaunchKernelGGL((k_describe<false, 256, true>), dim3(grid), dim3(256), 0, st, (const DevKp *)nullptr, (const unsigned *)nullptr, 0u, d_levels, d_luts,
		                   d_lutpool, (float *)nullptr, n, 0, 1, (const int *)nullptr, (const unsigned *)nullptr, d_work, dev_flags, 0u, 0u, DescSplit{}, pp);
}

void launch_describe_finish(const DevKp *recs, unsigned n, const LevelRef *d_levels, const WinLut *d_luts, int nparts, const int *const *d_hist, const float *const *d_mass,
                            const float *d_units, bool final_round, float *d_desc, int *d_redo, float *d_units_next, unsigned *d_counters,
                            hipStream_t st) {
	if (n == 0) return;
	FinishParts fp;
	fp.nparts = nparts;
	for (int p = 0; p < nparts; p++) { fp.hist[p] = d_hist[p]; fp.mass[p] = d_mass[p]; }
	const int dev_flags = (hook(SIFT3D_HOOK_DESC_MASS_SHIFT) & 63) << 8;
	hipLaunchKernelGGL(k_describe_finish, dim3((int)std::min(n, 4096u)), dim3(256), 0, st, recs, n, d_levels, d_luts, fp, d_units, final_round ? 1 : 0, dev_flags,
	                   d_desc, d_redo, d_units_next, d_counters);
}

// final keypoint records (Keypoint fields incl. rx,ry,rz = x*2^octave, Src/cSIFT3D.cc:1377-1379)
__global__ void __launch_bounds__(256) k_finalize(const DevKp *__restrict__ kps, const unsigned *__restrict__ d_count, unsigned cap,
                                                  int transposed, sift3d_keypoint *__restrict__ out, float *__restrict__ xyz,
                                                  unsigned kp_cap) {
	const unsigned count = min(d_count[0], cap);
	for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
		const DevKp kp = kps[k];
		if (kp.slot < 0 || (unsigned)kp.slot >= kp_cap) continue;
		sift3d_keypoint o;
		o.x = (float)kp.x; o.y = (float)kp.y; o.z = (float)kp.z;
		o.scale = kp.scale; o.octave = kp.octave; o.level = kp.level;
		if (transposed) {
			const float cf = (float)(1 << kp.octave);  // pow(2.0, octave)
			o.rx = o.x * cf; o.ry = o.y * cf; o.rz = o.z * cf;
		} else {
			o.rx = o.ry = o.rz = -1.0f;  // Initialize_Keypoint, Src/cUtil.cc:451
		}
#pragma unroll
		for (int j = 0; j < 3; j++) { o.win[j] = kp.win[j]; o.eigvalue[j] = kp.eigvalue[j]; }
#pragma unroll
		for (int j = 0; j < 9; j++) { o.eigvector[j] = kp.eigvector[j]; o.str_tensor[j] = kp.st[j]; }
#pragma unroll
		for (int r = 0; r < 3; r++)
#pragma unroll
			for (int c = 0; c < 3; c++) o.Rotation[3 * r + c] = transposed ? kp.rot[3 * c + r] : kp.rot[3 * r + c];
		out[kp.slot] = o;
		xyz[3 * (size_t)kp.slot + 0] = o.rx; xyz[3 * (size_t)kp.slot + 1] = o.ry; xyz[3 * (size_t)kp.slot + 2] = o.rz;
	}
}

void launch_finalize(const DevKp *kps, const unsigned *d_count, unsigned cap, int transposed, sift3d_keypoint *d_out,
                     float *d_xyz, unsigned kp_cap, hipStream_t st) {
	hipLaunchKernelGGL(k_finalize, dim3(256), dim3(256), 0, st, kps, d_count, cap, transposed, d_out, d_xyz, kp_cap);
}

void preload_desc_kernels() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_finalize)); }  // (see kernels_march.hip)

}  // namespace s3d
