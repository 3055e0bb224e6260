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
//   * a workgroup owns a TX x TY = 64 x 16 column of the volume and marches along z over a chunk
//   * per plane q:  global -> LDS tile with xy halo (coalesced 16-B loads)
//                   x-blur  LDS tile -> LDS (each thread slides a register window over 8 outputs)
//                   y-blur  LDS -> registers (each thread: 4 consecutive y of one x, lanes along x)
//                   the xy-blurred value enters a per-thread REGISTER RING of 2*HW+2 planes
//                   z-blur of plane p = q-HW straight from the ring, DoG against G[i-1](p), stores
//   * block -> tile mapping is XCD aware (neighbouring tiles share their halo in one XCD's L2)
//
// Parity: every output is the literal sum acc = acc + tap[d+hw]*term for d = -hw..+hw with separate
// IEEE multiply and add (no FMA: file built with -ffp-contract=off), interior term = src[p-d],
// boundary term = (1-frac)*src[lo] + frac*src[hi] with the reference's fp32 coordinate rule
// (Src/cSIFT3D.cc:736-764), i.e. bit-identical to the CPU path.  Boundary outputs take a slow path
// (x, y: per-output LDS gathers; z: wave-uniform ring selects) -- they are O(hw/n) of the volume.
#include "sift3d_internal.h"

namespace s3d {

__device__ __forceinline__ float absmax_step_f(float m, float v) {
	const float a = fabsf(v);
	return (a > m) ? a : m;
}

template <int HW>
struct FusedCfg {
	static constexpr int TX = 64, TY = 16, NT = 256;
	static constexpr int HXL = ((HW + 1 + 3) / 4) * 4;  // low-side x halo: the right-boundary rule reaches p-hw-1
	static constexpr int HXH = ((HW + 3) / 4) * 4;
	static constexpr int W = HXL + TX + HXH;            // tile row width (floats, multiple of 4)
	static constexpr int W4 = W / 4;
	static constexpr int PITCH = W + 4;                 // +16 B: rows shift by 4 banks
	static constexpr int ROWS = TY + 2 * HW + 1;        // low halo HW+1, high halo HW
	static constexpr int RING = 2 * HW + 2;             // planes p-hw-1 .. p+hw
	static constexpr int WSTART = (HXL - HW) & ~3;      // 16-B aligned start of a thread's x window
	static constexpr int WOFF = HXL - HW - WSTART;      // window index of input x-hw for output j=0
	static constexpr int WN4 = (WOFF + 8 + 2 * HW + 3) / 4;
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

template <int HW, bool DOG>
__global__ void __launch_bounds__(256, (HW <= 4 ? 4 : (HW <= 6 ? 3 : 2))) k_fused_level(const float *__restrict__ src, float *__restrict__ dst,
                                                     float *__restrict__ dog, unsigned *__restrict__ dogmax, int nx, int ny,
                                                     int nz, Taps t, int ntx, int nty, int nchunks, int cz) {
	using C = FusedCfg<HW>;
	__shared__ __attribute__((aligned(16))) float in_t[C::ROWS * C::PITCH];
	__shared__ __attribute__((aligned(16))) float xb[C::ROWS * C::TX];
	__shared__ float s_red[4];

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
	const int zc0 = chunk * cz, zc1 = min(nz, zc0 + cz);

	const int tid = threadIdx.x, lane = tid & 63, yq = tid >> 6;
	const bool vec_ok = (nx & 3) == 0;
	const bool edge_x = (x0 < HW) || (x0 + C::TX - 1 > nx - 2 - HW);
	const bool edge_y = (y0 < HW) || (y0 + C::TY - 1 > ny - 2 - HW);
	const size_t sy = (size_t)nx, sz = (size_t)nx * ny;

	float ring[4][C::RING];
#pragma unroll
	for (int j = 0; j < 4; j++)
#pragma unroll
		for (int k = 0; k < C::RING; k++) ring[j][k] = 0.0f;
	float mx = 0.0f;

	const int q_begin = zc0 - HW - 1, q_end = zc1 - 1 + HW;  // inclusive
	for (int q = q_begin; q <= q_end; q++) {
		float v[4] = {0.f, 0.f, 0.f, 0.f};
		if (q >= 0 && q < nz) {
			// ---------------- global -> LDS tile (plane q, xy halo) ----------------
			const float *plane = src + sz * (size_t)q;
			for (int item = tid; item < C::ROWS * C::W4; item += C::NT) {
				const int r = item / C::W4, c4 = item - r * C::W4;
				const int gy = y0 - HW - 1 + r, gx = x0 - C::HXL + 4 * c4;
				float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
				if (gy >= 0 && gy < ny) {
					const float *row = plane + sy * (size_t)gy;
					if (vec_ok && gx >= 0 && gx + 3 < nx) {
						val = *reinterpret_cast<const float4 *>(row + gx);
					} else {
						if (gx >= 0 && gx < nx) val.x = row[gx];
						if (gx + 1 >= 0 && gx + 1 < nx) val.y = row[gx + 1];
						if (gx + 2 >= 0 && gx + 2 < nx) val.z = row[gx + 2];
						if (gx + 3 >= 0 && gx + 3 < nx) val.w = row[gx + 3];
					}
				}
				*reinterpret_cast<float4 *>(&in_t[r * C::PITCH + 4 * c4]) = val;
			}
			__syncthreads();
			// ---------------- x-blur: in_t -> xb ----------------
			for (int item = tid; item < C::ROWS * 8; item += C::NT) {
				const int r = item >> 3, seg = item & 7;
				const int gy = y0 - HW - 1 + r;
				if (gy < 0 || gy >= ny) continue;
				const float *trow = &in_t[r * C::PITCH];
				float win[C::WN4 * 4];
#pragma unroll
				for (int k = 0; k < C::WN4; k++) {
					const float4 f = *reinterpret_cast<const float4 *>(trow + C::WSTART + seg * 8 + 4 * k);
					win[4 * k] = f.x; win[4 * k + 1] = f.y; win[4 * k + 2] = f.z; win[4 * k + 3] = f.w;
				}
				float o[8];
#pragma unroll
				for (int j = 0; j < 8; j++) {
					float acc = 0.0f;
#pragma unroll
					for (int d = -HW; d <= HW; d++) acc = acc + t.w[d + HW] * win[C::WOFF + j + HW - d];
					o[j] = acc;
				}
				float4 *xo = reinterpret_cast<float4 *>(&xb[r * C::TX + seg * 8]);
				xo[0] = make_float4(o[0], o[1], o[2], o[3]);
				xo[1] = make_float4(o[4], o[5], o[6], o[7]);
			}
			if (edge_x) {
				// slow path: boundary columns of this tile are recomputed with the reference's mirror / lerp rule
				__syncthreads();
				const int nleft = (x0 < HW) ? min(HW, nx) : 0;  // x0 < HW  =>  x0 == 0
				const int rstart = max(max(x0, nx - 1 - HW), nleft);
				const int rend = min(x0 + C::TX - 1, nx - 1);
				const int nb = nleft + max(0, rend - rstart + 1);
				for (int item = tid; item < C::ROWS * nb; item += C::NT) {
					const int r = item / nb, ci = item - r * nb;
					const int gy = y0 - HW - 1 + r;
					if (gy < 0 || gy >= ny) continue;
					const int gx = ci < nleft ? ci : rstart + (ci - nleft);
					const float *trow = &in_t[r * C::PITCH];
					float acc = 0.0f;
#pragma unroll 1
					for (int d = -HW; d <= HW; d++) {
						int lo, hi;
						float frac;
						boundary_src(gx, d, nx, lo, hi, frac);
						const float a = trow[lo - (x0 - C::HXL)], b = trow[hi - (x0 - C::HXL)];
						acc = acc + t.w[d + HW] * ((1.0f - frac) * a + frac * b);
					}
					xb[r * C::TX + (gx - x0)] = acc;
				}
			}
			__syncthreads();
			// ---------------- y-blur: xb -> registers (4 consecutive y of column x0+lane) ----------------
			{
				float yw[4 + 2 * HW];
#pragma unroll
				for (int k = 0; k < 4 + 2 * HW; k++) yw[k] = xb[(yq * 4 + 1 + k) * C::TX + lane];
#pragma unroll
				for (int j = 0; j < 4; j++) {
					float acc = 0.0f;
#pragma unroll
					for (int d = -HW; d <= HW; d++) acc = acc + t.w[d + HW] * yw[j + HW - d];
					v[j] = acc;
				}
				if (edge_y) {
#pragma unroll
					for (int j = 0; j < 4; j++) {
						const int gy = y0 + yq * 4 + j;
						if ((gy < HW || gy > ny - 2 - HW) && gy < ny) {
							float acc = 0.0f;
#pragma unroll 1
							for (int d = -HW; d <= HW; d++) {
								int lo, hi;
								float frac;
								boundary_src(gy, d, ny, lo, hi, frac);
								const float a = xb[(lo - (y0 - HW - 1)) * C::TX + lane], b = xb[(hi - (y0 - HW - 1)) * C::TX + lane];
								acc = acc + t.w[d + HW] * ((1.0f - frac) * a + frac * b);
							}
							v[j] = acc;
						}
					}
				}
			}
		}
		// ---------------- ring: slot k holds plane q - (RING-1) + k ----------------
#pragma unroll
		for (int j = 0; j < 4; j++) {
#pragma unroll
			for (int k = 0; k < C::RING - 1; k++) ring[j][k] = ring[j][k + 1];
			ring[j][C::RING - 1] = v[j];
		}
		// ---------------- z-blur of plane p = q - HW ----------------
		const int p = q - HW;
		if (p >= zc0 && p < zc1) {
			const bool z_interior = (p >= HW) && (p <= nz - 2 - HW);
			float out[4];
			if (z_interior) {
#pragma unroll
				for (int j = 0; j < 4; j++) {
					float acc = 0.0f;
#pragma unroll
					for (int d = -HW; d <= HW; d++) acc = acc + t.w[d + HW] * ring[j][HW + 1 - d];
					out[j] = acc;
				}
			} else {
				// wave-uniform tap sources; plane s sits in slot s - (p - HW - 1)
				float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
				for (int d = -HW; d <= HW; d++) {
					int lo, hi;
					float frac;
					boundary_src(p, d, nz, lo, hi, frac);
					const int slo = lo - (p - HW - 1), shi = hi - (p - HW - 1);
					const float tap = t.w[d + HW];
#pragma unroll
					for (int j = 0; j < 4; j++) {
						float a = ring[j][0], b = ring[j][0];
#pragma unroll
						for (int k = 1; k < C::RING; k++) {
							a = (slo == k) ? ring[j][k] : a;
							b = (shi == k) ? ring[j][k] : b;
						}
						acc[j] = acc[j] + tap * ((1.0f - frac) * a + frac * b);
					}
				}
#pragma unroll
				for (int j = 0; j < 4; j++) out[j] = acc[j];
			}
			const int gx = x0 + lane;
			if (gx < nx) {
#pragma unroll
				for (int j = 0; j < 4; j++) {
					const int gy = y0 + yq * 4 + j;
					if (gy < ny) {
						const size_t idx = (size_t)gx + sy * (size_t)gy + sz * (size_t)p;
						dst[idx] = out[j];
						if (DOG) {
							const float dg = (out[j] - src[idx]) * (-1.0f);
							dog[idx] = dg;
							mx = absmax_step_f(mx, dg);
						}
					}
				}
			}
		}
		__syncthreads();  // xb / in_t are rewritten by the next plane
	}
	if (DOG) {
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
		if (lane == 0) s_red[yq] = mx;
		__syncthreads();
		if (tid == 0) {
			const float r = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
			if (r > 0.0f) atomicMax(dogmax, __float_as_uint(r));
		}
	}
}

template <int HW>
static void launch_hw(const float *src, float *dst, float *dog, unsigned *dogmax, int nx, int ny, int nz, const Taps &t,
                      hipStream_t st) {
	using C = FusedCfg<HW>;
	const int ntx = (nx + C::TX - 1) / C::TX, nty = (ny + C::TY - 1) / C::TY;
	const int ntiles = ntx * nty;
	// enough workgroups to fill 256 CUs a few times over, but z chunks long enough to amortise the 2*HW+1 ramp
	int nchunks = (1024 + ntiles - 1) / ntiles;
	const int min_cz = 4 * (2 * HW + 1);
	int cz = (nz + nchunks - 1) / nchunks;
	if (cz < min_cz) cz = min_cz;
	if (cz > nz) cz = nz;
	nchunks = (nz + cz - 1) / cz;
	dim3 grid((unsigned)(ntiles * nchunks)), block(256);
	if (dog) hipLaunchKernelGGL((k_fused_level<HW, true>), grid, block, 0, st, src, dst, dog, dogmax, nx, ny, nz, t, ntx, nty, nchunks, cz);
	else hipLaunchKernelGGL((k_fused_level<HW, false>), grid, block, 0, st, src, dst, dog, dogmax, nx, ny, nz, t, ntx, nty, nchunks, cz);
}

// returns false when no fused instantiation exists for this half width (caller uses the generic
// separable kernels of kernels_pyramid.hip instead -- still the HIP path)
bool launch_fused_level(const float *src, float *dst, float *dog, unsigned *dogmax, int nx, int ny, int nz, const Taps &t,
                        hipStream_t st) {
	switch (t.hw) {
	case 2: launch_hw<2>(src, dst, dog, dogmax, nx, ny, nz, t, st); return true;
	case 3: launch_hw<3>(src, dst, dog, dogmax, nx, ny, nz, t, st); return true;
	case 4: launch_hw<4>(src, dst, dog, dogmax, nx, ny, nz, t, st); return true;
	case 5: launch_hw<5>(src, dst, dog, dogmax, nx, ny, nz, t, st); return true;
	case 6: launch_hw<6>(src, dst, dog, dogmax, nx, ny, nz, t, st); return true;
	case 8: launch_hw<8>(src, dst, dog, dogmax, nx, ny, nz, t, st); return true;
	default: return false;
	}
}

}  // namespace s3d
