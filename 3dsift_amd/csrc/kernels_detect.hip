// kernels_detect.hip -- DoG extrema scan with deterministic, order-preserving compaction.
//
// Restates Detect_KeyPoints + IsExtrema_neighbor (reference Src/cSIFT3D.cc:362-425, 884-911):
// for x,y,z in [1, n-2]: (val > thr || val < -thr) and val strictly below (or strictly above) all
// EIGHT neighbours: +-x, +-y, +-z in the current DoG level and the centre voxel of the previous and
// next level.  thr = peak_thresh * max|level| (max from the DoG-producing kernel, fp32 multiply).
//
// The reference emits extrema in (octave, level, z, y, x) scan order and the matcher output order
// depends on it, so instead of an atomic append + sort the scan is made order-preserving:
//   k_mark : one thread per voxel in linear order; each wave stores its 64-bit ballot, each block its count
//   k_scan : one workgroup turns block counts into exclusive offsets (+ running base over levels)
//   k_emit : each wave re-reads its ballot and writes its extrema at offset + rank (popcount prefix)
// No host synchronisation anywhere; the running total stays on the device.
#include "sift3d_internal.h"

namespace s3d {

constexpr int kBlock = 1024;  // 16 waves per block

__global__ void __launch_bounds__(kBlock) k_mark(const float *__restrict__ prev, const float *__restrict__ cur,
                                                 const float *__restrict__ next, int nx, int ny, int nz,
                                                 const unsigned *__restrict__ absmax_bits, float peak_thresh,
                                                 unsigned long long *__restrict__ masks, unsigned *__restrict__ block_counts) {
	__shared__ unsigned s_cnt[kBlock / 64];
	const size_t total = (size_t)nx * ny * nz;
	const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
	const float thr = peak_thresh * __uint_as_float(*absmax_bits);
	bool hit = false;
	if (i < total) {
		const int x = (int)(i % nx);
		const size_t r = i / nx;
		const int y = (int)(r % ny), z = (int)(r / ny);
		if (x >= 1 && x <= nx - 2 && y >= 1 && y <= ny - 2 && z >= 1 && z <= nz - 2) {
			const float val = cur[i];
			if (val > thr || val < -thr) {
				const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
				const float n0 = prev[i], n1 = cur[i - 1], n2 = cur[i + 1], n3 = cur[i + sy], n4 = cur[i - sy],
				            n5 = cur[i + sz], n6 = cur[i - sz], n7 = next[i];
				const bool mn = val < n0 && val < n1 && val < n2 && val < n3 && val < n4 && val < n5 && val < n6 && val < n7;
				const bool mx = val > n0 && val > n1 && val > n2 && val > n3 && val > n4 && val > n5 && val > n6 && val > n7;
				hit = mn || mx;
			}
		}
	}
	const unsigned long long b = __ballot(hit);
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	if (lane == 0) {
		masks[(size_t)blockIdx.x * (kBlock / 64) + wid] = b;
		s_cnt[wid] = (unsigned)__popcll(b);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned c = 0;
		for (int w = 0; w < kBlock / 64; w++) c += s_cnt[w];
		block_counts[blockIdx.x] = c;
	}
}

// exclusive scan of block_counts[0..nblocks) by ONE workgroup; offsets start at the running total
// total[0], which is then advanced.  nblocks <= ~1e6.
__global__ void __launch_bounds__(1024) k_scan(const unsigned *__restrict__ block_counts, unsigned *__restrict__ block_offsets,
                                               unsigned nblocks, unsigned *__restrict__ total) {
	__shared__ unsigned s_part[1024];
	__shared__ unsigned s_wave[16];
	const unsigned t = threadIdx.x;
	const unsigned chunk = (nblocks + 1023u) / 1024u;
	const unsigned lo = t * chunk, hi = min(lo + chunk, nblocks);
	unsigned sum = 0;
	for (unsigned i = lo; i < hi; i++) sum += block_counts[i];
	// inclusive scan of the 1024 partial sums: wave scan + scan of wave totals
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
	s_part[t] = incl - sum;  // exclusive prefix of this thread's chunk
	__syncthreads();
	const unsigned base = total[0];
	unsigned run = base + s_part[t];
	for (unsigned i = lo; i < hi; i++) { block_offsets[i] = run; run += block_counts[i]; }
	__syncthreads();
	if (t == 1023) total[0] = base + incl;
}

__global__ void __launch_bounds__(kBlock) k_emit(const unsigned long long *__restrict__ masks,
                                                 const unsigned *__restrict__ block_offsets, int nx, int ny, int octave,
                                                 int level, float scale, DevKp *__restrict__ out, unsigned cap,
                                                 unsigned *__restrict__ total) {
	__shared__ unsigned s_off[kBlock / 64];
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	const unsigned long long b = masks[(size_t)blockIdx.x * (kBlock / 64) + wid];
	if (lane == 0) s_off[wid] = (unsigned)__popcll(b);
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned a = block_offsets[blockIdx.x];
		for (int w = 0; w < kBlock / 64; w++) { unsigned c = s_off[w]; s_off[w] = a; a += c; }
	}
	__syncthreads();
	if ((b >> lane) & 1ull) {
		const unsigned pos = s_off[wid] + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
		if (pos < cap) {
			const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
			DevKp k;
			k.x = (int)(i % nx);
			const size_t r = i / nx;
			k.y = (int)(r % ny);
			k.z = (int)(r / ny);
			k.octave = octave; k.level = level; k.scale = scale; k.code = 0; k.slot = -1;
#pragma unroll
			for (int j = 0; j < 3; j++) { k.win[j] = 0.f; k.eigvalue[j] = 0.f; }
#pragma unroll
			for (int j = 0; j < 9; j++) { k.eigvector[j] = 0.f; k.rot[j] = 0.f; k.st[j] = 0.f; }
			out[pos] = k;
		} else {
			total[1] = 1u;  // overflow flag; the host regrows and reruns
		}
	}
}

void launch_detect_level(const float *prev, const float *cur, const float *next, int nx, int ny, int nz,
                         const unsigned *d_absmax_bits, float peak_thresh, int octave, int level, float scale,
                         const DetectBufs &b, DevKp *out, unsigned cap, hipStream_t st) {
	const size_t total = (size_t)nx * ny * nz;
	const unsigned nblocks = (unsigned)((total + kBlock - 1) / kBlock);
	hipLaunchKernelGGL(k_mark, dim3(nblocks), dim3(kBlock), 0, st, prev, cur, next, nx, ny, nz, d_absmax_bits, peak_thresh,
	                   b.masks, b.block_counts);
	hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, b.block_counts, b.block_offsets, nblocks, b.total);
	hipLaunchKernelGGL(k_emit, dim3(nblocks), dim3(kBlock), 0, st, b.masks, b.block_offsets, nx, ny, octave, level, scale, out,
	                   cap, b.total);
}

}  // namespace s3d
