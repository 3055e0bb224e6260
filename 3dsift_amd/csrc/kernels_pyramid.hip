// kernels_pyramid.hip -- scale-space kernels: max-abs normalisation, separable Gaussian with the
// reference boundary rule, DoG, decimation.  gfx950 only.
//
// Parity contract (bit-exact with the reference CPU path):
//   * taps come from the host (s3d::Taps), built like Src/cSIFT3D.cc:541-572
//   * every output is acc = 0; for d = -hw..+hw: acc = acc + tap[d+hw] * term, separate IEEE
//     multiply and add, never contracted into an FMA (this file is built with -ffp-contract=off)
//   * interior term = src[p-d]; boundary term = (1-frac)*src[lo] + frac*src[hi] with the mirror /
//     "2*dim_end - c - 0.1f" coordinate rule of Src/cSIFT3D.cc:722-788 evaluated in fp32
//   * DoG = (cur - prev) * (-1)  (Src/cSIFT3D.cc:875)
#include <algorithm>

#include "sift3d_internal.h"

namespace s3d {

// ---------------------------------------------------------------------------------------------
// wave / block max helpers (wave = 64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
	return v;
}

// |v| max-abs with the reference's "NaN never wins" comparison (Src/cUtil.cc:548, 598):
// max = (fabs(tmp) > max) ? fabs(tmp) : max
__device__ __forceinline__ float absmax_step(float m, float v) {
	float a = fabsf(v);
	return (a > m) ? a : m;
}

__device__ __forceinline__ void block_max_to_global(float m, unsigned *dst) {
	__shared__ float s[16];
	m = wave_max(m);
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	if (lane == 0) s[wid] = m;
	__syncthreads();
	if (threadIdx.x == 0) {
		const int nw = (blockDim.x + 63) >> 6;
		float r = s[0];
		for (int i = 1; i < nw; i++) r = fmaxf(r, s[i]);
		// non-negative floats order like their bit patterns
		if (r > 0.0f) atomicMax(dst, __float_as_uint(r));
	}
}

// ---------------------------------------------------------------------------------------------
// measured device-copy ceiling beside the spec peak (sift3d_debug_copy_bandwidth): the float4 copy the MI355X guide
// quotes 6.29 TB/s for -- one 16-byte load and store per lane, read + write counted
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_copy16(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4) {
	// ONE 16-byte piece per lane, as many workgroups as pieces: the workgroups in flight cover a contiguous, moving window of the
	// buffers (scripts/microbench/copy_bw.hip: 6.29 TB/s this way, 4.5-5.0 TB/s as a grid-stride loop of 4096 workgroups)
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n4) dst[i] = src[i];
}
void launch_copy16(const float *src, float *dst, size_t nfloats, hipStream_t st) {
	const size_t n4 = nfloats / 4;
	hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst), n4);
}

// ---------------------------------------------------------------------------------------------
// The simulated transport of the z-slab driver (csrc/sharded.hip, all ranks on one GPU): the "sends" of one exchange step -- up to
// kCopySegs plane ranges -- as ONE launch instead of one hipMemcpyAsync each (a level of eight simulated ranks posted 42 copies of a few
// planes: ~500 copy launches per step on the one stream the ranks share), and the MAX "all-reduce" of the DoG maxima on the device.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_copy_segments(CopySegs a) {
	const int sg = blockIdx.y;
	if (sg >= a.n) return;
	const float *__restrict__ src = a.src[sg];
	float *__restrict__ dst = a.dst[sg];
	const size_t nf = a.floats[sg];
	// (plane ranges of one level: both ends share the alignment of plane * nx * ny floats; a 16-byte body where both are aligned)
	const bool al = ((((size_t)src) | ((size_t)dst)) & 15) == 0;
	const size_t n4 = al ? nf / 4 : 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
		reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(src)[i];
	for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void launch_copy_segments(const CopySegs &a, hipStream_t st) {
	if (a.n <= 0) return;
	size_t mx = 0;
	for (int i = 0; i < a.n; i++) mx = std::max(mx, a.floats[i]);
	if (mx == 0) return;
	const unsigned gx = (unsigned)std::min<size_t>(std::max<size_t>(1, (mx / 4 + 255) / 256), 256);
	hipLaunchKernelGGL(k_copy_segments, dim3(gx, (unsigned)a.n), dim3(256), 0, st, a);
}
__global__ void __launch_bounds__(64) k_max_merge(MaxMerge a) {
	const int i = threadIdx.x;
	if (i >= a.n) return;
	float m = 0.0f;
	for (int r = 0; r < a.np; r++) m = fmaxf(m, a.p[r][i]);
	for (int r = 0; r < a.np; r++) a.p[r][i] = m;
}
void launch_max_merge(const MaxMerge &a, hipStream_t st) {
	if (a.np > 0 && a.n > 0) hipLaunchKernelGGL(k_max_merge, dim3(1), dim3(64), 0, st, a);
}

// ---------------------------------------------------------------------------------------------
// data_scale (Src/cUtil.cc:536-564): global max|v|, then v /= max
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_absmax(const float *__restrict__ src, size_t n, unsigned *dst) {
	float m = 0.0f;
	// scalar head up to the first 16-byte boundary (a z-slab's owned planes start at plane * nx * ny floats: any dword alignment),
	// float4 body, scalar tail
	const size_t mis = ((size_t)src >> 2) & 3;
	const size_t head = mis ? (4 - mis < n ? 4 - mis : n) : 0;
	const size_t n4 = (n - head) / 4;
	const float4 *s4 = reinterpret_cast<const float4 *>(src + head);
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
		float4 v = s4[i];
		m = absmax_step(m, v.x); m = absmax_step(m, v.y); m = absmax_step(m, v.z); m = absmax_step(m, v.w);
	}
	if (blockIdx.x == 0) {
		if (threadIdx.x < head) m = absmax_step(m, src[threadIdx.x]);
		const size_t tail0 = head + n4 * 4;
		if (threadIdx.x < n - tail0) m = absmax_step(m, src[tail0 + threadIdx.x]);
	}
	block_max_to_global(m, dst);
}

__global__ void __launch_bounds__(256) k_scale(float *__restrict__ data, size_t n, const unsigned *mx) {
	const float m = __uint_as_float(*mx);
	const size_t n4 = n / 4;
	float4 *d4 = reinterpret_cast<float4 *>(data);
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
		float4 v = d4[i];
		v.x = __fdiv_rn(v.x, m); v.y = __fdiv_rn(v.y, m); v.z = __fdiv_rn(v.z, m); v.w = __fdiv_rn(v.w, m);
		d4[i] = v;
	}
	if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
		size_t i = n4 * 4 + threadIdx.x;
		data[i] = __fdiv_rn(data[i], m);
	}
}

static int grid_for(size_t n, int block, int per_thread = 1) {
	size_t b = (n + (size_t)block * per_thread - 1) / ((size_t)block * per_thread);
	if (b < 1) b = 1;
	if (b > 256 * 8) b = 256 * 8;  // grid-stride the rest
	return (int)b;
}

void launch_absmax(const float *src, size_t n, unsigned *d_max_bits, hipStream_t st) {
	hipMemsetAsync(d_max_bits, 0, sizeof(unsigned), st);
	hipLaunchKernelGGL(k_absmax, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, src, n, d_max_bits);
}

// Sub (Src/cSIFT3D.cc:849-882) for a DoG level the pipeline did not materialise (checking accessor only): dog = (hi - lo) * (-1)
__global__ void __launch_bounds__(256) k_dog_from_gss(const float *__restrict__ hi, const float *__restrict__ lo, float *__restrict__ dog, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dog[i] = (hi[i] - lo[i]) * (-1.0f);
}
void launch_dog_from_gss(const float *hi, const float *lo, float *dog, size_t n, hipStream_t st) {
	hipLaunchKernelGGL(k_dog_from_gss, dim3(grid_for(n, 256)), dim3(256), 0, st, hi, lo, dog, n);
}

void launch_scale_by_max(float *data, size_t n, const unsigned *d_max_bits, hipStream_t st) {
	hipLaunchKernelGGL(k_scale, dim3(grid_for(n, 256, 4)), dim3(256), 0, st, data, n, d_max_bits);
}

// ---------------------------------------------------------------------------------------------
// one separable pass along AXIS, one thread per output voxel (x fastest => coalesced on every axis)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float boundary_term(const float *__restrict__ line, size_t stride, int p, int d, int n) {
	const int dim_end = n - 1;
	float c = (float)p - (float)d * 1.0f;
	if (c < 0.0f) c = -1.0f * c;
	else if (c >= (float)dim_end) c = (float)(2 * dim_end) - c - 0.1f;
	int lo = (int)c;
	const float frac = c - (float)lo;
	int hi = lo + 1;
	// the reference reads out of bounds when n <= 9 and hw == 8 (undefined values there); clamp
	lo = min(max(lo, 0), dim_end);
	hi = min(max(hi, 0), dim_end);
	const float a = line[(size_t)lo * stride], b = line[(size_t)hi * stride];
	return (1.0f - frac) * a + frac * b;
}

template <int AXIS, bool DOG>
__global__ void __launch_bounds__(256) k_conv_axis(const float *__restrict__ src, float *__restrict__ dst, int nx, int ny,
                                                   int nz, Taps t, const float *__restrict__ prev,
                                                   float *__restrict__ dog, unsigned *dogmax) {
	const size_t total = (size_t)nx * ny * nz;
	const int hw = t.hw;
	float m = 0.0f;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const int x = (int)(i % nx);
		const size_t r = i / nx;
		const int y = (int)(r % ny), z = (int)(r / ny);
		int p, n;
		size_t stride;
		if (AXIS == 0) { p = x; n = nx; stride = 1; }
		else if (AXIS == 1) { p = y; n = ny; stride = (size_t)nx; }
		else { p = z; n = nz; stride = (size_t)nx * ny; }
		const float *line = src + (i - (size_t)p * stride);
		float acc = 0.0f;
		if (p >= hw && p <= n - 2 - hw) {
			for (int d = -hw; d <= hw; d++) acc = acc + t.w[d + hw] * line[(size_t)(p - d) * stride];
		} else {
			for (int d = -hw; d <= hw; d++) acc = acc + t.w[d + hw] * boundary_term(line, stride, p, d, n);
		}
		dst[i] = acc;
		if (DOG) {
			const float dg = (acc - prev[i]) * (-1.0f);
			dog[i] = dg;
			m = absmax_step(m, dg);
		}
	}
	if (DOG) block_max_to_global(m, dogmax);
}

void launch_conv_axis(int axis, const float *src, float *dst, int nx, int ny, int nz, const Taps &t, const float *prev,
                      float *dog, unsigned *d_dogmax, hipStream_t st) {
	const size_t total = (size_t)nx * ny * nz;
	dim3 grid((unsigned)((total + 255) / 256 > 65535u * 64u ? 65535u * 64u : (total + 255) / 256)), block(256);
	if (axis == 0) hipLaunchKernelGGL((k_conv_axis<0, false>), grid, block, 0, st, src, dst, nx, ny, nz, t, nullptr, nullptr, nullptr);
	else if (axis == 1) hipLaunchKernelGGL((k_conv_axis<1, false>), grid, block, 0, st, src, dst, nx, ny, nz, t, nullptr, nullptr, nullptr);
	else if (dog) hipLaunchKernelGGL((k_conv_axis<2, true>), grid, block, 0, st, src, dst, nx, ny, nz, t, prev, dog, d_dogmax);
	else hipLaunchKernelGGL((k_conv_axis<2, false>), grid, block, 0, st, src, dst, nx, ny, nz, t, nullptr, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------
// DownSample_3D (Src/cSIFT3D.cc:506-533): dst(x,y,z) = src(2x,2y,2z)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_downsample(const float *__restrict__ src, int snx, int sny, float *__restrict__ dst,
                                                    int nx, int ny, int nz) {
	const size_t total = (size_t)nx * ny * nz;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const int x = (int)(i % nx);
		const size_t r = i / nx;
		const int y = (int)(r % ny), z = (int)(r / ny);
		dst[i] = src[(size_t)(2 * x) + (size_t)snx * ((size_t)(2 * y) + (size_t)sny * (size_t)(2 * z))];
	}
}

// widths that are multiples of 4: one thread makes 4 consecutive outputs from two 16-byte loads and one 16-byte store
// (0.75 vector-memory instructions per output instead of 2; this launch heads every octave's chain)
typedef float f4d __attribute__((ext_vector_type(4), aligned(4)));
__global__ void __launch_bounds__(256) k_downsample4(const float *__restrict__ src, int snx, int sny, float *__restrict__ dst,
                                                     int nx4, int ny, int nz) {
	const size_t total = (size_t)nx4 * ny * nz;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const int x4 = (int)(i % nx4);
		const size_t r = i / nx4;
		const int y = (int)(r % ny), z = (int)(r / ny);
		const float *s = src + (size_t)(8 * x4) + (size_t)snx * ((size_t)(2 * y) + (size_t)sny * (size_t)(2 * z));
		const f4d a = *reinterpret_cast<const f4d *>(s), b = *reinterpret_cast<const f4d *>(s + 4);
		*reinterpret_cast<f4d *>(dst + 4 * i) = f4d{a.x, a.z, b.x, b.z};
	}
}

void launch_downsample(const float *src, int snx, int sny, float *dst, int nx, int ny, int nz, hipStream_t st) {
	const size_t total = (size_t)nx * ny * nz;
	// 8*x4+7 <= 2*nx-1 <= snx-1: the two pieces stay inside the source row
	if ((nx & 3) == 0) hipLaunchKernelGGL(k_downsample4, dim3(grid_for(total / 4, 256)), dim3(256), 0, st, src, snx, sny, dst, nx / 4, ny, nz);
	else hipLaunchKernelGGL(k_downsample, dim3(grid_for(total, 256)), dim3(256), 0, st, src, snx, sny, dst, nx, ny, nz);
}

}  // namespace s3d
