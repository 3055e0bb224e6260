// read_bw.hip -- read-only streaming in the launch shape of k_mark (24 576 workgroups x 4 waves, a wave reads 8 rows of 2 KB one after the
// other): what rate does the SHAPE reach with 4-byte loads per lane, with 16-byte loads, with all rows requested at once?
//   hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
// MODE 0: per row 8 dword loads (512 floats), rows one after the other   1: all 8 rows (64 dword loads) requested at once
//      2: per row 2 dwordx4 loads   3: all rows, 16 dwordx4 loads at once     LDSB: bytes of (unused) LDS per workgroup -> occupancy
template <int MODE>
__global__ void __launch_bounds__(256) k_read(const float *__restrict__ s, float *__restrict__ out, int lds_words) {
	extern __shared__ float dyn[];
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	const float *base = s + ((size_t)blockIdx.x * 32 + wid * 8) * 512;
	float acc = 0.f;
	if (lds_words && threadIdx.x == 1000) dyn[lds_words - 1] = 1.f;
	if (MODE == 0) {
		for (int r = 0; r < 8; r++) {
			float v[8];
#pragma unroll
			for (int b = 0; b < 8; b++) v[b] = base[r * 512 + b * 64 + lane];
#pragma unroll
			for (int b = 0; b < 8; b++) acc += __builtin_popcountll(__ballot(fabsf(v[b]) > 0.5f));
		}
	} else if (MODE == 1) {
		float v[64];
#pragma unroll
		for (int i = 0; i < 64; i++) v[i] = base[i * 64 + lane];
#pragma unroll
		for (int i = 0; i < 64; i++) acc += __builtin_popcountll(__ballot(fabsf(v[i]) > 0.5f));
	} else if (MODE == 2) {
		for (int r = 0; r < 8; r++) {
			f4 v[2];
#pragma unroll
			for (int b = 0; b < 2; b++) v[b] = *reinterpret_cast<const f4 *>(base + r * 512 + b * 256 + lane * 4);
#pragma unroll
			for (int b = 0; b < 2; b++) acc += __builtin_popcountll(__ballot(fabsf(v[b].x) > 0.5f)) + __builtin_popcountll(__ballot(fabsf(v[b].y) > 0.5f)) +
			                                 __builtin_popcountll(__ballot(fabsf(v[b].z) > 0.5f)) + __builtin_popcountll(__ballot(fabsf(v[b].w) > 0.5f));
		}
	} else {
		f4 v[16];
#pragma unroll
		for (int i = 0; i < 16; i++) v[i] = *reinterpret_cast<const f4 *>(base + i * 256 + lane * 4);
#pragma unroll
		for (int i = 0; i < 16; i++) acc += __builtin_popcountll(__ballot(fabsf(v[i].x) > 0.5f)) + __builtin_popcountll(__ballot(fabsf(v[i].y) > 0.5f)) +
		                                  __builtin_popcountll(__ballot(fabsf(v[i].z) > 0.5f)) + __builtin_popcountll(__ballot(fabsf(v[i].w) > 0.5f));
	}
	if (acc == 12345.f) out[blockIdx.x] = acc;
}
template <int MODE>
static void run(const float *s, float *o, int grid, int lds_bytes) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9;
	for (int it = 0; it < 6; it++) {
		hipEventRecord(e0); hipLaunchKernelGGL((k_read<MODE>), dim3(grid), dim3(256), lds_bytes, 0, s, o, lds_bytes / 4); hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
	}
	printf("mode %d lds %6d B/wg grid %6d : %7.1f us  %.0f GB/s\n", MODE, lds_bytes, grid, best * 1e3, (double)grid * 32 * 512 * 4 / (best * 1e-3) / 1e9);
}
int main() {
	const int grid = 24576;                       // 3 levels x 512 planes x 16 row blocks
	const size_t bytes = (size_t)grid * 32 * 512 * 4;  // 1.6 GB
	float *s, *o; hipMalloc(&s, bytes); hipMalloc(&o, grid * 4); hipMemset(s, 0, bytes);
	for (int lds : {0, 17408, 40960}) { run<0>(s, o, grid, lds); run<1>(s, o, grid, lds); run<2>(s, o, grid, lds); run<3>(s, o, grid, lds); }
	return 0;
}
