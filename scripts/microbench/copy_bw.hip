// copy_bw.hip -- device-copy ceiling sweep (float4 copies, read + write counted).  hipcc --offload-arch=gfx950 -O3 -o copy_bw copy_bw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_copy(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n4) {
	const size_t stride = (size_t)gridDim.x * 256;
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	for (; i + (U - 1) * stride < n4; i += U * stride) {
		f4 v[U];
#pragma unroll
		for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
		for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u]; }
	}
	for (; i < n4; i += stride) d[i] = s[i];
}
template <int U, bool NT>
static void run(const f4 *s, f4 *d, size_t n4, int grid) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9;
	for (int it = 0; it < 6; it++) {
		hipEventRecord(e0); hipLaunchKernelGGL((k_copy<U, NT>), dim3(grid), dim3(256), 0, 0, s, d, n4); hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
	}
	printf("U %d nt %d grid %6d : %.0f GB/s\n", U, (int)NT, grid, 2.0 * n4 * 16 / (best * 1e-3) / 1e9);
}
int main() {
	const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
	f4 *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
	for (int grid : {1024, 2048, 4096, 8192, 16384, 65536, (int)(n4 / 256)}) {
		run<1, false>(s, d, n4, grid); run<2, false>(s, d, n4, grid); run<4, false>(s, d, n4, grid); run<8, false>(s, d, n4, grid);
		run<1, true>(s, d, n4, grid); run<4, true>(s, d, n4, grid);
	}
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int it = 0; it < 3; it++) { hipEventRecord(e0); hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); printf("hipMemcpy D2D: %.0f GB/s\n", 2.0 * bytes / (ms * 1e-3) / 1e9); }
	return 0;
}
