// LDS atomic throughput microbenchmark (gfx950): ds_add_f32 vs ds_add_u32 vs ds_add_u64, conflict-free
// (each lane its own address) and 4-way same-address.  Prints wave-instructions per microsecond per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE, int CONF>
__global__ void __launch_bounds__(256) k(float *out, int iters) {
	__shared__ float sf[4096];
	__shared__ unsigned long long s64[2048];
	for (int i = threadIdx.x; i < 4096; i += 256) sf[i] = 0.f;
	for (int i = threadIdx.x; i < 2048; i += 256) s64[i] = 0ull;
	__syncthreads();
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int idx = (CONF ? (lane / CONF) : lane) + w * 64;
	float v = 1.0f + lane;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int j = 0; j < 16; j++) {
			const int a = (idx + j * 256) & 4095;
			if (MODE == 0) atomicAdd(&sf[a], v);
			else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned *>(&sf[a]), (unsigned)lane + 1u);
			else atomicAdd(&s64[a & 2047], (unsigned long long)lane + 1ull);
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) out[blockIdx.x] = sf[5] + (float)s64[7];
}
template <int MODE, int CONF>
void run(const char *name) {
	float *d;
	hipMalloc(&d, 4096 * 4);
	const int iters = 2000, blocks = 256 * 4;
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL((k<MODE, CONF>), dim3(blocks), dim3(256), 0, 0, d, 10);
	hipEventRecord(a);
	hipLaunchKernelGGL((k<MODE, CONF>), dim3(blocks), dim3(256), 0, 0, d, iters);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	const double winstr = (double)blocks * 4 * iters * 16;  // wave-instructions
	printf("%-28s %8.3f ms  %7.1f wave-instr/us/CU  => %5.1f cycles per wave-instr per CU @2.4GHz\n", name, ms,
	       winstr / (ms * 1e3) / 256.0, 2400.0 / (winstr / (ms * 1e3) / 256.0));
	hipFree(d);
}
int main() {
	run<0, 0>("ds_add_f32 conflict-free");
	run<0, 4>("ds_add_f32 4 lanes/address");
	run<1, 0>("ds_add_u32 conflict-free");
	run<1, 4>("ds_add_u32 4 lanes/address");
	run<2, 0>("ds_add_u64 conflict-free");
	run<2, 4>("ds_add_u64 4 lanes/address");
	run<1, 2>("ds_add_u32 2 lanes/address");
	run<1, 8>("ds_add_u32 8 lanes/address");
	run<1, 16>("ds_add_u32 16 lanes/address");
	run<2, 2>("ds_add_u64 2 lanes/address");
	run<2, 8>("ds_add_u64 8 lanes/address");
	run<2, 16>("ds_add_u64 16 lanes/address");
	return 0;
}
