// Dependent-issue latency of fp32 VALU on gfx950: NCH independent chains of (mul, add) per wave, scalar vs packed,
// at 1 and 3 waves per SIMD.  Prints cycles per wave-instruction per SIMD (4 = full rate).
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang fp contract(off)
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PACKED, int NCH>
__global__ void k(float *out, int iters, float a, float b) {
	float x[NCH];
	f2 y[NCH];
	for (int i = 0; i < NCH; i++) { x[i] = threadIdx.x + i; y[i] = f2{(float)threadIdx.x + i, (float)i}; }
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 16 / NCH; r++) {
			if (PACKED) {
#pragma unroll
				for (int i = 0; i < NCH; i++) y[i] = y[i] * a + b;
			} else {
#pragma unroll
				for (int i = 0; i < NCH; i++) x[i] = x[i] * a + b;
			}
		}
	}
	float s = 0;
	for (int i = 0; i < NCH; i++) s += x[i] + y[i].x + y[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int PACKED, int NCH>
void run(int wps) {
	float *d;
	const int threads = 256, blocks = 256 * wps;
	hipMalloc(&d, sizeof(float) * threads * blocks);
	const int iters = 20000;
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL((k<PACKED, NCH>), dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0001f, 0.5f);
	hipEventRecord(e0);
	hipLaunchKernelGGL((k<PACKED, NCH>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	const double per_simd = (double)iters * 32 * wps;
	printf("%s chains=%d waves/SIMD=%d  %.2f cycles per wave-instruction per SIMD (@2.4 GHz)\n", PACKED ? "packed" : "scalar", NCH, wps,
	       ms * 1e-3 * 2.4e9 / per_simd);
	hipFree(d);
}
int main() {
	for (int w : {1, 3}) { run<0, 1>(w); run<0, 2>(w); run<0, 4>(w); run<0, 8>(w); }
	for (int w : {1, 3}) { run<1, 1>(w); run<1, 2>(w); run<1, 4>(w); run<1, 8>(w); }
	return 0;
}
