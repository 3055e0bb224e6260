// VALU throughput microbenchmark (gfx950): independent chains of v_mul_f32+v_add_f32 (no FMA contraction)
// versus packed v_pk_mul_f32+v_pk_add_f32, at 1..8 waves per SIMD.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang fp contract(off)
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PACKED>
__global__ void k(float *out, int iters, float a, float b) {
	float x[16];
	f2 y[8];
	for (int i = 0; i < 16; i++) x[i] = threadIdx.x + i;
	for (int i = 0; i < 8; i++) y[i] = f2{(float)threadIdx.x + i, (float)i};
	f2 a2 = {a, a}, b2 = {b, b};
	for (int it = 0; it < iters; it++) {
		if (PACKED) {
#pragma unroll
			for (int i = 0; i < 8; i++) y[i] = y[i] * a2 + b2;  // v_pk_mul_f32 + v_pk_add_f32
		} else {
#pragma unroll
			for (int i = 0; i < 16; i++) x[i] = x[i] * a + b;   // v_mul_f32 + v_add_f32
		}
	}
	float s = 0;
	for (int i = 0; i < 16; i++) s += x[i];
	for (int i = 0; i < 8; i++) s += y[i].x + y[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int PACKED>
void run(int waves_per_simd) {
	float *d;
	const int threads = 256, blocks = 256 * waves_per_simd;  // 4 waves per block -> waves_per_simd per SIMD
	hipMalloc(&d, sizeof(float) * threads * blocks);
	const int iters = 20000;
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<PACKED>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0001f, 0.5f);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<PACKED>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	const double instr_per_wave = (double)iters * (PACKED ? 16 : 32);  // mul + add per element (packed: per pair)
	const double per_simd = instr_per_wave * waves_per_simd;
	const double cyc = ms * 1e-3 * 2.4e9;
	printf("%s waves/SIMD=%d  %.3f ms  %.2f cycles per wave-instruction per SIMD (@2.4 GHz)  %.1f Gflop-lanes/s/SIMD\n",
	       PACKED ? "packed" : "scalar", waves_per_simd, ms, cyc / per_simd, (PACKED ? 2.0 : 1.0) * 64 * per_simd / (ms * 1e-3) / 1e9);
	hipFree(d);
}
int main() {
	for (int w : {1, 2, 4, 8}) run<0>(w);
	for (int w : {1, 2, 4, 8}) run<1>(w);
	return 0;
}
