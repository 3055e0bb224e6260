// fp32 VALU issue rate on gfx950 by OPERAND KIND (r06): v_mul_f32 / v_fma_f32 with vgpr x vgpr, vgpr x sgpr, vgpr x inline constant, and the packed forms,
// inline asm (nothing for the compiler to vectorise), 8 chains, 4 waves per SIMD.  ns per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a) {
	float x[8];
	f2 p[8];
	float y = a + threadIdx.x * 1e-6f;
	f2 y2 = f2{y, y};
	for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.37f + i + a; p[i] = f2{x[i], x[i] + 1.f}; }
	for (int it = 0; it < iters; it++) {
#define VV(i) "v_mul_f32 %" #i ", %" #i ", %8\n\t"
#define VS(i) "v_mul_f32 %" #i ", %8, %" #i "\n\t"
#define VC(i) "v_mul_f32 %" #i ", 0.5, %" #i "\n\t"
#define FV(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n\t"
#define FS(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n\t"
#define PV(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n\t"
#define PF(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %8\n\t"
#define AV(i) "v_add_f32 %" #i ", %" #i ", %8\n\t"
		if (MODE == 0) asm volatile(REP8(VV) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
		if (MODE == 1) asm volatile(REP8(VS) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "s"(a));
		if (MODE == 2) asm volatile(REP8(VC) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "s"(a));
		if (MODE == 3) asm volatile(REP8(FV) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
		if (MODE == 4) asm volatile(REP8(FS) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "s"(a));
		if (MODE == 5) asm volatile(REP8(PV) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(y2));
		if (MODE == 6) asm volatile(REP8(PF) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(y2));
		if (MODE == 7) asm volatile(REP8(AV) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
	}
	float s = 0;
	for (int i = 0; i < 8; i++) s += x[i] + p[i].x + p[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int wps) {
	float *d;
	const int threads = 256, blocks = 256 * wps;
	hipMalloc(&d, sizeof(float) * threads * blocks);
	const int iters = 20000;
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters / 4, 1.0001f);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	printf("%-34s waves/SIMD %d  %.3f ms  %.2f ns per wave-instruction per SIMD\n", name, wps, ms, ms * 1e6 / ((double)iters * 8 * wps));
	hipFree(d);
}
int main() {
	for (int w : {4, 8}) {
		run<0>("v_mul_f32 vgpr,vgpr", w); run<1>("v_mul_f32 sgpr,vgpr", w); run<2>("v_mul_f32 const,vgpr", w); run<3>("v_fma_f32 vgpr,vgpr,vgpr", w);
		run<4>("v_fma_f32 vgpr,sgpr,sgpr", w); run<5>("v_pk_mul_f32 (2 lanes per instr)", w); run<6>("v_pk_fma_f32 (2 lanes per instr)", w); run<7>("v_add_f32 vgpr,vgpr", w);
	}
	return 0;
}
