// issue rates of the instructions of k_describe's heavy part (gfx950): cycles per wave-instruction per SIMD at 4 waves per SIMD, 8 independent
// chains per wave.  hipcc --offload-arch=gfx950 -O3 -o instr_rates instr_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHAIN8(OP)                                                                                                             \
	asm volatile(OP " %0, %0\n\t" OP " %1, %1\n\t" OP " %2, %2\n\t" OP " %3, %3\n\t" OP " %4, %4\n\t" OP " %5, %5\n\t" OP " %6, %6\n\t" OP " %7, %7" \
	             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]))
#define CHAIN8B(OP)                                                                                                                    \
	asm volatile(OP " %0, %0, %8\n\t" OP " %1, %1, %8\n\t" OP " %2, %2, %8\n\t" OP " %3, %3, %8\n\t" OP " %4, %4, %8\n\t" OP " %5, %5, %8\n\t" OP " %6, %6, %8\n\t" OP " %7, %7, %8" \
	             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y))
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a) {
	float x[8];
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 p[5];
	for (int i = 0; i < 5; i++) p[i] = f2{threadIdx.x * 0.11f + i, a + i};
	float y = a;
	for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 0.37f + i + a;
	for (int it = 0; it < iters; it++) {
		if (MODE == 0) CHAIN8B("v_mul_f32");
		else if (MODE == 1) CHAIN8("v_cvt_rpi_i32_f32");
		else if (MODE == 2) CHAIN8("v_cvt_i32_f32");
		else if (MODE == 3) CHAIN8("v_floor_f32");
		else if (MODE == 4) CHAIN8("v_sqrt_f32");
		else if (MODE == 5) CHAIN8("v_rcp_f32");
		else if (MODE == 6) CHAIN8B("v_mul_i32_i24");
		else if (MODE == 7) CHAIN8B("v_mul_lo_u32");
		else if (MODE == 8) CHAIN8("v_cvt_f32_i32");
		else if (MODE == 9) CHAIN8B("v_pk_mul_f32");  // (pairs: reads the registers next to its operands; timing only)
		else if (MODE == 10) CHAIN8B("v_max_f32");
		else if (MODE == 11) CHAIN8("v_rndne_f32");
		else if (MODE == 12) CHAIN8B("v_add_u32");
		else if (MODE == 13) CHAIN8B("v_add_f32");
		else if (MODE == 14) CHAIN8B("v_sub_f32");
		else if (MODE == 15) asm volatile("v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\tv_fma_f32 %4, %4, %8, %8\n\tv_fma_f32 %5, %5, %8, %8\n\tv_fma_f32 %6, %6, %8, %8\n\tv_fma_f32 %7, %7, %8, %8"
		                           : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
		else if (MODE == 16) CHAIN8B("v_min_f32");
		else if (MODE == 17) CHAIN8B("v_and_b32");
		else if (MODE == 18) CHAIN8B("v_lshlrev_b32");
		else if (MODE == 19) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\tv_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
		                           : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y) : "vcc");
		else if (MODE == 20) asm volatile("v_cmp_lt_f32 vcc, %0, %8\n\tv_cmp_lt_f32 vcc, %1, %8\n\tv_cmp_lt_f32 vcc, %2, %8\n\tv_cmp_lt_f32 vcc, %3, %8\n\tv_cmp_lt_f32 vcc, %4, %8\n\tv_cmp_lt_f32 vcc, %5, %8\n\tv_cmp_lt_f32 vcc, %6, %8\n\tv_cmp_lt_f32 vcc, %7, %8"
		                           : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y) : "vcc");
		else if (MODE == 21) asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %4\n\tv_mov_b32 %4, %5\n\tv_mov_b32 %5, %6\n\tv_mov_b32 %6, %7\n\tv_mov_b32 %7, %8"
		                           : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
		else if (MODE == 22) asm volatile("v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\tv_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4"
		                           : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(p[4]));
		else if (MODE == 23) asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\tv_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4"
		                           : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(p[4]));
		else if (MODE == 24) asm volatile("v_mad_u32_u24 %0, %0, %8, %8\n\tv_mad_u32_u24 %1, %1, %8, %8\n\tv_mad_u32_u24 %2, %2, %8, %8\n\tv_mad_u32_u24 %3, %3, %8, %8\n\tv_mad_u32_u24 %4, %4, %8, %8\n\tv_mad_u32_u24 %5, %5, %8, %8\n\tv_mad_u32_u24 %6, %6, %8, %8\n\tv_mad_u32_u24 %7, %7, %8, %8"
		                           : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
		else if (MODE == 25) CHAIN8B("v_mul_f32_e64");
		else if (MODE == 27) asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\tv_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\tv_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\tv_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\tv_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\tv_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\tv_cndmask_b32_e64 %7, %7, %8, s[20:21]"
		                           : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y) : "s20", "s21");
		else if (MODE == 28) asm volatile("v_cmp_lt_f32 vcc, %0, %8\n\tv_cndmask_b32 %0, %0, %8, vcc\n\tv_cmp_lt_f32 vcc, %1, %8\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cmp_lt_f32 vcc, %2, %8\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cmp_lt_f32 vcc, %3, %8\n\tv_cndmask_b32 %3, %3, %8, vcc"
		                           : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y) : "vcc");
		else if (MODE == 29) asm volatile("v_cndmask_b32 %0, 0, %8, vcc\n\tv_cndmask_b32 %1, 0, %8, vcc\n\tv_cndmask_b32 %2, 0, %8, vcc\n\tv_cndmask_b32 %3, 0, %8, vcc\n\tv_cndmask_b32 %4, 0, %8, vcc\n\tv_cndmask_b32 %5, 0, %8, vcc\n\tv_cndmask_b32 %6, 0, %8, vcc\n\tv_cndmask_b32 %7, 0, %8, vcc"
		                           : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y) : "vcc");
		else if (MODE == 26) asm volatile("v_fmac_f32 %0, %8, %8\n\tv_fmac_f32 %1, %8, %8\n\tv_fmac_f32 %2, %8, %8\n\tv_fmac_f32 %3, %8, %8\n\tv_fmac_f32 %4, %8, %8\n\tv_fmac_f32 %5, %8, %8\n\tv_fmac_f32 %6, %8, %8\n\tv_fmac_f32 %7, %8, %8"
		                           : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
	}
	float s = 0;
	for (int i = 0; i < 8; i++) s += x[i];
	for (int i = 0; i < 5; i++) s += p[i].x + p[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name) {
	float *d;
	const int threads = 256, blocks = 256 * 4;
	hipMalloc(&d, sizeof(float) * threads * blocks);
	const int iters = 20000;
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0001f);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	const double per_simd = (double)iters * 8 * 4;
	printf("%-22s %.3f ms  %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / per_simd);
	hipFree(d);
}
int main() {
	run<0>("v_mul_f32"); run<1>("v_cvt_rpi_i32_f32"); run<2>("v_cvt_i32_f32"); run<3>("v_floor_f32"); run<4>("v_sqrt_f32"); run<5>("v_rcp_f32");
	run<6>("v_mul_i32_i24"); run<7>("v_mul_lo_u32"); run<8>("v_cvt_f32_i32"); run<10>("v_max_f32"); run<11>("v_rndne_f32"); run<12>("v_add_u32");
	run<13>("v_add_f32"); run<14>("v_sub_f32"); run<15>("v_fma_f32"); run<16>("v_min_f32"); run<17>("v_and_b32"); run<18>("v_lshlrev_b32"); run<19>("v_cndmask_b32");
	run<20>("v_cmp_lt_f32"); run<21>("v_mov_b32"); run<22>("v_pk_mul_f32 (2 lanes)"); run<23>("v_pk_add_f32 (2 lanes)"); run<24>("v_mad_u32_u24"); run<25>("v_mul_f32_e64"); run<26>("v_fmac_f32");
	run<19>("v_cndmask_b32 again"); run<27>("v_cndmask_b32_e64 sgpr"); run<28>("cmp+cndmask pairs (8 instr)"); run<29>("v_cndmask 0,y (no chain)"); run<0>("v_mul_f32 again");
	return 0;
}
