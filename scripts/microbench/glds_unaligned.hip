// glds_unaligned.hip -- does global_load_lds_dwordx4 accept a global address that is only dword aligned?  (k_march_level stages its tiles
// with it; volumes whose row pitch is not a multiple of 16 bytes need the answer.)  hipcc --offload-arch=gfx950 -O3 -o glds_unaligned glds_unaligned.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ void dma16(const float *base, unsigned voff, unsigned lds_dst) {
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
__global__ void k(const float *src, float *dst, int shift) {
	__shared__ __attribute__((aligned(1024))) float t[256];
	dma16(src, (unsigned)(threadIdx.x * 4 + shift) * 4u, (unsigned)(unsigned long long)t);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	for (int i = threadIdx.x; i < 256; i += 64) dst[i] = t[i];
}
int main() {
	float h[1024], o[256];
	for (int i = 0; i < 1024; i++) h[i] = (float)i;
	float *s, *d;
	hipMalloc(&s, sizeof(h)); hipMalloc(&d, sizeof(o));
	hipMemcpy(s, h, sizeof(h), hipMemcpyHostToDevice);
	for (int shift = 0; shift < 4; shift++) {
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, shift);
		hipError_t e = hipDeviceSynchronize();
		hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
		int bad = 0;
		for (int i = 0; i < 256; i++) bad += o[i] != (float)(i + shift);
		printf("shift %d floats: %s, %d of 256 wrong (first values %g %g %g %g)\n", shift, hipGetErrorString(e), bad, o[0], o[1], o[2], o[3]);
	}
	return 0;
}
