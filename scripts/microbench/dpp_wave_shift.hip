// does v_mov_b32_dpp wave_shr:1 / wave_shl:1 on gfx950 shift across the whole 64-lane wave (lane i <- lane i-1 / i+1), and do the
// lanes without a source keep `old`?   hipcc --offload-arch=gfx950 -o /tmp/dpp_ws dpp_wave_shift.hip && /tmp/dpp_ws
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *o, const float *in, float edge_l, float edge_r) {
	float v = in[threadIdx.x];
	float l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge_l), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
	float r = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge_r), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
	o[threadIdx.x] = l; o[64 + threadIdx.x] = r;
}
int main() {
	float h[64], out[128], *d, *o;
	for (int i = 0; i < 64; i++) h[i] = (float)(i + 1);
	hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(out));
	hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d, -7.f, -9.f);
	hipMemcpy(out, o, sizeof(out), hipMemcpyDeviceToHost);
	int bad = 0;
	for (int i = 0; i < 64; i++) {
		const float wl = i == 0 ? -7.f : h[i - 1], wr = i == 63 ? -9.f : h[i + 1];
		if (out[i] != wl || out[64 + i] != wr) { bad++; printf("lane %d: shr %g (want %g) shl %g (want %g)\n", i, out[i], wl, out[64 + i], wr); }
	}
	printf("wave shifts: %s\n", bad ? "UNEXPECTED" : "lane i <- i-1 / i+1 across the wave, edge lanes keep old");
	return bad != 0;
}
