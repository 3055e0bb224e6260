// Where do the waves of co-resident 256-thread workgroups land?  Prints (block, wave) -> XCC, SE, CU, SIMD, TG_ID, WAVE_ID for the
// workgroups of the first CU seen.  hipcc --offload-arch=gfx950 hwid_probe.hip -o hwid_probe && ./hwid_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
__global__ void __launch_bounds__(256, 3) k(unsigned *out, int spin) {
	__shared__ float pad[8192];  // 32 KB: three workgroups per CU
	const int wid = threadIdx.x >> 6;
	const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
	const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20); // HW_REG_XCC_ID
	float a = threadIdx.x;
	for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;
	pad[threadIdx.x] = a;
	__syncthreads();
	if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + wid) * 2] = hw; out[(blockIdx.x * 4 + wid) * 2 + 1] = xcc + (pad[(threadIdx.x + 1) & 255] == 123.f); }
}
int main() {
	const int nb = 768;
	unsigned *d; hipMalloc(&d, nb * 4 * 2 * 4);
	hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, 200000);
	std::vector<unsigned> h(nb * 8);
	hipMemcpy(h.data(), d, nb * 32, hipMemcpyDeviceToHost);
	struct R { unsigned key; int b, w, simd, tg, wave; };
	std::vector<R> r;
	for (int b = 0; b < nb; b++) for (int w = 0; w < 4; w++) {
		const unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 15;
		const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
		r.push_back({(xcc << 12) | (se << 8) | (sh << 4) | cu, b, w, (int)((hw >> 4) & 3), (int)((hw >> 16) & 15), (int)(hw & 15)});
	}
	std::sort(r.begin(), r.end(), [](const R &a, const R &b) { return a.key != b.key ? a.key < b.key : (a.b != b.b ? a.b < b.b : a.w < b.w); });
	unsigned last = ~0u; int shown = 0;
	for (auto &x : r) {
		if (x.key != last) { if (++shown > 6) break; printf("CU key %05x:\n", x.key); last = x.key; }
		printf("   block %4d wave %d  simd %d  tg %2d  waveslot %d\n", x.b, x.w, x.simd, x.tg, x.wave);
	}
	return 0;
}
