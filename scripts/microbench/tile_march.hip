// tile_march.hip -- the MEMORY SKELETON of k_march_level (no blur arithmetic): a workgroup owns a TX x TY column, marches along z,
// brings the (TX + 2 HX) x (TY + 2 HW) input tile of the next plane into an LDS double buffer by LDS-DMA, and stores its TX x TY
// centre to one or two output volumes with untracked non-temporal 16-byte stores; one barrier per plane, counted vmcnt wait.
// Question (r04): is the product's level time set by the tile GEOMETRY (halo amplification, row segment length, workgroup size)?
//   hipcc --offload-arch=gfx950 -O3 -o tile_march tile_march.hip && ./tile_march
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const float *base, unsigned voff, unsigned lds_dst) {
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void store16(float *base, unsigned voff, f4 d) {
	asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 2" ::"v"(voff), "v"(d), "s"(base));
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// TX, TY: tile; NT threads; HX: x halo (multiple of 4); HW: y/z halo; NOUT output volumes; PAD: extra LDS bytes (caps the residency)
template <int TX, int TY, int NT, int HX, int HW, int NOUT, int PAD, int CEN>
__global__ void __launch_bounds__(NT) k_tile(const float *__restrict__ src, float *__restrict__ d0, float *__restrict__ d1, int nx, int ny, int nz, int ntx, int nty, int cz) {
	constexpr int W4 = (TX + 2 * HX) / 4, R = TY + 2 * HW, ITEMS = W4 * R, NW = NT / 64;
	constexpr int NDMA = (ITEMS + NT - 1) / NT;
	constexpr int TILE_B = ((ITEMS * 16 + 1023) / 1024) * 1024;
	constexpr int PIECES = TX * TY / 4, PPT = PIECES / NT;  // 16-byte output pieces per thread
	static_assert(PIECES % NT == 0, "pieces per thread");
	__shared__ __attribute__((aligned(1024))) char tile[2 * TILE_B + PAD];
	__shared__ __attribute__((aligned(1024))) f4 cenb[CEN ? 2 * NT * PPT : 1];
	const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
	int lb;
	{
		const int nblocks = gridDim.x, b = blockIdx.x, xcd = b & 7, idx = b >> 3;
		const int per = nblocks >> 3, rem = nblocks & 7;
		lb = xcd * per + min(xcd, rem) + idx;
	}
	const int tile_x = lb % ntx, tile_y = (lb / ntx) % nty, chunk = lb / (ntx * nty);
	const int x0 = tile_x * TX, y0 = tile_y * TY;
	const int zc0 = chunk * cz, zc1 = min(nz, zc0 + cz);
	const int sy = nx, sz = nx * ny;
	unsigned goff[NDMA];
#pragma unroll
	for (int i = 0; i < NDMA; i++) {
		const int item = (wid + NW * i) * 64 + lane;
		const int r = item / W4, c = item - r * W4;
		const int gy = y0 - HW + r, gx = x0 - HX + 4 * c;
		const bool ok = item < ITEMS && gy >= 0 && gy < ny && gx >= 0 && gx + 3 < nx;
		goff[i] = (unsigned)(ok ? gy * sy + gx : y0 * sy + x0) * 4u;
	}
	const unsigned lds_tile = (unsigned)(unsigned long long)tile + (unsigned)wid * 1024u;
	const unsigned lds_cen = (unsigned)(unsigned long long)cenb + (unsigned)wid * 1024u;
	unsigned ooff[PPT];
	int lpos[PPT];
#pragma unroll
	for (int p = 0; p < PPT; p++) {
		const int piece = tid + NT * p, ty = piece / (TX / 4), xq = piece % (TX / 4);
		ooff[p] = (unsigned)((y0 + ty) * sy + x0 + 4 * xq) * 4u;
		lpos[p] = ((ty + HW) * W4 + HX / 4 + xq) * 16;
	}
	const int e_start = zc1 - 1 + HW + ((zc1 - 1 + HW >= nz - 1) ? 1 : 0), e_bot = zc0 - HW;   // the product's ramp: 2 HW (+1) extra planes per chunk
	const int nsteps = e_start - e_bot + 1;
	auto plane = [&](int e) { const int L = e < 0 ? -e : (e > nz - 1 ? 2 * (nz - 1) - e : e); return src + (size_t)sz * (size_t)min(max(L, 0), nz - 1); };
	auto issue = [&](int jn) {
		const float *pl = plane(e_start - jn);
		const unsigned dstb = lds_tile + (unsigned)(jn & 1) * (unsigned)TILE_B;
#pragma unroll
		for (int i = 0; i < NDMA; i++)
			if ((wid + NW * i) * 64 < ITEMS) dma16(pl, goff[i], dstb + (unsigned)(i * NW * 1024));
	};
	issue(0);
	wait_vmcnt<0>();
	__syncthreads();
	f4 acc[PPT];
#pragma unroll
	for (int p = 0; p < PPT; p++) acc[p] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
	for (int j = 0; j < nsteps; j++) {
		f4 cen[PPT];
		if (CEN) {
#pragma unroll
			for (int p = 0; p < PPT; p++) {
				cen[p] = cenb[((j & 1) * PPT + p) * NT + tid];
				const int pn = min(max(e_start - j - HW - 1, 0), nz - 1);
				dma16(src + (size_t)sz * (size_t)pn, ooff[p], lds_cen + (unsigned)((((j + 1) & 1) * PPT + p) * NT * 16));
			}
		}
		if (j + 1 < nsteps) issue(j + 1);
		const int p_loc = e_start - j - HW;  // (output plane "completed" by this feed)
		const bool emit = p_loc >= zc0 && p_loc < zc1;
#pragma unroll
		for (int p = 0; p < PPT; p++) {
			const f4 v = *reinterpret_cast<const f4 *>(tile + (j & 1) * TILE_B + lpos[p]);
			acc[p] = acc[p] * 0.25f + v * 0.75f;
		}
		if (emit) {
#pragma unroll
			for (int p = 0; p < PPT; p++) {
				store16(d0 + (size_t)sz * (size_t)p_loc, ooff[p], acc[p]);
				if (NOUT == 2) store16(d1 + (size_t)sz * (size_t)p_loc, ooff[p], CEN ? acc[p] - cen[p] : acc[p] - 1.0f);
			}
			wait_vmcnt<PPT * NOUT>();
		} else
			wait_vmcnt<0>();
		__syncthreads();
	}
	wait_vmcnt<0>();
}

template <int TX, int TY, int NT, int HX, int HW, int NOUT, int PAD, int CEN = 0>
static void run(const float *s, float *d0, float *d1, int n, int chunks, const char *tag) {
	const int ntx = n / TX, nty = n / TY, cz = (n + chunks - 1) / chunks;
	const int grid = ntx * nty * chunks;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9;
	for (int it = 0; it < 5; it++) {
		hipEventRecord(e0);
		// (each launch reads what the previous one wrote, like the level chain of the product)
		hipLaunchKernelGGL((k_tile<TX, TY, NT, HX, HW, NOUT, PAD, CEN>), dim3(grid), dim3(NT), 0, 0, (it & 1) ? d0 : s, (it & 1) ? const_cast<float *>(s) : d0, d1, n, n, n, ntx, nty, cz);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
	}
	int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_tile<TX, TY, NT, HX, HW, NOUT, PAD, CEN>, NT, 0);
	const double alg = (double)n * n * n * 4 * (1 + NOUT);
	printf("%-40s cen %d tile %3dx%-3d nt %4d hx %d hw %d out %d chunks %2d grid %5d occ %d : %7.1f us  %5.0f GB/s algorithmic\n", tag, CEN, TX, TY, NT, HX, HW, NOUT, chunks, grid, occ, best * 1e3, alg / (best * 1e-3) / 1e9);
	fflush(stdout);
	if (hipGetLastError() != hipSuccess) { printf("launch error\n"); exit(1); }
}

int main() {
	const int n = 512;
	const size_t bytes = (size_t)n * n * n * 4;
	float *s, *d0, *d1; hipMalloc(&s, bytes + (1 << 20)); hipMalloc(&d0, bytes + (1 << 20)); hipMalloc(&d1, bytes);
	{
		float *h = (float *)malloc(bytes);
		unsigned x = 12345u;
		for (size_t i = 0; i < bytes / 4; i++) { x = x * 1664525u + 1013904223u; h[i] = (float)(x >> 8) * (1.0f / 16777216.0f); }
		hipMemcpy(s, h, bytes, hipMemcpyHostToDevice); hipMemcpy(d0, h, bytes, hipMemcpyHostToDevice);
		free(h);
	}
	// the product's geometry: 32 x 32 tiles, 256 threads, three workgroups per CU (52 KB of LDS each)
	constexpr int P3 = 36 * 1024, P2 = 60 * 1024;
	for (int rep = 0; rep < 2; rep++) {
		// LDS pads chosen so that the residency is what the product's LDS / register budget would allow
		run<32, 32, 256, 4, 2, 1, P3>(s, d0, d1, n, 3, "product hw2 (L0), 3 per CU");
		run<64, 32, 512, 4, 2, 1, 40 * 1024>(s, d0, d1, n, 4, "  64x32 / 512 thr, 2 per CU, 512 wgs");
		run<64, 32, 512, 4, 2, 1, 40 * 1024>(s, d0, d1, n, 8, "  64x32 / 512 thr, 2 per CU, 1024 wgs");
		run<64, 32, 512, 4, 2, 1, 24 * 1024>(s, d0, d1, n, 6, "  64x32 / 512 thr, 3 per CU, 768 wgs");
		run<32, 32, 256, 4, 4, 2, P3>(s, d0, d1, n, 3, "product hw4 + DoG, 3 per CU");
		run<32, 32, 256, 4, 4, 2, P3, 1>(s, d0, d1, n, 3, "  same, centre by DMA");
		run<64, 32, 512, 4, 4, 2, 36 * 1024>(s, d0, d1, n, 4, "  64x32 / 512 thr, 2 per CU, 512 wgs");
		run<64, 32, 512, 4, 4, 2, 20 * 1024, 1>(s, d0, d1, n, 4, "  64x32 / 512 thr, 2 per CU, 512 wgs, centre by DMA");
		run<32, 32, 256, 8, 5, 2, P3>(s, d0, d1, n, 3, "product hw5 + DoG, 3 per CU");
		run<32, 32, 256, 8, 5, 2, P3, 1>(s, d0, d1, n, 3, "  same, centre by DMA");
		run<32, 32, 256, 8, 5, 2, 50 * 1024, 1>(s, d0, d1, n, 2, "  same, centre by DMA, 2 per CU, 512 wgs (hw 6 form)");
		run<64, 32, 512, 8, 5, 2, 36 * 1024>(s, d0, d1, n, 4, "  64x32 / 512 thr, 2 per CU, 512 wgs");
		run<64, 32, 512, 8, 5, 2, 20 * 1024, 1>(s, d0, d1, n, 4, "  64x32 / 512 thr, 2 per CU, 512 wgs, centre by DMA");
		run<64, 32, 512, 8, 5, 2, 20 * 1024, 1>(s, d0, d1, n, 8, "  64x32 / 512 thr, 2 per CU, 1024 wgs, centre by DMA");
		run<128, 16, 512, 8, 5, 2, 20 * 1024, 1>(s, d0, d1, n, 4, "  128x16 / 512 thr, 2 per CU, 512 wgs, centre by DMA");
	}
	return 0;
}
