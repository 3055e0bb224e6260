// kernels_small.hip -- the SMALL octaves of a pyramid (at most 16 x 16 x 16-class levels) in ONE launch: one workgroup keeps the octave
// in LDS and walks level after level, octave after octave.
//
// Replaces, for those octaves, the level loop of Build_Gaussian_Scale_Space / Build_DOG_Scale_Space (Src/cSIFT3D.cc:268-360) --
// GaussianSmooth_3D(_Imp) + Im_permute + Sub + im_max_abs + DownSample_3D (Src/cSIFT3D.cc:506-882, Src/cUtil.cc:587-605) -- which r01-r03
// ran as three separable launches per level plus a decimation launch per octave: ~16 launches of 5-8 us per octave on the stage's
// critical chain octave -> octave (the last two octaves of a 512^3 volume: 0.15 ms of the 2.25 ms stage on a nearly idle machine; the
// same chain bounds 128^3 / 256^3 volumes and the replicated tails of the multi-GPU path).  A single workgroup needs no cross-workgroup
// hand-off: every dependency is a workgroup barrier.
//
// Arithmetic contract (the same as kernels_pyramid.hip / kernels_march.hip; bit-identical levels): every output is the literal chain
// acc = acc + tap[d+hw] * term(p - d) for d = -hw..+hw with separate IEEE multiply and add (-ffp-contract=off); boundary terms come from
// the EXTENDED line E[] of kernels_march.hip's header (E[-k] = src[k]; E[dim_end+k] = (1-f_k) src[dim_end-k-1] + f_k src[dim_end-k]),
// valid while hw <= n - 2 on every axis (checked on the host).
//
// Form: one thread owns eight consecutive outputs of one line (along x, y or z).  The terms of output p arrive in the order
// E[p+hw] .. E[p-hw], so a thread that walks its extended segment DOWNWARDS can scatter: E[c] contributes tap[p-c+hw] * E[c] to every
// output p within hw of c, and because the taps are symmetric (tap[hw+d] == tap[hw-d] bit for bit: checked on the host) the product of
// E[c] with tap[hw-|d|] serves the outputs c+d and c-d: hw+1 multiplies and <= 2hw+1 adds per term instead of 2hw+1 of each.  Tasks are
// dealt segment-major with the line count padded to whole waves, so a wave's segment -- and with it every boundary decision -- is
// wave-uniform (scalar branches, no per-lane predicates).  Passes are out of place between two LDS images (x pitch odd: conflict-free
// for lines along x, y and z); the thread that owns a z segment keeps the previous level's values of that segment in registers for the
// DoG (Sub: (cur - prev) * (-1), Src/cSIFT3D.cc:875).
#include <string.h>

#include <algorithm>

#include "sift3d_internal.h"

#pragma clang fp contract(off)

namespace s3d {

namespace {
constexpr int kNOUT = 4;             // outputs per task (8: 512 tasks of ~1000 instructions per pass at 16^3 -- issue-bound on two waves per SIMD, 3.2 us per pass)
constexpr int kThreads = 1024;
constexpr int kRounds = 1;           // tasks per thread and pass (at most kRounds * kThreads tasks per pass)
constexpr int kCap = 4608;           // floats per LDS image: 16 x 16 x 16 with an x pitch of 17, and ragged octaves of that class
constexpr int kSeedCap = kCap / 8 + 64;
constexpr int kSMaxHW = 8;
constexpr int kMaxDim = 32;                       // longest line (the image capacity allows e.g. 32 x 12 x 11)
constexpr int kTabLen = kMaxDim + 2 * kSMaxHW + 4;  // coordinate table entries per axis: c = -8 .. n + 8

__device__ __forceinline__ float s_absmax(float m, float v) {
	const float a = fabsf(v);
	return (a > m) ? a : m;
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
// LDS-only barrier: __syncthreads() is a fence over ALL memory and would wait for the write acknowledgements of the level's global
// stores twice per level (2 x ~2 us of the 9.5 us a level took); nothing in this kernel reads back what it stored to global memory
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// untracked global store, SGPR base + 32-bit VGPR byte offset (hipcc would make later writes of the data register wait for the
// store to complete; the wait state covers the hardware's read of the data register)
__device__ __forceinline__ void g_store(float *base, unsigned byte_off, float v) {
	asm volatile("global_store_dword %0, %1, %2\n\ts_nop 0" ::"v"(byte_off), "v"(v), "s"(base));
}
typedef float __attribute__((address_space(1))) *gmem_p;  // global address space: global_load / global_store, never FLAT
__device__ __forceinline__ gmem_p as_gmem(float *p) { return (gmem_p)p; }

struct SmallGeom {
	int nx, ny, nz, px, pz;  // pz = px * ny
};

// One Gaussian level of the octave held in `in` (LDS image): x pass in -> tmp, y pass tmp -> in, z pass in -> tmp (+ global stores,
// DoG, abs-max, decimated seed).  On return `tmp` holds the new level.
template <int HW>
__device__ __forceinline__ void small_level(const SmallArgs &a, int o, int i, const SmallGeom &G, float *lds, int in, int tmp, int seedS,
                                            const int2 (*s_tab)[kTabLen], float (&prev)[kRounds][kNOUT], float &mx) {
	const int tid = threadIdx.x;
	float w[HW + 1];
#pragma unroll
	for (int k = 0; k <= HW; k++) w[k] = a.w[i][k];
	float *g_out = a.oct[o].g[i];
	const bool has_dog = ((a.dog_mask >> (i - 1)) & 1u) != 0;
	float *d_out = a.oct[o].dog[i - 1];
	const bool seed_level = i == a.seed && o + 1 < a.noct;
	float *next0 = seed_level ? a.oct[o + 1].g[0] : nullptr;
	const int nx2 = seed_level ? a.oct[o + 1].nx : 0, ny2 = seed_level ? a.oct[o + 1].ny : 0, nz2 = seed_level ? a.oct[o + 1].nz : 0;

#pragma unroll 1
	for (int pass = 0; pass < 3; pass++) {
		const float *src = lds + (pass == 1 ? tmp : in);  // (integer offsets into ONE LDS array: the accesses stay ds_read / ds_write)
		float *dst = lds + (pass == 1 ? in : tmp);
		// (everything that decides a branch below is forced into scalar registers: the decisions are wave-uniform by construction)
		const int n = rfl(pass == 0 ? G.nx : (pass == 1 ? G.ny : G.nz));
		const int stride = rfl(pass == 0 ? 1 : (pass == 1 ? G.px : G.pz));
		const int nlines = rfl(pass == 0 ? G.ny * G.nz : (pass == 1 ? G.nx * G.nz : G.nx * G.ny));
		const int lp = (nlines + 63) & ~63, nseg = (n + kNOUT - 1) / kNOUT;
		const int2 *tab = s_tab[pass];
#pragma unroll
		for (int r = 0; r < kRounds; r++) {
			const int t = tid + r * kThreads;
			const int seg = rfl(t / lp);     // wave-uniform: lp and kThreads are multiples of 64
			if (seg >= nseg) continue;       // uniform
			const int line = t - seg * lp;
			const bool valid = line < nlines;
			int base, lx = 0, ly = 0;
			if (pass == 0) base = line * G.px;                 // line = z * ny + y
			else if (pass == 1) { const int z = line / G.nx; lx = line - z * G.nx; base = z * G.pz + lx; }
			else { ly = line / G.nx; lx = line - ly * G.nx; base = ly * G.px + lx; }
			if (!valid) base = 0;
			const int s0 = seg * kNOUT;
			// ---- the extended segment E[s0 - HW .. s0 + NOUT - 1 + HW], branch-free, from the per-axis coordinate table (see k_small_octaves):
			// R[j] = src[refl(c_j)] with refl the reflection at both ends (|c| below 0, 2 dim_end - c above dim_end), so for a term beyond
			// the right end, c = dim_end + k, R[j] = src[dim_end - k] and R[j + 1] = src[dim_end - k - 1]: E = (1 - f_k) R[j+1] + f_k R[j].
			// Two batches of unconditional LDS reads (table entries at wave-uniform addresses, then the samples), then selects: a load
			// inside a wave-uniform branch would be waited for at the branch's join -- one LDS round trip per term.
			int2 T[kNOUT + 2 * HW + 1];
#pragma unroll
			for (int j = 0; j <= kNOUT + 2 * HW; j++) T[j] = tab[s0 - HW + j + kSMaxHW];  // (entries exist for c in [-8, n + 8])
			float R[kNOUT + 2 * HW + 1];
#pragma unroll
			for (int j = 0; j <= kNOUT + 2 * HW; j++) R[j] = src[base + T[j].x * stride];
			float E[kNOUT + 2 * HW];
#pragma unroll
			for (int j = 0; j < kNOUT + 2 * HW; j++) {
				const float fk = __int_as_float(T[j].y);  // f_k, or a negative value for the terms that are plain samples
				const float lerp = (1.0f - fk) * R[j + 1] + fk * R[j];
				E[j] = fk >= 0.0f ? lerp : R[j];
			}
			// ---- scatter, highest term first ----
			float out[kNOUT];
#pragma unroll
			for (int p = 0; p < kNOUT; p++) out[p] = 0.0f;
#pragma unroll
			for (int j = kNOUT + 2 * HW - 1; j >= 0; j--) {
				float m[HW + 1];
#pragma unroll
				for (int k = 0; k <= HW; k++) m[k] = w[k] * E[j];
#pragma unroll
				for (int p = 0; p < kNOUT; p++) {
					const int d = p + HW - j;  // output p - term coordinate
					if (d >= -HW && d <= HW) out[p] = out[p] + m[HW - (d < 0 ? -d : d)];
				}
			}
			if (!valid) continue;
			if (pass < 2) {
#pragma unroll
				for (int p = 0; p < kNOUT; p++)
					if (s0 + p < n) dst[base + (s0 + p) * stride] = out[p];
			} else {
#pragma unroll
				for (int p = 0; p < kNOUT; p++) {
					const int z = s0 + p;
					if (z < n) {
						const float v = out[p];
						const unsigned gi = (unsigned)((z * G.ny + ly) * G.nx + lx) * 4u;
						dst[base + z * stride] = v;
						g_store(g_out, gi, v);
						if (has_dog) {
							const float dg = (v - prev[r][p]) * (-1.0f);
							g_store(d_out, gi, dg);
							mx = s_absmax(mx, dg);
						}
						prev[r][p] = v;
						// DownSample_3D (Src/cSIFT3D.cc:506-533): level 0 of the next octave = every second voxel of the seed level
						if (seed_level && ((z | ly | lx) & 1) == 0) {
							const int hz = z >> 1, hy = ly >> 1, hx = lx >> 1;
							if (hz < nz2 && hy < ny2 && hx < nx2) {
								const int si = (hz * ny2 + hy) * nx2 + hx;
								g_store(next0, (unsigned)si * 4u, v);
								lds[seedS + si] = v;
							}
						}
					}
				}
			}
		}
		lds_barrier();
	}
}

}  // namespace

__global__ void __launch_bounds__(kThreads) k_small_octaves(SmallArgs a) {
	__shared__ float lds[2 * kCap + kSeedCap];  // two images of the octave (the passes go back and forth) + the decimated seed of the next one
	constexpr int bufA = 0, bufB = kCap, seedS = 2 * kCap;
	__shared__ int2 s_tab[3][kTabLen];
	__shared__ float s_red[kThreads / 64];
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	__builtin_amdgcn_s_setprio(3);  // the launch sits on the stage's critical chain, beside machine-filling launches of the big octaves

#pragma unroll 1
	for (int o = 0; o < a.noct; o++) {
		SmallGeom G;
		G.nx = rfl(a.oct[o].nx); G.ny = rfl(a.oct[o].ny); G.nz = rfl(a.oct[o].nz);
		G.px = G.nx | 1; G.pz = G.px * G.ny;
		// per-axis coordinate table of the extended line (Src/cSIFT3D.cc:722-788): entry e <-> coordinate c = e - 8;
		//   .x = the sample index refl(c) (reflection at 0 and at dim_end, clamped into the line)
		//   .y = bits of the right-boundary fraction f_k for c = dim_end + k (k = 0..8; the reference's fp32 arithmetic), -1.0f otherwise
		if (tid < 3 * kTabLen) {
			const int axis = tid / kTabLen, e = tid - axis * kTabLen;
			const int n = axis == 0 ? G.nx : (axis == 1 ? G.ny : G.nz), dim_end = n - 1;
			const int c = e - kSMaxHW;
			int idx = c < 0 ? -c : c;
			if (idx > dim_end) idx = 2 * dim_end - idx;
			idx = idx < 0 ? 0 : (idx > dim_end ? dim_end : idx);
			float f = -1.0f;
			if (c >= dim_end) {
				const int k = c - dim_end;
				const float cf = (float)(dim_end + k);
				const float cc = (float)(2 * dim_end) - cf - 0.1f;
				const int lo = (int)cc;
				f = cc - (float)lo;
			}
			s_tab[axis][e] = make_int2(idx, __float_as_int(f));
		}
		// level 0 of the octave: from global memory (written by the kernels of the octave above) for the first octave of the launch,
		// from the seed image the previous octave left in LDS otherwise; the z-segment owner keeps it as `prev`
		float prev[kRounds][kNOUT];
		{
			const int nlines = G.nx * G.ny, lp = (nlines + 63) & ~63, nseg = (G.nz + kNOUT - 1) / kNOUT;
			gmem_p g0 = as_gmem(a.oct[o].g[0]);
			// level 0 of the launch's first octave may still have to be formed: DownSample_3D (Src/cSIFT3D.cc:506-533) of the parent octave's
			// seed level (the parent's kernel wrote it already when its rows are whole 16-byte pieces: kernels_march.hip `half`)
			const bool decimate = o == 0 && a.parent != nullptr;
			gmem_p par = as_gmem(const_cast<float *>(a.parent));
			const int pnx = rfl(a.pnx), pny = rfl(a.pny);
#pragma unroll
			for (int r = 0; r < kRounds; r++) {
				const int t = tid + r * kThreads;
				const int seg = rfl(t / lp);
				const int line = t - seg * lp;
				const int ly = line / G.nx, lx = line - ly * G.nx;
				const bool mine = seg < nseg && line < nlines;
				float v[kNOUT];
#pragma unroll
				for (int p = 0; p < kNOUT; p++) {  // (all loads of the segment in flight together)
					const int z = seg * kNOUT + p;
					const int gi = (mine && z < G.nz) ? (z * G.ny + ly) * G.nx + lx : 0;
					v[p] = 0.0f;
					if (decimate) v[p] = par[(mine && z < G.nz) ? ((unsigned)(2 * z) * pny + 2 * ly) * pnx + 2 * lx : 0u];
					else if (o == 0) v[p] = g0[gi];
					else v[p] = lds[seedS + gi];
				}
#pragma unroll
				for (int p = 0; p < kNOUT; p++) {
					const int z = seg * kNOUT + p;
					if (mine && z < G.nz) {
						lds[bufA + z * G.pz + ly * G.px + lx] = v[p];
						if (decimate) g_store(a.oct[o].g[0], (unsigned)((z * G.ny + ly) * G.nx + lx) * 4u, v[p]);
					}
					prev[r][p] = v[p];
				}
			}
		}
		lds_barrier();
		int in = bufA, tmp = bufB;
#pragma unroll 1
		for (int i = 1; i < a.ng; i++) {
			if (!((a.build_mask >> i) & 1u)) continue;
			float mx = 0.0f;
			// ONE instantiation serves every level: the taps of a narrower kernel are padded with zeros to the launch's half width
			// (a term 0 * E adds +-0 to a sum that is never -0: the chain's value is unchanged bit for bit).  Straight-line code that runs
			// once is bound by instruction fetch: four half-width instantiations (44 KB touched once each) took 76 us for the two last
			// octaves of a 512^3 volume.
			if (a.hwp <= 6) small_level<6>(a, o, i, G, lds, in, tmp, seedS, s_tab, prev, mx);
			else small_level<8>(a, o, i, G, lds, in, tmp, seedS, s_tab, prev, mx);
			{ const int t2 = in; in = tmp; tmp = t2; }
			if ((a.dog_mask >> (i - 1)) & 1u) {  // max |DoG[i-1]| (im_max_abs, Src/cUtil.cc:587-605)
#pragma unroll
				for (int s = 32; s > 0; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s, 64));
				if (lane == 0) s_red[wid] = mx;
				lds_barrier();
				if (tid == 0) {
					float rmax = s_red[0];
					for (int k = 1; k < kThreads / 64; k++) rmax = fmaxf(rmax, s_red[k]);
					if (rmax > 0.0f) atomicMax(a.dogmax + (size_t)o * a.nd + (i - 1), __float_as_uint(rmax));
				}
				lds_barrier();
			}
		}
	}
}

// can the levels of an nx x ny x nz octave (and of every smaller octave behind it) run in k_small_octaves?
bool small_octave_fits(int nx, int ny, int nz, int max_hw) {
	max_hw = small_padded_hw(max_hw);
	if (max_hw < 0) return false;
	if (nx < 1 || ny < 1 || nz < 1 || nx > kMaxDim || ny > kMaxDim || nz > kMaxDim) return false;
	const int px = nx | 1;
	if ((size_t)px * ny * nz > (size_t)kCap) return false;
	if (max_hw > std::min(nx, std::min(ny, nz)) - 2) return false;  // extended-line form of the boundary rule
	auto tasks = [](int n, int lines) { return ((n + kNOUT - 1) / kNOUT) * ((lines + 63) & ~63); };
	const int cap = kRounds * kThreads;
	if (tasks(nx, ny * nz) > cap || tasks(ny, nx * nz) > cap || tasks(nz, nx * ny) > cap) return false;
	if ((size_t)(nx / 2) * (ny / 2) * (nz / 2) > (size_t)kSeedCap) return false;
	return true;
}

int small_padded_hw(int max_hw) { return max_hw <= 6 ? 6 : (max_hw <= 8 ? 8 : -1); }  // the half widths k_small_octaves is instantiated for

void launch_small_octaves(const SmallArgs &a, hipStream_t st) {
	hipLaunchKernelGGL(k_small_octaves, dim3(1), dim3(kThreads), 0, st, a);
}

void preload_small_kernels() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_small_octaves)); }

}  // namespace s3d
