// kernels_detect.hip -- DoG extrema scan with deterministic, order-preserving compaction.
//
// Restates Detect_KeyPoints + IsExtrema_neighbor (reference Src/cSIFT3D.cc:362-425, 884-911):
// for x,y,z in [1, n-2]: (val > thr || val < -thr) and val strictly below (or strictly above) all
// EIGHT neighbours: +-x, +-y, +-z in the current DoG level and the centre voxel of the previous and
// next level.  thr = peak_thresh * max|level| (max from the DoG-producing kernel, fp32 multiply).
//
// The reference emits extrema in (octave, level, z, y, x) scan order and the matcher output order
// depends on it, so instead of an atomic append + sort the scan is made order-preserving.  One launch
// of each kernel covers the THREE keypoint levels of an octave:
//   k_mark : block = (level, z, 16 rows) -- a contiguous range in scan order; lanes run along x (coalesced, no
//            index divisions); each wave stores the 64-bit ballot of its 64 voxels, each block its hit count
//   k_scan : one workgroup turns the block counts into exclusive offsets on top of the running total
//   k_emit : one thread per ballot word; words with hits (rare) write their extrema at offset + rank
// No host synchronisation anywhere; the running total stays on the device.
#include "sift3d_internal.h"

namespace s3d {

constexpr int kRows = 16;   // rows per block
constexpr int kThreads = 256;

// masks layout: word index = ((lvl * nz + z) * ny + y) * wpr + xw, wpr = ceil(nx / 64): scan order
__global__ void __launch_bounds__(kThreads) k_mark(DetectLevels L, int nx, int ny, ZRange zr, int nyb, float peak_thresh,
                                                   unsigned long long *__restrict__ masks, unsigned *__restrict__ block_counts) {
	__shared__ unsigned s_cnt[kThreads / 64];
	const int b = blockIdx.x;
	const int nz = zr.zo1 - zr.zo0;                              // planes scanned by this launch (local range [zo0, zo1))
	const int yb = b % nyb, zi = (b / nyb) % nz, lvl = b / (nyb * nz);
	const int z = zr.zo0 + zi;                                   // local plane in the buffers
	const int zg = z + zr.zoff;                                  // global plane (border rule, Src/cSIFT3D.cc:388)
	const float *__restrict__ cur = L.cur[lvl];
	const float *__restrict__ prev = L.prev[lvl];
	const float *__restrict__ next = L.next[lvl];
	const float thr = peak_thresh * __uint_as_float(*L.absmax_bits[lvl]);
	const int wpr = (nx + 63) >> 6;
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
	unsigned cnt = 0;
	const bool z_in = (zg >= 1 && zg <= zr.nzg - 2);
	const int y0 = yb * kRows;
	// The block's 16 rows are dealt to its 4 waves 4 rows each; a wave walks a row kBatch ballot words at a time: the
	// centre values of the kBatch words are requested together and only then examined, so a wave has kBatch independent
	// HBM reads in flight (the scan is a pure streaming read; with one load per iteration it ran at the memory latency).
	// Row / word indices are wave-uniform loop counters: no per-lane integer divisions.
	constexpr int kBatch = 8;
	const int nrows = min(kRows, ny - y0);
	const int swid = __builtin_amdgcn_readfirstlane(wid);
	for (int ry = swid * (kRows / 4); ry < min(nrows, (swid + 1) * (kRows / 4)); ry++) {
		const int y = y0 + ry;
		const bool y_in = z_in && y >= 1 && y <= ny - 2;
		const size_t rowbase = sy * (size_t)y + sz * (size_t)z;
		unsigned long long *mrow = masks + ((size_t)(lvl * nz + zi) * ny + y) * wpr;
		for (int xw0 = 0; xw0 < wpr; xw0 += kBatch) {
			float val[kBatch];
			bool in[kBatch];
#pragma unroll
			for (int b = 0; b < kBatch; b++) {
				const int x = (xw0 + b) * 64 + lane;
				in[b] = y_in && x >= 1 && x <= nx - 2;  // implies xw0 + b < wpr
				val[b] = cur[in[b] ? rowbase + (size_t)x : sz * (size_t)z];  // unconditional load, clamped address
			}
#pragma unroll
			for (int b = 0; b < kBatch; b++) {
				if (xw0 + b >= wpr) break;  // wave-uniform
				bool hit = false;
				const float v = val[b];
				if (in[b] && (v > thr || v < -thr)) {
					const size_t i = rowbase + (size_t)((xw0 + b) * 64 + lane);
					const float n0 = prev[i], n1 = cur[i - 1], n2 = cur[i + 1], n3 = cur[i + sy], n4 = cur[i - sy], n5 = cur[i + sz],
					            n6 = cur[i - sz], n7 = next[i];
					const bool mn = v < n0 && v < n1 && v < n2 && v < n3 && v < n4 && v < n5 && v < n6 && v < n7;
					const bool mx = v > n0 && v > n1 && v > n2 && v > n3 && v > n4 && v > n5 && v > n6 && v > n7;
					hit = mn || mx;
				}
				const unsigned long long m = __ballot(hit);
				if (lane == 0) mrow[xw0 + b] = m;
				cnt += (unsigned)__popcll(m);
			}
		}
	}
	if (lane == 0) s_cnt[wid] = cnt;
	__syncthreads();
	if (threadIdx.x == 0) block_counts[b] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// exclusive scan of block_counts[0..nblocks) by ONE workgroup; offsets start at the running total
// total[0], which is then advanced.
__global__ void __launch_bounds__(1024) k_scan(const unsigned *__restrict__ block_counts, unsigned *__restrict__ block_offsets,
                                               unsigned nblocks, unsigned *__restrict__ total) {
	__shared__ unsigned s_wave[16];
	const unsigned t = threadIdx.x;
	const unsigned chunk = (nblocks + 1023u) / 1024u;
	const unsigned lo = min(t * chunk, nblocks), hi = min(lo + chunk, nblocks);
	unsigned sum = 0;
	for (unsigned i = lo; i < hi; i++) sum += block_counts[i];
	unsigned v = sum;
	const int lane = t & 63, wid = t >> 6;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		unsigned u = __shfl_up(v, o, 64);
		if (lane >= o) v += u;
	}
	if (lane == 63) s_wave[wid] = v;
	__syncthreads();
	if (t == 0) {
		unsigned a = 0;
		for (int w = 0; w < 16; w++) { unsigned c = s_wave[w]; s_wave[w] = a; a += c; }
	}
	__syncthreads();
	const unsigned incl = v + s_wave[wid];
	const unsigned base = total[0];
	unsigned run = base + (incl - sum);
	for (unsigned i = lo; i < hi; i++) { block_offsets[i] = run; run += block_counts[i]; }
	__syncthreads();
	if (t == 1023) total[0] = base + incl;
}

// one thread per ballot word of a block (16 rows x wpr words, same block decomposition as k_mark)
__global__ void __launch_bounds__(kThreads) k_emit(const unsigned long long *__restrict__ masks,
                                                   const unsigned *__restrict__ block_offsets, int nx, int ny, ZRange zr, int nyb,
                                                   int octave, DetectLevels L, DevKp *__restrict__ out, unsigned cap,
                                                   unsigned *__restrict__ total) {
	__shared__ unsigned s_wave[kThreads / 64];
	const int b = blockIdx.x;
	const int nz = zr.zo1 - zr.zo0;
	const int yb = b % nyb, zi = (b / nyb) % nz, lvl = b / (nyb * nz);
	const int z = zr.zo0 + zi + zr.zoff;  // GLOBAL plane: keypoint coordinates are global
	const int wpr = (nx + 63) >> 6;
	const int y0 = yb * kRows;
	const int nwords = min(kRows, ny - y0) * wpr;
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	unsigned running = block_offsets[b];
	// words are consumed in scan order, 256 at a time; the rank of a word = hits of all earlier words
	for (int w0 = 0; w0 < nwords; w0 += kThreads) {
		const int wi = w0 + threadIdx.x;
		unsigned long long m = 0;
		int y = 0, xw = 0;
		if (wi < nwords) {
			const int ry = wi / wpr;
			xw = wi - ry * wpr;
			y = y0 + ry;
			// k_mark dealt the words to waves round-robin; the layout in memory is scan order
			m = masks[((size_t)(lvl * nz + zi) * ny + y) * wpr + xw];
		}
		const unsigned c = (unsigned)__popcll(m);
		unsigned v = c;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			unsigned u = __shfl_up(v, o, 64);
			if (lane >= o) v += u;
		}
		__syncthreads();
		if (lane == 63) s_wave[wid] = v;
		__syncthreads();
		unsigned before = v - c;
		for (int w = 0; w < wid; w++) before += s_wave[w];
		const unsigned all = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
		unsigned pos = running + before;
		while (m) {
			const int bit = __ffsll((long long)m) - 1;
			m &= m - 1;
			if (pos < cap) {
				DevKp k;
				k.x = xw * 64 + bit; k.y = y; k.z = z;
				k.octave = octave; k.level = L.level_id[lvl]; k.scale = L.scale[lvl]; k.code = 0; k.slot = -1;
#pragma unroll
				for (int j = 0; j < 3; j++) { k.win[j] = 0.f; k.eigvalue[j] = 0.f; }
#pragma unroll
				for (int j = 0; j < 9; j++) { k.eigvector[j] = 0.f; k.rot[j] = 0.f; k.st[j] = 0.f; }
				out[pos] = k;
			} else {
				total[1] = 1u;  // overflow flag; the host regrows and reruns
			}
			pos++;
		}
		running += all;
	}
}

void launch_detect_octave(const DetectLevels &L, int nlevels, int nx, int ny, const ZRange &zr, float peak_thresh, int octave,
                          const DetectBufs &b, DevKp *out, unsigned cap, hipStream_t st) {
	const int nyb = (ny + kRows - 1) / kRows;
	const int nzl = zr.zo1 - zr.zo0;
	if (nzl <= 0) return;
	const unsigned nblocks = (unsigned)(nlevels * nzl * nyb);
	if (nblocks == 0) return;
	hipLaunchKernelGGL(k_mark, dim3(nblocks), dim3(kThreads), 0, st, L, nx, ny, zr, nyb, peak_thresh, b.masks, b.block_counts);
	hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, b.block_counts, b.block_offsets, nblocks, b.total);
	hipLaunchKernelGGL(k_emit, dim3(nblocks), dim3(kThreads), 0, st, b.masks, b.block_offsets, nx, ny, zr, nyb, octave, L, out, cap,
	                   b.total);
}

}  // namespace s3d
