// staging.hip -- host <-> device copies of pageable host memory through a small pinned, double-buffered staging pool.
//
// CreateCSIFT3D(float*) hands over a pageable volume (Src/cSIFT3D.cc:146-163 copies it with memcpy) and GetKeypoints returns
// pageable vectors (Src/cSIFT3D.cc:1686-1688).  One hipMemcpy of pageable memory moved 0.5-0.8 GB/s on the MI355X boxes (r02:
// 650-990 ms for the 512 MB of a 512^3 volume); here the host side of the copy is a multi-threaded memcpy into / out of pinned
// chunks that overlap with the DMA of the previous chunk.  Host only, no kernels.
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sift3d_internal.h"

namespace s3d {

namespace {
constexpr size_t kChunk = (size_t)16 << 20;  // bytes per pinned buffer
constexpr int kBufs = 4;
constexpr int kSub = 4;  // the device -> host direction deals every buffer as kSub pieces (staged_d2h)
#ifndef S3D_D2H_THREADS
#define S3D_D2H_THREADS 12
#endif
constexpr int kD2HThreads = S3D_D2H_THREADS;
// One pool per device (r04; before: one process-wide pool behind one mutex whose events were re-created whenever the calling device
// changed -- the rank threads of the sharded driver took turns through it inside the timed region): 4 x 16 MB of pinned memory and
// four events per device that is ever used, created on first use by a thread whose current device is that device.
struct Pool {
	std::mutex mu;
	char *buf[kBufs] = {};
	hipEvent_t ev[kBufs] = {};
	hipEvent_t ev2[kBufs * kSub] = {};  // one per piece of the device -> host direction
};
constexpr int kMaxDev = 64;
Pool g_pools[kMaxDev];
Pool &pool_of(int device) { return g_pools[(device >= 0 && device < kMaxDev) ? device : 0]; }

int pool_ready(Pool &P) {  // the caller holds P.mu and has made the pool's device current
	for (int i = 0; i < kBufs; i++) {
		if (!P.buf[i]) S3D_HIP(hipHostMalloc(reinterpret_cast<void **>(&P.buf[i]), kChunk, hipHostMallocPortable));
		if (!P.ev[i]) S3D_HIP(hipEventCreateWithFlags(&P.ev[i], hipEventDisableTiming));
	}
	for (int i = 0; i < kBufs * kSub; i++)
		if (!P.ev2[i]) S3D_HIP(hipEventCreateWithFlags(&P.ev2[i], hipEventDisableTiming));
	return SIFT3D_OK;
}

}  // namespace

// Thread t of nt copies bytes [*o, *e) of an n-byte chunk: page-aligned parts of CEIL(n / nt) bytes, so that nt parts always cover
// the chunk (r05 took the floor: with n / nt a multiple of 4096 and n % nt != 0 the last n % nt bytes of a chunk were never copied --
// a 195 x 273 x 315 volume lost its last voxel to whatever an earlier upload had left in the pinned buffer).  tests/test_cabi_cpu.py
// walks the coverage through sift3d_test_staging_slice.
void staging_slice(size_t n, int nt, int t, size_t *o, size_t *e) {
	if (nt < 1) nt = 1;
	const size_t part = (((n + (size_t)nt - 1) / (size_t)nt) + 4095) & ~(size_t)4095;
	*o = std::min(n, part * (size_t)t);
	*e = std::min(n, *o + part);
}

// host (pageable) -> device; returns after every byte has been handed to the stream (the copies are still in flight: stream-ordered).
// r05: the copy INTO the pinned chunks was the bound (r04: four helper threads spawned per 16 MB chunk, 33.7 GB/s for the 512 MB of a
// 512^3 volume against ~55 of the DMA): now the copy threads live for the whole call, each fills its slice of chunk after chunk, four
// chunks rotate, and the calling thread only hands filled chunks to the stream and frees drained ones.
int staged_h2d(void *d_dst, const void *h_src, size_t bytes, int device, hipStream_t st) {
	if (bytes == 0) return SIFT3D_OK;
	if (bytes < ((size_t)1 << 20)) { S3D_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, st)); return SIFT3D_OK; }
	Pool &g_pool = pool_of(device);
	std::lock_guard<std::mutex> lock(g_pool.mu);
	int rc = pool_ready(g_pool);
	if (rc) return rc;
	const size_t nchunks = (bytes + kChunk - 1) / kChunk;
	const unsigned hc = std::thread::hardware_concurrency();
	const int nt = (int)std::min<size_t>(std::max(1u, std::min(8u, hc ? hc / 2 : 1u)), std::max<size_t>(1, bytes >> 22));
	// chunk i may be filled once chunk i - kBufs has left its buffer (freed counts the drained chunks); filled[i] counts the threads done with it
	std::atomic<size_t> freed{0};
	std::atomic<bool> stop{false};
	std::vector<std::atomic<int>> filled(nchunks);
	for (auto &f : filled) f.store(0);
	auto worker = [&](int t) {
		for (size_t i = 0; i < nchunks && !stop.load(std::memory_order_relaxed); i++) {
			while (i >= freed.load(std::memory_order_acquire) + (size_t)kBufs) {
				if (stop.load(std::memory_order_relaxed)) return;
				std::this_thread::yield();
			}
			const size_t off = i * kChunk, n = std::min(kChunk, bytes - off);
			size_t o, e;
			staging_slice(n, nt, t, &o, &e);
			if (e > o) memcpy(g_pool.buf[i % kBufs] + o, static_cast<const char *>(h_src) + off + o, e - o);
			filled[i].fetch_add(1, std::memory_order_release);
		}
	};
	std::vector<std::thread> th;
	for (int t = 1; t < nt; t++) th.emplace_back(worker, t);
	hipError_t err = hipSuccess;
	size_t issued = 0, drained = 0;
	auto drain_one = [&]() -> hipError_t {  // the DMA that read the oldest chunk still in a buffer is done: its buffer is free
		const hipError_t e = hipEventSynchronize(g_pool.ev[drained % kBufs]);
		drained++;
		freed.store(drained, std::memory_order_release);
		return e;
	};
	// the calling thread is copy thread 0 as well: it fills its slice of a chunk when the buffer is free, hands complete chunks to the stream
	for (size_t i = 0; i < nchunks && err == hipSuccess; i++) {
		while (i >= drained + (size_t)kBufs && err == hipSuccess) err = drain_one();
		if (err != hipSuccess) break;
		const size_t off = i * kChunk, n = std::min(kChunk, bytes - off);
		size_t o0, e0;
		staging_slice(n, nt, 0, &o0, &e0);
		memcpy(g_pool.buf[i % kBufs], static_cast<const char *>(h_src) + off, e0);
		filled[i].fetch_add(1, std::memory_order_release);
		while (filled[i].load(std::memory_order_acquire) < nt) std::this_thread::yield();
		err = hipMemcpyAsync(static_cast<char *>(d_dst) + off, g_pool.buf[i % kBufs], n, hipMemcpyHostToDevice, st);
		if (err == hipSuccess) err = hipEventRecord(g_pool.ev[i % kBufs], st);
		if (err == hipSuccess) issued = i + 1;
	}
	stop.store(true);
	for (auto &t : th) t.join();
	// the pool is free for the next caller.  Only the events this call recorded: another one may have been recorded last on a stream
	// that no longer exists (a closed handle's), and HIP refuses to synchronise with such an event (late r04: a one-chunk upload after a
	// two-chunk upload of a handle that had been closed failed with hipErrorCapturedEvent)
	while (drained < issued) { const hipError_t e = drain_one(); if (err == hipSuccess) err = e; }
	if (err != hipSuccess) { set_last_error(std::string("staged upload: ") + hipGetErrorString(err)); return SIFT3D_ERR_HIP; }
	return SIFT3D_OK;
}

// device -> host (pageable); synchronous: the data is in the destinations on return.
// r06: the mirror image of staged_h2d (r05's form spawned four threads per 16 MB chunk and waited for a whole chunk's DMA before the
// first byte moved on: GetKeypoints brought the 36.6 MB of a 512^3 run back at 11 GB/s).  The 64 MB of pinned memory are dealt as
// 16 slots of 4 MB -- a result of a few tens of MB is many pieces in flight -- the calling thread only issues DMAs and frees slots, the copy
// threads live for the whole call and each moves its slice of piece after piece out of the pinned slot into the caller's (pageable) memory.
// Several (destination, source) pairs travel in ONE call (GetKeypoints: the records and the descriptors): one set of threads, one pipeline.
int staged_d2h_v(const D2HSeg *segs, int nseg, int device, hipStream_t st) {
	size_t total = 0;
	for (int k = 0; k < nseg; k++) total += segs[k].bytes;
	if (total == 0) return SIFT3D_OK;
	if (total < ((size_t)1 << 20)) {
		for (int k = 0; k < nseg; k++)
			if (segs[k].bytes) S3D_HIP(hipMemcpyAsync(segs[k].h_dst, segs[k].d_src, segs[k].bytes, hipMemcpyDeviceToHost, st));
		S3D_HIP(hipStreamSynchronize(st));
		return SIFT3D_OK;
	}
	Pool &g_pool = pool_of(device);
	std::lock_guard<std::mutex> lock(g_pool.mu);
	int rc = pool_ready(g_pool);
	if (rc) return rc;
	constexpr size_t kPiece = kChunk / kSub;
	constexpr size_t kSlots = (size_t)kBufs * kSub;
	// pieces never straddle two pairs
	struct Piece { char *dst; const char *src; size_t n; };
	std::vector<Piece> pieces;
	for (int k = 0; k < nseg; k++)
		for (size_t off = 0; off < segs[k].bytes; off += kPiece)
			pieces.push_back(Piece{static_cast<char *>(segs[k].h_dst) + off, static_cast<const char *>(segs[k].d_src) + off, std::min(kPiece, segs[k].bytes - off)});
	const size_t npieces = pieces.size();
	const unsigned hc = std::thread::hardware_concurrency();
	const int nt = (int)std::min<size_t>(std::max(1u, std::min((unsigned)kD2HThreads, hc ? hc / 2 : 1u)), std::max<size_t>(1, total >> 21));
	auto slot = [&](size_t i) { return g_pool.buf[(i % kSlots) / kSub] + ((i % kSlots) % kSub) * kPiece; };
	// landed: pieces whose DMA has completed (the calling thread's event wait); copied[i]: copy threads done with piece i
	std::atomic<size_t> landed{0};
	std::atomic<bool> stop{false};
	std::vector<std::atomic<int>> copied(npieces);
	for (auto &f : copied) f.store(0);
	auto copy_piece = [&](size_t i, int t) {
		size_t o, e;
		staging_slice(pieces[i].n, nt, t, &o, &e);
		if (e > o) memcpy(pieces[i].dst + o, slot(i) + o, e - o);
		copied[i].fetch_add(1, std::memory_order_release);
	};
	auto worker = [&](int t) {
		for (size_t i = 0; i < npieces; i++) {
			while (i >= landed.load(std::memory_order_acquire)) {
				if (stop.load(std::memory_order_relaxed)) return;
				std::this_thread::yield();
			}
			copy_piece(i, t);
		}
	};
	hipError_t err = hipSuccess;
	size_t issued = 0;
	auto issue = [&](size_t i) -> hipError_t {
		hipError_t e = hipMemcpyAsync(slot(i), pieces[i].src, pieces[i].n, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess) e = hipEventRecord(g_pool.ev2[i % kSlots], st);
		return e;
	};
	// (the DMAs are on their way before the threads exist: spawning them costs as much as the first pieces take to land)
	while (issued < std::min(kSlots, npieces) && err == hipSuccess)
		if ((err = issue(issued)) == hipSuccess) issued++;
	std::vector<std::thread> th;
	for (int t = 1; t < nt; t++) th.emplace_back(worker, t);
	for (size_t i = 0; i < npieces && err == hipSuccess; i++) {
		if ((err = hipEventSynchronize(g_pool.ev2[i % kSlots])) != hipSuccess) break;
		landed.store(i + 1, std::memory_order_release);
		copy_piece(i, 0);  // the calling thread is copy thread 0
		// the slot of piece i is free for piece i + kSlots once every thread has copied its slice out
		if (issued < npieces) {
			while (copied[i].load(std::memory_order_acquire) < nt) std::this_thread::yield();
			if ((err = issue(issued)) == hipSuccess) issued++;
		}
	}
	if (err != hipSuccess) stop.store(true);
	for (auto &t : th) t.join();
	// (on an error the DMAs already handed to the stream still target the pool: drained before the next caller may use it)
	if (err != hipSuccess) { (void)hipStreamSynchronize(st); set_last_error(std::string("staged download: ") + hipGetErrorString(err)); return SIFT3D_ERR_HIP; }
	return SIFT3D_OK;
}
int staged_d2h(void *h_dst, const void *d_src, size_t bytes, int device, hipStream_t st) {
	const D2HSeg sg{h_dst, d_src, bytes};
	return staged_d2h_v(&sg, 1, device, st);
}

}  // namespace s3d

// test-only symbol (include/sift3d_hip_test.h): the slice of copy thread t
extern "C" int sift3d_test_staging_slice(size_t n, int nt, int t, size_t *o, size_t *e) {
	if (!o || !e || nt < 1 || t < 0 || t >= nt) return SIFT3D_ERR_ARG;
	s3d::staging_slice(n, nt, t, o, e);
	return SIFT3D_OK;
}
