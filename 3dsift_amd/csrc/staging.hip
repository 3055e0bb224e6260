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
// One pool per device (r04; before: one process-wide pool behind one mutex whose events were re-created whenever the calling device
// changed -- the rank threads of the sharded driver took turns through it inside the timed region): 4 x 16 MB of pinned memory and
// four events per device that is ever used, created on first use by a thread whose current device is that device.
struct Pool {
	std::mutex mu;
	char *buf[kBufs] = {};
	hipEvent_t ev[kBufs] = {};
};
constexpr int kMaxDev = 64;
Pool g_pools[kMaxDev];
Pool &pool_of(int device) { return g_pools[(device >= 0 && device < kMaxDev) ? device : 0]; }

int pool_ready(Pool &P) {  // the caller holds P.mu and has made the pool's device current
	for (int i = 0; i < kBufs; i++) {
		if (!P.buf[i]) S3D_HIP(hipHostMalloc(reinterpret_cast<void **>(&P.buf[i]), kChunk, hipHostMallocPortable));
		if (!P.ev[i]) S3D_HIP(hipEventCreateWithFlags(&P.ev[i], hipEventDisableTiming));
	}
	return SIFT3D_OK;
}

// memcpy with a few helper threads (a single core moves 8-12 GB/s; PCIe 5 x16 wants ~50) -- the device -> host direction
void par_memcpy(char *dst, const char *src, size_t bytes) {
	const unsigned hc = std::thread::hardware_concurrency();
	const int nt = (int)std::min<size_t>(std::max(1u, std::min(4u, hc ? hc / 2 : 1u)), std::max<size_t>(1, bytes >> 20));
	if (nt <= 1) { memcpy(dst, src, bytes); return; }
	const size_t part = ((bytes / nt) + 4095) & ~(size_t)4095;
	std::vector<std::thread> th;
	for (int t = 1; t < nt; t++) {
		const size_t o = std::min(bytes, part * (size_t)t), e = std::min(bytes, o + part);
		if (e > o) th.emplace_back([=] { memcpy(dst + o, src + o, e - o); });
	}
	memcpy(dst, src, std::min(bytes, part));
	for (auto &t : th) t.join();
}
}  // namespace

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
			const size_t part = ((n / (size_t)nt) + 4095) & ~(size_t)4095, o = std::min(n, part * (size_t)t), e = std::min(n, o + part);
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
		const size_t part = ((n / (size_t)nt) + 4095) & ~(size_t)4095, e0 = std::min(n, part);
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

// device -> host (pageable); synchronous: the data is in h_dst on return
int staged_d2h(void *h_dst, const void *d_src, size_t bytes, int device, hipStream_t st) {
	if (bytes == 0) return SIFT3D_OK;
	if (bytes < ((size_t)1 << 20)) {
		S3D_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st));
		S3D_HIP(hipStreamSynchronize(st));
		return SIFT3D_OK;
	}
	Pool &g_pool = pool_of(device);
	std::lock_guard<std::mutex> lock(g_pool.mu);
	int rc = pool_ready(g_pool);
	if (rc) return rc;
	const size_t nchunks = (bytes + kChunk - 1) / kChunk;
	auto issue = [&](size_t i) -> hipError_t {
		const size_t off = i * kChunk, n = std::min(kChunk, bytes - off);
		hipError_t e = hipMemcpyAsync(g_pool.buf[i % kBufs], static_cast<const char *>(d_src) + off, n, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess) e = hipEventRecord(g_pool.ev[i % kBufs], st);
		return e;
	};
	for (size_t i = 0; i < std::min<size_t>(kBufs, nchunks); i++) S3D_HIP(issue(i));
	for (size_t i = 0; i < nchunks; i++) {
		const size_t off = i * kChunk, n = std::min(kChunk, bytes - off);
		S3D_HIP(hipEventSynchronize(g_pool.ev[i % kBufs]));
		par_memcpy(static_cast<char *>(h_dst) + off, g_pool.buf[i % kBufs], n);
		if (i + kBufs < nchunks) S3D_HIP(issue(i + kBufs));  // the buffer is free again
	}
	return SIFT3D_OK;
}

}  // namespace s3d
