// sharded.hip -- ONE large volume over the GPUs of a node: the native (C++) driver of the z-slab sharding (SURVEY 8e,
// BASELINE.json configs[3]); host code only, the kernels are those of the slab / seeded contexts (context.hip).
//
// No reference counterpart: the reference is one process on one host (Src/cSIFT3D.cc:165-235).  What is sharded is exactly that
// pipeline, with results equal to the single-GPU run (pyramid / extrema bit for bit, descriptors to their fixed-point tolerance):
//
//   sharded octaves   rank r owns the global planes [z0_r, z1_r) of every level of the first S octaves; level buffers carry `halo`
//                     extra planes per side.  Per Gaussian level the z-neighbours exchange the planes the NEXT consumer reaches:
//                       G[i] -> level i+1 : hw_{i+1} + 1 planes per side       urgent: in front of level i+1 on the rank's stream
//                       G[1..3]           : up to `halo` planes (orientation / descriptor windows)   deferred: own stream + communicator
//                       DoG[1..3]         : 1 plane (the extremum test reads z +- 1)                  deferred
//                     The normalisation maximum and the DoG maxima are all-reduced (MAX); x / y blurs need no communication.
//   windows (r05,     sift3d_sharded_create_ex(..., SIFT3D_SHARDED_PARTIAL_WINDOWS)): the descriptor windows are split along z over the ranks --
//   opt-in)           records to the z-neighbours, every rank marches its part, 768 int32 + the mass back, the owner finishes (3dsift_amd/slab.py does
//                     the same and is where the protocol is tested over gloo): the halos of G[1..3] shrink from 24 / 30 / 38 planes to 8 / 10 / 12.
//   tail              level 0 of the first replicated octave is all-gathered (1 / 8^S of a level); the remaining octaves run
//                     replicated in a seeded context per rank; their orientation work is dealt by extremum index (an integer
//                     all-reduce(SUM) of zero-padded rows restores it exactly), their descriptor work by keypoint.
//
// Transport:
//   RCCL   one host thread per GPU, three communicators per rank (urgent / deferred / tail: operations of one communicator must be
//          issued in the same order on every rank, and the three flows run concurrently), point-to-point ncclSend / ncclRecv between
//          z-neighbours over xGMI inside ncclGroupStart / End, on the rank's own streams: no host synchronisation between the
//          levels.  librccl is opened at run time (dlopen), so the library loads on hosts without it.
//          Failure protocol (r04): the first rank (or tail thread) whose step fails aborts EVERY communicator of the handle
//          (ncclCommAbort), so that z-neighbours blocked in a receive / reduction kernel return instead of wedging the process; the
//          handle is then dead -- sift3d_sharded_run returns an error from now on, the caller destroys it and, if it wants to go on,
//          starts a fresh process (nothing is restarted in place).
//   SIM    all ranks in this process on ONE GPU and one stream: sends are device copies, reductions go through the host.  This is
//          how the 1-GPU test boxes check the driver (same code, same plan) against the single-volume result.
//
// 3dsift_amd/slab.py drives the same C-ABI slab contexts from python over torch.distributed; this file is what the C++ user gets:
// CSIFT3DFactory::CreateCSIFT3D with SIFT3D_DEVICES=0,1,...,7 (3dsift_amd/host/src/cSIFT3D.cpp).
#include <dlfcn.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "sift3d_internal.h"

using namespace s3d;

namespace {

enum { KIND_INPUT = 0, KIND_GSS = 1, KIND_DOG = 2 };
struct Transfer { int src, dst, kind, idx, zg0, zg1, stage; };
typedef std::vector<std::pair<int, int>> Bounds;

// ---- planning (identical to 3dsift_amd/slab.py, which the CPU tests cover) ---------------------------------------------
int octaves_total(int nx, int ny, int nz) {  // Src/cSIFT3D.cc:254-255
	const int mn = std::min(nx, std::min(ny, nz));
	return std::max(0, (int)log2f((float)mn) - 2);
}

// owned plane ranges per rank: contiguous, starts on multiples of `align`, as equal as possible, remainder to the last rank
bool slab_bounds(int nz, int world, int align, Bounds &out) {
	const int units = nz / align;
	if (units < world) return false;
	const int base = units / world, rem = units % world;
	out.clear();
	int z = 0;
	for (int r = 0; r < world; r++) {
		const int n = align * (base + (r < rem ? 1 : 0));
		out.push_back({z, z + n});
		z += n;
	}
	out.back().second = nz;
	return true;
}

Bounds halve_bounds(const Bounds &b, int nz) {  // plane k of octave o+1 is plane 2k of octave o (Src/cSIFT3D.cc:321-344)
	Bounds o;
	for (auto &p : b) o.push_back({p.first / 2, std::min(p.second / 2, nz / 2)});
	return o;
}

// transfers that fill, for every rank, the global planes at distance (lo, hi] outside its owned range, from whichever ranks own
// them; destination-major order, identical on every rank (matching sends and receives are posted in the same order)
std::vector<Transfer> halo_transfers(const Bounds &bounds, int nz, int kind, int idx, int lo, int hi, int stage) {
	std::vector<Transfer> out;
	if (hi <= lo) return out;
	for (int r = 0; r < (int)bounds.size(); r++) {
		const int z0 = bounds[r].first, z1 = bounds[r].second;
		const int seg[2][2] = {{std::max(0, z0 - hi), std::max(0, z0 - lo)}, {std::min(nz, z1 + lo), std::min(nz, z1 + hi)}};
		for (auto &ab : seg) {
			if (ab[1] <= ab[0]) continue;
			for (int q = 0; q < (int)bounds.size(); q++) {
				if (q == r) continue;
				const int s = std::max(ab[0], bounds[q].first), e = std::min(ab[1], bounds[q].second);
				if (e > s) out.push_back(Transfer{q, r, kind, idx, s, e, stage});
			}
		}
	}
	return out;
}

// neigh[r] = the ranks q != r whose owned planes a descriptor window of a keypoint of rank r can reach into (windows cover at most `reach`
// planes either side of the keypoint's plane); symmetric, ascending
std::vector<std::vector<int>> window_neighbours(const Bounds &bounds, int reach) {
	std::vector<std::vector<int>> out(bounds.size());
	for (size_t r = 0; r < bounds.size(); r++) {
		const int z0 = bounds[r].first, z1 = bounds[r].second, lo = z0 - reach, hi = z1 - 1 + reach;
		for (size_t q = 0; q < bounds.size(); q++)
			if (q != r && z1 > z0 && bounds[q].second > bounds[q].first && bounds[q].first <= hi && bounds[q].second - 1 >= lo) out[r].push_back((int)q);
	}
	return out;
}

// ---- librccl through dlopen ---------------------------------------------------------------------------------------------
struct Rccl {
	void *lib = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	bool load(std::string &err) {
		if (lib) return true;
		for (const char *name : {"librccl.so.1", "librccl.so"}) {
			lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (lib) break;
		}
		if (!lib) { err = std::string("librccl not found: ") + dlerror(); return false; }
#define S3D_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(lib, sym)); if (!field) { err = std::string("librccl lacks ") + sym; return false; }
		S3D_SYM(CommInitAll, "ncclCommInitAll") S3D_SYM(CommDestroy, "ncclCommDestroy") S3D_SYM(CommAbort, "ncclCommAbort")
		S3D_SYM(GroupStart, "ncclGroupStart") S3D_SYM(GroupEnd, "ncclGroupEnd") S3D_SYM(Send, "ncclSend") S3D_SYM(Recv, "ncclRecv")
		S3D_SYM(AllReduce, "ncclAllReduce") S3D_SYM(Broadcast, "ncclBroadcast") S3D_SYM(GetErrorString, "ncclGetErrorString")
#undef S3D_SYM
		return true;
	}
};
Rccl g_rccl;
std::mutex g_rccl_mu;

struct Stage {  // one sharded octave of one rank
	int octave = 0, nx = 0, ny = 0, nz = 0;
	Bounds bounds;
	int z0 = 0, z1 = 0;
	size_t plane = 0;
	float *arena = nullptr;
	size_t arena_floats = 0;
	sift3d_handle ctx = nullptr;
	struct Buf { size_t off; int planes, zoff; bool ok = false; };
	std::map<int, Buf> bufs;
	float *view(int kind, int idx, int zg0, int zg1) {
		Buf &b = bufs[kind * 64 + idx];
		if (!b.ok) { if (sift3d_slab_buffer(ctx, kind, idx, &b.off, &b.planes, &b.zoff) != SIFT3D_OK) return nullptr; b.ok = true; }
		if (!(b.zoff <= zg0 && zg0 < zg1 && zg1 <= b.zoff + b.planes)) return nullptr;
		return arena + b.off + (size_t)(zg0 - b.zoff) * plane;
	}
};

struct Worker {  // the sharded octaves + the seeded, replicated tail context of one rank
	int rank = 0, device = 0;
	hipStream_t stream = nullptr, dstream = nullptr;  // the rank's stream; the stream of its deferred halos (RCCL)
	bool own_stream = false;
	hipEvent_t ev_level = nullptr, ev_def = nullptr, ev_seed = nullptr;
	std::vector<Stage> stages;
	std::vector<float *> dogmax;  // per stage: 8 floats (device)
	sift3d_handle tail = nullptr;
	float *seed = nullptr, *seed_mine = nullptr;
	ncclComm_t c_urgent = nullptr, c_deferred = nullptr, c_tail = nullptr;
	char *pscratch = nullptr; size_t pscratch_bytes = 0;  // partial descriptor windows: records / histograms / masses of a stage (grow-only, device)
	std::vector<sift3d_keypoint> kp;      // results of this rank's sharded octaves, reference order per stage
	std::vector<float> desc;
	std::vector<int> kp_stage_end;        // prefix ends per stage in kp
	std::vector<sift3d_keypoint> tkp;     // tail records (complete on every rank)
	std::vector<float> tdesc;             // tail descriptors: only this rank's rows are filled
	std::string err;
};

}  // namespace

struct sift3d_sharded {
	int nx = 0, ny = 0, nz = 0, world = 1, S = 1, halo = 0, noct = 0, levels = 3, ng = 6;
	bool sim = false;
	sift3d_params p{};
	std::vector<int> devices;
	std::vector<Worker> workers;
	std::vector<int> need, hws;
	std::vector<int> counts2;   // planes of the tail's seed level owned per rank
	int sx = 0, sy = 0, sz = 0; // dims of the tail's seed level
	bool ran = false;
	std::vector<sift3d_keypoint> kp;
	std::vector<float> desc;
	double times[4] = {0, 0, 0, 0};
	std::string err;
	// failure protocol of the RCCL transport: `failed` is set once, by the first rank whose step failed, which then aborts every
	// communicator.  Ranks hold comm_mu shared while they ISSUE RCCL calls (short, host side) and the aborter takes it exclusively,
	// so that no thread is inside a call on a communicator while it is torn down.
	// r05, partial descriptor windows (opt-in): accepted keypoints / flagged records per sharded octave and rank, written by the rank
	// threads and read by all of them behind a rendezvous
	bool partial = false;
	std::vector<std::vector<int>> kp_count, redo_count;
	std::mutex rv_mu;
	std::condition_variable rv_cv;
	int rv_arrived = 0;
	unsigned rv_gen = 0;
	std::atomic<bool> failed{false};
	std::atomic<bool> comms_aborted{false};  // abort_all ran: the communicators are gone (their pointers are left alone)
	std::shared_timed_mutex comm_mu;
};

namespace {

// A worker's error slot keeps its FIRST error; the rank's thread and the tail's thread may both report (one mutex for all slots: errors are rare).
std::mutex g_err_mu;
template <class W>
void set_err(W &w, const std::string &msg) {
	std::lock_guard<std::mutex> g(g_err_mu);
	if (w.err.empty()) w.err = msg;
}
#define SH_HIP(w, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_err((w), std::string(#call) + ": " + hipGetErrorString(e_)); return SIFT3D_ERR_HIP; } } while (0)
#define SH_ABI(w, call) do { int r_ = (call); if (r_ != SIFT3D_OK) { set_err((w), std::string(#call) + ": " + sift3d_error_string(r_) + " (" + sift3d_last_error() + ")"); return r_; } } while (0)
#define SH_NCCL(w, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { set_err((w), std::string(#call) + ": " + g_rccl.GetErrorString(r_)); return SIFT3D_ERR_HIP; } } while (0)

// The first failing rank: mark the handle dead and abort every communicator so that no peer stays blocked in a receive, a
// reduction or a broadcast whose partner will never arrive.  The exclusive lock is awaited for a bounded time only: a peer stuck
// INSIDE a call (connection set-up waiting for the rank that failed) holds it shared, and aborting under it is what frees that peer.
void abort_all(sift3d_sharded *H) {
	if (H->sim || H->failed.exchange(true)) return;
	const bool locked = H->comm_mu.try_lock_for(std::chrono::seconds(2));
	// The communicator POINTERS are never written after creation: a rank that reads one (under the shared lock, after SH_LIVE) cannot see
	// a pointer in the middle of a store.  A rank already inside an RCCL call when the wait above timed out is what the abort is for.
	H->comms_aborted.store(true);  // (ncclCommAbort frees the communicator: destroy must not hand it to ncclCommDestroy again)
	for (Worker &w : H->workers)
		for (ncclComm_t c : {w.c_urgent, w.c_deferred, w.c_tail})
			if (c) (void)g_rccl.CommAbort(c);
	if (locked) H->comm_mu.unlock();
}
#define SH_LIVE(H, w) do { if ((H)->failed.load()) { set_err((w), "aborted: another rank failed"); return SIFT3D_ERR_STATE; } } while (0)

// posts the transfers this set of local workers takes part in.  SIM: device copies on the shared stream.  RCCL: one group of
// sends / receives of the one local rank on `comm` / `stream` of the given flow (0 urgent, 1 deferred).
int exchange(sift3d_sharded *H, std::vector<Worker *> &ws, const std::vector<Transfer> &ts, int flow) {
	if (ts.empty()) return SIFT3D_OK;
	if (H->sim) {
		Worker &w0 = *ws[0];
		for (const Transfer &t : ts) {
			Stage &s = H->workers[(size_t)t.src].stages[(size_t)t.stage], &d = H->workers[(size_t)t.dst].stages[(size_t)t.stage];
			float *sp = s.view(t.kind, t.idx, t.zg0, t.zg1), *dp = d.view(t.kind, t.idx, t.zg0, t.zg1);
			if (!sp || !dp) { set_err(w0, "halo transfer outside a level buffer"); return SIFT3D_ERR_STATE; }
			SH_HIP(w0, hipMemcpyAsync(dp, sp, sizeof(float) * s.plane * (size_t)(t.zg1 - t.zg0), hipMemcpyDeviceToDevice, w0.stream));
		}
		return SIFT3D_OK;
	}
	Worker &w = *ws[0];
	ncclComm_t comm = flow ? w.c_deferred : w.c_urgent;
	hipStream_t st = flow ? w.dstream : w.stream;
	bool any = false;
	for (const Transfer &t : ts) any = any || t.src == w.rank || t.dst == w.rank;
	if (!any) return SIFT3D_OK;
	std::shared_lock<std::shared_timed_mutex> live(H->comm_mu);
	SH_LIVE(H, w);
	SH_NCCL(w, g_rccl.GroupStart());
	for (const Transfer &t : ts) {
		if (t.src != w.rank && t.dst != w.rank) continue;
		Stage &s = w.stages[(size_t)t.stage];
		float *p = s.view(t.kind, t.idx, t.zg0, t.zg1);
		if (!p) { (void)g_rccl.GroupEnd(); set_err(w, "halo transfer outside a level buffer"); return SIFT3D_ERR_STATE; }
		const size_t cnt = s.plane * (size_t)(t.zg1 - t.zg0);
		if (t.src == w.rank) SH_NCCL(w, g_rccl.Send(p, cnt, ncclFloat, t.dst, comm, st));
		else SH_NCCL(w, g_rccl.Recv(p, cnt, ncclFloat, t.src, comm, st));
	}
	SH_NCCL(w, g_rccl.GroupEnd());
	return SIFT3D_OK;
}

// in-place MAX over the ranks of n device floats per worker (non-negative values: the DoG maxima), stream ordered for RCCL
int allreduce_max_dev(sift3d_sharded *H, std::vector<Worker *> &ws, int stage, int n) {
	if (H->sim) {
		Worker &w0 = *ws[0];
		std::vector<float> m((size_t)n, 0.f), t((size_t)n);
		SH_HIP(w0, hipStreamSynchronize(w0.stream));
		for (Worker *w : ws) {
			SH_HIP(w0, hipMemcpy(t.data(), w->dogmax[(size_t)stage], sizeof(float) * n, hipMemcpyDeviceToHost));
			for (int i = 0; i < n; i++) m[(size_t)i] = std::max(m[(size_t)i], t[(size_t)i]);
		}
		for (Worker *w : ws) SH_HIP(w0, hipMemcpy(w->dogmax[(size_t)stage], m.data(), sizeof(float) * n, hipMemcpyHostToDevice));
		return SIFT3D_OK;
	}
	Worker &w = *ws[0];
	std::shared_lock<std::shared_timed_mutex> live(H->comm_mu);
	SH_LIVE(H, w);
	SH_NCCL(w, g_rccl.AllReduce(w.dogmax[(size_t)stage], w.dogmax[(size_t)stage], (size_t)n, ncclFloat, ncclMax, w.c_urgent, w.stream));
	return SIFT3D_OK;
}

// every worker's seed level = the owned planes of all ranks, in rank order (uneven counts: one broadcast per rank in a group)
int allgather_seed(sift3d_sharded *H, std::vector<Worker *> &ws) {
	const size_t pl = (size_t)H->sx * H->sy;
	if (H->sim) {
		Worker &w0 = *ws[0];
		size_t off = 0;
		for (int r = 0; r < H->world; r++) {
			const size_t cnt = pl * (size_t)H->counts2[(size_t)r];
			for (Worker *w : ws)
				if (cnt) SH_HIP(w0, hipMemcpyAsync(w->seed + off, H->workers[(size_t)r].seed_mine, sizeof(float) * cnt, hipMemcpyDeviceToDevice, w0.stream));
			off += cnt;
		}
		return SIFT3D_OK;
	}
	Worker &w = *ws[0];
	std::shared_lock<std::shared_timed_mutex> live(H->comm_mu);
	SH_LIVE(H, w);
	SH_NCCL(w, g_rccl.GroupStart());
	size_t off = 0;
	for (int r = 0; r < H->world; r++) {
		const size_t cnt = pl * (size_t)H->counts2[(size_t)r];
		if (cnt) SH_NCCL(w, g_rccl.Broadcast(w.seed_mine, w.seed + off, cnt, ncclFloat, r, w.c_urgent, w.stream));
		off += cnt;
	}
	SH_NCCL(w, g_rccl.GroupEnd());
	return SIFT3D_OK;
}

// ---- partial descriptor windows (r05, opt-in; the protocol of 3dsift_amd/slab.py _describe_partial) ---------------------
// every rank thread of the RCCL transport arrives; values written before are visible to all after.  A dead handle lets the waiters go.
int rendezvous(sift3d_sharded *H, Worker &w) {
	if (H->sim) return SIFT3D_OK;
	std::unique_lock<std::mutex> lk(H->rv_mu);
	const unsigned gen = H->rv_gen;
	if (++H->rv_arrived == H->world) { H->rv_arrived = 0; H->rv_gen++; H->rv_cv.notify_all(); return SIFT3D_OK; }
	while (H->rv_gen == gen) {
		if (H->failed.load()) {
			// a waiter that leaves takes its arrival with it and wakes the others, so that the count is right whatever the caller does next
			// (today a failed handle is dead; the invariant is local to this function all the same)
			H->rv_arrived = std::max(0, H->rv_arrived - 1);
			H->rv_cv.notify_all();
			set_err(w, "aborted: another rank failed");
			return SIFT3D_ERR_STATE;
		}
		H->rv_cv.wait_for(lk, std::chrono::milliseconds(20));
	}
	return SIFT3D_OK;
}

struct RawXfer { int src, dst; const void *sp; void *dp; size_t bytes; };  // sp valid where src is local, dp where dst is local

int exchange_raw(sift3d_sharded *H, std::vector<Worker *> &ws, const std::vector<RawXfer> &ts) {
	if (ts.empty()) return SIFT3D_OK;
	if (H->sim) {
		Worker &w0 = *ws[0];
		for (const RawXfer &t : ts) SH_HIP(w0, hipMemcpyAsync(t.dp, t.sp, t.bytes, hipMemcpyDeviceToDevice, w0.stream));
		return SIFT3D_OK;
	}
	Worker &w = *ws[0];
	std::shared_lock<std::shared_timed_mutex> live(H->comm_mu);
	SH_LIVE(H, w);
	SH_NCCL(w, g_rccl.GroupStart());
	for (const RawXfer &t : ts) {
		if (t.src == w.rank) SH_NCCL(w, g_rccl.Send(t.sp, t.bytes, ncclInt8, t.dst, w.c_urgent, w.stream));
		else if (t.dst == w.rank) SH_NCCL(w, g_rccl.Recv(t.dp, t.bytes, ncclInt8, t.src, w.c_urgent, w.stream));
	}
	SH_NCCL(w, g_rccl.GroupEnd());
	return SIFT3D_OK;
}

// device scratch of one rank for one round of one stage: pointers into Worker::pscratch
struct PartLayout {
	char *recs = nullptr;                      // this rank's records, processing order (round 2: the flagged subset)
	float *units = nullptr;                    // round 2: their exact units
	int *redo = nullptr; float *units_next = nullptr;
	std::map<int, char *> recs_in;             // neighbour r's records
	std::map<int, float *> units_in;
	std::map<int, int *> part_h;               // this rank's part of the windows of r's records (r = itself or a neighbour)
	std::map<int, float *> part_m;
	std::map<int, int *> got_h;                // neighbour q's part of this rank's records
	std::map<int, float *> got_m;
};

int lay_out(Worker &w, const std::vector<int> &counts, const std::vector<int> &nb, size_t rb, PartLayout &L) {
	auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
	const size_t me = (size_t)counts[(size_t)w.rank];
	size_t need = al(me * rb) + 3 * al(me * 4) + al(me * 768 * 4) + al(me * 4);
	for (int r : nb) {
		const size_t n = (size_t)counts[(size_t)r];
		need += al(n * rb) + al(n * 4) + al(n * 768 * 4) + al(n * 4);  // r's records + units here, this rank's part of them
		need += al(me * 768 * 4) + al(me * 4);                        // r's part of this rank's records
	}
	need = std::max<size_t>(need, 256);
	if (need > w.pscratch_bytes) {
		SH_HIP(w, hipStreamSynchronize(w.stream));
		if (w.pscratch) SH_HIP(w, hipFree(w.pscratch));
		w.pscratch = nullptr; w.pscratch_bytes = 0;
		const size_t cap = need + need / 4;
		SH_HIP(w, hipMalloc(reinterpret_cast<void **>(&w.pscratch), cap));
		w.pscratch_bytes = cap;
	}
	char *p = w.pscratch;
	auto take = [&](size_t b) { char *q = p; p += al(b); return q; };
	L = PartLayout();
	L.recs = take(me * rb);
	L.units = reinterpret_cast<float *>(take(me * 4));
	L.redo = reinterpret_cast<int *>(take(me * 4));
	L.units_next = reinterpret_cast<float *>(take(me * 4));
	L.part_h[w.rank] = reinterpret_cast<int *>(take(me * 768 * 4));
	L.part_m[w.rank] = reinterpret_cast<float *>(take(me * 4));
	for (int r : nb) {
		const size_t n = (size_t)counts[(size_t)r];
		L.recs_in[r] = take(n * rb);
		L.units_in[r] = reinterpret_cast<float *>(take(n * 4));
		L.part_h[r] = reinterpret_cast<int *>(take(n * 768 * 4));
		L.part_m[r] = reinterpret_cast<float *>(take(n * 4));
		L.got_h[r] = reinterpret_cast<int *>(take(me * 768 * 4));
		L.got_m[r] = reinterpret_cast<float *>(take(me * 4));
	}
	return SIFT3D_OK;
}

// one round: records (round 2: + units) to the neighbours, every rank's part in one launch, the parts back, the owner's finish.
// counts[r]: records of rank r in this round (known to every rank); L[i] belongs to ws[i] and already holds its records (and units).
int partial_round(sift3d_sharded *H, std::vector<Worker *> &ws, int s, const std::vector<std::vector<int>> &neigh, const std::vector<int> &counts,
                  std::vector<PartLayout> &L, size_t rb, bool second, std::vector<int> &n_redo) {
	std::map<int, size_t> local;
	for (size_t i = 0; i < ws.size(); i++) local[ws[i]->rank] = i;
	auto is_local = [&](int r) { return local.count(r) != 0; };
	const Bounds &bounds = ws[0]->stages[(size_t)s].bounds;
	std::vector<RawXfer> ts;
	for (int r = 0; r < H->world; r++)
		for (int q : neigh[(size_t)r]) {
			const size_t n = (size_t)counts[(size_t)r];
			if (!n || (!is_local(r) && !is_local(q))) continue;
			const PartLayout *Lr = is_local(r) ? &L[local[r]] : nullptr;
			PartLayout *Lq = is_local(q) ? &L[local[q]] : nullptr;
			ts.push_back(RawXfer{r, q, Lr ? Lr->recs : nullptr, Lq ? Lq->recs_in[r] : nullptr, n * rb});
			if (second) ts.push_back(RawXfer{r, q, Lr ? Lr->units : nullptr, Lq ? Lq->units_in[r] : nullptr, n * 4});
		}
	int rc = exchange_raw(H, ws, ts);
	if (rc) return rc;
	for (size_t i = 0; i < ws.size(); i++) {
		Worker &w = *ws[i];
		const int q = w.rank;
		std::vector<int> owners{q};
		owners.insert(owners.end(), neigh[(size_t)q].begin(), neigh[(size_t)q].end());
		std::vector<const void *> recs; std::vector<int> n, o0, o1; std::vector<const float *> un; std::vector<int *> hh; std::vector<float *> mm;
		for (int r : owners) {
			if (!counts[(size_t)r]) continue;
			recs.push_back(r == q ? L[i].recs : L[i].recs_in[r]);
			un.push_back(!second ? nullptr : r == q ? L[i].units : L[i].units_in[r]);
			n.push_back(counts[(size_t)r]); hh.push_back(L[i].part_h[r]); mm.push_back(L[i].part_m[r]);
			o0.push_back(bounds[(size_t)r].first); o1.push_back(bounds[(size_t)r].second);
		}
		SH_HIP(w, hipSetDevice(w.device));
		SH_ABI(w, sift3d_slab_describe_partial(w.stages[(size_t)s].ctx, (int)recs.size(), recs.data(), n.data(), second ? un.data() : nullptr, hh.data(),
		                                       mm.data(), o0.data(), o1.data()));
	}
	ts.clear();
	for (int q = 0; q < H->world; q++)
		for (int r : neigh[(size_t)q]) {
			const size_t n = (size_t)counts[(size_t)r];
			if (!n || (!is_local(r) && !is_local(q))) continue;
			PartLayout *Lq = is_local(q) ? &L[local[q]] : nullptr, *Lr = is_local(r) ? &L[local[r]] : nullptr;
			ts.push_back(RawXfer{q, r, Lq ? Lq->part_h[r] : nullptr, Lr ? Lr->got_h[q] : nullptr, n * 768 * 4});
			ts.push_back(RawXfer{q, r, Lq ? Lq->part_m[r] : nullptr, Lr ? Lr->got_m[q] : nullptr, n * 4});
		}
	rc = exchange_raw(H, ws, ts);
	if (rc) return rc;
	for (size_t i = 0; i < ws.size(); i++) {
		Worker &w = *ws[i];
		const int r = w.rank, n = counts[(size_t)r];
		std::vector<int> from = neigh[(size_t)r];
		from.push_back(r);
		std::sort(from.begin(), from.end());  // the masses are added in ascending rank order, as the python driver does
		std::vector<const int *> hh; std::vector<const float *> mm;
		if (n) for (int q : from) { hh.push_back(q == r ? L[i].part_h[r] : L[i].got_h[q]); mm.push_back(q == r ? L[i].part_m[r] : L[i].got_m[q]); }
		SH_HIP(w, hipSetDevice(w.device));
		int nr = 0;
		SH_ABI(w, sift3d_slab_describe_finish(w.stages[(size_t)s].ctx, n ? L[i].recs : nullptr, n, (int)hh.size(), hh.data(), mm.data(), second && n ? L[i].units : nullptr,
		                                      second ? 1 : 0, n ? L[i].redo : nullptr, n ? L[i].units_next : nullptr, &nr));
		n_redo[i] = nr;
	}
	return SIFT3D_OK;
}

// orientation of the owned extrema, then the descriptors of sharded octave s from partial integer histograms
int describe_partial_stage(sift3d_sharded *H, std::vector<Worker *> &ws, int s) {
	Worker &w0 = *ws[0];
	int rbi = 0, reach = 0;
	SH_ABI(w0, sift3d_slab_record_bytes(&rbi));
	const size_t rb = (size_t)rbi;
	for (Worker *w : ws) { SH_HIP(*w, hipSetDevice(w->device)); SH_ABI(*w, sift3d_slab_orient_launch(w->stages[(size_t)s].ctx)); }
	for (Worker *w : ws) {
		int n = 0;
		SH_HIP(*w, hipSetDevice(w->device));
		SH_ABI(*w, sift3d_slab_orient_count(w->stages[(size_t)s].ctx, &n));
		H->kp_count[(size_t)s][(size_t)w->rank] = n;
	}
	int rc = rendezvous(H, w0);
	if (rc) return rc;
	const std::vector<int> counts = H->kp_count[(size_t)s];
	SH_ABI(w0, sift3d_slab_desc_reach(w0.stages[(size_t)s].ctx, &reach));
	const std::vector<std::vector<int>> neigh = window_neighbours(w0.stages[(size_t)s].bounds, reach);
	std::vector<PartLayout> L(ws.size());
	std::vector<int> n_redo(ws.size(), 0);
	for (size_t i = 0; i < ws.size(); i++) {
		Worker &w = *ws[i];
		SH_HIP(w, hipSetDevice(w.device));
		if ((rc = lay_out(w, counts, neigh[(size_t)w.rank], rb, L[i])) != SIFT3D_OK) return rc;
		if (counts[(size_t)w.rank]) SH_ABI(w, sift3d_slab_export_records(w.stages[(size_t)s].ctx, L[i].recs));
	}
	if ((rc = partial_round(H, ws, s, neigh, counts, L, rb, false, n_redo)) != SIFT3D_OK) return rc;
	for (size_t i = 0; i < ws.size(); i++) H->redo_count[(size_t)s][(size_t)ws[i]->rank] = n_redo[i];
	if ((rc = rendezvous(H, w0)) != SIFT3D_OK) return rc;
	const std::vector<int> tot = H->redo_count[(size_t)s];
	if (!std::any_of(tot.begin(), tot.end(), [](int v) { return v > 0; })) return SIFT3D_OK;
	// rare: records whose first fixed-point unit failed are repeated, by every part, with the exact unit.  The flagged subset is compacted
	// through the host (a few records), then the scratch is laid out again for the second round's counts.
	std::vector<std::vector<char>> recs2(ws.size());
	std::vector<std::vector<float>> units2(ws.size());
	for (size_t i = 0; i < ws.size(); i++) {
		Worker &w = *ws[i];
		const size_t n = (size_t)counts[(size_t)w.rank];
		if (!tot[(size_t)w.rank]) continue;
		std::vector<char> recs(n * rb);
		std::vector<int> redo(n);
		std::vector<float> un(n);
		SH_HIP(w, hipSetDevice(w.device));
		SH_HIP(w, hipMemcpyAsync(recs.data(), L[i].recs, n * rb, hipMemcpyDeviceToHost, w.stream));
		SH_HIP(w, hipMemcpyAsync(redo.data(), L[i].redo, n * 4, hipMemcpyDeviceToHost, w.stream));
		SH_HIP(w, hipMemcpyAsync(un.data(), L[i].units_next, n * 4, hipMemcpyDeviceToHost, w.stream));
		SH_HIP(w, hipStreamSynchronize(w.stream));
		for (size_t k = 0; k < n; k++)
			if (redo[k]) { recs2[i].insert(recs2[i].end(), recs.begin() + (ptrdiff_t)(k * rb), recs.begin() + (ptrdiff_t)((k + 1) * rb)); units2[i].push_back(un[k]); }
		if ((int)units2[i].size() != tot[(size_t)w.rank]) { set_err(w, "flagged records and their count disagree"); return SIFT3D_ERR_STATE; }
	}
	for (size_t i = 0; i < ws.size(); i++) {
		Worker &w = *ws[i];
		SH_HIP(w, hipSetDevice(w.device));
		SH_HIP(w, hipStreamSynchronize(w.stream));  // (simulated ranks share the stream: every rank's first round has drained before a scratch moves)
		if ((rc = lay_out(w, tot, neigh[(size_t)w.rank], rb, L[i])) != SIFT3D_OK) return rc;
		if (tot[(size_t)w.rank]) {
			SH_HIP(w, hipMemcpyAsync(L[i].recs, recs2[i].data(), recs2[i].size(), hipMemcpyHostToDevice, w.stream));
			SH_HIP(w, hipMemcpyAsync(L[i].units, units2[i].data(), units2[i].size() * 4, hipMemcpyHostToDevice, w.stream));
			SH_HIP(w, hipStreamSynchronize(w.stream));  // (the host vectors are pageable and go out of scope)
		}
	}
	return partial_round(H, ws, s, neigh, tot, L, rb, true, n_redo);
}

// the replicated tail of the local workers: pyramid + extrema of the remaining octaves on every rank, orientation dealt by extremum
// index and restored everywhere by an integer all-reduce(SUM) of zero-padded rows (exact), descriptors dealt by keypoint
int run_tail(sift3d_sharded *H, std::vector<Worker *> &ws) {
	Worker &w0 = *ws[0];
	for (Worker *w : ws) {
		SH_HIP(*w, hipSetDevice(w->device));
		SH_HIP(*w, hipEventSynchronize(w->ev_seed));  // the all-gathered seed level exists
		SH_ABI(*w, sift3d_seed_upload(w->tail, w->seed, 1));
		SH_ABI(*w, sift3d_run_partial_orientation(w->tail));
	}
	std::vector<int *> rows(ws.size(), nullptr);
	std::vector<int> next(ws.size(), 0);
	for (size_t i = 0; i < ws.size(); i++) {
		Worker *w = ws[i];
		SH_HIP(*w, hipSetDevice(w->device));
		SH_ABI(*w, sift3d_num_extrema(w->tail, &next[i]));
		if (next[i] > 0) {
			SH_HIP(*w, hipMalloc(&rows[i], sizeof(int) * (size_t)next[i] * SIFT3D_ORIENT_WORDS));
			SH_ABI(*w, sift3d_export_orientation_device(w->tail, rows[i]));
		}
	}
	int rc = SIFT3D_OK;
	if (H->sim) {
		const size_t n = (size_t)next[0] * SIFT3D_ORIENT_WORDS;
		std::vector<int> sum(n, 0), t(n);
		for (size_t i = 0; i < ws.size() && n; i++) {
			if (next[i] != next[0]) { set_err(w0, "replicated tails disagree on the number of extrema"); rc = SIFT3D_ERR_STATE; break; }
			if (hipMemcpy(t.data(), rows[i], sizeof(int) * n, hipMemcpyDeviceToHost) != hipSuccess) { rc = SIFT3D_ERR_HIP; break; }
			for (size_t k = 0; k < n; k++) sum[k] += t[k];
		}
		for (size_t i = 0; i < ws.size() && n && rc == SIFT3D_OK; i++)
			if (hipMemcpy(rows[i], sum.data(), sizeof(int) * n, hipMemcpyHostToDevice) != hipSuccess) rc = SIFT3D_ERR_HIP;
	} else if (next[0] > 0) {
		Worker &w = *ws[0];
		hipStream_t ts = nullptr;  // the tail's collective runs on the null stream of the rank's device, ordered behind the export above
		ncclResult_t r = ncclSuccess;
		{
			std::shared_lock<std::shared_timed_mutex> live(H->comm_mu);
			if (H->failed.load() || !w.c_tail) { set_err(w, "aborted: another rank failed"); rc = SIFT3D_ERR_STATE; }
			else r = g_rccl.AllReduce(rows[0], rows[0], (size_t)next[0] * SIFT3D_ORIENT_WORDS, ncclInt32, ncclSum, w.c_tail, ts);
		}
		if (rc != SIFT3D_OK) {}
		else if (r != ncclSuccess) { set_err(w, std::string("ncclAllReduce (tail): ") + g_rccl.GetErrorString(r)); rc = SIFT3D_ERR_HIP; }
		else if (hipStreamSynchronize(ts) != hipSuccess) { set_err(w, "tail all-reduce did not complete"); rc = SIFT3D_ERR_HIP; }
		if (rc == SIFT3D_OK && H->failed.load()) { set_err(w, "aborted: another rank failed"); rc = SIFT3D_ERR_STATE; }  // an aborted collective leaves garbage rows
	}
	for (size_t i = 0; i < ws.size(); i++) {
		Worker *w = ws[i];
		if (rc == SIFT3D_OK) {
			(void)hipSetDevice(w->device);
			if (next[i] > 0) rc = sift3d_import_orientation_device(w->tail, rows[i]);
			if (rc == SIFT3D_OK) rc = sift3d_run_describe(w->tail);
			if (rc != SIFT3D_OK) w->err = std::string("tail: ") + sift3d_error_string(rc) + " (" + sift3d_last_error() + ")";
		}
		if (rows[i]) (void)hipFree(rows[i]);
	}
	return rc;
}

// CSIFT3D::KpSiftAlgorithm (Src/cSIFT3D.cc:165-235) over the slabs of the local workers
int run_local(sift3d_sharded *H, std::vector<Worker *> &ws) {
	Worker &w0 = *ws[0];
	const int ng = H->ng;
	for (Worker *w : ws) SH_HIP(*w, hipSetDevice(w->device));
	for (int s = 0; s < H->S; s++) {
		const Stage &st0 = w0.stages[(size_t)s];
		const Bounds &bounds = st0.bounds;
		const int nzs = st0.nz;
		for (int i = 0; i < ng; i++) {
			for (Worker *w : ws) { SH_HIP(*w, hipSetDevice(w->device)); SH_ABI(*w, sift3d_slab_level(w->stages[(size_t)s].ctx, i)); }
			const int urgent_h = i + 1 < ng ? H->hws[(size_t)i + 1] + 1 : 0;  // planes p-hw-1 .. p+hw of the next level's z-march
			// urgent: ordered behind the level kernel on the rank's stream, in front of the next level
			int rc = exchange(H, ws, halo_transfers(bounds, nzs, KIND_GSS, i, 0, urgent_h, s), 0);
			if (rc) return rc;
			// deferred: the wider keypoint-window halo of G[1..levels] and the DoG plane behind it, on the deferred flow
			std::vector<Transfer> late = halo_transfers(bounds, nzs, KIND_GSS, i, urgent_h, H->need[(size_t)i], s);
			if (i - 1 >= 1 && i - 1 <= H->levels) {
				std::vector<Transfer> dg = halo_transfers(bounds, nzs, KIND_DOG, i - 1, 0, 1, s);
				late.insert(late.end(), dg.begin(), dg.end());
			}
			if (!late.empty()) {
				if (!H->sim)
					for (Worker *w : ws) {  // the deferred stream picks up behind the level kernel
						SH_HIP(*w, hipEventRecord(w->ev_level, w->stream));
						SH_HIP(*w, hipStreamWaitEvent(w->dstream, w->ev_level, 0));
					}
				rc = exchange(H, ws, late, 1);
				if (rc) return rc;
			}
			if (i == H->levels && s + 1 < H->noct) {
				// G[s+1][0] = DownSample_3D(G[s][levels]) (Src/cSIFT3D.cc:293-296, 321-344), owned planes only: straight into the next
				// sharded octave's level-0 buffer, or into this rank's piece of the tail's seed level
				for (Worker *w : ws) {
					SH_HIP(*w, hipSetDevice(w->device));
					if (s + 1 < H->S) {
						Stage &nst = w->stages[(size_t)s + 1];
						if (nst.z1 > nst.z0) {
							float *dstp = nst.view(KIND_GSS, 0, nst.z0, nst.z1);
							if (!dstp) { w->err = "decimation target outside the buffer"; return SIFT3D_ERR_STATE; }
							SH_ABI(*w, sift3d_slab_decimate_async(w->stages[(size_t)s].ctx, dstp));
						}
					} else {
						SH_ABI(*w, sift3d_slab_decimate_async(w->stages[(size_t)s].ctx, w->seed_mine));
					}
				}
			}
		}
		// DoG maxima -> global (threshold of Detect_KeyPoints, Src/cSIFT3D.cc:379-384)
		for (Worker *w : ws) { SH_HIP(*w, hipSetDevice(w->device)); SH_ABI(*w, sift3d_slab_export_dogmax_device(w->stages[(size_t)s].ctx, w->dogmax[(size_t)s])); }
		int rc = allreduce_max_dev(H, ws, s, 8);
		if (rc) return rc;
		for (Worker *w : ws) { SH_HIP(*w, hipSetDevice(w->device)); SH_ABI(*w, sift3d_slab_import_dogmax_device(w->stages[(size_t)s].ctx, w->dogmax[(size_t)s])); }
	}
	if (!H->sim)
		for (Worker *w : ws) {  // the deferred halos are complete before detection reads them
			SH_HIP(*w, hipEventRecord(w->ev_def, w->dstream));
			SH_HIP(*w, hipStreamWaitEvent(w->stream, w->ev_def, 0));
		}
	const bool has_tail = H->noct > H->S;
	if (has_tail) {
		int rc = allgather_seed(H, ws);
		if (rc) return rc;
		for (Worker *w : ws) SH_HIP(*w, hipEventRecord(w->ev_seed, w->stream));
	}
	// replicated tail on its own host thread (RCCL), beside the sharded detection and descriptors; inline for simulated ranks
	int tail_rc = SIFT3D_OK;
	std::thread tail_thread;
	if (has_tail && !H->sim) tail_thread = std::thread([&] { tail_rc = run_tail(H, ws); if (tail_rc != SIFT3D_OK) abort_all(H); });
	int rc = SIFT3D_OK;
	for (int s = 0; s < H->S && rc == SIFT3D_OK; s++) {
		for (Worker *w : ws) {
			if (hipSetDevice(w->device) != hipSuccess) { rc = SIFT3D_ERR_HIP; break; }
			rc = sift3d_slab_detect(w->stages[(size_t)s].ctx);
			if (rc == SIFT3D_OK && !H->partial) rc = sift3d_slab_describe(w->stages[(size_t)s].ctx);
			if (rc != SIFT3D_OK) { set_err(*w, std::string("sharded keypoints: ") + sift3d_error_string(rc) + " (" + sift3d_last_error() + ")"); break; }
		}
		if (rc == SIFT3D_OK && H->partial) rc = describe_partial_stage(H, ws, s);
	}
	if (rc != SIFT3D_OK) abort_all(H);               // (the tail thread may sit in its all-reduce waiting for ranks that will not come)
	if (tail_thread.joinable()) tail_thread.join();  // joined whatever happened above: it calls into contexts destroy would free
	if (rc == SIFT3D_OK && H->failed.load()) { set_err(w0, "aborted: another rank failed"); rc = SIFT3D_ERR_STATE; }  // halos of an aborted exchange are garbage
	if (has_tail && H->sim && rc == SIFT3D_OK) tail_rc = run_tail(H, ws);
	if (rc == SIFT3D_OK) rc = tail_rc;
	if (rc != SIFT3D_OK) return rc;
	// results of the local ranks to the host
	for (Worker *w : ws) {
		SH_HIP(*w, hipSetDevice(w->device));
		w->kp.clear(); w->desc.clear(); w->kp_stage_end.clear();
		for (int s = 0; s < H->S; s++) {
			int n = 0;
			SH_ABI(*w, sift3d_num_keypoints(w->stages[(size_t)s].ctx, &n));
			const size_t o = w->kp.size();
			w->kp.resize(o + (size_t)n); w->desc.resize((o + (size_t)n) * kDesc);
			if (n) SH_ABI(*w, sift3d_get_keypoints(w->stages[(size_t)s].ctx, w->kp.data() + o, w->desc.data() + o * kDesc));
			w->kp_stage_end.push_back((int)w->kp.size());
		}
		if (has_tail) {
			int n = 0;
			SH_ABI(*w, sift3d_num_keypoints(w->tail, &n));
			w->tkp.resize((size_t)n); w->tdesc.resize((size_t)n * kDesc);
			if (n) SH_ABI(*w, sift3d_get_keypoints(w->tail, w->tkp.data(), w->tdesc.data()));
		}
	}
	return SIFT3D_OK;
}

// phase 0: everything that may still enqueue on or wait for a stream; phase 1: the streams (simulated ranks SHARE rank 0's stream: it
// must outlive the contexts of every rank -- destroying it with rank 0 made the other ranks synchronise a dead stream, which hung
// about one run in ten)
void destroy_worker(Worker &w, int phase, bool comms_aborted) {
	(void)hipSetDevice(w.device);
	if (phase == 0) {
		if (w.stream) (void)hipStreamSynchronize(w.stream);
		if (w.dstream) (void)hipStreamSynchronize(w.dstream);
		if (w.tail) sift3d_destroy(w.tail);
		w.tail = nullptr;
		for (Stage &s : w.stages) {
			if (s.ctx) { (void)sift3d_set_stream(s.ctx, nullptr); sift3d_destroy(s.ctx); s.ctx = nullptr; }
			if (s.arena) (void)hipFree(s.arena);
			s.arena = nullptr;
		}
		for (float *&d : w.dogmax) { if (d) (void)hipFree(d); d = nullptr; }
		if (w.pscratch) (void)hipFree(w.pscratch);
		w.pscratch = nullptr; w.pscratch_bytes = 0;
		if (w.seed) (void)hipFree(w.seed);
		if (w.seed_mine) (void)hipFree(w.seed_mine);
		w.seed = w.seed_mine = nullptr;
		if (w.ev_level) (void)hipEventDestroy(w.ev_level);
		if (w.ev_def) (void)hipEventDestroy(w.ev_def);
		if (w.ev_seed) (void)hipEventDestroy(w.ev_seed);
		w.ev_level = w.ev_def = w.ev_seed = nullptr;
		if (!comms_aborted) {
			if (w.c_urgent) (void)g_rccl.CommDestroy(w.c_urgent);
			if (w.c_deferred) (void)g_rccl.CommDestroy(w.c_deferred);
			if (w.c_tail) (void)g_rccl.CommDestroy(w.c_tail);
		}
		w.c_urgent = w.c_deferred = w.c_tail = nullptr;
	} else {
		if (w.dstream) (void)hipStreamDestroy(w.dstream);
		if (w.own_stream && w.stream) (void)hipStreamDestroy(w.stream);
		w.dstream = w.stream = nullptr;
	}
}

// rows (keypoint slots, reference order) whose descriptor a partitioned handle computes: the library deals the accepted keypoints in
// its processing order -- keypoint level descending, stable (kernels_orient.hip k_slots) -- position p goes to rank p % world
std::vector<int> described_rows(const std::vector<sift3d_keypoint> &kp, int rank, int world) {
	std::vector<int> order(kp.size());
	for (size_t i = 0; i < kp.size(); i++) order[i] = (int)i;
	std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return kp[(size_t)a].level > kp[(size_t)b].level; });
	std::vector<int> rows;
	for (size_t p = (size_t)rank; p < order.size(); p += (size_t)world) rows.push_back(order[p]);
	std::sort(rows.begin(), rows.end());
	return rows;
}

}  // namespace

extern "C" int sift3d_sharded_destroy(sift3d_sharded_handle H) {
	if (!H) return SIFT3D_OK;
	for (int phase = 0; phase < 2; phase++)
		for (Worker &w : H->workers) destroy_worker(w, phase, H->comms_aborted.load());
	delete H;
	return SIFT3D_OK;
}

extern "C" const char *sift3d_sharded_error(sift3d_sharded_handle H) { return H ? H->err.c_str() : ""; }

extern "C" int sift3d_sharded_create(sift3d_sharded_handle *out, const float *volume, int nx, int ny, int nz, const sift3d_params *params,
                                     const int *devices, int ndev, int sim_ranks, int sharded_octaves) {
	return sift3d_sharded_create_ex(out, volume, nx, ny, nz, params, devices, ndev, sim_ranks, sharded_octaves, 0u);
}

extern "C" int sift3d_sharded_create_ex(sift3d_sharded_handle *out, const float *volume, int nx, int ny, int nz, const sift3d_params *params,
                                        const int *devices, int ndev, int sim_ranks, int sharded_octaves, unsigned flags) {
	if (!out) return SIFT3D_ERR_ARG;
	*out = nullptr;
	if (!volume || nx <= 0 || ny <= 0 || nz <= 0 || !devices || ndev < 1 || sim_ranks < 0 || (sim_ranks > 0 && ndev != 1)) {
		set_last_error("sift3d_sharded_create: bad argument (simulated ranks need exactly one device)");
		return SIFT3D_ERR_ARG;
	}
	int have = 0;
	if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) { set_last_error("no HIP device visible: this library has no CPU fallback"); return SIFT3D_ERR_NO_DEVICE; }
	for (int i = 0; i < ndev; i++) if (devices[i] < 0 || devices[i] >= have) { set_last_error("device index out of range"); return SIFT3D_ERR_ARG; }
	sift3d_sharded *H = new sift3d_sharded();
	auto fail = [&](int rc, const std::string &why) { set_last_error(why); sift3d_sharded_destroy(H); return rc; };
	H->nx = nx; H->ny = ny; H->nz = nz;
	if (params) H->p = *params; else sift3d_default_params(&H->p);
	H->sim = sim_ranks > 0;
	H->world = H->sim ? sim_ranks : ndev;
	H->devices.assign(devices, devices + ndev);
	H->levels = H->p.num_kp_levels; H->ng = H->levels + 3;
	H->partial = (flags & SIFT3D_SHARDED_PARTIAL_WINDOWS) != 0;
	if ((H->partial ? sift3d_slab_min_halo_partial(&H->p, &H->halo) : sift3d_slab_min_halo(&H->p, &H->halo)) != SIFT3D_OK) return fail(SIFT3D_ERR_ARG, "bad parameters");
	H->noct = octaves_total(nx, ny, nz);
	if (H->noct < 1) return fail(SIFT3D_ERR_ARG, "volume too small for one octave");
	// sharded octaves: as asked, but none whose planes are smaller than the level kernel's tile (+ widest half width) or thinner than the ranks
	int S = std::max(1, std::min(sharded_octaves > 0 ? sharded_octaves : 2, H->noct));
	auto fits = [](int n) { return n == 32 || n >= 40; };  // one 32 x 32 tile, or room for a shifted last tile behind the widest mirror zone
	while (S > 1 && (!fits(nx >> (S - 1)) || !fits(ny >> (S - 1)) || (nz >> S) < H->world)) S--;
	H->S = S;
	Bounds b;
	if (!slab_bounds(nz, H->world, 1 << S, b)) return fail(SIFT3D_ERR_ARG, "too few planes for this many slabs");
	if (!H->sim) {
		std::lock_guard<std::mutex> lk(g_rccl_mu);
		std::string e;
		if (!g_rccl.load(e)) return fail(SIFT3D_ERR_STATE, e);
	}
	H->workers.resize((size_t)H->world);
	hipStream_t shared = nullptr;
	for (int r = 0; r < H->world; r++) {
		Worker &w = H->workers[(size_t)r];
		w.rank = r; w.device = H->sim ? devices[0] : devices[r];
#define CR_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(SIFT3D_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
#define CR_ABI(call) do { int r_ = (call); if (r_ != SIFT3D_OK) return fail(r_, std::string(#call) + ": " + sift3d_last_error()); } while (0)
		CR_HIP(hipSetDevice(w.device));
		if (H->sim && shared) w.stream = shared;  // simulated ranks share ONE stream: their "sends" are copies ordered on it
		else { CR_HIP(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking)); w.own_stream = true; if (H->sim) shared = w.stream; }
		if (!H->sim) CR_HIP(hipStreamCreateWithFlags(&w.dstream, hipStreamNonBlocking));
		CR_HIP(hipEventCreateWithFlags(&w.ev_level, hipEventDisableTiming));
		CR_HIP(hipEventCreateWithFlags(&w.ev_def, hipEventDisableTiming));
		CR_HIP(hipEventCreateWithFlags(&w.ev_seed, hipEventDisableTiming));
		Bounds bb = b;
		int dx = nx, dy = ny, dz = nz;
		for (int o = 0; o < S; o++) {
			w.stages.emplace_back();
			Stage &st = w.stages.back();
			st.octave = o; st.nx = dx; st.ny = dy; st.nz = dz; st.bounds = bb; st.z0 = bb[(size_t)r].first; st.z1 = bb[(size_t)r].second;
			st.plane = (size_t)dx * dy;
			sift3d_slab_desc d{dx, dy, dz, st.z0, st.z1, H->halo, H->noct, o};
			if (st.z1 > st.z0) {
				CR_ABI(sift3d_slab_arena_floats(&d, &H->p, &st.arena_floats));
				CR_HIP(hipMalloc(&st.arena, sizeof(float) * st.arena_floats));
				CR_ABI(sift3d_slab_create(&st.ctx, &d, &H->p, w.device, st.arena, st.arena_floats));
				CR_ABI(sift3d_set_stream(st.ctx, w.stream));
				if (H->partial) CR_ABI(sift3d_slab_set_desc_partial(st.ctx, 1));
			} else {
				return fail(SIFT3D_ERR_ARG, "a rank would own no planes of a sharded octave");
			}
			float *dm = nullptr;
			CR_HIP(hipMalloc(&dm, sizeof(float) * 8));
			CR_HIP(hipMemset(dm, 0, sizeof(float) * 8));
			w.dogmax.push_back(dm);
			bb = halve_bounds(bb, dz);
			dx /= 2; dy /= 2; dz /= 2;
		}
		if (r == 0) {
			H->sx = dx; H->sy = dy; H->sz = dz;
			H->counts2.clear();
			for (auto &p : bb) H->counts2.push_back(p.second - p.first);
			sift3d_handle c0 = w.stages[0].ctx;
			for (int i = 0; i < H->ng; i++) { int v = 0; CR_ABI(sift3d_slab_halo_planes(c0, i, &v)); H->need.push_back(v); CR_ABI(sift3d_slab_level_hw(c0, i, &v)); H->hws.push_back(v); }
		}
		if (H->noct > S) {
			CR_ABI(sift3d_create_seeded(&w.tail, dx, dy, dz, S, H->noct, &H->p, w.device));
			CR_ABI(sift3d_set_describe_partition(w.tail, r, H->world));
			CR_HIP(hipMalloc(&w.seed, sizeof(float) * (size_t)dx * dy * std::max(dz, 1)));
			const int mine = *std::max_element(H->counts2.begin(), H->counts2.end());
			CR_HIP(hipMalloc(&w.seed_mine, sizeof(float) * (size_t)dx * dy * std::max(mine, 1)));
		}
	}
	if (H->partial) {
		// one finish launch adds at most kDescSegs parts: the owner's and those of five z-neighbours (slabs thinner than that take the whole-window path)
		H->kp_count.assign((size_t)S, std::vector<int>((size_t)H->world, 0));
		H->redo_count = H->kp_count;
		for (const Stage &st : H->workers[0].stages) {
			int reach = 0;
			CR_ABI(sift3d_slab_desc_reach(st.ctx, &reach));
			for (const std::vector<int> &nb : window_neighbours(st.bounds, reach))
				if ((int)nb.size() + 1 > kDescSegs)
					return fail(SIFT3D_ERR_ARG, "partial descriptor windows: a slab of octave " + std::to_string(st.octave) + " is so thin that a window spans more than " +
					                                std::to_string(kDescSegs) + " ranks; use fewer sharded octaves or whole windows");
		}
	}
	if (!H->sim) {
		// three communicators over the same devices: urgent halos + reductions, deferred halos, the tail's reduction
		std::vector<ncclComm_t> c((size_t)H->world);
		for (int k = 0; k < 3; k++) {
			ncclResult_t r = g_rccl.CommInitAll(c.data(), H->world, H->devices.data());
			if (r != ncclSuccess) return fail(SIFT3D_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
			for (int q = 0; q < H->world; q++) (k == 0 ? H->workers[(size_t)q].c_urgent : k == 1 ? H->workers[(size_t)q].c_deferred : H->workers[(size_t)q].c_tail) = c[(size_t)q];
		}
	}
	// ---- constructor work (Src/cSIFT3D.cc:146-163): copy the owned planes, max-abs normalise over the WHOLE volume, exchange the
	// input halo of the base blur
	const size_t pl = (size_t)nx * ny;
	float gmax = 0.f;
	std::vector<float> lmax((size_t)H->world, 0.f);
	if (H->sim) {
		for (Worker &w : H->workers) {
			CR_HIP(hipSetDevice(w.device));
			Stage &st = w.stages[0];
			CR_ABI(sift3d_slab_upload(st.ctx, volume + pl * (size_t)st.z0, st.z0, st.z1, 0));
			CR_ABI(sift3d_slab_input_absmax(st.ctx, &lmax[(size_t)w.rank]));
		}
	} else {
		// one host thread per GPU: every rank stages its own slab through its device's pinned pool (csrc/staging.hip), so the
		// constructor's H2D scales with the number of GPUs instead of running the slabs one after the other
		std::vector<std::thread> th;
		std::vector<int> rcs((size_t)H->world, SIFT3D_OK);
		std::vector<std::string> errs((size_t)H->world);
		for (int r = 0; r < H->world; r++)
			th.emplace_back([&, r] {
				Worker &w = H->workers[(size_t)r];
				Stage &st = w.stages[0];
				int rc = hipSetDevice(w.device) == hipSuccess ? SIFT3D_OK : SIFT3D_ERR_HIP;
				if (rc == SIFT3D_OK) rc = sift3d_slab_upload(st.ctx, volume + pl * (size_t)st.z0, st.z0, st.z1, 0);
				if (rc == SIFT3D_OK) rc = sift3d_slab_input_absmax(st.ctx, &lmax[(size_t)r]);
				if (rc != SIFT3D_OK) errs[(size_t)r] = sift3d_last_error();  // (the error text is thread-local)
				rcs[(size_t)r] = rc;
			});
		for (auto &t : th) t.join();
		for (int r = 0; r < H->world; r++) if (rcs[(size_t)r]) return fail(rcs[(size_t)r], "slab upload of rank " + std::to_string(r) + ": " + errs[(size_t)r]);
	}
	for (float v : lmax) gmax = std::max(gmax, v);  // (one process holds every rank: the MAX all-reduce is a host loop)
	for (Worker &w : H->workers) { CR_HIP(hipSetDevice(w.device)); CR_ABI(sift3d_slab_input_scale(w.stages[0].ctx, gmax)); }
	{
		const std::vector<Transfer> ts = halo_transfers(b, nz, KIND_INPUT, 0, 0, H->hws[0] + 1, 0);
		int rc = SIFT3D_OK;
		if (H->sim) {
			std::vector<Worker *> ws;
			for (Worker &w : H->workers) ws.push_back(&w);
			rc = exchange(H, ws, ts, 0);
			if (rc == SIFT3D_OK && hipStreamSynchronize(H->workers[0].stream) != hipSuccess) rc = SIFT3D_ERR_HIP;
		} else {
			std::vector<std::thread> th;
			std::vector<int> rcs((size_t)H->world, SIFT3D_OK);
			for (int r = 0; r < H->world; r++)
				th.emplace_back([&, r] {
					Worker &w = H->workers[(size_t)r];
					std::vector<Worker *> ws{&w};
					(void)hipSetDevice(w.device);
					rcs[(size_t)r] = exchange(H, ws, ts, 0);
					if (rcs[(size_t)r] != SIFT3D_OK) abort_all(H);
					else if (hipStreamSynchronize(w.stream) != hipSuccess) rcs[(size_t)r] = SIFT3D_ERR_HIP;
				});
			for (auto &t : th) t.join();
			for (int v : rcs) if (v) rc = v;
		}
		if (rc) return fail(rc, "input halo exchange failed: " + H->workers[0].err);
	}
#undef CR_HIP
#undef CR_ABI
	*out = H;
	return SIFT3D_OK;
}

extern "C" int sift3d_sharded_run(sift3d_sharded_handle H) {
	if (!H) return SIFT3D_ERR_ARG;
	if (H->failed.load()) {  // a rank failed in an earlier run and the communicators were aborted: nothing is restarted in place
		H->err = "this sharded extractor is dead (an earlier run failed and its communicators were aborted): destroy it";
		set_last_error(H->err);
		return SIFT3D_ERR_STATE;
	}
	const auto t0 = std::chrono::steady_clock::now();
	int rc = SIFT3D_OK;
	if (H->sim) {
		std::vector<Worker *> ws;
		for (Worker &w : H->workers) ws.push_back(&w);
		rc = run_local(H, ws);
		if (rc) for (Worker &w : H->workers) if (!w.err.empty()) { H->err = w.err; break; }
	} else {
		std::vector<std::thread> th;
		std::vector<int> rcs((size_t)H->world, SIFT3D_OK);
		for (int r = 0; r < H->world; r++)
			th.emplace_back([&, r] {
				std::vector<Worker *> ws{&H->workers[(size_t)r]};
				rcs[(size_t)r] = run_local(H, ws);
				if (rcs[(size_t)r] != SIFT3D_OK) abort_all(H);  // frees the z-neighbours blocked in a receive / reduction with this rank
			});
		for (auto &t : th) t.join();
		// the rank that failed FIRST carries the cause; the others report "aborted: another rank failed"
		for (int pass = 0; pass < 2 && rc == SIFT3D_OK; pass++)
			for (int r = 0; r < H->world; r++) {
				const std::string &e = H->workers[(size_t)r].err;
				if (rcs[(size_t)r] && (pass == 1 || e.compare(0, 8, "aborted:") != 0)) { rc = rcs[(size_t)r]; H->err = "rank " + std::to_string(r) + ": " + e; break; }
			}
	}
	if (rc) { set_last_error(H->err); return rc; }
	H->times[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	// ---- merge: reference order (octave, level, z, y, x) (Src/cSIFT3D.cc:373-416): per sharded octave the ranks' lists concatenated and
	// sorted by (level, z, y, x); then the tail, whose records are complete on every rank and whose descriptor rows are dealt
	H->kp.clear(); H->desc.clear();
	for (int s = 0; s < H->S; s++) {
		std::vector<std::pair<const sift3d_keypoint *, const float *>> items;
		for (Worker &w : H->workers) {
			const int a = s ? w.kp_stage_end[(size_t)s - 1] : 0, e = w.kp_stage_end[(size_t)s];
			for (int i = a; i < e; i++) items.push_back({&w.kp[(size_t)i], &w.desc[(size_t)i * kDesc]});
		}
		std::stable_sort(items.begin(), items.end(), [](const std::pair<const sift3d_keypoint *, const float *> &A, const std::pair<const sift3d_keypoint *, const float *> &B) {
			const sift3d_keypoint &a = *A.first, &b = *B.first;
			if (a.level != b.level) return a.level < b.level;
			if (a.z != b.z) return a.z < b.z;
			if (a.y != b.y) return a.y < b.y;
			return a.x < b.x;
		});
		for (auto &it : items) { H->kp.push_back(*it.first); H->desc.insert(H->desc.end(), it.second, it.second + kDesc); }
	}
	if (H->noct > H->S) {
		const Worker &w0 = H->workers[0];
		const size_t o = H->kp.size();
		H->kp.insert(H->kp.end(), w0.tkp.begin(), w0.tkp.end());
		H->desc.resize((o + w0.tkp.size()) * kDesc, 0.0f);
		for (const Worker &w : H->workers)
			for (int row : described_rows(w0.tkp, w.rank, H->world))
				memcpy(&H->desc[(o + (size_t)row) * kDesc], &w.tdesc[(size_t)row * kDesc], sizeof(float) * kDesc);
	}
	H->times[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	H->ran = true;
	return SIFT3D_OK;
}

extern "C" int sift3d_sharded_num_keypoints(sift3d_sharded_handle H, int *n) {
	if (!H || !n) return SIFT3D_ERR_ARG;
	*n = H->ran ? (int)H->kp.size() : 0;
	return SIFT3D_OK;
}

extern "C" int sift3d_sharded_get_keypoints(sift3d_sharded_handle H, sift3d_keypoint *out, float *desc) {
	if (!H) return SIFT3D_ERR_ARG;
	if (!H->ran) return SIFT3D_OK;
	if (out && !H->kp.empty()) memcpy(out, H->kp.data(), sizeof(sift3d_keypoint) * H->kp.size());
	if (desc && !H->desc.empty()) memcpy(desc, H->desc.data(), sizeof(float) * H->desc.size());
	return SIFT3D_OK;
}

extern "C" int sift3d_sharded_info(sift3d_sharded_handle H, int *world, int *sharded_octaves, int *halo, double seconds[2]) {
	if (!H) return SIFT3D_ERR_ARG;
	if (world) *world = H->world;
	if (sharded_octaves) *sharded_octaves = H->S;
	if (halo) *halo = H->halo;
	if (seconds) { seconds[0] = H->times[0]; seconds[1] = H->times[1]; }
	return SIFT3D_OK;
}
