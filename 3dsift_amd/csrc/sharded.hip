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
//   windows           the descriptor windows are split along z over the ranks (r05; the default since r06): records to the z-neighbours, every
//                     rank marches its part, 768 int32 + the mass back, the owner finishes (3dsift_amd/slab.py does the same and is where the
//                     protocol is tested over gloo): the halos of G[1..3] are 8 / 10 / 12 planes instead of 24 / 30 / 38.  Whole windows on the
//                     wide halos remain for slabs so thin that a window would span more than six ranks (and behind SIFT3D_SHARDED_WHOLE_WINDOWS).
//   tail              (r06) the remaining octaves run ONCE, on the last rank: every rank sends its planes of their seed level (1 / 8^S of a
//                     level) to that rank, which runs them as an ordinary seeded extractor on streams of its own beside its slab; that rank
//                     owns fewer planes of the sharded octaves in exchange (slab_bounds).  r05 ran the tail on every rank (all-gather, orientation
//                     and descriptors dealt and all-reduced): eight times the pyramid and the extrema of the tail for an eighth of its descriptors.
//   no read-backs     (r06) detection and orientation of every sharded octave are enqueued before the first count is read; the first round's
//                     finish leaves its count of flagged records in pinned memory and is only looked at when everything else is enqueued.
// Transport:
//   RCCL   one host thread per GPU, three communicators per rank (urgent / deferred / tail: operations of one communicator must be
//          issued in the same order on every rank, and the three flows run concurrently; the tail's carries the gather of the seed level
//          alone), point-to-point ncclSend / ncclRecv between
//          z-neighbours over xGMI inside ncclGroupStart / End, on the rank's own streams: no host synchronisation between the
//          levels.  librccl is opened at run time (dlopen), so the library loads on hosts without it.
//          Failure protocol (r04): the first rank (or tail thread) whose step fails aborts EVERY communicator of the handle
//          (ncclCommAbort), so that z-neighbours blocked in a receive / reduction kernel return instead of wedging the process; the
//          handle is then dead -- sift3d_sharded_run returns an error from now on, the caller destroys it and, if it wants to go on,
//          starts a fresh process (nothing is restarted in place).
//   SIM    all ranks in this process on ONE GPU and one stream: the sends of an exchange step are one copy launch (kernels_pyramid.hip
//          k_copy_segments), the MAX reduction a device kernel.  This is how the 1-GPU test boxes check the driver (same code, same
//          plan) against the single-volume result, and how the work of eight ranks is added up on one GPU (bench.py --sim-ranks).
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
#include <deque>
#include <memory>
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

// ---- planning (halo / window plans as in 3dsift_amd/slab.py, which the CPU tests cover; the slab boundaries: weighted, see slab_bounds_weighted) ----
int octaves_total(int nx, int ny, int nz) {  // Src/cSIFT3D.cc:254-255
	const int mn = std::min(nx, std::min(ny, nz));
	return std::max(0, (int)log2f((float)mn) - 2);
}

// owned plane ranges per rank: contiguous, starts on multiples of `align`, as equal as possible, remainder to the last rank.
// tail_planes (r06): the last rank also runs the replicated tail, which costs about as much as that many planes of the first sharded
// octave: it owns that many planes fewer than the others (never less than half an even share)
bool slab_bounds(int nz, int world, int align, Bounds &out, int tail_planes = 0) {
	const int units = nz / align;
	if (units < world) return false;
	std::vector<int> n((size_t)world, 0);
	const int tu = world > 1 ? std::max(0, tail_planes / align) : 0;
	if (tu == 0) {  // the even deal: remainders to the first ranks
		const int base = units / world, rem = units % world;
		for (int r = 0; r < world; r++) n[(size_t)r] = base + (r < rem ? 1 : 0);
	} else {
		// every rank carries (units + tu) / world units of work, the last rank tu of them as the tail: it owns that many units fewer (at
		// least half an even share, at least one unit); the others deal the rest evenly
		const double target = (double)(units + tu) / (double)world;
		int last = (int)lround(target - (double)tu);
		last = std::max(last, std::max(1, units / world / 2));
		last = std::min(last, units / world);
		const int rest = units - last, base = rest / (world - 1), rem = rest % (world - 1);
		for (int r = 0; r < world - 1; r++) n[(size_t)r] = base + (r < rem ? 1 : 0);
		n[(size_t)world - 1] = last;
	}
	out.clear();
	int z = 0;
	for (int r = 0; r < world; r++) {
		out.push_back({z, z + align * n[(size_t)r]});
		z += align * n[(size_t)r];
	}
	out.back().second = nz;
	return true;
}

// r06 (late): owned plane ranges by WEIGHT, any integer boundaries.  A rank's step costs a + b * planes, and a is not the same for every rank: a rank
// exchanges halos and window parts with one z-neighbour (the first and the last rank) or with two, and one rank also runs the tail -- measured at
// 1024 x 1024 x 512 over 2 / 4 / 8 ranks (profiles/r06aa_slab_sim.json): b = 43.6 us per plane, a side 0.4 ms = 9 planes, the tail 0.65 ms = 15 planes;
// the even deal left the first rank at 3.2 ms, the inner ones at 3.65 and the tail rank at 3.9-4.15.  side_w / tail_w: those costs in planes.  Every
// rank owns at least min_planes.
bool slab_bounds_weighted(int nz, int world, int min_planes, double side_w, double tail_w, int tail_rank, Bounds &out) {
	if (world < 1 || min_planes < 1 || (long)world * min_planes > nz) return false;
	std::vector<double> cost((size_t)world, 0.0), p((size_t)world, 0.0);
	for (int r = 0; r < world; r++) cost[(size_t)r] = side_w * (double)((r > 0 ? 1 : 0) + (r + 1 < world ? 1 : 0)) + (r == tail_rank ? tail_w : 0.0);
	std::vector<char> pinned((size_t)world, 0);  // ranks held at min_planes
	for (int it = 0; it <= world; it++) {
		double fixed = 0.0, held = 0.0;
		int nfree = 0;
		for (int r = 0; r < world; r++) { if (pinned[(size_t)r]) held += (double)min_planes; else { fixed += cost[(size_t)r]; nfree++; } }
		if (nfree == 0) break;
		const double T = ((double)nz - held + fixed) / (double)nfree;
		bool again = false;
		for (int r = 0; r < world; r++) {
			if (pinned[(size_t)r]) { p[(size_t)r] = (double)min_planes; continue; }
			p[(size_t)r] = T - cost[(size_t)r];
			if (p[(size_t)r] < (double)min_planes) { pinned[(size_t)r] = 1; again = true; }
		}
		if (!again) break;
	}
	out.clear();
	double cum = 0.0;
	int z = 0;
	for (int r = 0; r < world; r++) {
		cum += p[(size_t)r];
		int z1 = r + 1 == world ? nz : (int)lround(cum);
		z1 = std::max(z1, z + min_planes);
		z1 = std::min(z1, nz - (world - 1 - r) * min_planes);
		out.push_back({z, z1});
		z = z1;
	}
	return true;
}

// plane k of octave o+1 is plane 2k of octave o (Src/cSIFT3D.cc:321-344): a rank owns the planes k whose plane 2k it owns in the octave above --
// [ceil(z0 / 2), ceil(z1 / 2)), any integer boundaries (r06; the aligned deal made them even)
Bounds halve_bounds(const Bounds &b, int nz) {
	Bounds o;
	for (auto &p : b) o.push_back({std::min((p.first + 1) / 2, nz / 2), std::min((p.second + 1) / 2, nz / 2)});
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

struct Worker {  // the sharded octaves of one rank (+ the tail's extractor on the last rank)
	int rank = 0, device = 0;
	// r06: every sharded octave has a stream of its own (sstream[s]; stream == sstream[0]) -- octave s + 1 only depends on the seed level octave
	// s decimates (ev_next[s]), and its launches are small and latency-bound: they run beside the machine-filling launches of the octave
	// above, as the octaves of the single-GPU extractor do -- and, for RCCL, a stream for its deferred halos and a pair of communicators
	// (operations of one communicator are issued in one order; the octaves' flows interleave freely)
	hipStream_t stream = nullptr;
	std::vector<hipStream_t> sstream, sdstream;
	hipStream_t tstream = nullptr;                    // tail rank: the stream of the tail's extractor (the seed level is gathered on it)
	bool own_stream = false;                          // (simulated ranks share rank 0's streams)
	std::vector<hipEvent_t> ev_level, ev_def, ev_next;
	hipEvent_t ev_seed = nullptr;
	std::vector<Stage> stages;
	std::vector<float *> dogmax;  // per stage: 8 floats (device)
	sift3d_handle tail = nullptr; // tail rank only: the seeded extractor of the octaves >= S
	float *seed_dst = nullptr;    // tail rank: level 0 of the tail's first octave (inside the tail's arena)
	float *seed_mine = nullptr;   // this rank's planes of that level (other ranks: a buffer of their own; tail rank: a pointer into seed_dst)
	bool seed_mine_owned = false;
	std::vector<ncclComm_t> c_urgent, c_deferred;     // per sharded octave
	ncclComm_t c_tail = nullptr;
	// partial descriptor windows: records / histograms / masses, per stage (grow-only, device); [2 s] the first round's, [2 s + 1] the rare second
	// round's (its own memory: the first round's lists stay in place, which is what a solo re-run of a neighbour reads)
	std::vector<char *> pscratch; std::vector<size_t> pscratch_bytes;
	std::string err;
	// tail rank: the thread that enqueues the tail's pipeline beside this rank's own launches (start_tail), and what it came back with
	std::thread tail_thread;
	int tail_rc = 0;
	std::string tail_err;
	// copy transport: the events this rank has recorded for its sends in the current step (reused from step to step: a step ends with every
	// stream of every rank drained), and where its peers' DoG maxima land
	std::vector<hipEvent_t> evp;
	size_t evp_i = 0;
	std::vector<float *> dogmax_in;  // per stage: world x 8 floats (device)
};

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

// a stage's first round of partial windows: who exchanges with whom, how many records, where every rank's lists live
struct PartStage {
	std::vector<std::vector<int>> neigh;
	std::vector<int> counts;
	std::vector<PartLayout> L;
	size_t rb = 0;
};

}  // namespace

struct sift3d_sharded {
	int nx = 0, ny = 0, nz = 0, world = 1, S = 1, halo = 0, noct = 0, levels = 3, ng = 6;
	bool sim = false;
	sift3d_params p{};
	std::vector<int> devices;
	std::vector<Worker> workers;
	std::vector<std::vector<int>> need;  // per sharded octave: halo planes of every Gaussian level its consumers reach
	std::vector<int> hws;
	std::vector<char> stage_partial;     // per sharded octave: descriptor windows split along z (else whole windows on that octave's wide halos)
	std::vector<int> stage_halo;         // ... and the halo planes its level buffers carry
	std::vector<int> counts2;   // planes of the tail's seed level owned per rank
	int sx = 0, sy = 0, sz = 0; // dims of the tail's seed level
	int tail_rank = -1;         // the rank that runs the octaves >= S (-1: none)
	bool ran = false;
	double times[4] = {0, 0, 0, 0};
	std::string err;
	// descriptor windows split along z (the default): accepted keypoints / flagged records per sharded octave and rank, written by the rank
	// threads and read by all of them behind a rendezvous
	bool partial = false;
	std::vector<std::vector<int>> kp_count, redo_count;
	// simulated ranks only: the layouts of the last full run (one PartStage per sharded octave, L indexed by rank), and the flag of a SOLO
	// re-run of one rank on the buffers that run left behind (sift3d_test_sharded_time_rank: what ONE rank does in a step, timed alone)
	std::vector<PartStage> ps_last;
	bool solo = false;
	std::mutex rv_mu;
	std::condition_variable rv_cv;
	int rv_arrived = 0;
	unsigned rv_gen = 0;
	// failure protocol of the RCCL transport: `failed` is set once, by the first rank whose step failed, which then aborts every
	// communicator.  Ranks hold comm_mu shared while they ISSUE RCCL calls (short, host side) and the aborter takes it exclusively,
	// so that no thread is inside a call on a communicator while it is torn down.
	std::atomic<bool> failed{false};
	std::atomic<bool> comms_aborted{false};  // abort_all ran: the communicators are gone (their pointers are left alone)
	std::shared_timed_mutex comm_mu;
	// r06, the COPY transport (SIFT3D_SHARDED_COPY_TRANSPORT): the same rank threads, streams and plan as the RCCL transport, but a "send" is a message
	// -- an event recorded on the sender's stream behind the producer + the source address -- in the mailbox of the (sender, receiver) pair, and a
	// "receive" waits for the message (host), makes its stream wait for the event and copies (one copy launch per exchange step on one device,
	// hipMemcpyPeerAsync between devices).  Both sides walk the same transfer list in the same order, so a FIFO per pair matches them.  A step
	// writes every buffer it sends from once, and ends with every rank's streams drained: a sender never overwrites what a peer still reads.
	// `devices` may name one device several times: N rank THREADS on one GPU -- what a one-GPU box can run of the multi-threaded driver.
	bool ghost0 = false;  // SIFT3D_SHARDED_GHOST_OCTAVE0: octave 0's levels on ghost zones, nothing of octave 0 is exchanged
	bool copies = false;
	struct Msg { hipEvent_t ev; const void *p; };
	struct Mailbox { std::mutex mu; std::condition_variable cv; std::deque<Msg> q; };
	std::unique_ptr<Mailbox[]> mail;  // [src * world + dst]
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
	if (H->copies) {  // (no communicators: the flag alone frees the ranks that wait for a message or at a rendezvous)
		for (int i = 0; i < H->world * H->world; i++) H->mail[(size_t)i].cv.notify_all();
		return;
	}
	const bool locked = H->comm_mu.try_lock_for(std::chrono::seconds(2));
	// The communicator POINTERS are never written after creation: a rank that reads one (under the shared lock, after SH_LIVE) cannot see
	// a pointer in the middle of a store.  A rank already inside an RCCL call when the wait above timed out is what the abort is for.
	H->comms_aborted.store(true);  // (ncclCommAbort frees the communicator: destroy must not hand it to ncclCommDestroy again)
	for (Worker &w : H->workers)
		{
			std::vector<ncclComm_t> all = w.c_urgent;
			all.insert(all.end(), w.c_deferred.begin(), w.c_deferred.end());
			all.push_back(w.c_tail);
			for (ncclComm_t c : all)
				if (c) (void)g_rccl.CommAbort(c);
		}
	if (locked) H->comm_mu.unlock();
}
#define SH_LIVE(H, w) do { if ((H)->failed.load()) { set_err((w), "aborted: another rank failed"); return SIFT3D_ERR_STATE; } } while (0)

// ---- the copy transport's three moves --------------------------------------------------------------------------------------------
// an event of this rank, recorded on `st` (behind everything the rank has enqueued there)
int mb_mark(Worker &w, hipStream_t st, hipEvent_t &ev) {
	if (w.evp_i == w.evp.size()) {
		hipEvent_t e = nullptr;
		SH_HIP(w, hipEventCreateWithFlags(&e, hipEventDisableTiming));
		w.evp.push_back(e);
	}
	ev = w.evp[w.evp_i++];
	SH_HIP(w, hipEventRecord(ev, st));
	return SIFT3D_OK;
}
void mb_post(sift3d_sharded *H, int src, int dst, hipEvent_t ev, const void *p) {
	sift3d_sharded::Mailbox &mb = H->mail[(size_t)src * (size_t)H->world + (size_t)dst];
	{ std::lock_guard<std::mutex> g(mb.mu); mb.q.push_back(sift3d_sharded::Msg{ev, p}); }
	mb.cv.notify_all();
}
// the next message of src for this rank: `st` waits for its event; a dead handle lets the waiter go
int mb_take(sift3d_sharded *H, Worker &w, int src, hipStream_t st, const void *&p) {
	sift3d_sharded::Mailbox &mb = H->mail[(size_t)src * (size_t)H->world + (size_t)w.rank];
	sift3d_sharded::Msg m{nullptr, nullptr};
	{
		std::unique_lock<std::mutex> lk(mb.mu);
		while (mb.q.empty()) {
			if (H->failed.load()) { set_err(w, "aborted: another rank failed"); return SIFT3D_ERR_STATE; }
			mb.cv.wait_for(lk, std::chrono::milliseconds(20));
		}
		m = mb.q.front(); mb.q.pop_front();
	}
	SH_HIP(w, hipStreamWaitEvent(st, m.ev, 0));
	p = m.p;
	return SIFT3D_OK;
}
// n bytes from a peer's buffer into this rank's, on this rank's stream: collected into `cs` on one device, a peer copy between devices
int mb_copy(sift3d_sharded *H, Worker &w, int src, const void *sp, void *dp, size_t bytes, CopySegs &cs, hipStream_t st) {
	const int sdev = H->workers[(size_t)src].device;
	if (!bytes) return SIFT3D_OK;
	if (sdev != w.device) { SH_HIP(w, hipMemcpyPeerAsync(dp, w.device, sp, sdev, bytes, st)); return SIFT3D_OK; }
	if (bytes & 3) { SH_HIP(w, hipMemcpyAsync(dp, sp, bytes, hipMemcpyDeviceToDevice, st)); return SIFT3D_OK; }
	cs.src[cs.n] = static_cast<const float *>(sp); cs.dst[cs.n] = static_cast<float *>(dp); cs.floats[cs.n] = bytes / 4; cs.n++;
	if (cs.n == kCopySegs) { launch_copy_segments(cs, st); cs.n = 0; }
	return SIFT3D_OK;
}

// posts the transfers this set of local workers takes part in.  SIM: device copies on the shared stream.  RCCL: one group of
// sends / receives of the one local rank on `comm` / `stream` of the given flow (0 urgent, 1 deferred).
int exchange(sift3d_sharded *H, std::vector<Worker *> &ws, const std::vector<Transfer> &ts, int flow) {
	if (ts.empty()) return SIFT3D_OK;
	const size_t sgi = (size_t)ts[0].stage;  // (a call's transfers belong to one sharded octave)
	if (H->sim) {
		Worker &w0 = *ws[0];
		hipStream_t sst = w0.sstream[sgi];
		CopySegs cs;
		for (const Transfer &t : ts) {
			if (H->solo && t.dst != w0.rank) continue;  // (solo: only what this rank RECEIVES; its neighbours' buffers hold the last full run's planes)
			Stage &s = H->workers[(size_t)t.src].stages[(size_t)t.stage], &d = H->workers[(size_t)t.dst].stages[(size_t)t.stage];
			float *sp = s.view(t.kind, t.idx, t.zg0, t.zg1), *dp = d.view(t.kind, t.idx, t.zg0, t.zg1);
			if (!sp || !dp) { set_err(w0, "halo transfer outside a level buffer"); return SIFT3D_ERR_STATE; }
			cs.src[cs.n] = sp; cs.dst[cs.n] = dp; cs.floats[cs.n] = s.plane * (size_t)(t.zg1 - t.zg0); cs.n++;
			if (cs.n == kCopySegs) { launch_copy_segments(cs, sst); cs.n = 0; }
		}
		launch_copy_segments(cs, sst);
		SH_HIP(w0, hipGetLastError());
		return SIFT3D_OK;
	}
	Worker &w = *ws[0];
	hipStream_t st = flow ? w.sdstream[sgi] : w.sstream[sgi];
	bool any = false, sends = false;
	for (const Transfer &t : ts) { any = any || t.src == w.rank || t.dst == w.rank; sends = sends || t.src == w.rank; }
	if (!any) return SIFT3D_OK;
	if (H->copies) {
		SH_LIVE(H, w);
		hipEvent_t ev = nullptr;
		int rc;
		if (sends && (rc = mb_mark(w, st, ev)) != SIFT3D_OK) return rc;
		for (const Transfer &t : ts) {  // every send first (none of them waits), then the receives
			if (t.src != w.rank) continue;
			float *p = w.stages[(size_t)t.stage].view(t.kind, t.idx, t.zg0, t.zg1);
			if (!p) { set_err(w, "halo transfer outside a level buffer"); return SIFT3D_ERR_STATE; }
			mb_post(H, w.rank, t.dst, ev, p);
		}
		CopySegs cs;
		for (const Transfer &t : ts) {
			if (t.dst != w.rank) continue;
			Stage &s = w.stages[(size_t)t.stage];
			float *p = s.view(t.kind, t.idx, t.zg0, t.zg1);
			if (!p) { set_err(w, "halo transfer outside a level buffer"); return SIFT3D_ERR_STATE; }
			const void *sp = nullptr;
			if ((rc = mb_take(H, w, t.src, st, sp)) != SIFT3D_OK) return rc;
			if ((rc = mb_copy(H, w, t.src, sp, p, sizeof(float) * s.plane * (size_t)(t.zg1 - t.zg0), cs, st)) != SIFT3D_OK) return rc;
		}
		launch_copy_segments(cs, st);
		SH_HIP(w, hipGetLastError());
		return SIFT3D_OK;
	}
	ncclComm_t comm = flow ? w.c_deferred[sgi] : w.c_urgent[sgi];
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
	if (H->sim && H->solo) return SIFT3D_OK;  // (the rank's buffer still holds the global maxima of the last full run: run_local skips the export)
	if (H->sim) {
		Worker &w0 = *ws[0];
		if ((int)ws.size() <= kMaxMergePtrs && n <= 64) {  // on the device, in the shared stream's order (no host round trip)
			MaxMerge mm;
			for (Worker *w : ws) mm.p[mm.np++] = w->dogmax[(size_t)stage];
			mm.n = n;
			launch_max_merge(mm, w0.sstream[(size_t)stage]);
			SH_HIP(w0, hipGetLastError());
			return SIFT3D_OK;
		}
		std::vector<float> m((size_t)n, 0.f), t((size_t)n);
		SH_HIP(w0, hipStreamSynchronize(w0.sstream[(size_t)stage]));
		for (Worker *w : ws) {
			SH_HIP(w0, hipMemcpy(t.data(), w->dogmax[(size_t)stage], sizeof(float) * n, hipMemcpyDeviceToHost));
			for (int i = 0; i < n; i++) m[(size_t)i] = std::max(m[(size_t)i], t[(size_t)i]);
		}
		for (Worker *w : ws) SH_HIP(w0, hipMemcpy(w->dogmax[(size_t)stage], m.data(), sizeof(float) * n, hipMemcpyHostToDevice));
		return SIFT3D_OK;
	}
	Worker &w = *ws[0];
	if (H->copies) {
		// every rank's n values into this rank's scratch, then the MAX over them and its own (a peer's array may already hold ITS merge: a maximum
		// of maxima, the same result)
		SH_LIVE(H, w);
		hipStream_t st = w.sstream[(size_t)stage];
		hipEvent_t ev = nullptr;
		int rc = mb_mark(w, st, ev);
		if (rc) return rc;
		for (int r = 0; r < H->world; r++) if (r != w.rank) mb_post(H, w.rank, r, ev, w.dogmax[(size_t)stage]);
		MaxMerge mm;
		mm.p[mm.np++] = w.dogmax[(size_t)stage];
		mm.n = n;
		for (int r = 0; r < H->world; r++) {
			if (r == w.rank) continue;
			const void *sp = nullptr;
			if ((rc = mb_take(H, w, r, st, sp)) != SIFT3D_OK) return rc;
			float *slot = w.dogmax_in[(size_t)stage] + (size_t)r * 8;
			const int sdev = H->workers[(size_t)r].device;
			if (sdev != w.device) SH_HIP(w, hipMemcpyPeerAsync(slot, w.device, sp, sdev, sizeof(float) * (size_t)n, st));
			else SH_HIP(w, hipMemcpyAsync(slot, sp, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, st));
			mm.p[mm.np++] = slot;
		}
		launch_max_merge(mm, st);
		SH_HIP(w, hipGetLastError());
		return SIFT3D_OK;
	}
	std::shared_lock<std::shared_timed_mutex> live(H->comm_mu);
	SH_LIVE(H, w);
	SH_NCCL(w, g_rccl.AllReduce(w.dogmax[(size_t)stage], w.dogmax[(size_t)stage], (size_t)n, ncclFloat, ncclMax, w.c_urgent[(size_t)stage], w.sstream[(size_t)stage]));
	return SIFT3D_OK;
}

// The tail's seed level = the owned planes of all ranks, in rank order, gathered ON THE TAIL RANK straight into level 0 of its seeded
// extractor, on that extractor's stream.  The tail rank's own planes were decimated in place.
int gather_seed(sift3d_sharded *H, std::vector<Worker *> &ws) {
	const size_t pl = (size_t)H->sx * H->sy;
	std::vector<size_t> off((size_t)H->world + 1, 0);
	for (int r = 0; r < H->world; r++) off[(size_t)r + 1] = off[(size_t)r] + pl * (size_t)H->counts2[(size_t)r];
	Worker *tw = nullptr;
	for (Worker *w : ws) if (w->rank == H->tail_rank) tw = w;
	const size_t sl = (size_t)H->S - 1;  // the last sharded octave decimates the tail's seed level: on its stream
	if (H->sim) {
		Worker &w0 = *ws[0];
		if (!tw && H->solo) return SIFT3D_OK;  // (a solo rank that is not the tail rank: its piece was decimated, the send costs the GPU nothing)
		if (!tw) { set_err(w0, "no tail rank among the simulated ranks"); return SIFT3D_ERR_STATE; }
		CopySegs cs;
		std::vector<Worker *> srcs = ws;
		if (H->solo) { srcs.clear(); for (Worker &w : H->workers) srcs.push_back(&w); }  // (the other ranks' pieces of the last full run)
		for (Worker *w : srcs) {
			const size_t cnt = off[(size_t)w->rank + 1] - off[(size_t)w->rank];
			if (w == tw || !cnt) continue;
			cs.src[cs.n] = w->seed_mine; cs.dst[cs.n] = tw->seed_dst + off[(size_t)w->rank]; cs.floats[cs.n] = cnt; cs.n++;
			if (cs.n == kCopySegs) { launch_copy_segments(cs, w0.sstream[sl]); cs.n = 0; }
		}
		launch_copy_segments(cs, w0.sstream[sl]);
		SH_HIP(w0, hipEventRecord(tw->ev_seed, w0.sstream[sl]));
		SH_HIP(w0, hipStreamWaitEvent(tw->tstream, tw->ev_seed, 0));
		return SIFT3D_OK;
	}
	Worker &w = *ws[0];
	if (H->copies) {
		SH_LIVE(H, w);
		int rc;
		if (&w != tw) {  // behind the decimation on that octave's stream
			hipEvent_t ev = nullptr;
			if (off[(size_t)w.rank + 1] == off[(size_t)w.rank]) return SIFT3D_OK;
			if ((rc = mb_mark(w, w.sstream[sl], ev)) != SIFT3D_OK) return rc;
			mb_post(H, w.rank, H->tail_rank, ev, w.seed_mine);
			return SIFT3D_OK;
		}
		CopySegs cs;
		for (int r = 0; r < H->world; r++) {
			const size_t cnt = off[(size_t)r + 1] - off[(size_t)r];
			if (r == w.rank || !cnt) continue;
			const void *sp = nullptr;
			if ((rc = mb_take(H, w, r, w.tstream, sp)) != SIFT3D_OK) return rc;
			if ((rc = mb_copy(H, w, r, sp, w.seed_dst + off[(size_t)r], sizeof(float) * cnt, cs, w.tstream)) != SIFT3D_OK) return rc;
		}
		launch_copy_segments(cs, w.tstream);
		SH_HIP(w, hipGetLastError());
		SH_HIP(w, hipEventRecord(w.ev_seed, w.sstream[sl]));
		SH_HIP(w, hipStreamWaitEvent(w.tstream, w.ev_seed, 0));
		return SIFT3D_OK;
	}
	{
		std::shared_lock<std::shared_timed_mutex> live(H->comm_mu);
		SH_LIVE(H, w);
		if (&w != tw) {
			const size_t cnt = off[(size_t)w.rank + 1] - off[(size_t)w.rank];
			if (cnt) SH_NCCL(w, g_rccl.Send(w.seed_mine, cnt, ncclFloat, H->tail_rank, w.c_tail, w.sstream[sl]));  // behind the decimation on that octave's stream
			return SIFT3D_OK;
		}
		SH_NCCL(w, g_rccl.GroupStart());
		for (int r = 0; r < H->world; r++) {
			const size_t cnt = off[(size_t)r + 1] - off[(size_t)r];
			if (r != w.rank && cnt) SH_NCCL(w, g_rccl.Recv(w.seed_dst + off[(size_t)r], cnt, ncclFloat, r, w.c_tail, w.tstream));
		}
		SH_NCCL(w, g_rccl.GroupEnd());
	}
	SH_HIP(w, hipEventRecord(w.ev_seed, w.sstream[sl]));       // the tail rank's own planes (decimated in place on that octave's stream)
	SH_HIP(w, hipStreamWaitEvent(w.tstream, w.ev_seed, 0));
	return SIFT3D_OK;
}

// the tail's whole KpSiftAlgorithm enqueued on its own streams (sift3d_run_async), behind the gathered seed level -- by a thread of its own:
// ~70 launches and as many event calls take 0.4-0.6 ms of a host thread, and the rank's thread has its own launches to place meanwhile
// (enqueued inline right behind the gather it held back the rank's extrema by 0.45 ms; behind the rank's extrema and orientation launches
// it held back the rank's descriptor launches instead: the tail rank's step alone 4.1 / 3.9 ms against 3.3-3.7 for the others,
// profiles/r06q_solo_rank7.txt, r06q_solo_rank7b.txt).  join_tail collects the thread; every exit of run_local passes through it.
int start_tail(sift3d_sharded *H, std::vector<Worker *> &ws) {
	for (Worker *w : ws) {
		if (!w->tail || w->rank != H->tail_rank) continue;
		if (w->tail_thread.joinable()) w->tail_thread.join();  // (never: a step joins what it started)
		w->tail_rc = SIFT3D_OK; w->tail_err.clear();
		w->tail_thread = std::thread([w] {
			int rc = hipSetDevice(w->device) == hipSuccess ? SIFT3D_OK : SIFT3D_ERR_HIP;
			if (rc != SIFT3D_OK) w->tail_err = "hipSetDevice failed on the tail's thread";
			else if ((rc = sift3d_run_async(w->tail)) != SIFT3D_OK) w->tail_err = std::string(sift3d_error_string(rc)) + " (" + sift3d_last_error() + ")";  // (the error text is thread-local)
			w->tail_rc = rc;
		});
	}
	return SIFT3D_OK;
}
// the tail's enqueue is complete (or has failed): first error of the ranks in ws, SIFT3D_OK otherwise
int join_tail(std::vector<Worker *> &ws) {
	int rc = SIFT3D_OK;
	for (Worker *w : ws) {
		if (!w->tail_thread.joinable()) continue;
		w->tail_thread.join();
		if (w->tail_rc != SIFT3D_OK && rc == SIFT3D_OK) { set_err(*w, "tail: " + w->tail_err); rc = w->tail_rc; }
	}
	return rc;
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

int exchange_raw(sift3d_sharded *H, std::vector<Worker *> &ws, const std::vector<RawXfer> &ts, int stage) {
	if (ts.empty()) return SIFT3D_OK;
	if (H->sim) {
		Worker &w0 = *ws[0];
		hipStream_t sst = w0.sstream[(size_t)stage];
		CopySegs cs;
		for (const RawXfer &t : ts) {
			if (!t.bytes) continue;
			if (t.bytes & 3) { SH_HIP(w0, hipMemcpyAsync(t.dp, t.sp, t.bytes, hipMemcpyDeviceToDevice, sst)); continue; }  // (never: records, histograms and masses are words)
			cs.src[cs.n] = static_cast<const float *>(t.sp); cs.dst[cs.n] = static_cast<float *>(t.dp); cs.floats[cs.n] = t.bytes / 4; cs.n++;
			if (cs.n == kCopySegs) { launch_copy_segments(cs, sst); cs.n = 0; }
		}
		launch_copy_segments(cs, sst);
		SH_HIP(w0, hipGetLastError());
		return SIFT3D_OK;
	}
	Worker &w = *ws[0];
	if (H->copies) {
		SH_LIVE(H, w);
		hipStream_t st = w.sstream[(size_t)stage];
		hipEvent_t ev = nullptr;
		int rc;
		bool sends = false;
		for (const RawXfer &t : ts) sends = sends || (t.src == w.rank && t.bytes);
		if (sends && (rc = mb_mark(w, st, ev)) != SIFT3D_OK) return rc;
		for (const RawXfer &t : ts) if (t.src == w.rank && t.bytes) mb_post(H, w.rank, t.dst, ev, t.sp);
		CopySegs cs;
		for (const RawXfer &t : ts) {
			if (t.dst != w.rank || !t.bytes) continue;
			const void *sp = nullptr;
			if ((rc = mb_take(H, w, t.src, st, sp)) != SIFT3D_OK) return rc;
			if ((rc = mb_copy(H, w, t.src, sp, t.dp, t.bytes, cs, st)) != SIFT3D_OK) return rc;
		}
		launch_copy_segments(cs, st);
		SH_HIP(w, hipGetLastError());
		return SIFT3D_OK;
	}
	std::shared_lock<std::shared_timed_mutex> live(H->comm_mu);
	SH_LIVE(H, w);
	SH_NCCL(w, g_rccl.GroupStart());
	for (const RawXfer &t : ts) {
		if (t.src == w.rank) SH_NCCL(w, g_rccl.Send(t.sp, t.bytes, ncclInt8, t.dst, w.c_urgent[(size_t)stage], w.sstream[(size_t)stage]));
		else if (t.dst == w.rank) SH_NCCL(w, g_rccl.Recv(t.dp, t.bytes, ncclInt8, t.src, w.c_urgent[(size_t)stage], w.sstream[(size_t)stage]));
	}
	SH_NCCL(w, g_rccl.GroupEnd());
	return SIFT3D_OK;
}

int lay_out(Worker &w, int s, const std::vector<int> &counts, const std::vector<int> &nb, size_t rb, PartLayout &L, bool second = false) {
	const hipStream_t sst = w.sstream[(size_t)s];
	s = 2 * s + (second ? 1 : 0);  // (slot of the scratch)
	auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
	const size_t me = (size_t)counts[(size_t)w.rank];
	size_t need = al(me * rb) + 3 * al(me * 4) + al(me * 768 * 4) + al(me * 4);
	for (int r : nb) {
		const size_t n = (size_t)counts[(size_t)r];
		need += al(n * rb) + al(n * 4) + al(n * 768 * 4) + al(n * 4);  // r's records + units here, this rank's part of them
		need += al(me * 768 * 4) + al(me * 4);                        // r's part of this rank's records
	}
	need = std::max<size_t>(need, 256);
	if (w.pscratch.size() <= (size_t)s) { w.pscratch.resize((size_t)s + 1, nullptr); w.pscratch_bytes.resize((size_t)s + 1, 0); }
	if (need > w.pscratch_bytes[(size_t)s]) {
		// (a stage's scratch is only in use between the enqueue of its windows and the end of the run; the run before has drained)
		SH_HIP(w, hipStreamSynchronize(sst));
		if (w.pscratch[(size_t)s]) SH_HIP(w, hipFree(w.pscratch[(size_t)s]));
		w.pscratch[(size_t)s] = nullptr; w.pscratch_bytes[(size_t)s] = 0;
		const size_t cap = need + need / 4;
		SH_HIP(w, hipMalloc(reinterpret_cast<void **>(&w.pscratch[(size_t)s]), cap));
		w.pscratch_bytes[(size_t)s] = cap;
	}
	char *p = w.pscratch[(size_t)s];
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
// first round: the finish is only enqueued (sift3d_slab_describe_finish_launch; its count of flagged records is read when the whole step has been
// enqueued); second round: the blocking finish of the flagged subset.
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
			if (H->solo && !is_local(q)) continue;  // (solo: only what this rank receives, from the lists the last full run left in its neighbours' scratch)
			const PartLayout *Lr = is_local(r) ? &L[local[r]] : (H->solo ? &H->ps_last[(size_t)s].L[(size_t)r] : nullptr);
			PartLayout *Lq = is_local(q) ? &L[local[q]] : nullptr;
			ts.push_back(RawXfer{r, q, Lr ? Lr->recs : nullptr, Lq ? Lq->recs_in[r] : nullptr, n * rb});
			if (second) ts.push_back(RawXfer{r, q, Lr ? Lr->units : nullptr, Lq ? Lq->units_in[r] : nullptr, n * 4});
		}
	int rc = exchange_raw(H, ws, ts, s);
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
			if (H->solo && !is_local(r)) continue;
			PartLayout *Lq = is_local(q) ? &L[local[q]] : (H->solo ? &H->ps_last[(size_t)s].L[(size_t)q] : nullptr), *Lr = is_local(r) ? &L[local[r]] : nullptr;
			ts.push_back(RawXfer{q, r, Lq ? Lq->part_h[r] : nullptr, Lr ? Lr->got_h[q] : nullptr, n * 768 * 4});
			ts.push_back(RawXfer{q, r, Lq ? Lq->part_m[r] : nullptr, Lr ? Lr->got_m[q] : nullptr, n * 4});
		}
	rc = exchange_raw(H, ws, ts, s);
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
		if (second)
			SH_ABI(w, sift3d_slab_describe_finish(w.stages[(size_t)s].ctx, n ? L[i].recs : nullptr, n, (int)hh.size(), hh.data(), mm.data(), n ? L[i].units : nullptr,
			                                      1, n ? L[i].redo : nullptr, n ? L[i].units_next : nullptr, &nr));
		else
			SH_ABI(w, sift3d_slab_describe_finish_launch(w.stages[(size_t)s].ctx, n ? L[i].recs : nullptr, n, (int)hh.size(), hh.data(), mm.data(), n ? L[i].redo : nullptr,
			                                             n ? L[i].units_next : nullptr));
		n_redo[i] = nr;
	}
	return SIFT3D_OK;
}

// the descriptors of sharded octave s from partial integer histograms, first round, ENQUEUED (the keypoint counts of every rank are known:
// H->kp_count[s]); the state the rare second round needs stays in PS
int partial_stage_enqueue(sift3d_sharded *H, std::vector<Worker *> &ws, int s, PartStage &PS) {
	Worker &w0 = *ws[0];
	int rbi = 0, reach = 0;
	SH_ABI(w0, sift3d_slab_record_bytes(&rbi));
	PS.rb = (size_t)rbi;
	PS.counts = H->kp_count[(size_t)s];
	SH_ABI(w0, sift3d_slab_desc_reach(w0.stages[(size_t)s].ctx, &reach));
	PS.neigh = window_neighbours(w0.stages[(size_t)s].bounds, reach);
	PS.L.assign(ws.size(), PartLayout());
	std::vector<int> n_redo(ws.size(), 0);
	int rc;
	for (size_t i = 0; i < ws.size(); i++) {
		Worker &w = *ws[i];
		SH_HIP(w, hipSetDevice(w.device));
		if ((rc = lay_out(w, s, PS.counts, PS.neigh[(size_t)w.rank], PS.rb, PS.L[i])) != SIFT3D_OK) return rc;
		if (PS.counts[(size_t)w.rank]) SH_ABI(w, sift3d_slab_export_records(w.stages[(size_t)s].ctx, PS.L[i].recs));
	}
	return partial_round(H, ws, s, PS.neigh, PS.counts, PS.L, PS.rb, false, n_redo);
}

// rare: records whose first fixed-point unit failed (H->redo_count[s], known to every rank) are repeated, by every part, with the exact unit.
// The flagged subset is compacted through the host (a few records), then the scratch is laid out again for the second round's counts.
int partial_stage_second(sift3d_sharded *H, std::vector<Worker *> &ws, int s, PartStage &PS) {
	const std::vector<int> tot = H->redo_count[(size_t)s];
	const std::vector<int> &counts = PS.counts;
	const size_t rb = PS.rb;
	const std::vector<PartLayout> &L = PS.L;   // the first round's lists (read)
	std::vector<PartLayout> L2(ws.size());    // the second round's, in a scratch of their own
	int rc;
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
		SH_HIP(w, hipMemcpyAsync(recs.data(), L[i].recs, n * rb, hipMemcpyDeviceToHost, w.sstream[(size_t)s]));
		SH_HIP(w, hipMemcpyAsync(redo.data(), L[i].redo, n * 4, hipMemcpyDeviceToHost, w.sstream[(size_t)s]));
		SH_HIP(w, hipMemcpyAsync(un.data(), L[i].units_next, n * 4, hipMemcpyDeviceToHost, w.sstream[(size_t)s]));
		SH_HIP(w, hipStreamSynchronize(w.sstream[(size_t)s]));
		for (size_t k = 0; k < n; k++)
			if (redo[k]) { recs2[i].insert(recs2[i].end(), recs.begin() + (ptrdiff_t)(k * rb), recs.begin() + (ptrdiff_t)((k + 1) * rb)); units2[i].push_back(un[k]); }
		if ((int)units2[i].size() != tot[(size_t)w.rank]) { set_err(w, "flagged records and their count disagree"); return SIFT3D_ERR_STATE; }
	}
	std::vector<int> n_redo(ws.size(), 0);
	for (size_t i = 0; i < ws.size(); i++) {
		Worker &w = *ws[i];
		SH_HIP(w, hipSetDevice(w.device));
		SH_HIP(w, hipStreamSynchronize(w.sstream[(size_t)s]));  // (simulated ranks share the stream: every rank's first round has drained before a scratch moves)
		if ((rc = lay_out(w, s, tot, PS.neigh[(size_t)w.rank], rb, L2[i], true)) != SIFT3D_OK) return rc;
		if (tot[(size_t)w.rank]) {
			SH_HIP(w, hipMemcpyAsync(L2[i].recs, recs2[i].data(), recs2[i].size(), hipMemcpyHostToDevice, w.sstream[(size_t)s]));
			SH_HIP(w, hipMemcpyAsync(L2[i].units, units2[i].data(), units2[i].size() * 4, hipMemcpyHostToDevice, w.sstream[(size_t)s]));
			SH_HIP(w, hipStreamSynchronize(w.sstream[(size_t)s]));  // (the host vectors are pageable and go out of scope)
		}
	}
	return partial_round(H, ws, s, PS.neigh, tot, L2, rb, true, n_redo);
}

// CSIFT3D::KpSiftAlgorithm (Src/cSIFT3D.cc:165-235) over the slabs of the local workers.  Host synchronisations of a step: the keypoint
// counts of the sharded octaves (read once everything up to the orientation of the LAST sharded octave has been enqueued), the counts of
// flagged records (read once every window has been enqueued), the tail's wait and the final drain.
int run_local(sift3d_sharded *H, std::vector<Worker *> &ws) {
	Worker &w0 = *ws[0];
	const int ng = H->ng;
	const bool has_tail = H->noct > H->S;
	for (Worker *w : ws) SH_HIP(*w, hipSetDevice(w->device));
	struct TailJoin { std::vector<Worker *> &ws; ~TailJoin() { (void)join_tail(ws); } } tail_join{ws};  // (whatever way this function is left)
	for (int s = 0; s < H->S; s++) {
		const Stage &st0 = w0.stages[(size_t)s];
		const Bounds &bounds = st0.bounds;
		const int nzs = st0.nz;
		if (s > 0)
			for (Worker *w : ws) {  // this octave's stream starts behind the decimation that wrote its level 0 (on the stream of the octave above)
				SH_HIP(*w, hipSetDevice(w->device));
				SH_HIP(*w, hipStreamWaitEvent(w->sstream[(size_t)s], w->ev_next[(size_t)s - 1], 0));
			}
		for (int i = 0; i < ng; i++) {
			for (Worker *w : ws) { SH_HIP(*w, hipSetDevice(w->device)); SH_ABI(*w, sift3d_slab_level(w->stages[(size_t)s].ctx, i)); }
			const int urgent_h = i + 1 < ng ? H->hws[(size_t)i + 1] + 1 : 0;  // planes p-hw-1 .. p+hw of the next level's z-march
			const bool ghost = H->ghost0 && s == 0;  // (ghost zones: this octave's slabs computed what they would receive)
			// urgent: ordered behind the level kernel on the rank's stream, in front of the next level
			int rc = ghost ? SIFT3D_OK : exchange(H, ws, halo_transfers(bounds, nzs, KIND_GSS, i, 0, urgent_h, s), 0);
			if (rc) return rc;
			// deferred: the wider keypoint-window halo of G[1..levels] and the DoG plane behind it, on the deferred flow
			std::vector<Transfer> late;
			if (!ghost) late = halo_transfers(bounds, nzs, KIND_GSS, i, urgent_h, H->need[(size_t)s][(size_t)i], s);
			if (!ghost && i - 1 >= 1 && i - 1 <= H->levels) {
				std::vector<Transfer> dg = halo_transfers(bounds, nzs, KIND_DOG, i - 1, 0, 1, s);
				late.insert(late.end(), dg.begin(), dg.end());
			}
			if (!late.empty()) {
				if (!H->sim)
					for (Worker *w : ws) {  // the deferred stream picks up behind the level kernel
						SH_HIP(*w, hipEventRecord(w->ev_level[(size_t)s], w->sstream[(size_t)s]));
						SH_HIP(*w, hipStreamWaitEvent(w->sdstream[(size_t)s], w->ev_level[(size_t)s], 0));
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
				if (s + 1 < H->S)
					for (Worker *w : ws) { SH_HIP(*w, hipSetDevice(w->device)); SH_HIP(*w, hipEventRecord(w->ev_next[(size_t)s], w->sstream[(size_t)s])); }
				// the octaves behind the sharded ones, once, on the tail rank: the seed level gathered, the whole pipeline enqueued by a thread of its own
				if (s + 1 == H->S && has_tail) {
					if ((rc = gather_seed(H, ws)) != SIFT3D_OK) return rc;
					if ((rc = start_tail(H, ws)) != SIFT3D_OK) return rc;
				}
			}
		}
		// DoG maxima -> global (threshold of Detect_KeyPoints, Src/cSIFT3D.cc:379-384)
		if (!H->solo) for (Worker *w : ws) { SH_HIP(*w, hipSetDevice(w->device)); SH_ABI(*w, sift3d_slab_export_dogmax_device(w->stages[(size_t)s].ctx, w->dogmax[(size_t)s])); }
		int rc = allreduce_max_dev(H, ws, s, 8);
		if (rc) return rc;
		for (Worker *w : ws) { SH_HIP(*w, hipSetDevice(w->device)); SH_ABI(*w, sift3d_slab_import_dogmax_device(w->stages[(size_t)s].ctx, w->dogmax[(size_t)s])); }
		// extrema + orientation of this octave right behind its pyramid, BEFORE the next octave's levels are enqueued: on the GPU they then run
		// beside the next octaves' small level launches.  (Enqueued behind every octave's pyramid -- the order until late r06 -- octave 0's masks
		// started when the LAST octave's levels had been placed, 0.25 ms after its own pyramid had ended: profiles/r06x_solo3_all.txt.)  The counts
		// are read further down, when every octave's launches are in the queues.
		for (Worker *w : ws) {
			SH_HIP(*w, hipSetDevice(w->device));
			if (!H->sim) {  // the octave's deferred halos are complete before its detection reads them
				SH_HIP(*w, hipEventRecord(w->ev_def[(size_t)s], w->sdstream[(size_t)s]));
				SH_HIP(*w, hipStreamWaitEvent(w->sstream[(size_t)s], w->ev_def[(size_t)s], 0));
			}
			SH_ABI(*w, sift3d_slab_keypoints_launch(w->stages[(size_t)s].ctx));
		}
	}
	// SIFT3D_HOOK_SHARDED_FAIL_RANK (tests): this rank gives up here, with its pyramid enqueued and its peers on their way to the rendezvous
	for (Worker *w : ws)
		if (hook(SIFT3D_HOOK_SHARDED_FAIL_RANK) == w->rank + 1) { set_err(*w, "injected failure (SIFT3D_HOOK_SHARDED_FAIL_RANK)"); return SIFT3D_ERR_STATE; }
	int rc = SIFT3D_OK;
	auto say = [&](Worker &w, const char *what, int r) { set_err(w, std::string(what) + ": " + sift3d_error_string(r) + " (" + sift3d_last_error() + ")"); };
	{
		// the counts of every sharded octave's extrema + orientation launches (enqueued above, octave by octave: the GPU is busy with the later
		// octaves while the host waits for the first), shared with the other ranks' threads behind one rendezvous
		for (int s = 0; s < H->S && rc == SIFT3D_OK; s++)
			for (Worker *w : ws) {
				int n = 0;
				if (hipSetDevice(w->device) != hipSuccess) { rc = SIFT3D_ERR_HIP; break; }
				if ((rc = sift3d_slab_keypoints_count(w->stages[(size_t)s].ctx, &n)) != SIFT3D_OK) { say(*w, "sharded keypoint counts", rc); break; }
				H->kp_count[(size_t)s][(size_t)w->rank] = n;
			}
		if (rc == SIFT3D_OK) rc = rendezvous(H, w0);
		// the descriptors: an octave whose windows are split along z exchanges records and partial histograms; an octave of slabs too thin for
		// that (its level buffers carry the whole windows' reach) describes its own keypoints from its own buffers
		std::vector<PartStage> PS((size_t)H->S);
		for (int s = 0; s < H->S && rc == SIFT3D_OK; s++) {
			if (H->stage_partial[(size_t)s]) { rc = partial_stage_enqueue(H, ws, s, PS[(size_t)s]); continue; }
			for (Worker *w : ws) {
				if (hipSetDevice(w->device) != hipSuccess) { rc = SIFT3D_ERR_HIP; break; }
				if ((rc = sift3d_slab_describe_launch(w->stages[(size_t)s].ctx)) != SIFT3D_OK) { say(*w, "sharded descriptors (whole windows)", rc); break; }
			}
		}
		for (int s = 0; s < H->S && rc == SIFT3D_OK; s++)
			for (Worker *w : ws) {
				int nr = 0;
				if (!H->stage_partial[(size_t)s]) { H->redo_count[(size_t)s][(size_t)w->rank] = 0; continue; }
				if (hipSetDevice(w->device) != hipSuccess) { rc = SIFT3D_ERR_HIP; break; }
				if ((rc = sift3d_slab_describe_finish_count(w->stages[(size_t)s].ctx, &nr)) != SIFT3D_OK) { say(*w, "sharded descriptors", rc); break; }
				H->redo_count[(size_t)s][(size_t)w->rank] = nr;
			}
		if (rc == SIFT3D_OK) rc = rendezvous(H, w0);
		for (int s = 0; s < H->S && rc == SIFT3D_OK; s++) {
			const std::vector<int> &tot = H->redo_count[(size_t)s];
			if (!std::any_of(tot.begin(), tot.end(), [](int v) { return v > 0; })) continue;
			if (H->solo) {
				// a solo re-run times the first round only: the flagged rows keep what the full run's second round stored, and an empty
				// final round marks the rank's results complete again
				for (Worker *w : ws) {
					int nr = 0;
					if (hipSetDevice(w->device) != hipSuccess) { rc = SIFT3D_ERR_HIP; break; }
					if ((rc = sift3d_slab_describe_finish(w->stages[(size_t)s].ctx, nullptr, 0, 0, nullptr, nullptr, nullptr, 1, nullptr, nullptr, &nr)) != SIFT3D_OK) { say(*w, "solo finish", rc); break; }
				}
				continue;
			}
			rc = partial_stage_second(H, ws, s, PS[(size_t)s]);
		}
		if (H->sim && !H->solo && rc == SIFT3D_OK) H->ps_last = PS;  // (simulated ranks: what a solo re-run of one rank reads its neighbours' lists from)
	}
	if (rc != SIFT3D_OK) abort_all(H);  // (peers may sit in a receive waiting for this rank)
	// the tail's run is completed whatever happened above (an extractor with a run in flight must not be destroyed under it)
	{
		const int jrc = join_tail(ws);
		if (jrc != SIFT3D_OK && rc == SIFT3D_OK) { rc = jrc; abort_all(H); }
	}
	for (Worker *w : ws)
		if (w->tail && has_tail && w->rank == H->tail_rank && w->tail_rc == SIFT3D_OK) {
			(void)hipSetDevice(w->device);
			const int trc = sift3d_wait(w->tail);
			if (trc != SIFT3D_OK && rc == SIFT3D_OK) { say(*w, "tail", trc); rc = trc; abort_all(H); }
		}
	for (Worker *w : ws) {
		(void)hipSetDevice(w->device);
		for (hipStream_t sst : w->sstream) {
			const hipError_t e = hipStreamSynchronize(sst);
			if (e != hipSuccess && rc == SIFT3D_OK) { set_err(*w, std::string("hipStreamSynchronize: ") + hipGetErrorString(e)); rc = SIFT3D_ERR_HIP; }
		}
	}
	if (rc == SIFT3D_OK && H->failed.load()) { set_err(w0, "aborted: another rank failed"); rc = SIFT3D_ERR_STATE; }  // halos of an aborted exchange are garbage
	return rc;
}

// phase 0: everything that may still enqueue on or wait for a stream; phase 1: the streams (simulated ranks SHARE rank 0's stream: it
// must outlive the contexts of every rank -- destroying it with rank 0 made the other ranks synchronise a dead stream, which hung
// about one run in ten)
void destroy_worker(Worker &w, int phase, bool comms_aborted) {
	(void)hipSetDevice(w.device);
	if (phase == 0) {
		for (hipStream_t st : w.sstream) if (st) (void)hipStreamSynchronize(st);
		for (hipStream_t st : w.sdstream) if (st) (void)hipStreamSynchronize(st);
		if (w.tail_thread.joinable()) w.tail_thread.join();
		if (w.tstream) (void)hipStreamSynchronize(w.tstream);
		if (w.tail) { (void)sift3d_set_stream(w.tail, nullptr); sift3d_destroy(w.tail); }
		w.tail = nullptr; w.seed_dst = nullptr;
		for (Stage &s : w.stages) {
			if (s.ctx) { (void)sift3d_set_stream(s.ctx, nullptr); sift3d_destroy(s.ctx); s.ctx = nullptr; }
			if (s.arena) (void)hipFree(s.arena);
			s.arena = nullptr;
		}
		for (float *&d : w.dogmax) { if (d) (void)hipFree(d); d = nullptr; }
		for (float *&d : w.dogmax_in) { if (d) (void)hipFree(d); d = nullptr; }
		for (hipEvent_t e : w.evp) if (e) (void)hipEventDestroy(e);
		w.evp.clear(); w.evp_i = 0;
		for (char *&ps : w.pscratch) { if (ps) (void)hipFree(ps); ps = nullptr; }
		w.pscratch.clear(); w.pscratch_bytes.clear();
		if (w.seed_mine && w.seed_mine_owned) (void)hipFree(w.seed_mine);
		w.seed_mine = nullptr; w.seed_mine_owned = false;
		for (auto *v : {&w.ev_level, &w.ev_def, &w.ev_next}) { for (hipEvent_t e : *v) if (e) (void)hipEventDestroy(e); v->clear(); }
		if (w.ev_seed) (void)hipEventDestroy(w.ev_seed);
		w.ev_seed = nullptr;
		if (!comms_aborted) {
			for (ncclComm_t c : w.c_urgent) if (c) (void)g_rccl.CommDestroy(c);
			for (ncclComm_t c : w.c_deferred) if (c) (void)g_rccl.CommDestroy(c);
			if (w.c_tail) (void)g_rccl.CommDestroy(w.c_tail);
		}
		w.c_urgent.clear(); w.c_deferred.clear(); w.c_tail = nullptr;
	} else {
		for (hipStream_t st : w.sdstream) if (st) (void)hipStreamDestroy(st);
		if (w.tstream) (void)hipStreamDestroy(w.tstream);
		if (w.own_stream) for (hipStream_t st : w.sstream) if (st) (void)hipStreamDestroy(st);
		w.sstream.clear(); w.sdstream.clear(); w.tstream = w.stream = nullptr;
	}
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
	H->copies = !H->sim && (flags & SIFT3D_SHARDED_COPY_TRANSPORT) != 0;
	if (H->copies) {
		if (H->world > kMaxMergePtrs) return fail(SIFT3D_ERR_ARG, "the copy transport takes at most " + std::to_string(kMaxMergePtrs) + " ranks");
		H->mail.reset(new sift3d_sharded::Mailbox[(size_t)H->world * (size_t)H->world]);
		// peer access between the devices of ranks that exchange (every pair: the DoG maxima travel between all of them), so that the peer copies go
		// over xGMI directly; a refusal (already enabled, or no peer path: the copies are then staged by the runtime) is not an error
		for (int i = 0; i < ndev; i++)
			for (int j = 0; j < ndev; j++)
				if (devices[i] != devices[j] && hipSetDevice(devices[i]) == hipSuccess) { (void)hipDeviceEnablePeerAccess(devices[j], 0); (void)hipGetLastError(); }
	} else if (!H->sim) {
		for (int i = 0; i < ndev; i++)
			for (int j = 0; j < i; j++)
				if (devices[i] == devices[j]) return fail(SIFT3D_ERR_ARG, "RCCL takes one rank per device (several ranks on one device: SIFT3D_SHARDED_COPY_TRANSPORT, or simulated ranks)");
	}
	H->devices.assign(devices, devices + ndev);
	H->levels = H->p.num_kp_levels; H->ng = H->levels + 3;
	int halo_whole = 0, halo_partial = 0;
	if (sift3d_slab_min_halo(&H->p, &halo_whole) != SIFT3D_OK || sift3d_slab_min_halo_partial(&H->p, &halo_partial) != SIFT3D_OK) return fail(SIFT3D_ERR_ARG, "bad parameters");
	H->noct = octaves_total(nx, ny, nz);
	if (H->noct < 1) return fail(SIFT3D_ERR_ARG, "volume too small for one octave");
	// sharded octaves: as asked, but none whose planes are smaller than the level kernel's tile (+ widest half width) or thinner than the ranks.
	// Not asked (0): every octave that is worth it -- at least 2^22 voxels and 16 planes per rank (the 256 x 256 x 128 octave 2 of a
	// 1024 x 1024 x 512 volume holds 45 % of that volume's keypoints: r05 ran it replicated on every rank, early r06 once on one rank, which then
	// took 6.5 ms of a 3.5 ms step) -- and never fewer than two where two fit.
	auto fits = [](int n) { return n == 32 || n >= 40; };  // one 32 x 32 tile, or room for a shifted last tile behind the widest mirror zone
	int S;
	if (sharded_octaves > 0) S = std::max(1, std::min(sharded_octaves, H->noct));
	else {
		S = std::max(1, std::min(2, H->noct));
		while (S < H->noct && ((size_t)(nx >> S) * (size_t)(ny >> S) * (size_t)(nz >> S)) >= ((size_t)1 << 22) && ((nz >> S) / H->world) >= 16) S++;
	}
	while (S > 1 && (!fits(nx >> (S - 1)) || !fits(ny >> (S - 1)) || (nz >> S) < H->world)) S--;
	// ... and none the slab contexts cannot hold (a level thinner than its kernel's column: no separable fallback for slabs)
	for (;;) {
		int ok = 0;
		if (sift3d_slab_admits(&H->p, nx >> (S - 1), ny >> (S - 1), nz >> (S - 1), S == 1, &ok) != SIFT3D_OK) return fail(SIFT3D_ERR_ARG, "bad parameters");
		if (ok) break;
		if (S == 1) return fail(SIFT3D_ERR_ARG, "this volume / these parameters do not fit the slab kernels (half widths 2 .. 8, planes of 32 or >= 32 + hw voxels per side, at least 2 hw + 2 planes)");
		S--;
	}
	H->S = S;
	// the tail (octaves >= S) runs once, on the last rank, which owns fewer planes in exchange: the tail is a volume of nz / 2^S planes of
	// 1 / 4^S the size, i.e. nz / 8^S planes of the first octave (x 8/7 for its own octaves), and small volumes cost ~2.3x as much per voxel as
	// the big levels (256^3: 1.7 ms against 24 ms for 32 times the voxels)
	H->tail_rank = H->noct > S ? H->world - 1 : -1;
	const int tail_planes = H->tail_rank >= 0 ? (int)lround(2.6 * (double)nz / (double)(1 << (3 * S))) : 0;
	Bounds b;
	static const int balance = dev_tune_i("S3D_BALANCE", 1);  // 0: the aligned deal of r05 / early r06 (multiples of 2^S planes, even shares)
	if (balance && H->world > 1) {
		// weights in planes of octave 0 (see slab_bounds_weighted): a side = 0.7 x the orientation windows' halo (13 planes by default: 9 planes); the tail = its
		// latency-bound pipeline beside the rank's own launches (0.7 ms) + its voxels at the big levels' rate, over the time of one plane (41.6 ps per voxel)
		const int halo0 = halo_partial;  // (whole windows on the wide halos: the same 0.4 ms per side, measured -- with 0.7 x 38 planes the first rank took 4.7 ms of a 3.8 ms step)
		const double plane_s = 41.6e-12 * (double)nx * (double)ny;
		const double tail_vox = (double)(nx >> S) * (double)(ny >> S) * (double)(nz >> S) * 8.0 / 7.0;
		const double tail_w = H->tail_rank >= 0 ? (0.7e-3 + 51e-12 * tail_vox) / plane_s : 0.0;
		const int min_planes = std::max(1 << S, nz / H->world / 2);
		// (octave 0 on ghost zones: a side also costs the levels' work on its ghost planes -- on average 0.62 x the ghost halo, at about half the pyramid's 0.35 share
		// of a plane's time (the rest hides under the chain of the octaves below): 3.9 planes more with the defaults; measured at 8 ranks: 4.12 ms (inner) / 3.91 (first) with no extra weight, 4.03 / 4.19 with twice this one)
		double side_w = 0.7 * (double)halo0;
		if (flags & SIFT3D_SHARDED_GHOST_OCTAVE0) {
			int gh = 0;
			if (sift3d_slab_min_halo_ghost(&H->p, (flags & SIFT3D_SHARDED_WHOLE_WINDOWS) ? 0 : 1, &gh) == SIFT3D_OK) side_w += 0.11 * (double)gh;
		}
		if (!slab_bounds_weighted(nz, H->world, min_planes, side_w, tail_w, H->tail_rank, b)) return fail(SIFT3D_ERR_ARG, "too few planes for this many slabs");
	} else if (!slab_bounds(nz, H->world, 1 << S, b, tail_planes)) return fail(SIFT3D_ERR_ARG, "too few planes for this many slabs");
	// descriptor windows, per sharded octave: split along z over the ranks (partial integer histograms) unless the caller asks for whole windows
	// -- or the octave's slabs are so thin that a window would span more ranks than one finish launch adds parts (the owner's and five
	// z-neighbours'): that octave carries whole windows on the wide halos instead, unless partial windows were asked for by name (refused)
	{
		const int reach = halo_whole - 1;  // planes a window reaches beyond its keypoint, in voxels of ITS octave: the same in every octave (scale / unit)
		Bounds bo = b;
		int dz = nz;
		H->stage_partial.assign((size_t)S, 0);
		H->stage_halo.assign((size_t)S, halo_whole);
		for (int o = 0; o < S; o++) {
			bool part = (flags & SIFT3D_SHARDED_WHOLE_WINDOWS) == 0;
			static const int max_ranks = dev_tune_i("S3D_PARTIAL_MAX_RANKS", kDescSegs);
			for (const std::vector<int> &nb : window_neighbours(bo, reach))
				if (part && (int)nb.size() + 1 > std::min(max_ranks, kDescSegs)) {
					if (flags & SIFT3D_SHARDED_PARTIAL_WINDOWS)
						return fail(SIFT3D_ERR_ARG, "partial descriptor windows: a slab of octave " + std::to_string(o) + " is so thin that a window spans more than " +
						                                std::to_string(kDescSegs) + " ranks; use fewer sharded octaves or whole windows");
					part = false;
				}
			H->stage_partial[(size_t)o] = part ? 1 : 0;
			H->stage_halo[(size_t)o] = part ? halo_partial : halo_whole;
			bo = halve_bounds(bo, dz);
			dz /= 2;
		}
		H->partial = std::all_of(H->stage_partial.begin(), H->stage_partial.end(), [](char c) { return c != 0; });
	}
	H->ghost0 = (flags & SIFT3D_SHARDED_GHOST_OCTAVE0) != 0;
	if (H->ghost0) {
		int gh = 0;
		if (sift3d_slab_min_halo_ghost(&H->p, H->stage_partial[0] ? 1 : 0, &gh) != SIFT3D_OK) return fail(SIFT3D_ERR_ARG, "bad parameters");
		H->stage_halo[0] = std::max(H->stage_halo[0], gh);
	}
	H->halo = H->stage_halo[0];
	H->need.assign((size_t)S, std::vector<int>());
	if (!H->sim && !H->copies) {
		std::lock_guard<std::mutex> lk(g_rccl_mu);
		std::string e;
		if (!g_rccl.load(e)) return fail(SIFT3D_ERR_STATE, e);
	}
	H->workers.resize((size_t)H->world);
	std::vector<hipStream_t> shared;  // simulated ranks share ONE stream per sharded octave: their "sends" are copies ordered on it
	for (int r = 0; r < H->world; r++) {
		Worker &w = H->workers[(size_t)r];
		w.rank = r; w.device = H->sim ? devices[0] : devices[r];
#define CR_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(SIFT3D_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
#define CR_ABI(call) do { int r_ = (call); if (r_ != SIFT3D_OK) return fail(r_, std::string(#call) + ": " + sift3d_last_error()); } while (0)
		CR_HIP(hipSetDevice(w.device));
		if (H->sim && !shared.empty()) w.sstream = shared;
		else {
			w.sstream.assign((size_t)S, nullptr);
			for (int o = 0; o < S; o++) CR_HIP(hipStreamCreateWithFlags(&w.sstream[(size_t)o], hipStreamNonBlocking));
			w.own_stream = true;
			if (H->sim) shared = w.sstream;
		}
		w.stream = w.sstream[0];
		w.sdstream.assign((size_t)S, nullptr);
		w.ev_level.assign((size_t)S, nullptr); w.ev_def.assign((size_t)S, nullptr); w.ev_next.assign((size_t)S, nullptr);
		for (int o = 0; o < S; o++) {
			if (!H->sim) CR_HIP(hipStreamCreateWithFlags(&w.sdstream[(size_t)o], hipStreamNonBlocking));
			CR_HIP(hipEventCreateWithFlags(&w.ev_level[(size_t)o], hipEventDisableTiming));
			CR_HIP(hipEventCreateWithFlags(&w.ev_def[(size_t)o], hipEventDisableTiming));
			CR_HIP(hipEventCreateWithFlags(&w.ev_next[(size_t)o], hipEventDisableTiming));
		}
		CR_HIP(hipEventCreateWithFlags(&w.ev_seed, hipEventDisableTiming));
		Bounds bb = b;
		int dx = nx, dy = ny, dz = nz;
		for (int o = 0; o < S; o++) {
			w.stages.emplace_back();
			Stage &st = w.stages.back();
			st.octave = o; st.nx = dx; st.ny = dy; st.nz = dz; st.bounds = bb; st.z0 = bb[(size_t)r].first; st.z1 = bb[(size_t)r].second;
			st.plane = (size_t)dx * dy;
			sift3d_slab_desc d{dx, dy, dz, st.z0, st.z1, H->stage_halo[(size_t)o], H->noct, o};
			if (st.z1 > st.z0) {
				CR_ABI(sift3d_slab_arena_floats(&d, &H->p, &st.arena_floats));
				CR_HIP(hipMalloc(&st.arena, sizeof(float) * st.arena_floats));
				CR_ABI(sift3d_slab_create(&st.ctx, &d, &H->p, w.device, st.arena, st.arena_floats));
				CR_ABI(sift3d_set_stream(st.ctx, w.sstream[(size_t)o]));
				if (H->stage_partial[(size_t)o]) CR_ABI(sift3d_slab_set_desc_partial(st.ctx, 1));
				if (H->ghost0 && o == 0) CR_ABI(sift3d_slab_set_ghost(st.ctx, 1));  // (behind the window form: the ghost extents depend on it)
			} else {
				return fail(SIFT3D_ERR_ARG, "a rank would own no planes of a sharded octave");
			}
			float *dm = nullptr;
			CR_HIP(hipMalloc(&dm, sizeof(float) * 8));
			CR_HIP(hipMemset(dm, 0, sizeof(float) * 8));
			w.dogmax.push_back(dm);
			if (H->copies) {
				float *di = nullptr;
				CR_HIP(hipMalloc(&di, sizeof(float) * 8 * (size_t)H->world));
				CR_HIP(hipMemset(di, 0, sizeof(float) * 8 * (size_t)H->world));
				w.dogmax_in.push_back(di);
			}
			bb = halve_bounds(bb, dz);
			dx /= 2; dy /= 2; dz /= 2;
		}
		if (r == 0) {
			H->sx = dx; H->sy = dy; H->sz = dz;
			H->counts2.clear();
			for (auto &p : bb) H->counts2.push_back(p.second - p.first);
			for (int o = 0; o < S; o++) {
				sift3d_handle co = w.stages[(size_t)o].ctx;
				for (int i = 0; i < H->ng; i++) {
					int v = 0;
					CR_ABI(sift3d_slab_halo_planes(co, i, &v)); H->need[(size_t)o].push_back(v);
					if (o == 0) { CR_ABI(sift3d_slab_level_hw(co, i, &v)); H->hws.push_back(v); }
				}
			}
		}
		if (H->tail_rank >= 0) {
			const size_t pl2 = (size_t)dx * dy;
			if (r == H->tail_rank) {
				// the tail: an ordinary seeded extractor of the octaves >= S on a stream of its own; its level 0 is where the seed level is gathered
				CR_ABI(sift3d_create_seeded(&w.tail, dx, dy, dz, S, H->noct, &H->p, w.device));
				// ... created with a CU mask of ALL compute units: HIP gives such a stream a hardware queue of its own (the others are dealt onto four
				// shared ones, and launches that share a queue run one after the other -- the tail's pyramid sat behind octave 1's extrema in
				// theirs): the tail rank's step alone 3.95-3.97 -> 3.83-3.88 ms (S3D_TAIL_OWNQ=0 in a -DS3D_DEV_SWITCHES build; profiles/r06q_ownq.txt).
				// Measured and not kept: the tail's streams at the lowest / highest queue priority (4.8 / 5.3 ms: every wait across priorities
				// is slow, r06q_tail_prio.txt), more hardware queues for everybody (GPU_MAX_HW_QUEUES 5 .. 8: 4.4-4.6 ms, r06q_hwq2.txt), partial-
				// window workgroups that leave after 1 / 2 / 4 records instead of staying to the end (4.28 / 3.93 / 3.86 ms, r06q_perwg.txt)
				{
					static const int own_q = dev_tune_i("S3D_TAIL_OWNQ", 1);
					hipDeviceProp_t pr;
					if (own_q && hipGetDeviceProperties(&pr, w.device) == hipSuccess && pr.multiProcessorCount > 0) {
						std::vector<uint32_t> mask((size_t)(pr.multiProcessorCount + 31) / 32, 0u);
						for (int cu = 0; cu < pr.multiProcessorCount; cu++) mask[(size_t)cu / 32] |= 1u << (cu % 32);
						if (hipExtStreamCreateWithCUMask(&w.tstream, (uint32_t)mask.size(), mask.data()) != hipSuccess) { (void)hipGetLastError(); w.tstream = nullptr; }  // (a pooled stream then)
					}
					if (!w.tstream) CR_HIP(hipStreamCreateWithFlags(&w.tstream, hipStreamNonBlocking));
				}
				CR_ABI(sift3d_set_stream(w.tail, w.tstream));
				size_t nf = 0;
				CR_ABI(sift3d_seed_buffer(w.tail, &w.seed_dst, &nf));
				size_t off = 0;
				for (int q = 0; q < r; q++) off += pl2 * (size_t)H->counts2[(size_t)q];
				if (!w.seed_dst || off + pl2 * (size_t)H->counts2[(size_t)r] > nf) return fail(SIFT3D_ERR_STATE, "the tail's seed level is smaller than the ranks' pieces");
				w.seed_mine = w.seed_dst + off;  // the tail rank decimates its own planes in place
			} else {
				CR_HIP(hipMalloc(&w.seed_mine, sizeof(float) * pl2 * (size_t)std::max(H->counts2[(size_t)r], 1)));
				w.seed_mine_owned = true;
			}
		}
	}
	H->kp_count.assign((size_t)S, std::vector<int>((size_t)H->world, 0));
	H->redo_count = H->kp_count;
	if (!H->sim && !H->copies) {
		// communicators over the same devices: per sharded octave one for its urgent halos, reductions and window exchange and one for its
		// deferred halos; one for the gather of the tail's seed level
		std::vector<ncclComm_t> c((size_t)H->world);
		for (Worker &w : H->workers) { w.c_urgent.assign((size_t)S, nullptr); w.c_deferred.assign((size_t)S, nullptr); }
		for (int k = 0; k < 2 * S + 1; k++) {
			ncclResult_t r = g_rccl.CommInitAll(c.data(), H->world, H->devices.data());
			if (r != ncclSuccess) return fail(SIFT3D_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
			for (int q = 0; q < H->world; q++) {
				Worker &w = H->workers[(size_t)q];
				if (k < S) w.c_urgent[(size_t)k] = c[(size_t)q];
				else if (k < 2 * S) w.c_deferred[(size_t)k - (size_t)S] = c[(size_t)q];
				else w.c_tail = c[(size_t)q];
			}
		}
	}
	// ---- constructor work (Src/cSIFT3D.cc:146-163): copy the owned planes, max-abs normalise over the WHOLE volume, exchange the
	// input halo of the base blur
	const size_t pl = (size_t)nx * ny;
	float gmax = 0.f;
	std::vector<float> lmax((size_t)H->world, 0.f);
	// (ghost zones: a rank uploads the planes its input buffer holds on either side of its own -- the halo it would otherwise receive from its
	// z-neighbours, and more: the host has the whole volume)
	auto up0 = [&](const Stage &st) { return H->ghost0 ? std::max(0, st.z0 - H->stage_halo[0]) : st.z0; };
	auto up1 = [&](const Stage &st) { return H->ghost0 ? std::min(nz, st.z1 + H->stage_halo[0]) : st.z1; };
	if (H->sim) {
		for (Worker &w : H->workers) {
			CR_HIP(hipSetDevice(w.device));
			Stage &st = w.stages[0];
			CR_ABI(sift3d_slab_upload(st.ctx, volume + pl * (size_t)up0(st), up0(st), up1(st), 0));
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
				if (rc == SIFT3D_OK) rc = sift3d_slab_upload(st.ctx, volume + pl * (size_t)up0(st), up0(st), up1(st), 0);
				if (rc == SIFT3D_OK) rc = sift3d_slab_input_absmax(st.ctx, &lmax[(size_t)r]);
				if (rc != SIFT3D_OK) errs[(size_t)r] = sift3d_last_error();  // (the error text is thread-local)
				rcs[(size_t)r] = rc;
			});
		for (auto &t : th) t.join();
		for (int r = 0; r < H->world; r++) if (rcs[(size_t)r]) return fail(rcs[(size_t)r], "slab upload of rank " + std::to_string(r) + ": " + errs[(size_t)r]);
	}
	for (float v : lmax) gmax = std::max(gmax, v);  // (one process holds every rank: the MAX all-reduce is a host loop)
	for (Worker &w : H->workers) { CR_HIP(hipSetDevice(w.device)); CR_ABI(sift3d_slab_input_scale(w.stages[0].ctx, gmax)); }
	if (!H->ghost0) {
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
	H->ran = false;
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
				H->workers[(size_t)r].evp_i = 0;  // (copy transport: the last step's events are free again, every stream has drained)
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
	// like the single-GPU extractor's, the results stay on the devices until they are asked for (sift3d_sharded_get_keypoints)
	H->times[0] = H->times[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	H->ran = true;
	return SIFT3D_OK;
}

extern "C" int sift3d_sharded_num_keypoints(sift3d_sharded_handle H, int *n) {
	if (!H || !n) return SIFT3D_ERR_ARG;
	*n = 0;
	if (!H->ran) return SIFT3D_OK;
	int tot = 0;
	for (Worker &w : H->workers) {
		for (Stage &st : w.stages) { int k = 0; int rc = sift3d_num_keypoints(st.ctx, &k); if (rc) return rc; tot += k; }
		if (w.tail) { int k = 0; int rc = sift3d_num_keypoints(w.tail, &k); if (rc) return rc; tot += k; }
	}
	*n = tot;
	return SIFT3D_OK;
}

// GetKeypoints (Src/cSIFT3D.cc:1686-1688) in reference order (octave, level, z, y, x) (Src/cSIFT3D.cc:373-416).  A rank's list of a sharded
// octave is in that order for the planes it owns, and the ranks own ascending, disjoint plane ranges: the merged list of an octave is, level
// by level, the ranks' runs of that level one after the other -- no sort, and every run's descriptors travel from their GPU straight to
// their place in the caller's array (r05 copied every rank's results into vectors of the handle, sorted pointers and copied twice more).
// The tail's list, complete on the tail rank, follows the sharded octaves'.
extern "C" int sift3d_sharded_get_keypoints(sift3d_sharded_handle H, sift3d_keypoint *out, float *desc) {
	if (!H) return SIFT3D_ERR_ARG;
	if (!H->ran || (!out && !desc)) return SIFT3D_OK;
	const auto t0 = std::chrono::steady_clock::now();
	const int W = H->world, S = H->S;
	auto bad = [&](Worker &w, int rc, const char *what) { H->err = std::string(what) + ": " + sift3d_error_string(rc) + " (" + sift3d_last_error() + ")"; (void)w; set_last_error(H->err); return rc; };
	// 1. the records of every (rank, stage): small (168 bytes each)
	std::vector<std::vector<std::vector<sift3d_keypoint>>> recs((size_t)W, std::vector<std::vector<sift3d_keypoint>>((size_t)S));
	for (Worker &w : H->workers) {
		if (hipSetDevice(w.device) != hipSuccess) return SIFT3D_ERR_HIP;
		for (int s = 0; s < S; s++) {
			int n = 0, rc = sift3d_num_keypoints(w.stages[(size_t)s].ctx, &n);
			if (rc) return bad(w, rc, "sift3d_num_keypoints");
			recs[(size_t)w.rank][(size_t)s].resize((size_t)n);
			if (n && (rc = sift3d_get_keypoints(w.stages[(size_t)s].ctx, recs[(size_t)w.rank][(size_t)s].data(), nullptr)) != SIFT3D_OK) return bad(w, rc, "sift3d_get_keypoints");
		}
	}
	// 2. the runs: (stage, level, rank) -> [a, b) of that rank's list, and where the run starts in the merged list
	std::vector<std::vector<D2HSeg>> segs((size_t)W);
	size_t pos = 0;
	for (int s = 0; s < S; s++) {
		std::vector<size_t> at((size_t)W, 0);
		for (int lv = 0; lv < 16; lv++)
			for (int r = 0; r < W; r++) {
				const std::vector<sift3d_keypoint> &L = recs[(size_t)r][(size_t)s];
				size_t a = at[(size_t)r], b = a;
				while (b < L.size() && L[b].level == lv) b++;
				if (b == a) continue;
				at[(size_t)r] = b;
				if (out) memcpy(out + pos, L.data() + a, sizeof(sift3d_keypoint) * (b - a));
				if (desc) {
					const float *dd = nullptr;
					int n = 0;
					Worker &w = H->workers[(size_t)r];
					const int rc = sift3d_device_results(w.stages[(size_t)s].ctx, &dd, nullptr, &n);
					if (rc) return bad(w, rc, "sift3d_device_results");
					segs[(size_t)r].push_back(D2HSeg{desc + pos * kDesc, dd + a * kDesc, sizeof(float) * kDesc * (b - a)});
				}
				pos += b - a;
			}
		for (int r = 0; r < W; r++)
			if (at[(size_t)r] != recs[(size_t)r][(size_t)s].size()) { H->err = "a rank's keypoint list is not ordered by level"; set_last_error(H->err); return SIFT3D_ERR_STATE; }
	}
	// 3. the tail's results behind them
	if (H->tail_rank >= 0) {
		Worker &w = H->workers[(size_t)H->tail_rank];
		if (hipSetDevice(w.device) != hipSuccess) return SIFT3D_ERR_HIP;
		int n = 0, rc = sift3d_num_keypoints(w.tail, &n);
		if (rc) return bad(w, rc, "sift3d_num_keypoints (tail)");
		if (n && out && (rc = sift3d_get_keypoints(w.tail, out + pos, nullptr)) != SIFT3D_OK) return bad(w, rc, "sift3d_get_keypoints (tail)");
		if (n && desc) {
			const float *dd = nullptr;
			if ((rc = sift3d_device_results(w.tail, &dd, nullptr, nullptr)) != SIFT3D_OK) return bad(w, rc, "sift3d_device_results (tail)");
			segs[(size_t)w.rank].push_back(D2HSeg{desc + pos * kDesc, dd, sizeof(float) * kDesc * (size_t)n});
		}
		pos += (size_t)n;
	}
	// 4. the descriptors: one staged pipeline per GPU (its own pinned pool), the GPUs side by side.  Ranks that share a device (simulated ranks, rank
	// threads of the copy transport) share ONE pipeline: eight calls one after the other took 6.7 ms for 112 MB, one takes what the bytes take.
	if (desc) {
		std::map<int, std::vector<D2HSeg>> by_dev;
		std::map<int, hipStream_t> dev_stream;
		for (int r = 0; r < W; r++) {
			Worker &w = H->workers[(size_t)r];
			if (segs[(size_t)r].empty()) continue;
			std::vector<D2HSeg> &v = by_dev[w.device];
			v.insert(v.end(), segs[(size_t)r].begin(), segs[(size_t)r].end());
			if (!dev_stream.count(w.device)) dev_stream[w.device] = w.stream;
		}
		std::map<int, int> rcs;
		std::map<int, std::string> errs;
		for (auto &kv : by_dev) { rcs[kv.first] = SIFT3D_OK; errs[kv.first] = std::string(); }  // (no insertion from the threads below)
		auto pull = [&](int dev) {
			if (hipSetDevice(dev) != hipSuccess) { rcs[dev] = SIFT3D_ERR_HIP; return; }
			std::vector<D2HSeg> &v = by_dev[dev];
			rcs[dev] = staged_d2h_v(v.data(), (int)v.size(), dev, dev_stream[dev]);
			if (rcs[dev]) errs[dev] = sift3d_last_error();
		};
		if (by_dev.size() <= 1) { for (auto &kv : by_dev) pull(kv.first); }
		else {
			std::vector<std::thread> th;
			for (auto &kv : by_dev) th.emplace_back(pull, kv.first);
			for (auto &t : th) t.join();
		}
		for (auto &kv : rcs) if (kv.second) { H->err = "device " + std::to_string(kv.first) + ": " + errs[kv.first]; set_last_error(H->err); return kv.second; }
	}
	H->times[1] = H->times[0] + std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
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

extern "C" int sift3d_sharded_plan(sift3d_sharded_handle H, int *partial_windows, int *tail_rank, int *planes /* [world] or NULL */, int *stage_partial /* [sharded octaves] or NULL */) {
	if (!H) return SIFT3D_ERR_ARG;
	if (partial_windows) *partial_windows = H->partial ? 1 : 0;
	if (stage_partial) for (int o = 0; o < H->S; o++) stage_partial[o] = H->stage_partial[(size_t)o] ? 1 : 0;
	if (tail_rank) *tail_rank = H->tail_rank;
	if (planes && !H->workers.empty())
		for (int r = 0; r < H->world; r++) planes[r] = H->workers[(size_t)r].stages[0].z1 - H->workers[(size_t)r].stages[0].z0;
	return SIFT3D_OK;
}

// TEST / MEASUREMENT entry (include/sift3d_hip_test.h): what ONE rank of a simulated run does in a step, alone on the GPU.  After a full run
// (sift3d_sharded_run) every rank's level buffers, halos, record lists and partial histograms are still in place; the rank's whole step is
// enqueued again -- its levels, the copies of the planes / records / histograms it RECEIVES (from what the last run left in its neighbours'
// buffers: the same bytes), its extrema, orientation, its part of every window that reaches it, the finish of its own keypoints, the tail on
// the tail rank -- and timed from the first launch to the drain.  That is the GPU time of that rank on a node of `world` GPUs, short of what
// its transfers wait for.  Results are not touched (the rank recomputes what it held).  Partial windows only.
extern "C" int sift3d_test_sharded_time_rank(sift3d_sharded_handle H, int rank, double *seconds) {
	if (!H || !seconds || rank < 0 || rank >= H->world) return SIFT3D_ERR_ARG;
	if (!H->sim || !H->ran || H->ps_last.size() != (size_t)H->S) { set_last_error("needs a simulated extractor that has run"); return SIFT3D_ERR_STATE; }
	Worker &w = H->workers[(size_t)rank];
	if (hipSetDevice(w.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return SIFT3D_ERR_HIP;
	std::vector<Worker *> ws{&w};
	H->solo = true;
	const auto t0 = std::chrono::steady_clock::now();
	const int rc = run_local(H, ws);
	H->solo = false;
	*seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (rc) { H->err = w.err; set_last_error(H->err); }
	return rc;
}

extern "C" int sift3d_test_slab_plan(int nz, int world, int min_planes, double side_w, double tail_w, int tail_rank, int halvings, int *z0, int *z1) {
	if (nz <= 0 || world < 1 || halvings < 0 || !z0 || !z1) return SIFT3D_ERR_ARG;
	Bounds b;
	if (!slab_bounds_weighted(nz, world, min_planes, side_w, tail_w, tail_rank, b)) return SIFT3D_ERR_ARG;
	int dz = nz;
	for (int o = 0; o <= halvings; o++) {
		for (int r = 0; r < world; r++) { z0[o * world + r] = b[(size_t)r].first; z1[o * world + r] = b[(size_t)r].second; }
		b = halve_bounds(b, dz);
		dz /= 2;
	}
	return SIFT3D_OK;
}

// bytes every rank RECEIVES per step: the plane halos of the plan (what run_local posts per level and sharded octave) and, from the keypoint
// counts of the last run, the records and partial histograms of the octaves whose windows are split along z.  halo[r], window[r]: bytes.
extern "C" int sift3d_sharded_traffic(sift3d_sharded_handle H, double *halo /* [world] */, double *window /* [world] */) {
	if (!H || !halo || !window) return SIFT3D_ERR_ARG;
	for (int r = 0; r < H->world; r++) { halo[r] = 0.0; window[r] = 0.0; }
	const Worker &w0 = H->workers[0];
	int rbi = 0;
	(void)sift3d_slab_record_bytes(&rbi);
	for (int s = 0; s < H->S; s++) {
		const Stage &st = w0.stages[(size_t)s];
		std::vector<Transfer> all;

		for (int i = 0; i < H->ng; i++) {
			const int urgent_h = i + 1 < H->ng ? H->hws[(size_t)i + 1] + 1 : 0;
			for (const std::vector<Transfer> &ts : {halo_transfers(st.bounds, st.nz, KIND_GSS, i, 0, urgent_h, s), halo_transfers(st.bounds, st.nz, KIND_GSS, i, urgent_h, H->need[(size_t)s][(size_t)i], s),
			                                         (i - 1 >= 1 && i - 1 <= H->levels) ? halo_transfers(st.bounds, st.nz, KIND_DOG, i - 1, 0, 1, s) : std::vector<Transfer>()})
				all.insert(all.end(), ts.begin(), ts.end());
		}
		if (!(H->ghost0 && s == 0))  // (ghost zones: no plane of octave 0 travels)
			for (const Transfer &t : all) halo[t.dst] += (double)(t.zg1 - t.zg0) * (double)st.plane * 4.0;
		if (!H->ran || !H->stage_partial[(size_t)s]) continue;
		int reach = 0;
		if (sift3d_slab_desc_reach(st.ctx, &reach) != SIFT3D_OK) continue;
		const std::vector<std::vector<int>> neigh = window_neighbours(st.bounds, reach);
		const std::vector<int> &cnt = H->kp_count[(size_t)s];
		for (int r = 0; r < H->world; r++)
			for (int q : neigh[(size_t)r]) {
				window[r] += (double)cnt[(size_t)q] * (double)rbi;                 // q's records
				window[r] += (double)cnt[(size_t)r] * (768.0 * 4.0 + 4.0);         // q's part of r's windows
			}
	}
	return SIFT3D_OK;
}
