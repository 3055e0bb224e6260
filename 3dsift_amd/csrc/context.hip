// context.hip -- host orchestration + the C-ABI (include/sift3d_hip.h).
//
// One sift3d_ctx owns a device arena (both pyramids, scratch, keypoint lists), a HIP stream and
// the host-built constant tables.  sift3d_run enqueues the whole KpSiftAlgorithm pipeline
// (reference Src/cSIFT3D.cc:165-235) on that stream with no host synchronisation inside; the
// only sync is the final one that also brings the keypoint count back.
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>

#include "sift3d_internal.h"

#pragma clang fp contract(off)

namespace s3d {

static thread_local std::string g_last_error;
void set_last_error(const std::string &s) { g_last_error = s; }

// test hooks (include/sift3d_hip.h) and development switches (sift3d_internal.h)
static int g_hooks[SIFT3D_HOOK_COUNT] = {0};
int hook(int which) { return which >= 0 && which < SIFT3D_HOOK_COUNT ? g_hooks[which] : 0; }
#ifdef S3D_DEV_SWITCHES
// measurement builds only (scripts/build_variant.sh): the old environment switches
int dev_tune_i(const char *env_name, int dflt) { const char *e = getenv(env_name); return e ? atoi(e) : dflt; }
double dev_tune_d(const char *env_name, double dflt) { const char *e = getenv(env_name); return e ? atof(e) : dflt; }
static const bool g_env_hooks = [] {
	static const char *names[SIFT3D_HOOK_COUNT] = {"S3D_DOG_EAGER", "S3D_GLAST_EAGER", "S3D_DET_SERIAL", "S3D_SEPARABLE", "S3D_DESC_NOCACHE",
	                                               "S3D_MATCH_NODMA", "S3D_ONE_STREAM", "S3D_DESC_MASS_SHIFT", "S3D_LIST_CAP", "S3D_PEER_COPY", "S3D_DESC_NOSPLIT",
	                                               "S3D_MARCH_TILES", "S3D_DESC_EXACT_CELLS", "S3D_LAZY_GENERIC"};
	static_assert(sizeof(names) / sizeof(names[0]) == SIFT3D_HOOK_COUNT, "one environment name per hook");
	for (int i = 0; i < SIFT3D_HOOK_COUNT; i++) { const char *e = names[i] ? getenv(names[i]) : nullptr; if (e) g_hooks[i] = atoi(e); }
	return true;
}();
#else
int dev_tune_i(const char *, int dflt) { return dflt; }
double dev_tune_d(const char *, double dflt) { return dflt; }
#endif

// ---------------------------------------------------------------------------------------------
// host-side constant builders (each mirrors a reference routine; same fp32/fp64 mix)
// ---------------------------------------------------------------------------------------------
// GaussianSmooth_3D kernel generation, Src/cSIFT3D.cc:541-572
static bool build_taps(float sigma, Taps &t) {
	sigma = sigma > 0 ? sigma : 0;
	int hw = 1;
	if (sigma > 0) {
		hw = (int)ceil((double)sigma * 3.0);
		if (hw < 1) hw = 1;
	}
	if (hw > kMaxHW) return false;
	t.hw = hw;
	const int width = 2 * hw + 1;
	float acc = 0;
	for (int i = 0; i < width; i++) {
		float x = (float)(i - hw);
		x = (float)((double)x / ((double)sigma + DBL_EPSILON));
		t.w[i] = (float)exp(-0.5 * (double)x * (double)x);
		acc += t.w[i];
	}
	for (int i = 0; i < width; i++) t.w[i] /= acc;
	for (int i = width; i < kMaxTaps; i++) t.w[i] = 0.f;
	return true;
}

// incremental blur schedule, Src/cSIFT3D.cc:272-287, 299
static void level_sigmas(const sift3d_params &p, std::vector<float> &sig, float &base_sigma) {
	const int ng = p.num_kp_levels + 3;
	sig.assign(ng, 0.f);
	const float k = (float)pow(2.0, 1.0 / (double)p.num_kp_levels);
	const float base = (float)((double)p.sigma_default * pow(2.0, -1.0 / 3.0));
	sig[0] = base;
	for (int i = 1; i < ng; i++) {
		const float sig_prev = (float)(pow((double)k, (double)(i - 1)) * (double)base);
		const float sig_total = sig_prev * k;
		sig[i] = sqrtf(sig_total * sig_total - sig_prev * sig_prev);
	}
	base_sigma = sqrtf(sig[0] * sig[0] - p.sigma_n_default * p.sigma_n_default);
}

// icosahedron + hoisted cart2bary constants, Src/cUtil.cc:19-55, 113-175; Src/cSIFT3D.cc:1599-1619
static void cross3(const float *a, const float *b, float *o) {
	o[0] = a[1] * b[2] - a[2] * b[1];
	o[1] = a[2] * b[0] - a[0] * b[2];
	o[2] = a[0] * b[1] - a[1] * b[0];
}
static float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static void build_faces(FaceConst *F) {
	const double gr = 1.6180339887;
	const double vert[12][3] = {{0, 1, gr}, {0, -1, gr}, {0, 1, -gr}, {0, -1, -gr}, {1, gr, 0}, {-1, gr, 0},
	                            {1, -gr, 0}, {-1, -gr, 0}, {gr, 0, 1}, {-gr, 0, 1}, {gr, 0, -1}, {-gr, 0, -1}};
	static const int faces[kFaces][3] = {{0, 1, 8}, {0, 8, 4}, {0, 4, 5}, {0, 5, 9}, {0, 9, 1}, {1, 6, 8}, {8, 6, 10},
	                                     {8, 10, 4}, {4, 10, 2}, {4, 2, 5}, {5, 2, 11}, {5, 11, 9}, {9, 11, 7}, {9, 7, 1},
	                                     {1, 7, 6}, {3, 6, 7}, {3, 7, 11}, {3, 11, 2}, {3, 2, 10}, {3, 10, 6}};
	for (int f = 0; f < kFaces; f++) {
		float v[3][3];
		for (int j = 0; j < 3; j++) {
			F[f].idx[j] = faces[f][j];
			float raw[3] = {(float)vert[faces[f][j]][0], (float)vert[faces[f][j]][1], (float)vert[faces[f][j]][2]};
			const double mag = (double)sqrtf(dot3(raw, raw));
			const double sca = 1.0 / mag;
			for (int c = 0; c < 3; c++) v[j][c] = (float)((double)raw[c] * sca);
		}
		float a[3], b[3], n[3];
		for (int c = 0; c < 3; c++) { a[c] = v[2][c] - v[1][c]; b[c] = v[1][c] - v[0][c]; }
		cross3(a, b, n);
		if (dot3(n, v[0]) < 0)
			for (int c = 0; c < 3; c++) std::swap(v[0][c], v[1][c]);
		for (int c = 0; c < 3; c++) {
			F[f].e1[c] = v[1][c] - v[0][c];
			F[f].e2[c] = v[2][c] - v[0][c];
			F[f].t[c] = (float)((double)v[0][c] * (-1.0));
		}
		cross3(F[f].t, F[f].e1, F[f].q);
		F[f].qe2 = dot3(F[f].q, F[f].e2);
		for (int c = 0; c < 3; c++) F[f].centre[c] = (v[0][c] + v[1][c] + v[2][c]) / 3.0f;
	}
}

// symmetry table of the face lookup (see FaceSym): for every (type, sign bits) find the mesh face whose three vertices are the
// sign-flipped canonical ones, and which of its vertices plays which role
static bool build_facesym(const FaceConst *F, FaceSym *S) {
	const double gr = 1.6180339887, nrm = sqrt(1.0 + gr * gr);
	for (int type = 0; type < 4; type++)
		for (int bits = 0; bits < 8; bits++) {
			const double sx = (bits & 1) ? -1.0 : 1.0, sy = (bits & 2) ? -1.0 : 1.0, sz = (bits & 4) ? -1.0 : 1.0;
			const double A[3] = {0, sy, sz * gr}, B[3] = {sx, sy * gr, 0}, C[3] = {sx * gr, 0, sz};
			const double Am[3] = {0, -sy, sz * gr}, Bm[3] = {-sx, sy * gr, 0}, Cm[3] = {sx * gr, 0, -sz};  // mirrored across the straddled axis
			const double *role[3];
			switch (type) {
			case 0: role[0] = A; role[1] = B; role[2] = C; break;       // octant face
			case 1: role[0] = C; role[1] = Cm; role[2] = B; break;      // lambda_A < 0: across edge BC, the face that straddles z
			case 2: role[0] = A; role[1] = Am; role[2] = C; break;      // lambda_B < 0: across edge AC, straddles y
			default: role[0] = B; role[1] = Bm; role[2] = A; break;     // lambda_C < 0: across edge AB, straddles x
			}
			int found = -1, slot[3] = {-1, -1, -1};
			for (int f = 0; f < kFaces && found < 0; f++) {
				// geometric vertices of the face as the intersection test sees them (after the winding fix): v0 = -t, v1 = v0 + e1, v2 = v0 + e2
				double v[3][3];
				for (int c = 0; c < 3; c++) { v[0][c] = -(double)F[f].t[c]; v[1][c] = v[0][c] + (double)F[f].e1[c]; v[2][c] = v[0][c] + (double)F[f].e2[c]; }
				int sl[3] = {-1, -1, -1}, hit = 0;
				for (int r = 0; r < 3; r++)
					for (int j = 0; j < 3; j++) {
						double d = 0;
						for (int c = 0; c < 3; c++) d += fabs(v[j][c] - role[r][c] / nrm);
						if (d < 1e-4) { sl[r] = j; hit++; }
					}
				if (hit == 3 && sl[0] != sl[1] && sl[1] != sl[2] && sl[0] != sl[2]) { found = f; for (int r = 0; r < 3; r++) slot[r] = sl[r]; }
			}
			if (found < 0) return false;
			const int key = type * 8 + bits;
			S->face[key] = found;
			for (int r = 0; r < 3; r++) { S->slot[key][r] = slot[r]; S->vert[key][r] = F[found].idx[slot[r]]; }
		}
	return true;
}

}  // namespace s3d

using namespace s3d;

// ---------------------------------------------------------------------------------------------
// the context
// ---------------------------------------------------------------------------------------------
struct sift3d_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t own_stream = nullptr;  // the stream this context created (stream may be replaced by sift3d_set_stream)
	sift3d_params p{};
	int nx = 0, ny = 0, nz = 0;     // dims of the first octave this context holds (global)
	int noct = 0, ng = 0, nd = 0;
	int octave_base = 0;            // absolute index of that octave (seeded contexts of the multi-GPU path start at 1)
	bool seeded = false;            // level (0,0) is written by the caller (sift3d_seed_upload); no input volume, no base blur
	// z-slab mode (multi-GPU sharding of octave 0): this context owns global planes [own0, own1) and every level buffer
	// holds planes [own0-halo, own1+halo); halo planes are filled by the caller (neighbour exchange)
	bool slab = false;
	int own0 = 0, own1 = 0, halo = 0;
	// single-volume path: DoG[o][0] and DoG[o][nd-1] are not written by the pyramid (see DetectLevels); copy_level forms them
	bool dog_elide = false;
	bool g_last_elide = false;      // the last Gaussian level of every octave is not built (DetectLevels::lazy_src); implies dog_elide
	std::vector<char> g_last_built; // per octave: built on request (sift3d_copy_level)
	unsigned *d_prov = nullptr;     // parked candidates of the lazy level + [prov_cap] = their count
	bool desc_lut_lds = true;       // every descriptor window weight table fits the LDS copy (kMaxDescLut)
	bool ext_arena = false;         // level buffers live in memory owned by the caller
	int part_rank = 0, part_world = 1;  // descriptor work split of replicated octaves

	// device memory
	float *arena = nullptr;       // input + pyramids + scratch (one allocation)
	// r05: a HOST volume gets its own input buffer, allocated first, and a thread that stages the volume into it (staging.hip) while the
	// constructor allocates everything else (the 8 GB arena, the lists, the tables: 4 ms that used to come in front of the 10 ms copy)
	float *in_own = nullptr;
	hipStream_t up_stream = nullptr;
	std::thread uploader;
	int upload_rc = SIFT3D_OK;
	std::string upload_err;
	size_t arena_floats = 0;
	Level in;
	std::vector<Level> gss, dog;
	std::vector<float *> tmpA, tmpB;        // per-octave scratch of the generic separable passes (octaves may overlap)
	// octave o >= 1 only depends on G[o-1][num_kp_levels]: each octave chain runs on its own stream so the small
	// octaves fill the machine next to the tail of the big ones (ostream[0] == stream)
	std::vector<hipStream_t> ostream;
	// r04: the HEADS of the octaves >= 1 (levels up to the seed level) run on ONE stream, in order: each head waits for the seed level of
	// the octave above anyway, and a dependency across streams costs 13-16 us (event -> barrier packet on another queue) where
	// consecutive launches of one stream follow each other without a gap -- four hops of the 512^3 chain; the levels behind the seed
	// level stay on the octave's own stream (cstream == nullptr: every octave wholly on its own stream, as before)
	hipStream_t cstream = nullptr;
	std::vector<hipEvent_t> ev_seed, ev_done;
	hipEvent_t ev_fork = nullptr;
	unsigned *d_words = nullptr;  // [0] input max bits, [1..] per-DoG-level max bits, then counters
	unsigned *d_slots_part = nullptr;  // scratch of launch_slots (per-wave totals)
	unsigned *h_words = nullptr;  // pinned: the five counters a run reads back (a pageable destination makes the copy a staged, synchronous one)
	unsigned *d_inmax = nullptr, *d_dogmax = nullptr, *d_total = nullptr, *d_nkp = nullptr;
	DetectBufs det{};
	size_t det_blocks = 0;
	// octaves >= 1 own a slice of a second detection scratch: their masks are formed on a second stream beside octave 0's, only
	// the ordered compaction into the extrema list stays serial (S3D_DET_SERIAL=1: everything on one stream, one scratch)
	std::vector<DetectBufs> det_o;
	unsigned long long *d_masks2 = nullptr;
	unsigned *d_counts2 = nullptr, *d_offsets2 = nullptr, *d_prov2 = nullptr;
	hipEvent_t ev_det_fork = nullptr, ev_det_join = nullptr, ev_det_fork2 = nullptr, ev_det_join2 = nullptr;
	DevKp *d_ext = nullptr;
	int *d_codes = nullptr, *d_order = nullptr;  // d_order: slot -> extremum index
	unsigned ext_cap = 0, kp_cap = 0;
	LevelRef *d_levels = nullptr;
	WinLut *d_luts = nullptr;
	float *d_lutpool = nullptr;
	sift3d_keypoint *d_kpout = nullptr;
	float *d_desc = nullptr, *d_xyz = nullptr;
	DescSplit dsplit{};             // scratch of the split descriptor windows (runs with few keypoints), one allocation at dsplit.gacc
	bool desc_partial = false;      // slab contexts (r05): descriptor windows are split along z over the ranks -- the halo of G[1..levels] only has to carry the orientation windows
	bool dsplit_dirty = false;      // a run ended in an error: the "every run leaves the scratch clean" invariant is re-established by the next run
	float *d_peer = nullptr;        // sift3d_match_handles: a target's descriptors + coordinates copied from another GPU (grow-only)
	size_t peer_floats = 0;

	// host tables
	std::vector<Taps> taps;  // per GSS level index within an octave
	Taps base_taps{};

	bool use_fused = true;  // SIFT3D_HOOK_SEPARABLE forces the generic three-pass kernels (parity cross-check)
	int n_regrow = 0, n_desc_redo = 0;  // sift3d_debug_counters: list regrows / second descriptor passes of the last run

	// results / state
	int stage = 0;  // highest stage run
	bool pending = false;  // sift3d_run_async enqueued a run that sift3d_wait has not completed yet
	hipEvent_t gate = nullptr;  // sift3d_run_async_after: the next enqueue starts behind this event (another handle's orientation stage); one shot
	unsigned n_ext = 0, n_kp = 0;
	hipEvent_t ev[8] = {};
	double times[8] = {};
};

static int set_device(int device) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
		set_last_error("no HIP device visible: this library has no CPU fallback");
		return SIFT3D_ERR_NO_DEVICE;
	}
	if (device < 0 || device >= n) { set_last_error("device index out of range"); return SIFT3D_ERR_ARG; }
	S3D_HIP(hipSetDevice(device));
	return SIFT3D_OK;
}

static void free_lists(sift3d_ctx *c) {
	hipFree(c->d_ext); c->d_ext = nullptr;
	hipFree(c->d_codes); c->d_codes = nullptr;
	hipFree(c->d_order); c->d_order = nullptr;
	hipFree(c->d_kpout); c->d_kpout = nullptr;
	hipFree(c->d_desc); c->d_desc = nullptr;
	hipFree(c->d_xyz); c->d_xyz = nullptr;
	hipFree(c->d_prov); c->d_prov = nullptr;
	hipFree(c->d_prov2); c->d_prov2 = nullptr;
	hipFree(c->dsplit.gacc); c->dsplit = DescSplit{};
}

static int alloc_lists(sift3d_ctx *c, unsigned ext_cap) {
	free_lists(c);
	c->ext_cap = ext_cap;
	c->kp_cap = ext_cap;  // every extremum could survive orientation
	S3D_HIP(hipMalloc(&c->d_ext, sizeof(DevKp) * (size_t)c->ext_cap));
	S3D_HIP(hipMalloc(&c->d_codes, sizeof(int) * (size_t)c->ext_cap));
	S3D_HIP(hipMalloc(&c->d_order, sizeof(int) * (size_t)c->kp_cap));
	S3D_HIP(hipMalloc(&c->d_kpout, sizeof(sift3d_keypoint) * (size_t)c->kp_cap));
	S3D_HIP(hipMalloc(&c->d_desc, sizeof(float) * kDesc * (size_t)c->kp_cap));
	S3D_HIP(hipMalloc(&c->d_xyz, sizeof(float) * 3 * (size_t)c->kp_cap));
	{
		// split descriptor windows: [cap][768] int accumulators | [cap][8] masses | [cap] arrivals (zeroed once: every run leaves them clean)
		const unsigned scap = 4096;
		const size_t words = (size_t)scap * (kDesc + 8 + 1);
		int *base = nullptr;
		S3D_HIP(hipMalloc(&base, sizeof(int) * words));
		// on the handle's own stream: the legacy null stream would synchronise with every blocking stream of the host application
		hipError_t me = hipMemsetAsync(base, 0, sizeof(int) * words, c->stream);
		if (me == hipSuccess) me = hipStreamSynchronize(c->stream);
		if (me != hipSuccess) {
			hipFree(base);
			set_last_error(std::string("HIP error: ") + hipGetErrorString(me) + " (clearing the split-window scratch)");
			return SIFT3D_ERR_HIP;
		}
		c->dsplit.gacc = base;
		c->dsplit.gmass = reinterpret_cast<float *>(base + (size_t)scap * kDesc);
		c->dsplit.gdone = reinterpret_cast<unsigned *>(base + (size_t)scap * (kDesc + 8));
		c->dsplit.cap = scap;
	}
	S3D_HIP(hipMalloc(&c->d_prov, sizeof(unsigned) * ((size_t)c->ext_cap + 1)));
	c->det.prov = c->d_prov; c->det.prov_count = c->d_prov + c->ext_cap; c->det.prov_cap = c->ext_cap;
	if (c->det_o.size() > 1) {  // octaves >= 1: equal slices of a second parking list, each followed by its counter
		const size_t n2 = c->det_o.size() - 1;
		const unsigned slice = std::max(64u, (unsigned)(c->ext_cap / n2));
		S3D_HIP(hipMalloc(&c->d_prov2, sizeof(unsigned) * ((size_t)slice + 1) * n2));
		for (size_t o = 1; o < c->det_o.size(); o++) {
			c->det_o[o].prov = c->d_prov2 + (o - 1) * ((size_t)slice + 1);
			c->det_o[o].prov_count = c->det_o[o].prov + slice;
			c->det_o[o].prov_cap = slice;
		}
	}
	return SIFT3D_OK;
}

extern "C" int sift3d_wait(sift3d_handle c);

extern "C" int sift3d_test_hook(int which, int value) {
	if (which < 0 || which >= SIFT3D_HOOK_COUNT) return -1;
	const int prev = g_hooks[which];
	g_hooks[which] = value;
	return prev;
}

extern "C" void sift3d_default_params(sift3d_params *p) {
	if (!p) return;
	p->num_kp_levels = 3;
	p->sigma_default = 1.6f;
	p->sigma_n_default = 1.15f;
	p->peak_thresh = 0.1f;
	p->max_eig_thres = 0.9f;
	p->corner_thresh = 0.4f;
}

extern "C" int sift3d_device_count(int *n) {
	int k = 0;
	if (hipGetDeviceCount(&k) != hipSuccess) k = 0;
	if (n) *n = k;
	return SIFT3D_OK;
}

extern "C" const char *sift3d_error_string(int code) {
	switch (code) {
	case SIFT3D_OK: return "ok";
	case SIFT3D_ERR_ARG: return "bad argument";
	case SIFT3D_ERR_NO_DEVICE: return "no usable HIP device (no CPU fallback)";
	case SIFT3D_ERR_HIP: return "HIP runtime error";
	case SIFT3D_ERR_STATE: return "call out of order";
	case SIFT3D_ERR_CAPACITY: return "device list capacity exceeded";
	default: return "unknown error";
	}
}

extern "C" const char *sift3d_last_error(void) { return g_last_error.c_str(); }

extern "C" int sift3d_destroy(sift3d_handle c) {
	if (!c) return SIFT3D_OK;
	hipSetDevice(c->device);
	c->pending = false;  // (an asynchronous run in flight is drained below, its results dropped)
	if (c->uploader.joinable()) c->uploader.join();  // (a constructor that failed beside its upload)
	if (c->up_stream) { hipStreamSynchronize(c->up_stream); hipStreamDestroy(c->up_stream); c->up_stream = nullptr; }
	if (c->stream) hipStreamSynchronize(c->stream);
	if (c->own_stream && c->own_stream != c->stream) hipStreamSynchronize(c->own_stream);
	free_lists(c);
	if (!c->ext_arena) hipFree(c->arena);
	hipFree(c->in_own);
	hipFree(c->d_peer);
	hipFree(c->d_words);
	if (c->h_words) (void)hipHostFree(c->h_words);
	hipFree(c->d_slots_part);
	hipFree(c->det.masks); hipFree(c->det.block_counts); hipFree(c->det.block_offsets);
	hipFree(c->d_masks2); hipFree(c->d_counts2); hipFree(c->d_offsets2);
	if (c->ev_det_fork) hipEventDestroy(c->ev_det_fork);
	if (c->ev_det_join) hipEventDestroy(c->ev_det_join);
	if (c->ev_det_fork2) hipEventDestroy(c->ev_det_fork2);
	if (c->ev_det_join2) hipEventDestroy(c->ev_det_join2);
	hipFree(c->d_levels); hipFree(c->d_luts); hipFree(c->d_lutpool);
	for (auto &e : c->ev) if (e) hipEventDestroy(e);
	for (auto &e : c->ev_seed) if (e) hipEventDestroy(e);
	for (auto &e : c->ev_done) if (e) hipEventDestroy(e);
	if (c->ev_fork) hipEventDestroy(c->ev_fork);
	for (size_t o = 1; o < c->ostream.size(); o++) if (c->ostream[o] && c->ostream[o] != c->stream && c->ostream[o] != c->own_stream) hipStreamDestroy(c->ostream[o]);
	if (c->cstream) hipStreamDestroy(c->cstream);
	if (c->own_stream) hipStreamDestroy(c->own_stream);
	delete c;
	return SIFT3D_OK;
}

// Initialize + Initialize_Pyramid geometry, Src/cSIFT3D.cc:237-266, Src/cUtil.cc:177-235
static int octaves_of(int nx, int ny, int nz) {  // Src/cSIFT3D.cc:254-255
	const int mn = std::min(nx, std::min(ny, nz));
	return std::max(0, (int)log2f((float)mn) - 3 + 1);
}

static void plan_pyramid(sift3d_ctx *c, int noct_total) {
	c->noct = noct_total >= 0 ? std::max(0, noct_total - c->octave_base) : octaves_of(c->nx, c->ny, c->nz);
	if (c->slab) c->noct = std::min(c->noct, 1);
	c->ng = c->p.num_kp_levels + 3;
	c->nd = c->p.num_kp_levels + 2;
	c->gss.assign((size_t)c->noct * c->ng, Level());
	c->dog.assign((size_t)c->noct * c->nd, Level());
	const double sigma0 = (double)c->p.sigma_default * pow(2.0, -1.0 / 3.0);
	for (int pyr = 0; pyr < 2; pyr++) {
		const int interval = pyr ? c->nd : c->ng;
		std::vector<Level> &P = pyr ? c->dog : c->gss;
		int nx = c->nx, ny = c->ny, nz = c->nz;
		float u = (float)(1 << c->octave_base);  // units double per octave starting at 1 (Src/cUtil.cc:215-225)
		for (int o = 0; o < c->noct; o++) {
			for (int s = 0; s < interval; s++) {
				Level &L = P[(size_t)o * interval + s];
				L.nx = nx; L.ny = ny; L.nz = nz; L.unit = u;
				const double scale_factor = pow(2.0, (double)(o + c->octave_base) + (double)s / (double)c->p.num_kp_levels);
				L.scale = (float)(scale_factor * sigma0);
				if (c->slab) { L.bz = c->own1 - c->own0 + 2 * c->halo; L.zoff = c->own0 - c->halo; }
			}
			nx /= 2; ny /= 2; nz /= 2;
			u *= 2;
		}
	}
}

// Gaussian window tables (see WinLut): orientation (Src/cSIFT3D.cc:915, 968-971) and descriptor
// (Src/cSIFT3D.cc:1155-1156, 1270, 1312) windows of every (octave, keypoint level).
// one table: which = 0 the orientation window (sigma, radius = 3 sigma), 1 the descriptor window of a keypoint of scale `scale`
// (sigma = 7.0711 scale, radius = 2 sigma) on a level of unit u; appended to `pool`.  Returns false when a descriptor table is too long for the LDS.
static bool append_lut(std::vector<float> &pool, WinLut &L, int which, float sigma, float radius, float u, float scale) {
	bool fits_lds = true;
	const float r2 = radius * radius, uu = u * u;
	const int len = (int)floor((double)r2 / (double)uu) + 2;
	L.off = (int)pool.size(); L.len = len; L.nin = -1; L.radius = radius; L.sigma = sigma; L.fix_scale = 1.0f; L.list_off = -1; L.list_R = 0;
	if (which == 1) {
		// 32-bit histogram bins: a bin (cell, vertex) collects wgt * |g| * bary over the voxels within one cell of its centre;
		// the trilinear weights of those voxels sum to at most (cw + 2)^3 (cw = cell width in voxels = desc_width / (4u) =
		// 5 scale / u), |g| <= sqrt(3) (normalised data, |0.5 (a - b) / u| <= 1 per axis, weight <= 1), bary <= 1 + 2e-6
		const double cw = 5.0 * (double)scale / (double)u, bound = (cw + 2.0) * (cw + 2.0) * (cw + 2.0) * 1.7321 * 1.001;
		int k = (int)floor(log2(2147483647.0 / bound));
		k = std::max(0, std::min(k, 29));
		L.fix_scale = (float)ldexp(1.0, k);
	}
	if (which == 1 && len > kMaxDescLut) fits_lds = false;  // k_describe<false>: table read from global memory
	for (int n = 0; n < len; n++) {
		const float sq = (float)n * uu;  // exact: equals the reference's fp32 sum of squares
		float w;
		if (!(sq > r2)) L.nin = n;
		if (sq > r2) w = -1.0f;
		else if (which == 0) w = expf((float)(-0.5 * (double)sq / (double)(sigma * sigma)));
		else w = expf(-0.5f * sq / (sigma * sigma)) * (0.5f / u);  // exact scaling (u = 2^octave), see WinLut
		pool.push_back(w);
	}
	// sum of the weights over the lattice points of the sphere (k_describe's first guess of the gradient mass)
	const int R = (int)floor(sqrt((double)std::max(L.nin, 0)));
	double ws = 0.0;
	for (int dz = -R; dz <= R; dz++)
		for (int dy = -R; dy <= R; dy++)
			for (int dx = -R; dx <= R; dx++) {
				const int n = dx * dx + dy * dy + dz * dz;
				if (n <= L.nin) ws += (double)pool[(size_t)L.off + n] * (which == 1 ? (double)u / 0.5 : 1.0);
			}
	L.wsum = (float)std::max(ws, 1.0);
	if (which == 0 && R <= 127 && L.nin < 65536) {  // lattice points of the orientation sphere (WinLut::list_off)
		std::vector<unsigned> words((size_t)2 * R + 2);
		for (int dz = -R; dz <= R; dz++) {
			words[(size_t)(dz + R)] = (unsigned)(words.size() - ((size_t)2 * R + 2));
			for (int dy = -R; dy <= R; dy++)
				for (int dx = -R; dx <= R; dx++) {
					const int n = dx * dx + dy * dy + dz * dz;
					if (n <= L.nin) words.push_back((unsigned)(dx + 128) | (unsigned)(dy + 128) << 8 | (unsigned)n << 16);
				}
		}
		words[(size_t)2 * R + 1] = (unsigned)(words.size() - ((size_t)2 * R + 2));
		L.list_off = (int)pool.size(); L.list_R = R;
		pool.resize(pool.size() + words.size());
		memcpy(pool.data() + L.list_off, words.data(), words.size() * sizeof(unsigned));
	}
	return fits_lds;
}

static int upload_luts(sift3d_ctx *c, const std::vector<WinLut> &luts, std::vector<float> &pool) {
	if (pool.empty()) pool.push_back(-1.0f);
	if (c->d_luts) { S3D_HIP(hipFree(c->d_luts)); c->d_luts = nullptr; }
	if (c->d_lutpool) { S3D_HIP(hipFree(c->d_lutpool)); c->d_lutpool = nullptr; }
	S3D_HIP(hipMalloc(&c->d_luts, sizeof(WinLut) * luts.size()));
	S3D_HIP(hipMalloc(&c->d_lutpool, sizeof(float) * pool.size()));
	S3D_HIP(hipMemcpy(c->d_luts, luts.data(), sizeof(WinLut) * luts.size(), hipMemcpyHostToDevice));
	S3D_HIP(hipMemcpy(c->d_lutpool, pool.data(), sizeof(float) * pool.size(), hipMemcpyHostToDevice));
	return SIFT3D_OK;
}

static std::vector<WinLut> blank_luts(const sift3d_ctx *c) {
	std::vector<WinLut> luts((size_t)std::max(1, c->noct + c->octave_base) * 8 * 2);
	for (auto &l : luts) { l.off = 0; l.len = 0; l.nin = -1; l.radius = 0; l.sigma = 0; l.fix_scale = 1.0f; l.wsum = 1.0f; l.list_off = -1; l.list_R = 0; }
	return luts;
}

static int build_luts(sift3d_ctx *c) {
	std::vector<WinLut> luts = blank_luts(c);
	std::vector<float> pool;
	for (int o = 0; o < c->noct; o++)
		for (int lv = 1; lv <= c->p.num_kp_levels && lv < 8; lv++) {
			const Level &D = c->dog[(size_t)o * c->nd + lv];  // keypoint scale = DoG level scale (Src/cSIFT3D.cc:407)
			const float u = D.unit, scale = D.scale;
			for (int which = 0; which < 2; which++) {
				float sigma, radius;
				if (which == 0) { sigma = 1.5f * scale; radius = sigma * 3.0f; }
				else { sigma = scale * 7.071067812f; radius = 2.0f * sigma; }
				if (!append_lut(pool, luts[((size_t)(o + c->octave_base) * 8 + lv) * 2 + which], which, sigma, radius, u, scale)) c->desc_lut_lds = false;
			}
		}
	return upload_luts(c, luts, pool);
}

// how a context is built: the classic whole-volume extractor, a SEEDED tail (octaves >= octave_base starting from a
// caller-provided G[octave_base][0]) or a z-SLAB of octave 0 (multi-GPU sharding)
struct CreateCfg {
	int nx = 0, ny = 0, nz = 0;   // global dims of the first octave held
	int octave_base = 0;
	int noct_total = -1;          // octaves counted on the ORIGINAL volume (Src/cSIFT3D.cc:254-255); -1: from nx,ny,nz
	bool seeded = false;
	bool slab = false;
	int z0 = 0, z1 = 0, halo = 0;
	const float *host_volume = nullptr;  // plain extractor on a pageable host volume: uploaded beside the allocations (sift3d_ctx::uploader)
	float *ext_arena = nullptr;   // slab: caller-owned device memory for the level buffers (so that the caller's
	size_t ext_arena_floats = 0;  // communication layer can address halo planes directly), see sift3d_slab_arena_floats
};

static size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }

// floats needed for: input | per-octave scratch A,B | GSS levels | DoG levels
static size_t arena_floats_of(const sift3d_ctx *c) {
	size_t total = c->in_own ? 0 : al64(c->in.n());
	if (!c->slab) for (int o = 0; o < c->noct; o++) total += 2 * al64(c->gss[(size_t)o * c->ng].n());  // slabs only run the fused kernel
	for (auto &L : c->gss) total += al64(L.n());
	for (auto &L : c->dog) total += al64(L.n());
	return total;
}

static int create_common(sift3d_handle *out, const CreateCfg &cfg, const sift3d_params *params, int device) {
	*out = nullptr;
	if (cfg.nx <= 0 || cfg.ny <= 0 || cfg.nz <= 0) { set_last_error("bad dimensions"); return SIFT3D_ERR_ARG; }
	int rc = set_device(device);
	if (rc) return rc;
	sift3d_ctx *c = new sift3d_ctx();
	c->device = device;
	if (params) c->p = *params; else sift3d_default_params(&c->p);
	if (c->p.num_kp_levels < 1 || c->p.num_kp_levels > 5) { delete c; set_last_error("num_kp_levels must be in [1,5]"); return SIFT3D_ERR_ARG; }
	c->nx = cfg.nx; c->ny = cfg.ny; c->nz = cfg.nz;
	c->octave_base = cfg.octave_base; c->seeded = cfg.seeded;
	c->slab = cfg.slab; c->own0 = cfg.z0; c->own1 = cfg.z1; c->halo = cfg.halo;
	c->use_fused = !hook(SIFT3D_HOOK_SEPARABLE);
	plan_pyramid(c, cfg.noct_total);
	c->in.nx = cfg.nx; c->in.ny = cfg.ny; c->in.nz = cfg.nz; c->in.unit = (float)(1 << c->octave_base); c->in.scale = 1.f;
	if (c->slab) { c->in.bz = c->own1 - c->own0 + 2 * c->halo; c->in.zoff = c->own0 - c->halo; }
	if (c->seeded) c->in.bz = 1;  // no input volume: keep a token plane
	for (auto &L : c->gss) if ((size_t)L.nx * L.ny * L.planes() >= ((size_t)1 << 31)) { delete c; set_last_error("level too large for int32 voxel indices"); return SIFT3D_ERR_ARG; }

	// host tables
	std::vector<float> sig;
	float base_sigma;
	level_sigmas(c->p, sig, base_sigma);
	c->taps.resize(c->ng);
	bool ok = build_taps(base_sigma, c->base_taps);
	for (int i = 1; i < c->ng; i++) ok = ok && build_taps(sig[i], c->taps[i]);
	if (!ok) { delete c; set_last_error("Gaussian kernel wider than the supported 129 taps"); return SIFT3D_ERR_ARG; }

#define CHECKED(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); sift3d_destroy(c); return SIFT3D_ERR_HIP; } } while (0)
	CHECKED(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	for (auto &e : c->ev) CHECKED(hipEventCreate(&e));
	CHECKED(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
	c->ostream.assign((size_t)std::max(1, c->noct), nullptr);
	c->ev_seed.assign(c->ostream.size(), nullptr);
	c->ev_done.assign(c->ostream.size(), nullptr);
	c->own_stream = c->stream;
	c->ostream[0] = c->stream;
	if (cfg.host_volume && !cfg.ext_arena && !c->slab && !c->seeded) {
		const size_t V0 = c->in.n();
		CHECKED(hipMalloc(&c->in_own, sizeof(float) * al64(V0)));
		CHECKED(hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
		const float *vol = cfg.host_volume;
		// (measured, 512^3: 14.5 ms with the copy behind the allocations, 15.6 with its own buffer but no thread, 12.0 like this)
		c->uploader = std::thread([c, vol, V0, device] {
			int r = set_device(device);
			if (r == SIFT3D_OK) r = staged_h2d(c->in_own, vol, sizeof(float) * V0, device, c->up_stream);
			if (r != SIFT3D_OK) c->upload_err = sift3d_last_error();  // (the error text is thread-local)
			c->upload_rc = r;
		});
	}
	// SIFT3D_HOOK_ONE_STREAM (profiling): every octave on the main stream, so a kernel trace shows isolated launch durations
	const bool one_stream = hook(SIFT3D_HOOK_ONE_STREAM) != 0;
	for (size_t o = 0; o < c->ostream.size(); o++) {
		if (o > 0 && one_stream) c->ostream[o] = c->stream;
		else if (o > 0) {
#ifndef S3D_STREAM_PRIO
#define S3D_STREAM_PRIO 0  /* first octave whose stream is created with the highest queue priority (0: none) */
#endif
			int pr_lo = 0, pr_hi = 0;
			if (S3D_STREAM_PRIO > 0 && (int)o >= S3D_STREAM_PRIO && hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi) == hipSuccess && pr_hi != pr_lo)
				CHECKED(hipStreamCreateWithPriority(&c->ostream[o], hipStreamNonBlocking, pr_hi));
			else
				CHECKED(hipStreamCreateWithFlags(&c->ostream[o], hipStreamNonBlocking));
		}
		if (o == 1 && !one_stream && !c->slab) {
#ifndef S3D_CHAIN_STREAM
#define S3D_CHAIN_STREAM 1
#endif
			// (only for the volumes that use it, see run_enqueue: HIP deals streams to its four hardware queues in creation order, so one more
			// stream changes which octave streams share a queue -- 480 x 500 x 300: pyramid 1.43 -> 1.61 ms, the simulated 8-rank run 40.5 -> 43.6 ms)
			if (S3D_CHAIN_STREAM && (size_t)c->nx * c->ny * c->nz <= ((size_t)1 << 22)) CHECKED(hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
		}
		CHECKED(hipEventCreateWithFlags(&c->ev_seed[o], hipEventDisableTiming));
		CHECKED(hipEventCreateWithFlags(&c->ev_done[o], hipEventDisableTiming));
	}

	// ---- arena: input | per-octave scratch A,B | GSS levels | DoG levels (each 256-B aligned) ----
	const size_t total = arena_floats_of(c);
	c->arena_floats = total;
	if (cfg.ext_arena) {
		if (cfg.ext_arena_floats < total) { set_last_error("caller arena too small"); sift3d_destroy(c); return SIFT3D_ERR_ARG; }
		c->arena = cfg.ext_arena;
		c->ext_arena = true;
	} else {
		CHECKED(hipMalloc(&c->arena, sizeof(float) * total));
	}
	float *p = c->arena;
	if (c->in_own) c->in.d = c->in_own;
	else { c->in.d = p; p += al64(c->in.n()); }
	c->tmpA.assign((size_t)std::max(1, c->noct), nullptr);
	c->tmpB.assign((size_t)std::max(1, c->noct), nullptr);
	for (int o = 0; o < c->noct && !c->slab; o++) {
		const size_t vo = al64(c->gss[(size_t)o * c->ng].n());
		c->tmpA[o] = p; p += vo;
		c->tmpB[o] = p; p += vo;
	}
	for (auto &L : c->gss) { L.d = p; p += al64(L.n()); }
	for (auto &L : c->dog) { L.d = p; p += al64(L.n()); }

	const size_t nwords = 1 + (size_t)std::max(1, c->noct * c->nd) + 8;  // ... + total, overflow, nkp, describe work counter, describe redo counter, orientation redo counter
	CHECKED(hipMalloc(&c->d_words, sizeof(unsigned) * nwords));
	CHECKED(hipMemset(c->d_words, 0, sizeof(unsigned) * nwords));
	CHECKED(hipHostMalloc(&c->h_words, sizeof(unsigned) * 8, hipHostMallocDefault));
	CHECKED(hipMalloc(&c->d_slots_part, sizeof(unsigned) * slots_scratch_words()));
	c->d_inmax = c->d_words;
	c->d_dogmax = c->d_words + 1;
	c->d_total = c->d_words + 1 + std::max(1, c->noct * c->nd);  // [0] extrema total, [1] overflow flag
	c->d_nkp = c->d_total + 2;

	// detection scratch sized for the first octave (the largest): one ballot word per 64 voxels of a row, one count per 16 rows
	const int scan_planes = c->slab ? (c->own1 - c->own0) : cfg.nz;
	{
		const size_t kl = (size_t)c->p.num_kp_levels;
		const size_t words = kl * (size_t)scan_planes * cfg.ny * ((cfg.nx + 63) / 64);
		c->det_blocks = kl * (size_t)scan_planes * ((cfg.ny + 15) / 16);
		CHECKED(hipMalloc(&c->det.masks, sizeof(unsigned long long) * std::max<size_t>(words, 1)));
		CHECKED(hipMalloc(&c->det.block_counts, sizeof(unsigned) * std::max<size_t>(c->det_blocks, 1)));
		CHECKED(hipMalloc(&c->det.block_offsets, sizeof(unsigned) * std::max<size_t>(c->det_blocks, 1)));
	}
	c->det.total = c->d_total;
	{
		const bool det_serial = hook(SIFT3D_HOOK_DET_SERIAL) != 0;
		if (!c->slab && c->noct > 1 && !det_serial && c->ostream.size() > 1 && c->ostream[1] != c->stream) {
			const size_t kl = (size_t)c->p.num_kp_levels;
			c->det_o.assign((size_t)c->noct, DetectBufs{});
			size_t words = 0, blocks = 0;
			std::vector<size_t> woff((size_t)c->noct, 0), boff((size_t)c->noct, 0);
			for (int o = 1; o < c->noct; o++) {
				const Level &D = c->dog[(size_t)o * c->nd + 1];
				woff[(size_t)o] = words; boff[(size_t)o] = blocks;
				words += kl * (size_t)D.nz * D.ny * ((D.nx + 63) / 64);
				blocks += kl * (size_t)D.nz * ((D.ny + 15) / 16);
			}
			CHECKED(hipMalloc(&c->d_masks2, sizeof(unsigned long long) * std::max<size_t>(words, 1)));
			CHECKED(hipMalloc(&c->d_counts2, sizeof(unsigned) * std::max<size_t>(blocks, 1)));
			CHECKED(hipMalloc(&c->d_offsets2, sizeof(unsigned) * std::max<size_t>(blocks, 1)));
			for (int o = 1; o < c->noct; o++) {
				DetectBufs &b = c->det_o[(size_t)o];
				b.masks = c->d_masks2 + woff[(size_t)o]; b.block_counts = c->d_counts2 + boff[(size_t)o];
				b.block_offsets = c->d_offsets2 + boff[(size_t)o]; b.total = c->d_total;
			}
			CHECKED(hipEventCreateWithFlags(&c->ev_det_fork, hipEventDisableTiming));
			CHECKED(hipEventCreateWithFlags(&c->ev_det_join, hipEventDisableTiming));
			CHECKED(hipEventCreateWithFlags(&c->ev_det_fork2, hipEventDisableTiming));
			CHECKED(hipEventCreateWithFlags(&c->ev_det_join2, hipEventDisableTiming));
		}
	}

	// level table for the keypoint kernels, indexed by ABSOLUTE octave
	std::vector<LevelRef> lr((size_t)std::max(1, c->noct + c->octave_base) * 8, LevelRef{nullptr, 0, 0, 0, 1.f, 0});
	for (int o = 0; o < c->noct; o++)
		for (int i = 0; i < c->ng && i < 8; i++) {
			const Level &L = c->gss[(size_t)o * c->ng + i];
			lr[(size_t)(o + c->octave_base) * 8 + i] = LevelRef{L.d, L.nx, L.ny, L.nz, L.unit, L.zoff};
		}
	CHECKED(hipMalloc(&c->d_levels, sizeof(LevelRef) * lr.size()));
	CHECKED(hipMemcpy(c->d_levels, lr.data(), sizeof(LevelRef) * lr.size(), hipMemcpyHostToDevice));
	rc = build_luts(c);
	if (rc) { sift3d_destroy(c); return rc; }
	FaceConst faces[kFaces];
	build_faces(faces);
	FaceSym sym;
	if (!build_facesym(faces, &sym)) { set_last_error("icosahedron symmetry table: no matching face"); sift3d_destroy(c); return SIFT3D_ERR_STATE; }
	CHECKED(upload_faces(faces, &sym));

	// keypoint lists: synthetic blob volumes give ~6e-4*V extrema; leave 8x headroom, regrow on overflow
	const size_t V0 = (size_t)cfg.nx * cfg.ny * scan_planes;
	unsigned cap = (unsigned)std::min<size_t>(std::max<size_t>(4096, V0 / 256), 4u << 20);
	if (hook(SIFT3D_HOOK_LIST_CAP) > 0) cap = (unsigned)hook(SIFT3D_HOOK_LIST_CAP);  // tests: overflow -> regrow -> rerun
	rc = alloc_lists(c, cap);
	if (rc) { sift3d_destroy(c); return rc; }
	CHECKED(hipStreamSynchronize(c->stream));
#undef CHECKED
	{
		// first create on this device: load the kernels of every translation unit now (HIP loads a unit's code object at the first
		// launch of one of its kernels: ~1 ms of the first KpSiftAlgorithm of a process went there, scripts/step_times_probe.py)
		static std::atomic<unsigned long long> loaded{0};
		const unsigned long long bit = 1ull << (device & 63);
		if (!(loaded.fetch_or(bit) & bit)) { preload_march_kernels(); preload_small_kernels(); preload_detect_kernels(); preload_orient_kernels(); preload_desc_kernels(); preload_match_kernels(); }
	}
	*out = c;
	return SIFT3D_OK;
}

extern "C" int sift3d_create(sift3d_handle *out, const float *volume, int nx, int ny, int nz, const sift3d_params *params,
                             int device, int volume_on_device) {
	if (!out || !volume || nx <= 0 || ny <= 0 || nz <= 0) { set_last_error("sift3d_create: bad argument"); return SIFT3D_ERR_ARG; }
	if ((size_t)nx * ny * nz >= ((size_t)1 << 31)) { set_last_error("volume too large for int32 voxel indices"); return SIFT3D_ERR_ARG; }
	CreateCfg cfg;
	cfg.nx = nx; cfg.ny = ny; cfg.nz = nz;
	if (!volume_on_device) cfg.host_volume = volume;  // its upload starts inside create_common, beside the allocations
	int rc = create_common(out, cfg, params, device);
	if (rc) return rc;
	sift3d_ctx *c = *out;
	// ---- constructor work proper: copy + data_scale (Src/cSIFT3D.cc:161-162) ----
	const size_t V0 = (size_t)nx * ny * nz;
	hipError_t e = hipSuccess;
	if (volume_on_device) e = hipMemcpyAsync(c->in.d, volume, sizeof(float) * V0, hipMemcpyDeviceToDevice, c->stream);
	else {
		if (c->uploader.joinable()) c->uploader.join();
		if (c->upload_rc != SIFT3D_OK) { rc = c->upload_rc; set_last_error(c->upload_err); sift3d_destroy(c); *out = nullptr; return rc; }
		e = hipStreamSynchronize(c->up_stream);  // every chunk has landed: the kernels below run on the handle's stream
	}
	if (e == hipSuccess) {
		launch_absmax(c->in.d, V0, c->d_inmax, c->stream);
		launch_scale_by_max(c->in.d, V0, c->d_inmax, c->stream);
		e = hipStreamSynchronize(c->stream);
	}
	if (e == hipSuccess) e = hipGetLastError();
	if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); sift3d_destroy(c); *out = nullptr; return SIFT3D_ERR_HIP; }
	return SIFT3D_OK;
}

// GaussianSmooth_3D (Src/cSIFT3D.cc:535-622) on device buffers: X -> Y -> Z(+DoG)
#ifndef S3D_FUSED_MIN_DEFAULT
#define S3D_FUSED_MIN_DEFAULT 33  /* levels with a dimension <= 32 (octaves 4+ of a 512^3 volume) take the generic separable kernels: 3.88 vs 4.02 ms */
#endif
#ifndef S3D_FUSED_HALF
#define S3D_FUSED_HALF 1  /* the seed level's march kernel also writes level 0 of the next octave (no decimation launch on the octave -> octave chain) */
#endif
#ifndef S3D_O0_TAIL_SLOTS_DEFAULT
#define S3D_O0_TAIL_SLOTS_DEFAULT 512  /* r02 (march kernel): 2.92 vs 3.05 ms with bg 256 */
#endif
#ifndef S3D_BG_SLOTS_DEFAULT
#define S3D_BG_SLOTS_DEFAULT 256
#endif
// half_out (optional): level 0 of the next octave; returns true when the march kernel wrote it together with dst (the caller then
// skips the decimation launch)
static bool smooth_level(sift3d_ctx *c, int o, const float *src, const Level &dst, const Taps &t, const float *prev, float *dog,
                         unsigned *dogmax, int level = 0, const Level *half_out = nullptr, hipStream_t st_override = nullptr) {
	hipStream_t st = st_override ? st_override : c->ostream[o];
	// Slot planning across the octave streams (single-round launches keep every slot they take until they end): the levels of
	// octave 0 behind the seed level G[0][num_kp_levels] leave a third of the machine to the chains of the smaller octaves, which
	// are planned for that third; otherwise those chains starve and run as a tail after octave 0 has finished.
	static const int tail_slots = dev_tune_i("S3D_O0_TAIL_SLOTS", S3D_O0_TAIL_SLOTS_DEFAULT);
	static const int bg_slots = dev_tune_i("S3D_BG_SLOTS", S3D_BG_SLOTS_DEFAULT);
	// wave priority: the launches of octaves >= 2 (1/64 of the work, but each octave waits for level 3 of the one above, and beside the
	// big launches their workgroups crawl) run at the highest wave priority: 2.82 -> 2.71 ms per 512^3 pyramid.  Raising octave 1 too
	// (S3D_PRIO=1) costs octave 0 as much as it gains.
	// r04 (scripts/sweep_sched3.sh, with the small octaves in one launch): octave 1 at the top priority too -- 2.20 vs 2.23 ms; its slot
	// plan (S3D_BG1_SLOTS 256 / 384 / 512) and more slots for the octaves behind it (S3D_BG_SLOTS 512 / 768: 2.23-2.27) change nothing
	static const int prio_mode = dev_tune_i("S3D_PRIO", 3);  // 1: octave 1 at wave priority 1; 2: at 0; 3: octave 1 at the top priority like the octaves behind it
	static const int bg1_slots = dev_tune_i("S3D_BG1_SLOTS", 0);  // slot plan of octave 1's levels (0: bg_slots)
	const int prio = (prio_mode && c->noct > 1) ? (o >= 2 ? 2 : (o == 1 ? (prio_mode == 1 ? 1 : (prio_mode == 3 ? 2 : 0)) : 0)) : 0;
	const int plan_slots = c->noct > 1 ? (o == 0 ? (level > c->p.num_kp_levels ? tail_slots : 0) : (o == 1 && bg1_slots > 0 ? bg1_slots : bg_slots)) : 0;
	// hot path: one fused pass (x, y, z blur + DoG + abs-max; kernels_march.hip); prev == src for every DoG-producing level
	static const int fused_min = dev_tune_i("S3D_FUSED_MIN", S3D_FUSED_MIN_DEFAULT);
	// r04 (shape cliff #2): the plane size decides, not the depth -- a thin volume (512 x 512 x 24: common MR / CT slabs) has planes of
	// many tiles and marches its few planes like any other chunk (the z ends are a feed order); launch_march_level declines a level
	// whose column is shorter than its kernel (nz < 2 hw + 2), which then takes the separable passes.  Cubes of <= 32 stay separable.
	if (c->use_fused && (prev == nullptr || prev == src) && std::min(dst.nx, dst.ny) >= fused_min) {
		MarchHalf hf;
		const bool want_half = S3D_FUSED_HALF && half_out != nullptr && march_half_ok(dst.nx, dst.ny, dst.zr_all());
		if (want_half) { hf.d = half_out->d; hf.nx = half_out->nx; hf.ny = half_out->ny; hf.nz = half_out->nz; }
		if (launch_march_level(src, dst.d, dog, dogmax, dst.nx, dst.ny, dst.zr_all(), t, st, plan_slots, prio, want_half ? &hf : nullptr))
			return want_half;
	}
	launch_conv_axis(0, src, c->tmpA[o], dst.nx, dst.ny, dst.nz, t, nullptr, nullptr, nullptr, st);
	launch_conv_axis(1, c->tmpA[o], c->tmpB[o], dst.nx, dst.ny, dst.nz, t, nullptr, nullptr, nullptr, st);
	launch_conv_axis(2, c->tmpB[o], dst.d, dst.nx, dst.ny, dst.nz, t, prev, dog, dogmax, st);
	return false;
}

// One run = prepare (flags of the run) -> enqueue (the whole pipeline on the handle's streams, up to the asynchronous read-back of the
// five counters into pinned memory; no host synchronisation) -> finish (synchronise, check the list capacity, fill state and times;
// on overflow: regrow and enqueue again).  sift3d_run / sift3d_run_stages do all three; sift3d_run_async stops behind the first
// enqueue and sift3d_wait finishes (r04: one host thread keeps several handles in flight).
static int run_prepare(sift3d_ctx *c, int &upto) {
	if (c->slab) { set_last_error("a z-slab context is driven stage by stage (sift3d_slab_*)"); return SIFT3D_ERR_STATE; }
	if (c->pending) { set_last_error("the handle has an asynchronous run in flight: call sift3d_wait first"); return SIFT3D_ERR_STATE; }
	int rc = set_device(c->device);
	if (rc) return rc;
	if (upto < 1) upto = 1;
	if (upto > 5) upto = 5;
	// SIFT3D_HOOK_DOG_EAGER: write every DoG level
	const bool dog_eager = hook(SIFT3D_HOOK_DOG_EAGER) != 0;
	c->dog_elide = !dog_eager && c->nd >= 3;
	// the last Gaussian level is only ever read at the voxels that pass seven of the eight extremum tests of the last keypoint level:
	// it is not built at all; those voxels get the value from k_lazy_next (DetectLevels::lazy_src).  SIFT3D_HOOK_GLAST_EAGER builds it.
	const bool glast_eager = hook(SIFT3D_HOOK_GLAST_EAGER) != 0;
	c->g_last_elide = c->dog_elide && !glast_eager && c->use_fused && 2 * (2 * c->taps[c->ng - 1].hw + 1) <= kLazySlots;
	c->g_last_built.assign((size_t)std::max(1, c->noct), 0);
	c->n_regrow = 0;
	return SIFT3D_OK;
}

static int run_enqueue(sift3d_ctx *c, int upto, bool part_orient) {
	int rc = set_device(c->device);
	if (rc) return rc;
	hipStream_t st = c->stream;
	if (c->dsplit_dirty && c->dsplit.gacc) {  // (a run cut short by an error may have left partial sums / arrival counts behind)
		S3D_HIP(hipMemsetAsync(c->dsplit.gacc, 0, sizeof(int) * (size_t)c->dsplit.cap * (kDesc + 8 + 1), st));
		c->dsplit_dirty = false;
	}
	{
		if (c->gate) { S3D_HIP(hipStreamWaitEvent(st, c->gate, 0)); c->gate = nullptr; }
		S3D_HIP(hipMemsetAsync(c->d_dogmax, 0, sizeof(unsigned) * (size_t)(std::max(1, c->noct * c->nd) + 6), st));
		S3D_HIP(hipEventRecord(c->ev[0], st));
		// ---- Build_Gaussian_Scale_Space (Src/cSIFT3D.cc:268-319) with the DoG (346-360) fused into the z pass ----
		// fork: every octave stream starts after the main stream reached this point; octave o is seeded by
		// G[o-1][num_kp_levels] (DownSample_3D), everything else of octave o-1 overlaps with octave o
		S3D_HIP(hipEventRecord(c->ev_fork, st));
		// Enqueue order: head(0), head(1), tail(0), head(2), tail(1), ... -- head(o) = the levels of octave o up to its seed level
		// G[o][num_kp_levels], tail(o) = the levels behind it.  The critical path of the stage is the chain of heads (every octave
		// waits for the seed level of the one above); the tail of octave 0 (its widest Gaussian) is machine-filling work that is off
		// that path and starts together with head(1).  Measured and rejected (r03, again r04): tail(0) waiting for the seed of
		// octave 1, so that head(1) -- 351 us alone, 604 us beside tail(0) in the rocprofv3 timeline -- runs undisturbed and tail(0)
		// fills the machine under the launch-latency chain of the small octaves instead: 2.36 -> 2.51 ms at 512^3 for every slot
		// planning tried (the tail is longer than that chain).
		std::vector<char> half_written((size_t)c->noct + 1, 0);  // level 0 of octave o was written by the seed level's kernel of octave o - 1
		// r04: the SMALL octaves (16^3-class and below) run in ONE launch of one workgroup that keeps the octave in LDS
		// (kernels_small.hip) instead of ~16 launches of a few microseconds per octave on the stage's critical chain
		int small_first = -1;
		SmallArgs sa;
		{
			static const int small_on = dev_tune_i("S3D_SMALL_OCT", 1);
			unsigned build_mask = 0, dog_mask = 0;
			int max_hw = 0;
			bool ok = small_on != 0 && c->use_fused && !c->slab && c->ng <= kSmallMaxLv;
			for (int i = 1; i < c->ng && ok; i++) {
				if (c->g_last_elide && i == c->ng - 1) continue;
				build_mask |= 1u << i;
				if (!(c->dog_elide && (i - 1 == 0 || i - 1 == c->nd - 1))) dog_mask |= 1u << (i - 1);
				const Taps &t = c->taps[i];
				for (int d = 1; d <= t.hw && ok; d++) ok = memcmp(&t.w[t.hw + d], &t.w[t.hw - d], sizeof(float)) == 0;  // symmetric bit for bit
				max_hw = std::max(max_hw, t.hw);
			}
			if (ok && small_padded_hw(max_hw) > 0)
				for (int o = 0; o < c->noct; o++) {
					const Level &L = c->gss[(size_t)o * c->ng];
					if (c->noct - o <= kSmallMaxOct && small_octave_fits(L.nx, L.ny, L.nz, max_hw)) { small_first = o; break; }
				}
			// EVERY octave of the launch must satisfy hw <= n - 2 (the extended-line form of the boundary rule): the octaves behind the first
			// are smaller, which makes the capacity limits easier and THIS one harder (sigma_default 1.7: hw 7 at the 8^3 octave of a
			// power-of-two volume; the glast_eager hook: hw 8).  One that fails sends the whole chain down the separable kernels.
			for (int o = small_first; small_first >= 0 && o < c->noct; o++) {
				const Level &L = c->gss[(size_t)o * c->ng];
				if (!small_octave_fits(L.nx, L.ny, L.nz, max_hw)) small_first = -1;
			}
			if (small_first >= 0) {
				memset(&sa, 0, sizeof(sa));
				sa.noct = c->noct - small_first; sa.ng = c->ng; sa.nd = c->nd; sa.seed = c->p.num_kp_levels;
				sa.build_mask = build_mask; sa.dog_mask = dog_mask;
				sa.dogmax = c->d_dogmax + (size_t)small_first * c->nd;
				sa.hwp = small_padded_hw(max_hw);
				for (int i = 1; i < c->ng; i++) {
					const Taps &t = c->taps[i];
					if (t.hw > sa.hwp) continue;  // (a level the launch does not build)
					for (int k = 0; k <= sa.hwp; k++) sa.w[i][k] = k >= sa.hwp - t.hw ? t.w[k - (sa.hwp - t.hw)] : 0.0f;
				}
				for (int o = small_first; o < c->noct; o++) {
					SmallOct &so = sa.oct[o - small_first];
					const Level &L = c->gss[(size_t)o * c->ng];
					so.nx = L.nx; so.ny = L.ny; so.nz = L.nz;
					for (int i = 0; i < c->ng; i++) so.g[i] = c->gss[(size_t)o * c->ng + i].d;
					for (int i = 0; i < c->nd; i++) so.dog[i] = c->dog[(size_t)o * c->nd + i].d;
				}
			}
		}
		auto enqueue = [&](int o, bool head) -> int {
			// (the octaves of the small launch share the stream of the first of them; heads of octaves >= 1: the chain stream)
			// measured (scripts/small_volume_times.py, A/B in one process per setting): 64^3 / 128^3 pyramid 0.238 / 0.348 -> 0.220 / 0.320 ms,
			// but 256^3 0.600 -> 0.656 and 512^3 2.13 -> 2.16 ms (the heads of the big octaves are long launches that gain nothing from a
			// gap of 8 instead of 15 us and lose the hardware queue they had to themselves): volumes of at most 4 M voxels only
			static const int chain_mode = dev_tune_i("S3D_CHAIN", 1);  // 0 never, 1 small volumes (the stream exists for them only)
			const bool chain_on = chain_mode != 0;
			const bool chained = c->cstream != nullptr && chain_on && o >= 1 && (head || (small_first >= 0 && o >= small_first));
			hipStream_t so = chained ? c->cstream : c->ostream[(small_first >= 0 && o > small_first) ? small_first : o];
			// late r04: HIP deals the streams onto its four hardware queues in the pattern 1 2 3 4 4 3 2 1 (profiles/r04e_timeline.txt, queue
			// ids), so with the small launch on stream 4 (256^3-class volumes: six octaves) the levels behind the seed level of octave 3 sat in
			// the same hardware queue BEHIND that launch and ended the stage 25 us after it.  They go to the stream of the octave above
			// (idle by then: its own last level is a 128^3-class launch that ended long before) -- off the critical chain either way
			const bool tail_moved = !head && small_first >= 3 && o == small_first - 1 && c->cstream == nullptr;
			if (tail_moved) so = c->ostream[o - 1];
			if (small_first >= 0 && o >= small_first) {
				if (head && o > 0) {
					S3D_HIP(hipStreamWaitEvent(so, c->ev_fork, 0));
					S3D_HIP(hipStreamWaitEvent(so, c->ev_seed[o - 1], 0));
				}
				if (head && o == small_first) {
					const Level &L = c->gss[(size_t)o * c->ng];
					if (o == 0) {
						if (!c->seeded) smooth_level(c, o, c->in.d, L, c->base_taps, nullptr, nullptr, nullptr);
					} else if (!half_written[(size_t)o]) {  // the launch decimates the parent's seed level itself (no decimation launch)
						const Level &P = c->gss[(size_t)(o - 1) * c->ng + c->p.num_kp_levels];
						sa.parent = P.d; sa.pnx = P.nx; sa.pny = P.ny;
					}
					launch_small_octaves(sa, so);
				}
				if (head) S3D_HIP(hipEventRecord(c->ev_seed[o], so));
				if (!head && o > 0) S3D_HIP(hipEventRecord(c->ev_done[o], so));
				return SIFT3D_OK;
			}
			if (head && o > 0) {
				S3D_HIP(hipStreamWaitEvent(so, c->ev_fork, 0));
				S3D_HIP(hipStreamWaitEvent(so, c->ev_seed[o - 1], 0));
			}
			if (!head && o >= 1 && ((c->cstream != nullptr && chain_on) || tail_moved)) {  // the levels behind the seed level: the octave's own stream, behind its head on the chain stream
				S3D_HIP(hipStreamWaitEvent(so, c->ev_fork, 0));
				S3D_HIP(hipStreamWaitEvent(so, c->ev_seed[o], 0));
			}
			const int i0 = head ? 0 : c->p.num_kp_levels + 1, i1 = head ? c->p.num_kp_levels + 1 : c->ng;
			for (int i = i0; i < i1; i++) {
				const Level &L = c->gss[(size_t)o * c->ng + i];
				if (o == 0 && i == 0) {
					if (!c->seeded) smooth_level(c, o, c->in.d, L, c->base_taps, nullptr, nullptr, nullptr);
					// seeded: G[octave_base][0] was written by sift3d_seed_upload
				} else if (i == 0) {
					const Level &P = c->gss[(size_t)(o - 1) * c->ng + c->p.num_kp_levels];
					if (!half_written[(size_t)o]) launch_downsample(P.d, P.nx, P.ny, L.d, L.nx, L.ny, L.nz, so);
				} else {
					const Level &P = c->gss[(size_t)o * c->ng + i - 1];
					const Level &D = c->dog[(size_t)o * c->nd + i - 1];
					if (c->g_last_elide && i == c->ng - 1) continue;  // never built (k_lazy_next / sift3d_copy_level form what is asked for)
					// the seed level also leaves decimated, as level 0 of the next octave (whole volumes, not seeded / partitioned contexts' inputs)
					const Level *half = (i == c->p.num_kp_levels && o + 1 < c->noct) ? &c->gss[(size_t)(o + 1) * c->ng] : nullptr;
					bool hw_ = false;
					if (c->dog_elide && (i - 1 == 0 || i - 1 == c->nd - 1)) hw_ = smooth_level(c, o, P.d, L, c->taps[i], nullptr, nullptr, nullptr, i, half, so);
					else hw_ = smooth_level(c, o, P.d, L, c->taps[i], P.d, D.d, c->d_dogmax + (size_t)o * c->nd + i - 1, i, half, so);
					if (half) half_written[(size_t)o + 1] = hw_ ? 1 : 0;
				}
				if (i == c->p.num_kp_levels) S3D_HIP(hipEventRecord(c->ev_seed[o], so));
			}
			if (!head && o > 0) S3D_HIP(hipEventRecord(c->ev_done[o], so));
			return SIFT3D_OK;
		};
		for (int o = 0; o <= c->noct; o++) {
			if (o < c->noct && (rc = enqueue(o, true)) != SIFT3D_OK) return rc;
			if (o >= 1 && (rc = enqueue(o - 1, false)) != SIFT3D_OK) return rc;
		}
		// r03, an option that is OFF (S3D_DET_EARLY_DEFAULT) -- octave 0's extremum masks start right behind its last level, BESIDE the chains of the small octaves: after octave 0's
		// widest level the stage used to end with ~0.3 ms of launch-latency-bound chain (octaves 2..6: 60 launches of a few
		// microseconds, each octave waiting for the seed level of the one above) on a nearly idle machine, while k_mark of octave 0
		// -- 0.38 ms, memory bound -- waited behind the join.  The pyramid stage still ends when EVERY octave's pyramid has
		// (ev[1] is recorded on the second detection stream after it has waited for all of them), so the stage times stay honest:
		// the pyramid's is its wall time, the detection's is what is left of it behind the pyramid.
		const bool two = upto >= 3 && c->det_o.size() > 1;  // masks of octaves >= 1 on a second stream beside octave 0's
#ifndef S3D_DET_EARLY_DEFAULT
#define S3D_DET_EARLY_DEFAULT 1  /* r04: ON -- with the small octaves in one launch the chain no longer starves behind k_mark's workgroups: detection 0.87 -> 0.82 ms, pyramid 2.18 -> 2.19 ms, step 7.41 -> 7.38 ms (r03, ~60 chain launches: detection 0.87 -> 0.73 but pyramid 2.31 -> 2.40, off) */
#endif
		static const int det_early_mode = dev_tune_i("S3D_DET_EARLY", S3D_DET_EARLY_DEFAULT);
		const bool early = two && det_early_mode != 0 && c->noct > 1 && c->ostream.size() > 1 && c->ostream[1] != st;
		std::vector<DetectLevels> DLs((size_t)c->noct);
		const int nl = c->nd - 2;  // DoG levels 1 .. nd-2 (Src/cSIFT3D.cc:376)
		const Taps *lt = c->g_last_elide ? &c->taps[c->ng - 1] : nullptr;
		if (upto >= 3)
			for (int o = 0; o < c->noct; o++) {
				DetectLevels &DL = DLs[(size_t)o];
				memset(&DL, 0, sizeof(DL));
				for (int i = 1; i <= nl; i++) {
					DL.cur[i - 1] = c->dog[(size_t)o * c->nd + i].d;
					DL.prev[i - 1] = c->dog[(size_t)o * c->nd + i - 1].d;
					DL.next[i - 1] = c->dog[(size_t)o * c->nd + i + 1].d;
					DL.absmax_bits[i - 1] = c->d_dogmax + (size_t)o * c->nd + i;
					DL.level_id[i - 1] = i;
					DL.scale[i - 1] = c->dog[(size_t)o * c->nd + i].scale;
				}
				if (c->dog_elide) {
					DL.prev0_hi = c->gss[(size_t)o * c->ng + 1].d; DL.prev0_lo = c->gss[(size_t)o * c->ng].d;
					DL.nextl_hi = c->gss[(size_t)o * c->ng + c->nd].d; DL.nextl_lo = c->gss[(size_t)o * c->ng + c->nd - 1].d;
					DL.nextl_slot = nl - 1;
					if (c->g_last_elide) { DL.nextl_hi = nullptr; DL.lazy_src = DL.nextl_lo; }
				}
			}
		hipStream_t sb = two ? c->ostream[1] : st;
		if (early) {
			const Level &C0 = c->dog[(size_t)0 * c->nd + 1];
			S3D_HIP(hipEventRecord(c->ev_det_fork, st));  // octave 0's pyramid is complete (the main stream is its stream)
			launch_detect_mark(DLs[0], nl, C0.nx, C0.ny, C0.zr_all(), c->p.peak_thresh, 0 + c->octave_base, c->det, st, lt);
			S3D_HIP(hipStreamWaitEvent(sb, c->ev_det_fork, 0));
			// (octave 1 too: its levels are in sb's own order only when neither the chain stream nor the small-octave launch took them)
			for (int o = 1; o < c->noct; o++) S3D_HIP(hipStreamWaitEvent(sb, c->ev_done[o], 0));
			S3D_HIP(hipEventRecord(c->ev[1], sb));  // every octave's pyramid is complete
			S3D_HIP(hipEventRecord(c->ev[2], sb));  // DoG is fused: zero-length stage
		} else {
			for (int o = 1; o < c->noct; o++) S3D_HIP(hipStreamWaitEvent(st, c->ev_done[o], 0));  // join
			S3D_HIP(hipEventRecord(c->ev[1], st));
			S3D_HIP(hipEventRecord(c->ev[2], st));  // DoG is fused: zero-length stage
		}
		// ---- Detect_KeyPoints (Src/cSIFT3D.cc:362-425) ----
		if (upto >= 3) {
			if (two && !early) { S3D_HIP(hipEventRecord(c->ev_det_fork, st)); S3D_HIP(hipStreamWaitEvent(sb, c->ev_det_fork, 0)); }
			if (two) {
				// late r04: the masks of the octaves >= 2 (twelve launches of a few microseconds at 512^3) on a THIRD stream beside octave
				// 1's: they used to queue behind octave 1's candidate pass (0.35 ms beside octave 0's masks) and ended the stage 0.13 ms
				// after octave 0's emit (profiles/r04e_timeline_full.txt); every octave has its own scratch
				hipStream_t sc = (c->noct > 2 && c->ostream.size() > 2 && c->ostream[2] != st && c->ostream[2] != sb) ? c->ostream[2] : sb;
				// r05: with the early start the masks of an octave >= 2 follow ITS pyramid (the event its last level recorded), not every
				// octave's: the chains of these octaves -- three launches of a few microseconds each per octave, the last thing the stage waited
				// for whenever HIP dealt sb and sc onto one hardware queue -- start beside the tail of the pyramid.  Measured (scripts/ab_full.py,
				// three pairs): extrema 0.75 -> 0.69 ms, pyramid 2.02 -> 2.02; octave 1's masks behind octave 1's pyramid as well: extrema 0.67
				// but pyramid 2.06 (its 0.2 ms of k_mark beside octave 0's widest level) -- not taken
				const bool own = early && sc != sb;
				if (sc != sb && !own) { S3D_HIP(hipEventRecord(c->ev_det_fork2, sb)); S3D_HIP(hipStreamWaitEvent(sc, c->ev_det_fork2, 0)); }  // every pyramid is complete
				for (int o = 1; o < c->noct; o++) {
					const Level &C = c->dog[(size_t)o * c->nd + 1];
					if (own && o >= 2) S3D_HIP(hipStreamWaitEvent(sc, c->ev_done[o], 0));
					launch_detect_mark(DLs[(size_t)o], nl, C.nx, C.ny, C.zr_all(), c->p.peak_thresh, o + c->octave_base, c->det_o[(size_t)o], o == 1 ? sb : sc, lt);
				}
				S3D_HIP(hipEventRecord(c->ev_det_join, sb));
				if (sc != sb) { S3D_HIP(hipEventRecord(c->ev_det_join2, sc)); S3D_HIP(hipStreamWaitEvent(sb, c->ev_det_join2, 0)); S3D_HIP(hipEventRecord(c->ev_det_join, sb)); }
			}
			std::vector<DetectEmitItem> rest;  // two streams: the octaves behind the first one are emitted by one scan + one emit launch
			for (int o = 0; o < c->noct; o++) {
				const Level &C = c->dog[(size_t)o * c->nd + 1];
				const DetectBufs &b = (two && o > 0) ? c->det_o[(size_t)o] : c->det;
				if (!(two && o > 0) && !(early && o == 0)) launch_detect_mark(DLs[(size_t)o], nl, C.nx, C.ny, C.zr_all(), c->p.peak_thresh, o + c->octave_base, b, st, lt);
				if (two && o == 1) S3D_HIP(hipStreamWaitEvent(st, c->ev_det_join, 0));
				if (two && o > 0 && c->noct - 1 <= 8) rest.push_back(DetectEmitItem{&DLs[(size_t)o], nl, C.nx, C.ny, C.zr_all(), o + c->octave_base, &b});
				else launch_detect_emit(DLs[(size_t)o], nl, C.nx, C.ny, C.zr_all(), o + c->octave_base, b, c->d_ext, c->ext_cap, st);
			}
			if (!rest.empty()) launch_detect_emit_multi(rest.data(), (int)rest.size(), c->d_ext, c->ext_cap, c->d_total, st);
		}
		S3D_HIP(hipEventRecord(c->ev[3], st));
		// ---- Assign_Orientation (Src/cSIFT3D.cc:427-482) ----
		if (upto >= 4) {
			launch_orient(c->d_ext, c->d_codes, c->d_total, c->ext_cap, c->d_levels, c->d_luts, c->d_lutpool, c->p.max_eig_thres,
			              c->p.corner_thresh, part_orient ? c->part_rank : 0, part_orient ? c->part_world : 1, c->d_order, c->d_nkp + 3, st);
			launch_slots(c->d_ext, c->d_codes, c->d_total, c->ext_cap, c->d_nkp, c->d_order, c->kp_cap, c->d_slots_part, st);
		}
		S3D_HIP(hipEventRecord(c->ev[4], st));
		// ---- Extract_Description (Src/cSIFT3D.cc:484-502) ----
		if (upto >= 5) {
			launch_describe(c->d_ext, c->d_total, c->ext_cap, c->d_levels, c->d_luts, c->d_lutpool, c->d_desc, c->kp_cap, c->part_rank,
			                c->part_world, c->d_order, c->d_nkp, c->d_nkp + 1, st, c->desc_lut_lds, &c->dsplit);
		}
		if (upto >= 4) launch_finalize(c->d_ext, c->d_total, c->ext_cap, upto >= 5, c->d_kpout, c->d_xyz, c->kp_cap, st);
		S3D_HIP(hipEventRecord(c->ev[5], st));
		// total, overflow, nkp, describe work counter, describe second passes -> pinned host words
		S3D_HIP(hipMemcpyAsync(c->h_words, c->d_total, sizeof(unsigned) * 5, hipMemcpyDeviceToHost, st));
	}
	return SIFT3D_OK;
}

// synchronise with the enqueued run and take its results; again = the lists overflowed and were regrown: enqueue once more
static int run_finish(sift3d_ctx *c, int upto, bool &again) {
	again = false;
	int rc = set_device(c->device);
	if (rc) return rc;
	hipStream_t st = c->stream;
	{
		unsigned *host_words = c->h_words;
		S3D_HIP(hipStreamSynchronize(st));
		S3D_HIP(hipGetLastError());
		if (host_words[1] != 0 || host_words[0] > c->ext_cap) {
			// list overflow: regrow to fit and rerun (rare; sizes are generous).  The detection kernels keep counting past the
			// capacity, so host_words[0] is the number of extrema of the volume (short of candidates dropped from a full parking list).
			const unsigned need = std::max(host_words[0], c->ext_cap) * 2u;
			rc = alloc_lists(c, need);
			if (rc) return rc;
			c->n_regrow++;
			again = true;
			return SIFT3D_OK;
		}
		c->n_ext = host_words[0];
		c->n_kp = upto >= 4 ? host_words[2] : 0;
		c->n_desc_redo = upto >= 5 ? (int)host_words[4] : 0;
		c->stage = upto;
		float ms = 0;
		auto dt = [&](int a, int b) { hipEventElapsedTime(&ms, c->ev[a], c->ev[b]); return (double)ms * 1e-3; };
		c->times[0] = dt(0, 5); c->times[1] = 0; c->times[2] = dt(0, 1); c->times[3] = dt(1, 2);
		c->times[4] = dt(2, 3); c->times[5] = dt(3, 4); c->times[6] = dt(4, 5); c->times[7] = 0;
		return SIFT3D_OK;
	}
}

static int run_complete(sift3d_ctx *c, int upto, bool part_orient) {  // behind an enqueue: finish, re-enqueueing while the lists overflow
	for (int attempt = 0; attempt < 8; attempt++) {
		bool again = false;
		int rc = run_finish(c, upto, again);
		if (rc) return rc;
		if (!again) return SIFT3D_OK;
		if ((rc = run_enqueue(c, upto, part_orient)) != SIFT3D_OK) return rc;
	}
	(void)hipStreamSynchronize(c->stream);
	set_last_error("extrema list kept overflowing");
	return SIFT3D_ERR_CAPACITY;
}

static int run_impl(sift3d_ctx *c, int upto, bool part_orient = false) {
	int rc = run_prepare(c, upto);
	if (rc) return rc;
	if ((rc = run_enqueue(c, upto, part_orient)) == SIFT3D_OK) rc = run_complete(c, upto, part_orient);
	if (rc) c->dsplit_dirty = true;
	return rc;
}

extern "C" int sift3d_run(sift3d_handle c) {
	if (!c) return SIFT3D_ERR_ARG;
	return run_impl(c, 5);
}

// KpSiftAlgorithm without the wait: the whole pipeline is enqueued on the handle's own streams and the call returns; sift3d_wait
// completes it.  One host thread can so keep several handles (volumes) in flight on one GPU -- the pyramid of one volume is bound by
// memory, the descriptors of another by instruction issue (BASELINE configs[2] / [4]: several volumes per GPU).
extern "C" int sift3d_run_async(sift3d_handle c) {
	if (!c) return SIFT3D_ERR_ARG;
	int upto = 5;
	int rc = run_prepare(c, upto);
	if (rc) return rc;
	if ((rc = run_enqueue(c, upto, false)) != SIFT3D_OK) { (void)hipStreamSynchronize(c->stream); c->dsplit_dirty = true; return rc; }
	c->pending = true;
	return SIFT3D_OK;
}

// Two volumes back to back on one GPU (Example.cpp:21-44 extracts a reference and a target volume one after the other): the second
// volume's pipeline starts when the FIRST volume's orientation stage has ended, i.e. its memory-bound front (pyramid, extrema, orientation)
// runs beside the first volume's descriptor stage, which is bound by instruction issue and the LDS.  `after` must have a run in flight
// (sift3d_run_async) on the same device; otherwise this is sift3d_run_async.
extern "C" int sift3d_run_async_after(sift3d_handle c, sift3d_handle after) {
	if (!c) return SIFT3D_ERR_ARG;
	if (after && after != c && after->pending && after->device == c->device) c->gate = after->ev[4];
	const int rc = sift3d_run_async(c);
	c->gate = nullptr;
	return rc;
}

extern "C" int sift3d_wait(sift3d_handle c) {
	if (!c) return SIFT3D_ERR_ARG;
	if (!c->pending) return SIFT3D_OK;  // nothing in flight (a blocking run has completed already)
	c->pending = false;
	const int rc = run_complete(c, 5, false);
	if (rc) c->dsplit_dirty = true;
	return rc;
}

extern "C" int sift3d_run_stages(sift3d_handle c, int upto) {
	if (!c) return SIFT3D_ERR_ARG;
	return run_impl(c, upto);
}

extern "C" int sift3d_stage_times(sift3d_handle c, double t[8]) {
	if (!c || !t) return SIFT3D_ERR_ARG;
	if (c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	if (c->stage < 1) return SIFT3D_ERR_STATE;
	memcpy(t, c->times, sizeof(double) * 8);
	return SIFT3D_OK;
}

extern "C" int sift3d_num_keypoints(sift3d_handle c, int *n) {
	if (!c || !n) return SIFT3D_ERR_ARG;
	if (c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	*n = (c->stage >= 4) ? (int)c->n_kp : 0;  // GetKeypoints before KpSiftAlgorithm returns empty
	return SIFT3D_OK;
}

extern "C" int sift3d_get_keypoints(sift3d_handle c, sift3d_keypoint *out, float *desc) {
	if (!c) return SIFT3D_ERR_ARG;
	if (c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	if (c->stage < 4 || c->n_kp == 0) return SIFT3D_OK;
	int rc = set_device(c->device);
	if (rc) return rc;
	if (desc && c->stage < 5) return SIFT3D_ERR_STATE;
	// results were complete when the run returned; the copies go through the pinned staging pool on the handle's own stream
	D2HSeg sg[2];
	int ns = 0;
	if (out) sg[ns++] = D2HSeg{out, c->d_kpout, sizeof(sift3d_keypoint) * (size_t)c->n_kp};
	if (desc) sg[ns++] = D2HSeg{desc, c->d_desc, sizeof(float) * kDesc * (size_t)c->n_kp};
	return staged_d2h_v(sg, ns, c->device, c->own_stream);
}

extern "C" int sift3d_device_results(sift3d_handle c, const float **d_desc, const float **d_xyz, int *n) {
	if (!c) return SIFT3D_ERR_ARG;
	if (c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	if (c->stage < 5) return SIFT3D_ERR_STATE;
	if (d_desc) *d_desc = c->d_desc;
	if (d_xyz) *d_xyz = c->d_xyz;
	if (n) *n = (int)c->n_kp;
	return SIFT3D_OK;
}

extern "C" int sift3d_match_handles(sift3d_handle ref, sift3d_handle tar, double thresHold, int mode, int *gIdx, int *sIdx, float *gDist,
                                    float *sDist, float *pairs6, int *npairs, double *seconds) {
	if (!ref || !tar) return SIFT3D_ERR_ARG;
	int rc;
	if (ref->pending && (rc = sift3d_wait(ref)) != SIFT3D_OK) return rc;
	if (tar->pending && (rc = sift3d_wait(tar)) != SIFT3D_OK) return rc;
	if (ref->stage < 5 || tar->stage < 5) { set_last_error("sift3d_match_handles: both extractors must have run"); return SIFT3D_ERR_STATE; }
	const int n = (int)ref->n_kp, m = (int)tar->n_kp;
	const float *td = tar->d_desc, *tx = tar->d_xyz;
	if ((tar->device != ref->device || hook(SIFT3D_HOOK_PEER_COPY)) && m > 0) {
		// the target's results live on another GPU: one peer-to-peer copy (xGMI) into a scratch on the reference's device
		if ((rc = set_device(ref->device)) != SIFT3D_OK) return rc;
		const size_t need = (size_t)m * (kDesc + 3);
		if (need > ref->peer_floats) {
			if (ref->d_peer) (void)hipFree(ref->d_peer);
			ref->d_peer = nullptr; ref->peer_floats = 0;
			S3D_HIP(hipMalloc(&ref->d_peer, sizeof(float) * (need + need / 4)));
			ref->peer_floats = need + need / 4;
		}
		S3D_HIP(hipMemcpyPeer(ref->d_peer, ref->device, tar->d_desc, tar->device, sizeof(float) * (size_t)m * kDesc));
		S3D_HIP(hipMemcpyPeer(ref->d_peer + (size_t)m * kDesc, ref->device, tar->d_xyz, tar->device, sizeof(float) * (size_t)m * 3));
		td = ref->d_peer; tx = ref->d_peer + (size_t)m * kDesc;
	}
	return sift3d_match(ref->d_desc, ref->d_xyz, n, td, tx, m, thresHold, mode, 1, ref->device, gIdx, sIdx, gDist, sDist, pairs6, npairs, seconds);
}

extern "C" int sift3d_num_octaves(sift3d_handle c, int *n) {
	if (!c || !n) return SIFT3D_ERR_ARG;
	*n = c->noct;
	return SIFT3D_OK;
}

static const Level *pick_level(sift3d_ctx *c, int is_dog, int idx) {
	const std::vector<Level> &P = is_dog ? c->dog : c->gss;
	if (idx < 0 || (size_t)idx >= P.size()) return nullptr;
	return &P[idx];
}

extern "C" int sift3d_level_info(sift3d_handle c, int is_dog, int idx, int dims3[3], float units3[3], float *scale) {
	if (!c) return SIFT3D_ERR_ARG;
	const Level *L = pick_level(c, is_dog, idx);
	if (!L) return SIFT3D_ERR_ARG;
	if (dims3) { dims3[0] = L->nx; dims3[1] = L->ny; dims3[2] = L->nz; }
	if (units3) units3[0] = units3[1] = units3[2] = L->unit;
	if (scale) *scale = L->scale;
	return SIFT3D_OK;
}

extern "C" int sift3d_copy_level(sift3d_handle c, int is_dog, int idx, float *out) {
	if (!c || !out) return SIFT3D_ERR_ARG;
	if (c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	if (c->stage < 1) return SIFT3D_ERR_STATE;
	const Level *L = pick_level(c, is_dog, idx);
	if (!L) return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	if (c->g_last_elide && c->slab) {
		// z-slab context: build the owned planes of the last Gaussian level on request (G[nd-1] holds its halo)
		const int i = is_dog ? idx % c->nd : idx % c->ng;
		if (((is_dog && i == c->nd - 1) || (!is_dog && i == c->ng - 1)) && (c->g_last_built.empty() || !c->g_last_built[0])) {
			const Level &G = c->gss[c->ng - 1];
			if (!launch_march_level(c->gss[c->ng - 2].d, G.d, nullptr, nullptr, G.nx, G.ny, G.zr(c->own0 - G.zoff, c->own1 - G.zoff), c->taps[c->ng - 1],
			                        c->stream)) return SIFT3D_ERR_STATE;
			S3D_HIP(hipStreamSynchronize(c->stream));
			c->g_last_built.assign(1, 1);
		}
	}
	if (c->g_last_elide && !c->slab) {
		// the last Gaussian level of the octave (and the DoG level behind it) was never built: build it now, once
		const int o = is_dog ? idx / c->nd : idx / c->ng, i = is_dog ? idx % c->nd : idx % c->ng;
		if (((is_dog && i == c->nd - 1) || (!is_dog && i == c->ng - 1)) && !c->g_last_built[(size_t)o]) {
			const Level &G = c->gss[(size_t)o * c->ng + c->ng - 1], &P = c->gss[(size_t)o * c->ng + c->ng - 2];
			smooth_level(c, o, P.d, G, c->taps[c->ng - 1], nullptr, nullptr, nullptr, c->ng - 1);
			S3D_HIP(hipStreamSynchronize(c->ostream[o]));
			c->g_last_built[(size_t)o] = 1;
		}
	}
	if (is_dog && c->dog_elide && (idx % c->nd == 0 || idx % c->nd == c->nd - 1)) {
		// an elided DoG level (never written by the pipeline): form it now, exactly like Sub, into its arena slot
		const int o = idx / c->nd, i = idx % c->nd;
		launch_dog_from_gss(c->gss[(size_t)o * c->ng + i + 1].d, c->gss[(size_t)o * c->ng + i].d, L->d, L->n(), c->stream);
		S3D_HIP(hipStreamSynchronize(c->stream));
	}
	return staged_d2h(out, L->d, sizeof(float) * L->n(), c->device, c->own_stream);
}

extern "C" int sift3d_copy_input(sift3d_handle c, float *out) {
	if (!c || !out) return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	return staged_d2h(out, c->in.d, sizeof(float) * c->in.n(), c->device, c->own_stream);
}

extern "C" int sift3d_num_extrema(sift3d_handle c, int *n) {
	if (!c || !n) return SIFT3D_ERR_ARG;
	if (c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	*n = c->stage >= 3 ? (int)c->n_ext : 0;
	return SIFT3D_OK;
}

static int fetch_ext(sift3d_ctx *c, std::vector<DevKp> &h) {
	int rc = set_device(c->device);
	if (rc) return rc;
	h.resize(c->n_ext);
	if (c->n_ext) S3D_HIP(hipMemcpy(h.data(), c->d_ext, sizeof(DevKp) * (size_t)c->n_ext, hipMemcpyDeviceToHost));
	return SIFT3D_OK;
}

extern "C" int sift3d_get_extrema(sift3d_handle c, sift3d_keypoint *out) {
	if (!c || !out) return SIFT3D_ERR_ARG;
	if (c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	if (c->stage < 3) return SIFT3D_ERR_STATE;
	std::vector<DevKp> h;
	int rc = fetch_ext(c, h);
	if (rc) return rc;
	for (size_t i = 0; i < h.size(); i++) {
		sift3d_keypoint &o = out[i];
		memset(&o, 0, sizeof(o));
		o.x = (float)h[i].x; o.y = (float)h[i].y; o.z = (float)h[i].z;
		o.scale = h[i].scale; o.octave = h[i].octave; o.level = h[i].level;
		o.rx = o.ry = o.rz = -1.0f;
		memcpy(o.win, h[i].win, sizeof(o.win));
		memcpy(o.eigvalue, h[i].eigvalue, sizeof(o.eigvalue));
		memcpy(o.eigvector, h[i].eigvector, sizeof(o.eigvector));
		memcpy(o.Rotation, h[i].rot, sizeof(o.Rotation));
		memcpy(o.str_tensor, h[i].st, sizeof(o.str_tensor));
	}
	return SIFT3D_OK;
}

extern "C" int sift3d_get_orientation_codes(sift3d_handle c, int *codes) {
	if (!c || !codes) return SIFT3D_ERR_ARG;
	if (c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	if (c->stage < 4) return SIFT3D_ERR_STATE;
	std::vector<DevKp> h;
	int rc = fetch_ext(c, h);
	if (rc) return rc;
	for (size_t i = 0; i < h.size(); i++) codes[i] = h[i].code;
	return SIFT3D_OK;
}

extern "C" int sift3d_gaussian_smooth(const float *src, int nx, int ny, int nz, float sigma, float *dst, int device) {
	if (!src || !dst || nx <= 0 || ny <= 0 || nz <= 0) return SIFT3D_ERR_ARG;
	int rc = set_device(device);
	if (rc) return rc;
	Taps t;
	if (!build_taps(sigma, t)) { set_last_error("kernel too wide"); return SIFT3D_ERR_ARG; }
	const size_t n = (size_t)nx * ny * nz;
	float *d = nullptr;
	S3D_HIP(hipMalloc(&d, sizeof(float) * n * 3));
	hipError_t e = hipMemcpy(d, src, sizeof(float) * n, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		launch_conv_axis(0, d, d + n, nx, ny, nz, t, nullptr, nullptr, nullptr, nullptr);
		launch_conv_axis(1, d + n, d + 2 * n, nx, ny, nz, t, nullptr, nullptr, nullptr, nullptr);
		launch_conv_axis(2, d + 2 * n, d, nx, ny, nz, t, nullptr, nullptr, nullptr, nullptr);
		e = hipDeviceSynchronize();
	}
	if (e == hipSuccess) e = hipMemcpy(dst, d, sizeof(float) * n, hipMemcpyDeviceToHost);
	hipFree(d);
	if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return SIFT3D_ERR_HIP; }
	return SIFT3D_OK;
}


// GaussianSmooth_3D_Imp (Src/cSIFT3D.cc:624-788): ONE pass along `dim` with the caller's taps (width odd: the reference reads
// weight[0 .. 2 (width / 2)]), interior and boundary rule of the pipeline's generic pass (k_conv_axis)
extern "C" int sift3d_conv_axis(const float *src, int nx, int ny, int nz, int dim, const float *weight, int width, float *dst, int device) {
	if (!src || !dst || !weight || nx <= 0 || ny <= 0 || nz <= 0 || dim < 0 || dim > 2) return SIFT3D_ERR_ARG;
	if (width < 1 || !(width & 1) || width / 2 > kMaxHW) { set_last_error("sift3d_conv_axis: the kernel width must be odd and at most 2 * 64 + 1"); return SIFT3D_ERR_ARG; }
	int rc = set_device(device);
	if (rc) return rc;
	Taps t;
	t.hw = width / 2;
	for (int i = 0; i < kMaxTaps; i++) t.w[i] = i < width ? weight[i] : 0.f;
	const size_t n = (size_t)nx * ny * nz;
	float *d = nullptr;
	S3D_HIP(hipMalloc(&d, sizeof(float) * n * 2));
	hipError_t e = hipMemcpy(d, src, sizeof(float) * n, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		launch_conv_axis(dim, d, d + n, nx, ny, nz, t, nullptr, nullptr, nullptr, nullptr);
		e = hipDeviceSynchronize();
	}
	if (e == hipSuccess) e = hipMemcpy(dst, d + n, sizeof(float) * n, hipMemcpyDeviceToHost);
	hipFree(d);
	if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return SIFT3D_ERR_HIP; }
	return SIFT3D_OK;
}

// ---- one keypoint on a caller-provided level: Assign_Orientation_Imp / Extract_Descriptor_Imp (Src/cSIFT3D.cc:913-1138, 1152-1381) as
// free functions (Include/cSIFT3D.h:224, 228).  The pipeline's own kernels run on the BOX of the level the window reaches -- the clipped
// window bounds of Src/cSIFT3D.cc:939-955 / 1184-1200 plus the plane either side the central differences read -- with the keypoint's
// coordinates shifted into it: the kernels' own clipping of a window to [1, n - 2] of that box gives the same voxel set as the
// reference's on the whole level, and nothing else of the level is read.  The box lives in level slot 1 of a small one-octave context that
// is kept for the next call (a loop over keypoints, as the reference's callers run, pays for it once); its tables are rebuilt when sigma / scale /
// unit change.  The keypoint must sit on a voxel and the level's unit must be a power of two, as in the pipeline: anything else is refused.
namespace {
struct OneKp {
	sift3d_ctx *c = nullptr;
	int device = -1, octave_base = -1, edge = 0;
	float ori_sigma = -1.f, scale = -1.f;
};
std::mutex g_onekp_mu;
OneKp g_onekp;  // (never destroyed at exit: the HIP runtime may be gone by then)

struct Box { int lo[3], n[3]; };  // first voxel of the box in the level, box dimensions

// window bounds like win_bounds (kernels_orient.hip / kernels_desc.hip), then one voxel either side
bool window_box(const int c[3], const int dims[3], float radius, float unit, Box &b) {
	for (int a = 0; a < 3; a++) {
		if (dims[a] < 3) return false;
		const int s = (int)floorf((float)c[a] - radius / unit), e = (int)ceilf((float)c[a] + radius / unit);
		const int lo = s > 1 ? s : 1, hi = e < dims[a] - 2 ? e : dims[a] - 2;
		if (hi < lo) return false;
		b.lo[a] = lo - 1; b.n[a] = hi - lo + 3;
	}
	return true;
}

int onekp_prepare(OneKp &K, int device, float unit, int edge, float ori_sigma, float scale) {
	int ex = 0;
	const float m = frexpf(unit, &ex);
	if (!(unit >= 1.0f) || m != 0.5f || ex - 1 > 20) { set_last_error("the level's unit must be a power of two >= 1 (2^octave)"); return SIFT3D_ERR_ARG; }
	const int ob = ex - 1;
	if (!K.c || K.device != device || K.octave_base != ob || K.edge < edge) {
		if (K.c) { sift3d_destroy(K.c); K = OneKp(); }
		CreateCfg cfg;
		const int e = std::max(96, (edge + 31) & ~31);
		cfg.nx = cfg.ny = cfg.nz = e; cfg.octave_base = ob; cfg.noct_total = ob + 1; cfg.seeded = true;
		sift3d_ctx *c = nullptr;
		int rc = create_common(&c, cfg, nullptr, device);
		if (rc) return rc;
		K.c = c; K.device = device; K.octave_base = ob; K.edge = e;
	}
	if (K.ori_sigma != ori_sigma || K.scale != scale) {
		sift3d_ctx *c = K.c;
		std::vector<WinLut> luts = blank_luts(c);
		std::vector<float> pool;
		c->desc_lut_lds = true;
		const size_t at = ((size_t)K.octave_base * 8 + 1) * 2;
		(void)append_lut(pool, luts[at], 0, ori_sigma, ori_sigma * 3.0f, unit, scale);
		const float dsig = scale * 7.071067812f;
		if (!append_lut(pool, luts[at + 1], 1, dsig, 2.0f * dsig, unit, scale)) c->desc_lut_lds = false;
		S3D_HIP(hipStreamSynchronize(c->stream));
		int rc = upload_luts(c, luts, pool);
		if (rc) return rc;
		K.ori_sigma = ori_sigma; K.scale = scale;
	}
	return SIFT3D_OK;
}

// the box -> level slot 1, its dimensions -> the level table, the record -> extremum 0
int onekp_load(OneKp &K, const float *level, int nx, int ny, float unit, const Box &b, const DevKp &rec) {
	sift3d_ctx *c = K.c;
	hipStream_t st = c->stream;
	std::vector<float> box((size_t)b.n[0] * b.n[1] * b.n[2]);
	for (int z = 0; z < b.n[2]; z++)
		for (int y = 0; y < b.n[1]; y++)
			memcpy(&box[((size_t)z * b.n[1] + y) * b.n[0]], level + ((size_t)(b.lo[2] + z) * ny + (size_t)(b.lo[1] + y)) * nx + b.lo[0], sizeof(float) * b.n[0]);
	Level &L = c->gss[1];
	const LevelRef ref{L.d, b.n[0], b.n[1], b.n[2], unit, 0};
	const unsigned words[3] = {1u, 0u, 0u};  // extrema, overflow flag, keypoints
	const int code = rec.code;
	S3D_HIP(hipMemcpyAsync(L.d, box.data(), sizeof(float) * box.size(), hipMemcpyHostToDevice, st));
	S3D_HIP(hipMemcpyAsync(c->d_levels + ((size_t)K.octave_base * 8 + 1), &ref, sizeof(ref), hipMemcpyHostToDevice, st));
	S3D_HIP(hipMemcpyAsync(c->d_ext, &rec, sizeof(rec), hipMemcpyHostToDevice, st));
	S3D_HIP(hipMemcpyAsync(c->d_codes, &code, sizeof(int), hipMemcpyHostToDevice, st));
	S3D_HIP(hipMemcpyAsync(c->d_total, words, sizeof(words), hipMemcpyHostToDevice, st));
	S3D_HIP(hipStreamSynchronize(st));  // (the sources are pageable host memory of this frame)
	return SIFT3D_OK;
}

int onekp_check(const float *level, int nx, int ny, int nz, const sift3d_keypoint *kp, int c[3]) {
	if (!level || !kp || nx < 3 || ny < 3 || nz < 3) { set_last_error("bad level / keypoint"); return SIFT3D_ERR_ARG; }
	const float f[3] = {kp->x, kp->y, kp->z};
	const int dims[3] = {nx, ny, nz};
	for (int a = 0; a < 3; a++) {
		c[a] = (int)f[a];
		if ((float)c[a] != f[a] || c[a] < 0 || c[a] >= dims[a]) { set_last_error("the keypoint must sit on a voxel of the level (integral x, y, z inside it)"); return SIFT3D_ERR_ARG; }
	}
	if (!(kp->scale > 0.0f)) { set_last_error("keypoint scale must be positive"); return SIFT3D_ERR_ARG; }
	return SIFT3D_OK;
}
}  // namespace

// kp in: x, y, z (voxel of the level), scale; out: win, eigvalue, eigvector, Rotation (as Assign_Orientation_Imp leaves it: not
// transposed), str_tensor (computed from zero: the reference accumulates into what Initialize_Keypoint zeroed).  *code: the reference's
// return value (1 accepted, -1 weak gradient, -2 eigenvalue ratio / not distinct, -3 corner).
extern "C" int sift3d_orient_keypoint(const float *level, int nx, int ny, int nz, float unit, sift3d_keypoint *kp, float sigma, float max_eig_ratio,
                                      float corner_thresh, int device, int *code) {
	int ctr[3];
	int rc = onekp_check(level, nx, ny, nz, kp, ctr);
	if (rc) return rc;
	if (!code || !(sigma > 0.0f)) { set_last_error("sift3d_orient_keypoint: bad argument"); return SIFT3D_ERR_ARG; }
	if ((rc = set_device(device)) != SIFT3D_OK) return rc;
	const int dims[3] = {nx, ny, nz};
	Box b;
	if (!window_box(ctr, dims, sigma * 3.0f, unit, b)) { set_last_error("the orientation window is empty"); return SIFT3D_ERR_ARG; }
	std::lock_guard<std::mutex> lk(g_onekp_mu);
	OneKp &K = g_onekp;
	if ((rc = onekp_prepare(K, device, unit, std::max(b.n[0], std::max(b.n[1], b.n[2])), sigma, kp->scale)) != SIFT3D_OK) return rc;
	sift3d_ctx *c = K.c;
	DevKp rec;
	memset(&rec, 0, sizeof(rec));
	rec.x = ctr[0] - b.lo[0]; rec.y = ctr[1] - b.lo[1]; rec.z = ctr[2] - b.lo[2];
	rec.octave = K.octave_base; rec.level = 1; rec.scale = kp->scale; rec.slot = -1;
	if ((rc = onekp_load(K, level, nx, ny, unit, b, rec)) != SIFT3D_OK) return rc;
	hipStream_t st = c->stream;
	launch_orient(c->d_ext, c->d_codes, c->d_total, c->ext_cap, c->d_levels, c->d_luts, c->d_lutpool, max_eig_ratio, corner_thresh, 0, 1, c->d_order,
	              c->d_nkp + 3, st);
	S3D_HIP(hipMemcpyAsync(&rec, c->d_ext, sizeof(rec), hipMemcpyDeviceToHost, st));
	S3D_HIP(hipStreamSynchronize(st));
	S3D_HIP(hipGetLastError());
	*code = rec.code;
	for (int i = 0; i < 3; i++) { kp->win[i] = rec.win[i]; kp->eigvalue[i] = rec.eigvalue[i]; }
	for (int i = 0; i < 9; i++) { kp->eigvector[i] = rec.eigvector[i]; kp->Rotation[i] = rec.rot[i]; kp->str_tensor[i] = rec.st[i]; }
	return SIFT3D_OK;
}

// kp in: x, y, z, scale, Rotation (as the orientation stage leaves it), str_tensor (first guess of the histogram's fixed-point unit only);
// out: Rotation TRANSPOSED (Src/cSIFT3D.cc:1214 inverts it in place), desc768 = the normalised descriptor (cc:1350-1358)
extern "C" int sift3d_describe_keypoint(const float *level, int nx, int ny, int nz, float unit, sift3d_keypoint *kp, float *desc768, int device) {
	int ctr[3];
	int rc = onekp_check(level, nx, ny, nz, kp, ctr);
	if (rc) return rc;
	if (!desc768) return SIFT3D_ERR_ARG;
	if ((rc = set_device(device)) != SIFT3D_OK) return rc;
	const int dims[3] = {nx, ny, nz};
	const float dsig = kp->scale * 7.071067812f;
	Box b;
	if (!window_box(ctr, dims, 2.0f * dsig, unit, b)) { set_last_error("the descriptor window is empty"); return SIFT3D_ERR_ARG; }
	std::lock_guard<std::mutex> lk(g_onekp_mu);
	OneKp &K = g_onekp;
	// the orientation table of the pair is ALWAYS the pipeline's for this scale (sigma = 1.5 scale): its weight sum enters the first guess of the
	// fixed-point unit, so a table left behind by an earlier sift3d_orient_keypoint with another sigma would make the descriptor's low bits
	// depend on the call history; onekp_prepare rebuilds the pair when it differs
	if ((rc = onekp_prepare(K, device, unit, std::max(b.n[0], std::max(b.n[1], b.n[2])), 1.5f * kp->scale, kp->scale)) != SIFT3D_OK)
		return rc;
	sift3d_ctx *c = K.c;
	DevKp rec;
	memset(&rec, 0, sizeof(rec));
	rec.x = ctr[0] - b.lo[0]; rec.y = ctr[1] - b.lo[1]; rec.z = ctr[2] - b.lo[2];
	rec.octave = K.octave_base; rec.level = 1; rec.scale = kp->scale; rec.code = 1; rec.slot = -1;
	for (int i = 0; i < 3; i++) { rec.win[i] = kp->win[i]; rec.eigvalue[i] = kp->eigvalue[i]; }
	for (int i = 0; i < 9; i++) { rec.eigvector[i] = kp->eigvector[i]; rec.rot[i] = kp->Rotation[i]; rec.st[i] = kp->str_tensor[i]; }
	if ((rc = onekp_load(K, level, nx, ny, unit, b, rec)) != SIFT3D_OK) return rc;
	hipStream_t st = c->stream;
	launch_slots(c->d_ext, c->d_codes, c->d_total, c->ext_cap, c->d_nkp, c->d_order, c->kp_cap, c->d_slots_part, st);
	launch_describe(c->d_ext, c->d_total, c->ext_cap, c->d_levels, c->d_luts, c->d_lutpool, c->d_desc, c->kp_cap, 0, 1, c->d_order, c->d_nkp, c->d_nkp + 1, st,
	                c->desc_lut_lds, &c->dsplit);
	launch_finalize(c->d_ext, c->d_total, c->ext_cap, 1, c->d_kpout, c->d_xyz, c->kp_cap, st);
	sift3d_keypoint out;
	S3D_HIP(hipMemcpyAsync(desc768, c->d_desc, sizeof(float) * kDesc, hipMemcpyDeviceToHost, st));
	S3D_HIP(hipMemcpyAsync(&out, c->d_kpout, sizeof(out), hipMemcpyDeviceToHost, st));
	S3D_HIP(hipStreamSynchronize(st));
	S3D_HIP(hipGetLastError());
	for (int i = 0; i < 9; i++) kp->Rotation[i] = out.Rotation[i];
	return SIFT3D_OK;
}

extern "C" int sift3d_downsample(const float *src, int snx, int sny, int snz, float *dst, int nx, int ny, int nz, int device) {
	if (!src || !dst || nx <= 0 || ny <= 0 || nz <= 0 || snx <= 0 || sny <= 0 || snz <= 0 || 2 * (nx - 1) >= snx || 2 * (ny - 1) >= sny ||
	    2 * (nz - 1) >= snz)
		return SIFT3D_ERR_ARG;
	int rc = set_device(device);
	if (rc) return rc;
	const size_t ns = (size_t)snx * sny * snz, nd = (size_t)nx * ny * nz;
	float *d = nullptr;
	S3D_HIP(hipMalloc(&d, sizeof(float) * (ns + nd)));
	hipError_t e = hipMemcpy(d, src, sizeof(float) * ns, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		launch_downsample(d, snx, sny, d + ns, nx, ny, nz, nullptr);
		e = hipDeviceSynchronize();
	}
	if (e == hipSuccess) e = hipMemcpy(dst, d + ns, sizeof(float) * nd, hipMemcpyDeviceToHost);
	hipFree(d);
	if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return SIFT3D_ERR_HIP; }
	return SIFT3D_OK;
}

extern "C" int sift3d_dog_sub(const float *prev, const float *cur, size_t n, float *dog, int device) {
	if (!prev || !cur || !dog || n == 0) return SIFT3D_ERR_ARG;
	int rc = set_device(device);
	if (rc) return rc;
	float *d = nullptr;
	S3D_HIP(hipMalloc(&d, sizeof(float) * n * 3));
	hipError_t e = hipMemcpy(d, prev, sizeof(float) * n, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(d + n, cur, sizeof(float) * n, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		launch_dog_from_gss(d + n, d, d + 2 * n, n, nullptr);  // (cur - prev) * (-1)
		e = hipDeviceSynchronize();
	}
	if (e == hipSuccess) e = hipMemcpy(dog, d + 2 * n, sizeof(float) * n, hipMemcpyDeviceToHost);
	hipFree(d);
	if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return SIFT3D_ERR_HIP; }
	return SIFT3D_OK;
}

// best-of-iters bandwidth (GB/s, read + write) of a float4 device-to-device copy of `bytes` bytes: the achievable HBM ceiling
// bench.py reports beside the 8 TB/s spec peak
extern "C" int sift3d_debug_copy_bandwidth(size_t bytes, int iters, int device, double *gbs) {
	if (!gbs || bytes < 4096 || iters < 1) return SIFT3D_ERR_ARG;
	int rc = set_device(device);
	if (rc) return rc;
	const size_t nf = (bytes / 16) * 4;
	float *d = nullptr;
	S3D_HIP(hipMalloc(&d, sizeof(float) * nf * 2));
	hipEvent_t e0 = nullptr, e1 = nullptr;
	hipError_t e = hipMemset(d, 0, sizeof(float) * nf * 2);
	if (e == hipSuccess) e = hipEventCreate(&e0);
	if (e == hipSuccess) e = hipEventCreate(&e1);
	double best = 0.0;
	for (int i = 0; i < iters + 1 && e == hipSuccess; i++) {  // first pass warms up
		(void)hipEventRecord(e0, nullptr);
		launch_copy16(d, d + nf, nf, nullptr);
		(void)hipEventRecord(e1, nullptr);
		e = hipEventSynchronize(e1);
		float ms = 0;
		if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
		if (i > 0 && ms > 0) best = std::max(best, 2.0 * (double)nf * 4.0 / ((double)ms * 1e-3) / 1e9);
	}
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	(void)hipFree(d);
	if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return SIFT3D_ERR_HIP; }
	*gbs = best;
	return SIFT3D_OK;
}

extern "C" int sift3d_debug_counters(sift3d_handle c, int out[4]) {
	if (!out) return SIFT3D_ERR_ARG;
	if (c && c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	out[0] = c ? c->n_regrow : 0;
	out[1] = c ? c->n_desc_redo : 0;
	out[2] = match_redo_rows();
	out[3] = 0;
	return SIFT3D_OK;
}

extern "C" int sift3d_debug_face_lookup(const float *grad3, int n, int route, int *face, float *bary3, int device) {
	if (!grad3 || !face || !bary3 || n < 0) return SIFT3D_ERR_ARG;
	int rc = set_device(device);
	if (rc) return rc;
	if (n == 0) return SIFT3D_OK;
	FaceConst faces[kFaces];
	build_faces(faces);
	FaceSym sym;
	if (!build_facesym(faces, &sym)) { set_last_error("icosahedron symmetry table: no matching face"); return SIFT3D_ERR_STATE; }
	S3D_HIP(upload_faces(faces, &sym));
	float *d = nullptr;
	S3D_HIP(hipMalloc(&d, sizeof(float) * (size_t)n * 7));
	int *d_face = reinterpret_cast<int *>(d + (size_t)n * 6);
	hipError_t e = hipMemcpy(d, grad3, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		launch_face_lookup(d, n, route, d_face, d + (size_t)n * 3, nullptr);
		e = hipDeviceSynchronize();
	}
	if (e == hipSuccess) e = hipMemcpy(bary3, d + (size_t)n * 3, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost);
	if (e == hipSuccess) e = hipMemcpy(face, d_face, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return SIFT3D_ERR_HIP; }
	return SIFT3D_OK;
}


// =============================================================================================================
// Multi-GPU sharding entry points (include/sift3d_hip.h): seeded tail contexts and z-slab contexts of octave 0
// =============================================================================================================
extern "C" int sift3d_create_seeded(sift3d_handle *out, int nx, int ny, int nz, int octave_base, int noct_total,
                                    const sift3d_params *params, int device) {
	if (!out || octave_base < 0 || octave_base > 20) { set_last_error("sift3d_create_seeded: bad argument"); return SIFT3D_ERR_ARG; }
	CreateCfg cfg;
	cfg.nx = nx; cfg.ny = ny; cfg.nz = nz; cfg.octave_base = octave_base; cfg.noct_total = noct_total; cfg.seeded = true;
	return create_common(out, cfg, params, device);
}

extern "C" int sift3d_seed_upload(sift3d_handle c, const float *level0, int on_device) {
	if (!c || !level0 || !c->seeded) return SIFT3D_ERR_ARG;
	if (c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	if (c->noct <= 0) return SIFT3D_OK;
	int rc = set_device(c->device);
	if (rc) return rc;
	const Level &L = c->gss[0];
	S3D_HIP(hipMemcpyAsync(L.d, level0, sizeof(float) * L.n(), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
	S3D_HIP(hipStreamSynchronize(c->stream));
	return SIFT3D_OK;
}

extern "C" int sift3d_set_describe_partition(sift3d_handle c, int rank, int world) {
	if (!c || world < 1 || rank < 0 || rank >= world) return SIFT3D_ERR_ARG;
	c->part_rank = rank; c->part_world = world;
	return SIFT3D_OK;
}

// planes a keypoint's descriptor window can reach along z in octave 0 (Src/cSIFT3D.cc:1155-1156, 1276-1290: window
// [floor(c-r), ceil(c+r)] plus the central-difference neighbours), and never less than the widest Gaussian + 1
extern "C" int sift3d_slab_min_halo(const sift3d_params *params, int *halo) {
	if (!halo) return SIFT3D_ERR_ARG;
	sift3d_params p;
	if (params) p = *params; else sift3d_default_params(&p);
	if (p.num_kp_levels < 1 || p.num_kp_levels > 5) return SIFT3D_ERR_ARG;
	const double sigma0 = (double)p.sigma_default * pow(2.0, -1.0 / 3.0);
	const float scale = (float)(pow(2.0, (double)p.num_kp_levels / (double)p.num_kp_levels) * sigma0);  // DoG level num_kp_levels, octave 0
	const float radius = 2.0f * (scale * 7.071067812f);
	int h = (int)ceilf(radius) + 2;
	std::vector<float> sig;
	float base_sigma;
	level_sigmas(p, sig, base_sigma);
	for (size_t i = 1; i < sig.size(); i++) {
		Taps t;
		if (!build_taps(sig[i], t)) return SIFT3D_ERR_ARG;
		h = std::max(h, t.hw + 1);
	}
	*halo = h;
	return SIFT3D_OK;
}

// D2D copy of the results into caller-owned device buffers (n*768 and n*3 floats), so that a communication layer that
// only knows its own allocations can reduce / gather them
extern "C" int sift3d_export_device(sift3d_handle c, float *d_desc_dst, float *d_xyz_dst) {
	if (!c) return SIFT3D_ERR_ARG;
	if (c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	if (c->stage < 5) return SIFT3D_ERR_STATE;
	int rc = set_device(c->device);
	if (rc) return rc;
	if (d_desc_dst && c->n_kp) S3D_HIP(hipMemcpyAsync(d_desc_dst, c->d_desc, sizeof(float) * kDesc * (size_t)c->n_kp, hipMemcpyDeviceToDevice, c->stream));
	if (d_xyz_dst && c->n_kp) S3D_HIP(hipMemcpyAsync(d_xyz_dst, c->d_xyz, sizeof(float) * 3 * (size_t)c->n_kp, hipMemcpyDeviceToDevice, c->stream));
	S3D_HIP(hipStreamSynchronize(c->stream));
	return SIFT3D_OK;
}

// the inverse: overwrite the descriptors of this handle with rows reduced elsewhere (after the all-reduce of a
// partitioned describe), so that sift3d_get_keypoints / sift3d_device_results / sift3d_match see complete rows
extern "C" int sift3d_import_descriptors_device(sift3d_handle c, const float *d_desc_src) {
	if (!c || !d_desc_src) return SIFT3D_ERR_ARG;
	if (c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	if (c->stage < 5) return SIFT3D_ERR_STATE;
	int rc = set_device(c->device);
	if (rc) return rc;
	if (c->n_kp) S3D_HIP(hipMemcpyAsync(c->d_desc, d_desc_src, sizeof(float) * kDesc * (size_t)c->n_kp, hipMemcpyDeviceToDevice, c->stream));
	S3D_HIP(hipStreamSynchronize(c->stream));
	return SIFT3D_OK;
}

// ---- partitioned orientation of a replicated context: each rank orients the extrema k with k % world == rank, the
// caller all-reduces (integer SUM) the packed rows and hands them back, then every rank describes its share of slots
extern "C" int sift3d_run_partial_orientation(sift3d_handle c) {
	if (!c) return SIFT3D_ERR_ARG;
	return run_impl(c, 4, true);
}

extern "C" int sift3d_export_orientation_device(sift3d_handle c, int *d_dst) {
	if (!c || !d_dst) return SIFT3D_ERR_ARG;
	if (c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	if (c->stage < 4) return SIFT3D_ERR_STATE;
	int rc = set_device(c->device);
	if (rc) return rc;
	launch_orient_pack(c->d_ext, c->d_total, c->ext_cap, d_dst, c->part_rank, c->part_world, c->stream);
	S3D_HIP(hipStreamSynchronize(c->stream));
	return SIFT3D_OK;
}

extern "C" int sift3d_import_orientation_device(sift3d_handle c, const int *d_src) {
	if (!c || !d_src) return SIFT3D_ERR_ARG;
	if (c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	if (c->stage < 4) return SIFT3D_ERR_STATE;
	int rc = set_device(c->device);
	if (rc) return rc;
	launch_orient_unpack(c->d_ext, c->d_codes, c->d_total, c->ext_cap, d_src, c->stream);
	S3D_HIP(hipStreamSynchronize(c->stream));
	return SIFT3D_OK;
}

// Extract_Description (Src/cSIFT3D.cc:484-502) on the current orientation results (after an import): slots, this
// handle's share of the descriptors, final records
extern "C" int sift3d_run_describe(sift3d_handle c) {
	if (!c) return SIFT3D_ERR_ARG;
	if (c->pending) { int wrc = sift3d_wait(c); if (wrc) return wrc; }  // an asynchronous run in flight is completed first
	if (c->stage < 4 || c->slab) return SIFT3D_ERR_STATE;
	int rc = set_device(c->device);
	if (rc) return rc;
	hipStream_t st = c->stream;
	S3D_HIP(hipEventRecord(c->ev[6], st));
	launch_slots(c->d_ext, c->d_codes, c->d_total, c->ext_cap, c->d_nkp, c->d_order, c->kp_cap, c->d_slots_part, st);
	launch_describe(c->d_ext, c->d_total, c->ext_cap, c->d_levels, c->d_luts, c->d_lutpool, c->d_desc, c->kp_cap, c->part_rank,
	                c->part_world, c->d_order, c->d_nkp, c->d_nkp + 1, st, c->desc_lut_lds, &c->dsplit);
	launch_finalize(c->d_ext, c->d_total, c->ext_cap, 1, c->d_kpout, c->d_xyz, c->kp_cap, st);
	S3D_HIP(hipEventRecord(c->ev[7], st));
	unsigned host_words[3] = {0, 0, 0};
	S3D_HIP(hipMemcpyAsync(host_words, c->d_total, sizeof(unsigned) * 3, hipMemcpyDeviceToHost, st));
	S3D_HIP(hipStreamSynchronize(st));
	S3D_HIP(hipGetLastError());
	c->n_kp = host_words[2];
	c->stage = 5;
	float ms = 0;
	hipEventElapsedTime(&ms, c->ev[6], c->ev[7]);
	c->times[6] = (double)ms * 1e-3;
	c->times[0] = c->times[2] + c->times[3] + c->times[4] + c->times[5] + c->times[6];
	return SIFT3D_OK;
}

static int slab_cfg(const sift3d_slab_desc *d, CreateCfg &cfg) {
	// even start so that DownSample_3D's plane 2k stays inside one slab; an odd end is only possible at the top of the volume
	if (!d || d->nx <= 0 || d->ny <= 0 || d->nz <= 0 || d->z0 < 0 || d->z1 > d->nz || d->z1 <= d->z0 || (d->z0 & 1) ||
	    ((d->z1 & 1) && d->z1 != d->nz) || d->halo < 1) {
		set_last_error("bad slab description (owned range must be non-empty, start and end on even planes, inside the volume)");
		return SIFT3D_ERR_ARG;
	}
	if (d->octave < 0 || d->octave > 20) { set_last_error("bad slab octave"); return SIFT3D_ERR_ARG; }
	cfg.nx = d->nx; cfg.ny = d->ny; cfg.nz = d->nz; cfg.noct_total = d->noct_total; cfg.slab = true;
	cfg.z0 = d->z0; cfg.z1 = d->z1; cfg.halo = d->halo;
	cfg.octave_base = d->octave; cfg.seeded = d->octave > 0;
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_arena_floats(const sift3d_slab_desc *d, const sift3d_params *params, size_t *n) {
	if (!n) return SIFT3D_ERR_ARG;
	CreateCfg cfg;
	int rc = slab_cfg(d, cfg);
	if (rc) return rc;
	sift3d_ctx tmp;  // geometry only, no device work
	if (params) tmp.p = *params; else sift3d_default_params(&tmp.p);
	tmp.nx = cfg.nx; tmp.ny = cfg.ny; tmp.nz = cfg.nz; tmp.slab = true; tmp.own0 = cfg.z0; tmp.own1 = cfg.z1; tmp.halo = cfg.halo;
	tmp.octave_base = cfg.octave_base; tmp.seeded = cfg.seeded;
	plan_pyramid(&tmp, cfg.noct_total);
	tmp.in.nx = cfg.nx; tmp.in.ny = cfg.ny; tmp.in.nz = cfg.nz; tmp.in.bz = cfg.seeded ? 1 : cfg.z1 - cfg.z0 + 2 * cfg.halo;
	*n = arena_floats_of(&tmp);
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_create(sift3d_handle *out, const sift3d_slab_desc *d, const sift3d_params *params, int device,
                                  float *d_arena, size_t arena_floats) {
	if (!out || !d_arena) return SIFT3D_ERR_ARG;
	CreateCfg cfg;
	int rc = slab_cfg(d, cfg);
	if (rc) return rc;
	cfg.ext_arena = d_arena; cfg.ext_arena_floats = arena_floats;
	rc = create_common(out, cfg, params, device);
	if (rc) return rc;
	sift3d_ctx *c = *out;
	if (c->noct < 1) { sift3d_destroy(c); *out = nullptr; set_last_error("volume too small for one octave"); return SIFT3D_ERR_ARG; }
	// the march kernel is the only slab-aware Gaussian: default half widths, planes of at least 32 x 32 voxels (and not 33 .. 31 + hw)
	for (int i = c->seeded ? 1 : 0; i < c->ng; i++) {
		const int hw = i == 0 ? c->base_taps.hw : c->taps[i].hw;
		const bool inst = hw >= 2 && hw <= 8;
		auto fits = [&](int n) { return n == 32 || n >= 32 + hw; };
		if (!inst || !fits(c->nx) || !fits(c->ny) || hw + 1 > c->halo) {
			sift3d_destroy(c); *out = nullptr;
			set_last_error("slab mode needs the fused level kernel (default sigma schedule, nx, ny >= 40) and halo > hw");
			return SIFT3D_ERR_ARG;
		}
	}
	// planes outside the volume (rank 0 below z = 0, last rank above nz-1) are never read; zero everything once so that
	// halo planes that are never exchanged hold defined values
	hipError_t e = hipMemsetAsync(c->arena, 0, sizeof(float) * c->arena_floats, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); sift3d_destroy(c); *out = nullptr; return SIFT3D_ERR_HIP; }
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_buffer(sift3d_handle c, int kind, int idx, size_t *offset_floats, int *planes, int *zoff) {
	if (!c || !c->slab) return SIFT3D_ERR_ARG;
	const Level *L = nullptr;
	if (kind == 0) L = &c->in;
	else if (kind == 1 && idx >= 0 && idx < c->ng) L = &c->gss[idx];
	else if (kind == 2 && idx >= 0 && idx < c->nd) L = &c->dog[idx];
	if (!L) return SIFT3D_ERR_ARG;
	if (offset_floats) *offset_floats = (size_t)(L->d - c->arena);
	if (planes) *planes = L->planes();
	if (zoff) *zoff = L->zoff;
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_upload(sift3d_handle c, const float *planes, int zg0, int zg1, int on_device) {
	if (!c || !c->slab || c->seeded || !planes || zg0 < c->in.zoff || zg1 > c->in.zoff + c->in.planes() || zg0 < 0 || zg1 > c->nz || zg1 <= zg0)
		return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	const size_t pl = (size_t)c->nx * c->ny;
	float *dst = c->in.d + pl * (size_t)(zg0 - c->in.zoff);
	if (on_device) S3D_HIP(hipMemcpyAsync(dst, planes, sizeof(float) * pl * (size_t)(zg1 - zg0), hipMemcpyDeviceToDevice, c->stream));
	else if ((rc = staged_h2d(dst, planes, sizeof(float) * pl * (size_t)(zg1 - zg0), c->device, c->stream)) != SIFT3D_OK) return rc;  // pageable host planes: pinned staging (r04)
	S3D_HIP(hipStreamSynchronize(c->stream));
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_input_absmax(sift3d_handle c, float *local_max) {
	if (!c || !c->slab || c->seeded || !local_max) return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	const size_t pl = (size_t)c->nx * c->ny;
	launch_absmax(c->in.d + pl * (size_t)(c->own0 - c->in.zoff), pl * (size_t)(c->own1 - c->own0), c->d_inmax, c->stream);
	unsigned bits = 0;
	S3D_HIP(hipMemcpyAsync(&bits, c->d_inmax, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
	S3D_HIP(hipStreamSynchronize(c->stream));
	memcpy(local_max, &bits, sizeof(float));
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_input_scale(sift3d_handle c, float global_max) {
	if (!c || !c->slab || c->seeded) return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	unsigned bits;
	memcpy(&bits, &global_max, sizeof(float));
	S3D_HIP(hipMemcpyAsync(c->d_inmax, &bits, sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
	launch_scale_by_max(c->in.d, c->in.n(), c->d_inmax, c->stream);  // halo planes included (0/max = 0 where never uploaded)
	S3D_HIP(hipStreamSynchronize(c->stream));
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_halo_planes(sift3d_handle c, int i, int *planes) {
	if (!c || !c->slab || !planes || i < 0 || i >= c->ng) return SIFT3D_ERR_ARG;
	int need = 0;
	// input reach of the next Gaussian level: its z-march loads planes p-hw-1 .. p+hw (the extra low plane feeds the
	// right-boundary lerp of the last planes of the volume, Src/cSIFT3D.cc:751-760)
	if (i + 1 < c->ng) need = c->taps[i + 1].hw + 1;
	if (i >= 1 && i <= c->p.num_kp_levels) {
		// orientation / descriptor windows of the keypoints of level i live on G[i]: the z reach of ITS descriptor window (r03; before:
		// the reach of the widest level for all of them: 3 x 38 planes per side instead of 24 + 30 + 38 with the default parameters)
		const float radius = 2.0f * (c->dog[(size_t)i].scale * 7.071067812f);
		const int desc_reach = (int)ceilf(__builtin_fabsf(radius) / c->dog[(size_t)i].unit) + 2;
		// r05, partial descriptor windows: every rank marches the window planes it OWNS (plus one plane either side for the central
		// difference), so the halo of G[i] only carries the orientation window of a keypoint on the slab's face: sphere of
		// 3 * 1.5 * scale (Src/cSIFT3D.cc:925-955), i.e. floor(4.5 scale / unit) planes, + 1 for the central difference
		const int ori_reach = (int)floorf(4.5f * c->dog[(size_t)i].scale / c->dog[(size_t)i].unit) + 1;
		need = std::max(need, c->desc_partial ? ori_reach : desc_reach);
	}
	*planes = std::min(need, c->halo);
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_level_hw(sift3d_handle c, int i, int *hw) {
	if (!c || !c->slab || !hw || i < 0 || i >= c->ng) return SIFT3D_ERR_ARG;
	*hw = i == 0 ? c->base_taps.hw : c->taps[i].hw;
	return SIFT3D_OK;
}

// The caller's stream (a hipStream_t of the context's device, e.g. torch.cuda.Stream().cuda_stream) becomes the stream every later
// call of this handle enqueues on: a communication layer that orders its transfers behind / in front of that stream (RCCL through
// torch.distributed does) then needs no host synchronisation between the levels.  nullptr restores the context's own stream.
extern "C" int sift3d_set_stream(sift3d_handle c, void *stream) {
	if (!c) return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	S3D_HIP(hipStreamSynchronize(c->stream));
	hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
	for (auto &o : c->ostream) if (o == c->stream) o = st;
	c->stream = st;
	return SIFT3D_OK;
}

// local max|DoG| of the octave's levels as nd floats in device memory, and back after the caller's MAX all-reduce: D2D copies on
// the handle's stream, no host round trip (the host forms sift3d_slab_get/set_dogmax stay for tests)
extern "C" int sift3d_slab_export_dogmax_device(sift3d_handle c, float *d_dst) {
	if (!c || !c->slab || !d_dst) return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	S3D_HIP(hipMemcpyAsync(d_dst, c->d_dogmax, sizeof(unsigned) * (size_t)c->nd, hipMemcpyDeviceToDevice, c->stream));
	return SIFT3D_OK;
}
extern "C" int sift3d_slab_import_dogmax_device(sift3d_handle c, const float *d_src) {
	if (!c || !c->slab || !d_src) return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	S3D_HIP(hipMemcpyAsync(c->d_dogmax, d_src, sizeof(unsigned) * (size_t)c->nd, hipMemcpyDeviceToDevice, c->stream));
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_level(sift3d_handle c, int i) {
	if (!c || !c->slab || i < 0 || i >= c->ng) return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	const Level &L = c->gss[i];
	const ZRange zr = L.zr(c->own0 - L.zoff, c->own1 - L.zoff);
	bool ok;
	if (i == 0) {
		c->g_last_built.assign(1, 0);
		S3D_HIP(hipMemsetAsync(c->d_dogmax, 0, sizeof(unsigned) * (size_t)(c->nd + 4), c->stream));
		// octave > 0: level 0 is the decimated G[octave-1][num_kp_levels], written by the caller
		ok = c->seeded ? true : launch_march_level(c->in.d, L.d, nullptr, nullptr, L.nx, L.ny, zr, c->base_taps, c->stream);
	} else {
		// like the single-volume path, the first and last DoG level of the octave are not materialised (read only as the centre-voxel
		// neighbour of extremum candidates: no halo, no abs-max)
		const bool dog_eager = hook(SIFT3D_HOOK_DOG_EAGER) != 0, glast_eager = hook(SIFT3D_HOOK_GLAST_EAGER) != 0;
		c->dog_elide = !dog_eager && c->nd >= 3;
		// ... and the last Gaussian level is not built at all (k_lazy_next evaluates it at the parked extremum candidates; its source
		// level G[nd-1] holds the hw+1 halo planes the caller exchanged for this level)
		c->g_last_elide = c->dog_elide && !glast_eager && 2 * (2 * c->taps[c->ng - 1].hw + 1) <= kLazySlots;
		if (c->g_last_elide && i == c->ng - 1) { c->stage = std::max(c->stage, 1); return SIFT3D_OK; }
		const bool elided = c->dog_elide && (i - 1 == 0 || i - 1 == c->nd - 1);
		ok = launch_march_level(c->gss[i - 1].d, L.d, elided ? nullptr : c->dog[i - 1].d, elided ? nullptr : c->d_dogmax + (i - 1), L.nx, L.ny,
		                        zr, c->taps[i], c->stream);
	}
	if (!ok) { set_last_error("no fused kernel for this level"); return SIFT3D_ERR_STATE; }
	c->stage = std::max(c->stage, 1);
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_sync(sift3d_handle c) {
	if (!c) return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	S3D_HIP(hipStreamSynchronize(c->stream));
	S3D_HIP(hipGetLastError());
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_get_dogmax(sift3d_handle c, float *max5) {
	if (!c || !c->slab || !max5) return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	unsigned bits[8] = {0};
	S3D_HIP(hipMemcpyAsync(bits, c->d_dogmax, sizeof(unsigned) * (size_t)c->nd, hipMemcpyDeviceToHost, c->stream));
	S3D_HIP(hipStreamSynchronize(c->stream));
	memcpy(max5, bits, sizeof(float) * (size_t)c->nd);
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_set_dogmax(sift3d_handle c, const float *max5) {
	if (!c || !c->slab || !max5) return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	unsigned bits[8] = {0};
	memcpy(bits, max5, sizeof(float) * (size_t)c->nd);
	S3D_HIP(hipMemcpyAsync(c->d_dogmax, bits, sizeof(unsigned) * (size_t)c->nd, hipMemcpyHostToDevice, c->stream));
	S3D_HIP(hipStreamSynchronize(c->stream));
	return SIFT3D_OK;
}

static int slab_count_and_regrow(sift3d_ctx *c, bool &again) {
	unsigned host_words[3] = {0, 0, 0};
	S3D_HIP(hipMemcpyAsync(host_words, c->d_total, sizeof(unsigned) * 3, hipMemcpyDeviceToHost, c->stream));
	S3D_HIP(hipStreamSynchronize(c->stream));
	S3D_HIP(hipGetLastError());
	again = false;
	if (host_words[1] != 0 || host_words[0] > c->ext_cap) {
		int rc = alloc_lists(c, std::max(host_words[0], c->ext_cap) * 2u);
		if (rc) return rc;
		c->n_regrow++;
		again = true;
		return SIFT3D_OK;
	}
	c->n_ext = host_words[0];
	c->n_kp = host_words[2];
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_detect(sift3d_handle c) {
	if (!c || !c->slab) return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	for (int attempt = 0; attempt < 4; attempt++) {
		S3D_HIP(hipMemsetAsync(c->d_total, 0, sizeof(unsigned) * 3, c->stream));
		DetectLevels DL;
		memset(&DL, 0, sizeof(DL));
		const int nl = c->nd - 2;
		for (int i = 1; i <= nl; i++) {
			DL.cur[i - 1] = c->dog[i].d; DL.prev[i - 1] = c->dog[i - 1].d; DL.next[i - 1] = c->dog[i + 1].d;
			DL.absmax_bits[i - 1] = c->d_dogmax + i;
			DL.level_id[i - 1] = i;
			DL.scale[i - 1] = c->dog[i].scale;
		}
		if (c->dog_elide) {
			DL.prev0_hi = c->gss[1].d; DL.prev0_lo = c->gss[0].d;
			DL.nextl_hi = c->gss[c->nd].d; DL.nextl_lo = c->gss[c->nd - 1].d;
			DL.nextl_slot = nl - 1;
			if (c->g_last_elide) { DL.nextl_hi = nullptr; DL.lazy_src = DL.nextl_lo; }
		}
		const Level &C = c->dog[1];
		launch_detect_octave(DL, nl, C.nx, C.ny, C.zr(c->own0 - C.zoff, c->own1 - C.zoff), c->p.peak_thresh, c->octave_base, c->det,
		                     c->d_ext, c->ext_cap, c->stream, c->g_last_elide ? &c->taps[c->ng - 1] : nullptr);
		bool again;
		rc = slab_count_and_regrow(c, again);
		if (rc) return rc;
		if (!again) { c->stage = 3; c->n_kp = 0; return SIFT3D_OK; }
	}
	set_last_error("extrema list kept overflowing");
	return SIFT3D_ERR_CAPACITY;
}

// planes per side the level buffers must carry for the keypoint windows of this context: whole descriptor windows, or (partial) the
// orientation windows only
static int slab_window_halo(const sift3d_ctx *c, bool whole_descriptor_windows) {
	int need = 0;
	for (int i = 1; i <= c->p.num_kp_levels; i++) {
		const float sc = c->dog[(size_t)i].scale, u = c->dog[(size_t)i].unit;
		const int desc_reach = (int)ceilf(2.0f * (sc * 7.071067812f) / u) + 2, ori_reach = (int)floorf(4.5f * sc / u) + 1;
		need = std::max(need, whole_descriptor_windows ? desc_reach : ori_reach);
	}
	return need;
}

extern "C" int sift3d_slab_describe(sift3d_handle c) {
	if (!c || !c->slab || c->stage < 3) return SIFT3D_ERR_STATE;
	if (c->desc_partial || c->halo < slab_window_halo(c, true)) {
		// (sift3d_slab_halo_planes caps its answers at the buffers' halo: whole windows on a smaller halo would read planes nobody exchanged)
		set_last_error("sift3d_slab_describe marches whole descriptor windows: the level buffers' halo is too small for them (or the context is in partial-window mode) -- use sift3d_slab_orient / describe_partial / describe_finish");
		return SIFT3D_ERR_STATE;
	}
	int rc = set_device(c->device);
	if (rc) return rc;
	hipStream_t st = c->stream;
	launch_orient(c->d_ext, c->d_codes, c->d_total, c->ext_cap, c->d_levels, c->d_luts, c->d_lutpool, c->p.max_eig_thres,
	              c->p.corner_thresh, 0, 1, c->d_order, c->d_nkp + 3, st);
	launch_slots(c->d_ext, c->d_codes, c->d_total, c->ext_cap, c->d_nkp, c->d_order, c->kp_cap, c->d_slots_part, st);
	launch_describe(c->d_ext, c->d_total, c->ext_cap, c->d_levels, c->d_luts, c->d_lutpool, c->d_desc, c->kp_cap, 0, 1, c->d_order, c->d_nkp,
	                c->d_nkp + 1, st, c->desc_lut_lds, &c->dsplit);
	launch_finalize(c->d_ext, c->d_total, c->ext_cap, 1, c->d_kpout, c->d_xyz, c->kp_cap, st);
	bool again;
	rc = slab_count_and_regrow(c, again);
	if (rc) return rc;
	c->stage = 5;
	return SIFT3D_OK;
}

// ---- r05: descriptor windows split along z over the ranks of a sharded volume (partial integer histograms) ----------------
// No reference counterpart (the reference is one process, Src/cSIFT3D.cc:484-502 walks whole windows).  Instead of shipping the 24 / 30 /
// 38-plane halos of G[1..3] that whole descriptor windows reach, the ranks ship RECORDS (a keypoint's coordinates, level, scale,
// rotation, structure tensor: 164 bytes) to the z-neighbours a window reaches into; every rank marches, for its own and for the foreign
// records, the window planes it OWNS and returns 768 int32 sums + the part's gradient mass; the owner adds the parts -- the same
// integers the single-volume run adds in its LDS histogram -- and normalises.  kernels_desc.hip: k_describe<.., PARTIAL>, k_describe_finish.
extern "C" int sift3d_slab_set_desc_partial(sift3d_handle c, int on) {
	if (!c || !c->slab) return SIFT3D_ERR_ARG;
	c->desc_partial = on != 0;
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_min_halo_partial(const sift3d_params *params, int *halo) {
	if (!halo) return SIFT3D_ERR_ARG;
	sift3d_params p;
	if (params) p = *params; else sift3d_default_params(&p);
	if (p.num_kp_levels < 1 || p.num_kp_levels > 5) return SIFT3D_ERR_ARG;
	const double sigma0 = (double)p.sigma_default * pow(2.0, -1.0 / 3.0);
	const float scale = (float)(pow(2.0, (double)p.num_kp_levels / (double)p.num_kp_levels) * sigma0);  // DoG level num_kp_levels, octave 0
	int h = (int)floorf(4.5f * scale) + 2;  // orientation window of the widest keypoint level + the central difference (+ 1 spare)
	std::vector<float> sig;
	float base_sigma;
	level_sigmas(p, sig, base_sigma);
	for (size_t i = 1; i < sig.size(); i++) {
		Taps t;
		if (!build_taps(sig[i], t)) return SIFT3D_ERR_ARG;
		h = std::max(h, t.hw + 1);
	}
	*halo = h;
	return SIFT3D_OK;
}

extern "C" int sift3d_slab_record_bytes(int *bytes) {
	if (!bytes) return SIFT3D_ERR_ARG;
	*bytes = (int)sizeof(DevKp);
	return SIFT3D_OK;
}

// planes (of this context's octave) the widest descriptor window reaches beyond its keypoint, central difference included: ranks whose
// owned planes lie within this distance of a slab take part in its keypoints' windows
extern "C" int sift3d_slab_desc_reach(sift3d_handle c, int *planes) {
	if (!c || !c->slab || !planes) return SIFT3D_ERR_ARG;
	int reach = 0;
	for (int i = 1; i <= c->p.num_kp_levels; i++) {
		const float radius = 2.0f * (c->dog[(size_t)i].scale * 7.071067812f);
		reach = std::max(reach, (int)ceilf(__builtin_fabsf(radius) / c->dog[(size_t)i].unit) + 1);
	}
	*planes = reach;
	return SIFT3D_OK;
}

// Assign_Orientation (Src/cSIFT3D.cc:427-482) of the slab's extrema; the accepted keypoints are counted (sift3d_num_keypoints)
extern "C" int sift3d_slab_orient_launch(sift3d_handle c);
extern "C" int sift3d_slab_orient_count(sift3d_handle c, int *n_kp);
extern "C" int sift3d_slab_orient(sift3d_handle c) {
	int n = 0;
	const int rc = sift3d_slab_orient_launch(c);
	return rc ? rc : sift3d_slab_orient_count(c, &n);
}

// the accepted keypoints' records in PROCESSING order (large windows first, kernels_orient.hip k_slots) -> n_kp * record_bytes at d_dst
extern "C" int sift3d_slab_export_records(sift3d_handle c, void *d_dst) {
	if (!c || !c->slab || !d_dst) return SIFT3D_ERR_ARG;
	if (c->stage < 4) return SIFT3D_ERR_STATE;
	int rc = set_device(c->device);
	if (rc) return rc;
	if (c->n_kp) launch_export_records(c->d_ext, c->d_order, c->n_kp, static_cast<DevKp *>(d_dst), c->stream);
	return SIFT3D_OK;
}

// this rank's z part of the windows of nlists record lists -- its own keypoints and those of the z-neighbours whose windows reach into it
// (the tables of a neighbour's context of the same octave are the same) -- in ONE launch.  List i: n[i] records at d_records[i], whose
// owner owns the planes [owner_z0[i], owner_z1[i]) (this context's own range marks its own list); out: d_hist[i][n[i]][768] int32,
// d_mass[i][n[i]].  d_units: null, or per list null / the units of the second round (entries <= 0: first-pass rule).  Which planes of a
// window a rank marches: DescPartial (sift3d_internal.h).
extern "C" int sift3d_slab_describe_partial(sift3d_handle c, int nlists, const void *const *d_records, const int *n, const float *const *d_units,
                                            int *const *d_hist, float *const *d_mass, const int *owner_z0, const int *owner_z1) {
	if (!c || !c->slab || nlists < 0 || (nlists > 0 && (!d_records || !n || !d_hist || !d_mass || !owner_z0 || !owner_z1))) return SIFT3D_ERR_ARG;
	if (c->stage < 3) return SIFT3D_ERR_STATE;
	int rc = set_device(c->device);
	if (rc) return rc;
	for (int i0 = 0; i0 < nlists; i0 += kDescSegs) {  // (more lists than a launch takes: slabs much thinner than a window's reach)
		DescPartial pp;
		pp.zc0 = c->own0; pp.zc1 = c->own1;
		for (int l = 1; l <= c->p.num_kp_levels && l < 8; l++) {
			int planes = 0;
			if ((rc = sift3d_slab_halo_planes(c, l, &planes)) != SIFT3D_OK) return rc;
			pp.H[l] = std::max(0, planes - 1);  // (the outermost halo plane only serves the central difference)
		}
		unsigned first = 0;
		for (int i = i0; i < std::min(nlists, i0 + kDescSegs); i++) {
			if (n[i] < 0 || (n[i] > 0 && (!d_records[i] || !d_hist[i] || !d_mass[i]))) return SIFT3D_ERR_ARG;
			if (n[i] == 0) continue;
			DescSeg &sg = pp.seg[pp.nseg++];
			sg.recs = static_cast<const DevKp *>(d_records[i]); sg.units = d_units ? d_units[i] : nullptr;
			sg.hist = d_hist[i]; sg.mass = d_mass[i]; sg.first = first; sg.n = (unsigned)n[i]; sg.o0 = owner_z0[i]; sg.o1 = owner_z1[i];
			first += (unsigned)n[i];
		}
		launch_describe_partial(c->d_levels, c->d_luts, c->d_lutpool, pp, c->d_nkp + 1, c->stream, c->desc_lut_lds);
	}
	return SIFT3D_OK;
}

// the owner's finish of n of its records (all of them, or the second round's subset): nparts partial results (its own part and its
// z-neighbours', in ascending rank order: the integer histograms are added, the masses in that order).  Rows go to the descriptor table
// (sift3d_get_keypoints / sift3d_device_results); records whose unit failed are flagged (d_redo[k] = 1, d_units_next[k]) and counted in
// *n_redo unless final_round.  With nothing left to redo the keypoint records are finalised.
extern "C" int sift3d_slab_describe_finish(sift3d_handle c, const void *d_records, int n, int nparts, const int *const *d_hist,
                                           const float *const *d_mass, const float *d_units, int final_round, int *d_redo, float *d_units_next,
                                           int *n_redo) {
	if (!c || !c->slab || n < 0 || !n_redo || nparts < 0 || nparts > kDescSegs || (n > 0 && (!d_records || !d_hist || !d_mass || nparts < 1)))
		return SIFT3D_ERR_ARG;
	if (!final_round && n > 0 && (!d_redo || !d_units_next)) return SIFT3D_ERR_ARG;
	if (c->stage < 4) return SIFT3D_ERR_STATE;
	int rc = set_device(c->device);
	if (rc) return rc;
	hipStream_t st = c->stream;
	unsigned *counter = c->d_nkp + 4;  // (d_total + 6: a spare word behind the orientation's redo counter)
	S3D_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned), st));
	launch_describe_finish(static_cast<const DevKp *>(d_records), (unsigned)n, c->d_levels, c->d_luts, nparts, d_hist, d_mass, d_units, final_round != 0, c->d_desc,
	                       d_redo, d_units_next, counter, st);
	unsigned host = 0;
	if (!final_round) {  // (a final round flags nothing)
		S3D_HIP(hipMemcpyAsync(&host, counter, sizeof(unsigned), hipMemcpyDeviceToHost, st));
		S3D_HIP(hipStreamSynchronize(st));
		S3D_HIP(hipGetLastError());
	}
	*n_redo = (int)host;
	if (final_round) c->n_desc_redo = n; else c->n_desc_redo = 0;
	if (host == 0) {
		launch_finalize(c->d_ext, c->d_total, c->ext_cap, 1, c->d_kpout, c->d_xyz, c->kp_cap, st);
		S3D_HIP(hipStreamSynchronize(st));  // results are complete when the call returns (sift3d_get_keypoints copies on the handle's own stream)
		S3D_HIP(hipGetLastError());
		c->stage = 5;
	}
	return SIFT3D_OK;
}

// the launch and the read-back of sift3d_slab_orient as two calls: a driver with several ranks in one process (simulated ranks) enqueues
// every rank's orientation before it waits for the first count
extern "C" int sift3d_slab_orient_launch(sift3d_handle c) {
	if (!c || !c->slab || c->stage < 3) return SIFT3D_ERR_STATE;
	if (c->halo < slab_window_halo(c, false)) { set_last_error("the level buffers' halo is smaller than the orientation windows' reach"); return SIFT3D_ERR_STATE; }
	int rc = set_device(c->device);
	if (rc) return rc;
	hipStream_t st = c->stream;
	launch_orient(c->d_ext, c->d_codes, c->d_total, c->ext_cap, c->d_levels, c->d_luts, c->d_lutpool, c->p.max_eig_thres,
	              c->p.corner_thresh, 0, 1, c->d_order, c->d_nkp + 3, st);
	launch_slots(c->d_ext, c->d_codes, c->d_total, c->ext_cap, c->d_nkp, c->d_order, c->kp_cap, c->d_slots_part, st);
	return SIFT3D_OK;
}
extern "C" int sift3d_slab_orient_count(sift3d_handle c, int *n_kp) {
	if (!c || !c->slab || c->stage < 3 || !n_kp) return SIFT3D_ERR_STATE;
	int rc = set_device(c->device);
	if (rc) return rc;
	bool again;
	rc = slab_count_and_regrow(c, again);
	if (rc) return rc;
	c->stage = 4;
	*n_kp = (int)c->n_kp;
	return SIFT3D_OK;
}

// ---- r06: the same stages without a host read-back between them.  A driver enqueues detection + orientation of every sharded octave (and of
// every simulated rank), then asks for the counts; the GPU works on the later launches while the host learns the earlier counts.  Rare events
// (a list that overflowed, a record whose fixed-point unit failed) are found when the counts are read and take the blocking forms above.
static int slab_keypoints_enqueue(sift3d_ctx *c) {
	hipStream_t st = c->stream;
	S3D_HIP(hipMemsetAsync(c->d_total, 0, sizeof(unsigned) * 3, st));
	DetectLevels DL;
	memset(&DL, 0, sizeof(DL));
	const int nl = c->nd - 2;
	for (int i = 1; i <= nl; i++) {
		DL.cur[i - 1] = c->dog[i].d; DL.prev[i - 1] = c->dog[i - 1].d; DL.next[i - 1] = c->dog[i + 1].d;
		DL.absmax_bits[i - 1] = c->d_dogmax + i;
		DL.level_id[i - 1] = i;
		DL.scale[i - 1] = c->dog[i].scale;
	}
	if (c->dog_elide) {
		DL.prev0_hi = c->gss[1].d; DL.prev0_lo = c->gss[0].d;
		DL.nextl_hi = c->gss[c->nd].d; DL.nextl_lo = c->gss[c->nd - 1].d;
		DL.nextl_slot = nl - 1;
		if (c->g_last_elide) { DL.nextl_hi = nullptr; DL.lazy_src = DL.nextl_lo; }
	}
	const Level &C = c->dog[1];
	launch_detect_octave(DL, nl, C.nx, C.ny, C.zr(c->own0 - C.zoff, c->own1 - C.zoff), c->p.peak_thresh, c->octave_base, c->det,
	                     c->d_ext, c->ext_cap, st, c->g_last_elide ? &c->taps[c->ng - 1] : nullptr);
	launch_orient(c->d_ext, c->d_codes, c->d_total, c->ext_cap, c->d_levels, c->d_luts, c->d_lutpool, c->p.max_eig_thres,
	              c->p.corner_thresh, 0, 1, c->d_order, c->d_nkp + 3, st);
	launch_slots(c->d_ext, c->d_codes, c->d_total, c->ext_cap, c->d_nkp, c->d_order, c->kp_cap, c->d_slots_part, st);
	S3D_HIP(hipMemcpyAsync(c->h_words, c->d_total, sizeof(unsigned) * 3, hipMemcpyDeviceToHost, st));
	S3D_HIP(hipEventRecord(c->ev[6], st));
	return SIFT3D_OK;
}

// Detect_KeyPoints + Assign_Orientation (Src/cSIFT3D.cc:362-482) of the slab's owned planes, enqueued; the counts travel to pinned memory behind them
extern "C" int sift3d_slab_keypoints_launch(sift3d_handle c) {
	if (!c || !c->slab) return SIFT3D_ERR_ARG;
	if (c->stage < 1) return SIFT3D_ERR_STATE;
	if (c->halo < slab_window_halo(c, false)) { set_last_error("the level buffers' halo is smaller than the orientation windows' reach"); return SIFT3D_ERR_STATE; }
	int rc = set_device(c->device);
	if (rc) return rc;
	return slab_keypoints_enqueue(c);
}

// waits for the counts of sift3d_slab_keypoints_launch; a list that overflowed is regrown and the two stages run again (blocking: rare)
extern "C" int sift3d_slab_keypoints_count(sift3d_handle c, int *n_kp) {
	if (!c || !c->slab || !n_kp) return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	for (int attempt = 0; attempt < 4; attempt++) {
		S3D_HIP(hipEventSynchronize(c->ev[6]));
		S3D_HIP(hipGetLastError());
		const unsigned n_ext = c->h_words[0], over = c->h_words[1], n_acc = c->h_words[2];
		if (over == 0 && n_ext <= c->ext_cap) {
			c->n_ext = n_ext; c->n_kp = n_acc; c->stage = 4;
			*n_kp = (int)n_acc;
			return SIFT3D_OK;
		}
		S3D_HIP(hipStreamSynchronize(c->stream));  // (nothing of this handle may still use the lists that are about to be replaced)
		if ((rc = alloc_lists(c, std::max(n_ext, c->ext_cap) * 2u)) != SIFT3D_OK) return rc;
		c->n_regrow++;
		if ((rc = slab_keypoints_enqueue(c)) != SIFT3D_OK) return rc;
	}
	set_last_error("extrema list kept overflowing");
	return SIFT3D_ERR_CAPACITY;
}

// Extract_Description (Src/cSIFT3D.cc:484-502) of the slab's own keypoints with WHOLE windows from its own level buffers (slabs too thin to
// split the windows along z carry the windows' whole reach as halo), enqueued behind sift3d_slab_keypoints_launch / _count; the results are
// complete when the stream has drained
extern "C" int sift3d_slab_describe_launch(sift3d_handle c) {
	if (!c || !c->slab) return SIFT3D_ERR_ARG;
	if (c->stage < 4) return SIFT3D_ERR_STATE;
	if (c->desc_partial || c->halo < slab_window_halo(c, true)) {
		set_last_error("sift3d_slab_describe_launch marches whole descriptor windows: the level buffers' halo is too small for them (or the context is in partial-window mode)");
		return SIFT3D_ERR_STATE;
	}
	int rc = set_device(c->device);
	if (rc) return rc;
	hipStream_t st = c->stream;
	launch_describe(c->d_ext, c->d_total, c->ext_cap, c->d_levels, c->d_luts, c->d_lutpool, c->d_desc, c->kp_cap, 0, 1, c->d_order, c->d_nkp,
	                c->d_nkp + 1, st, c->desc_lut_lds, &c->dsplit);
	launch_finalize(c->d_ext, c->d_total, c->ext_cap, 1, c->d_kpout, c->d_xyz, c->kp_cap, st);
	c->stage = 5;
	return SIFT3D_OK;
}

// sift3d_slab_describe_finish of the FIRST round without its read-back: the finish, the final records and the count of flagged records
// (to pinned memory) are enqueued; sift3d_slab_describe_finish_count waits for that count.  Nothing flagged (the rule): the results are complete.
extern "C" int sift3d_slab_describe_finish_launch(sift3d_handle c, const void *d_records, int n, int nparts, const int *const *d_hist,
                                                  const float *const *d_mass, int *d_redo, float *d_units_next) {
	if (!c || !c->slab || n < 0 || nparts < 0 || nparts > kDescSegs || (n > 0 && (!d_records || !d_hist || !d_mass || nparts < 1 || !d_redo || !d_units_next)))
		return SIFT3D_ERR_ARG;
	if (c->stage < 4) return SIFT3D_ERR_STATE;
	int rc = set_device(c->device);
	if (rc) return rc;
	hipStream_t st = c->stream;
	unsigned *counter = c->d_nkp + 4;
	S3D_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned), st));
	launch_describe_finish(static_cast<const DevKp *>(d_records), (unsigned)n, c->d_levels, c->d_luts, nparts, d_hist, d_mass, nullptr, false, c->d_desc,
	                       d_redo, d_units_next, counter, st);
	launch_finalize(c->d_ext, c->d_total, c->ext_cap, 1, c->d_kpout, c->d_xyz, c->kp_cap, st);
	S3D_HIP(hipMemcpyAsync(c->h_words + 5, counter, sizeof(unsigned), hipMemcpyDeviceToHost, st));
	S3D_HIP(hipEventRecord(c->ev[7], st));
	c->n_desc_redo = 0;
	return SIFT3D_OK;
}
extern "C" int sift3d_slab_describe_finish_count(sift3d_handle c, int *n_redo) {
	if (!c || !c->slab || !n_redo) return SIFT3D_ERR_ARG;
	if (c->stage < 4) return SIFT3D_ERR_STATE;
	int rc = set_device(c->device);
	if (rc) return rc;
	S3D_HIP(hipEventSynchronize(c->ev[7]));
	S3D_HIP(hipGetLastError());
	*n_redo = (int)c->h_words[5];
	if (c->h_words[5] == 0) c->stage = 5;
	return SIFT3D_OK;
}

// level 0 of a seeded context's first octave in device memory (nx * ny * nz floats): a driver that gathers the seed level writes it in
// place, on the stream it gave the handle (sift3d_set_stream), and follows with sift3d_run_async -- no staging copy, no host synchronisation
extern "C" int sift3d_seed_buffer(sift3d_handle c, float **d_level0, size_t *floats) {
	if (!c || !c->seeded || c->slab || !d_level0) return SIFT3D_ERR_ARG;
	if (c->noct <= 0) { *d_level0 = nullptr; if (floats) *floats = 0; return SIFT3D_OK; }
	*d_level0 = c->gss[0].d;
	if (floats) *floats = c->gss[0].n();
	return SIFT3D_OK;
}

static int slab_decimate_impl(sift3d_handle c, float *d_dst, bool sync);
extern "C" int sift3d_slab_decimate(sift3d_handle c, float *d_dst) { return slab_decimate_impl(c, d_dst, true); }
extern "C" int sift3d_slab_decimate_async(sift3d_handle c, float *d_dst) { return slab_decimate_impl(c, d_dst, false); }
static int slab_decimate_impl(sift3d_handle c, float *d_dst, bool sync) {
	if (!c || !c->slab || !d_dst) return SIFT3D_ERR_ARG;
	int rc = set_device(c->device);
	if (rc) return rc;
	const Level &P = c->gss[c->p.num_kp_levels];
	const size_t pl = (size_t)P.nx * P.ny;
	// owned planes start at an even global z, so dst plane k = src global plane own0 + 2k (Src/cSIFT3D.cc:321-344)
	const int nz2 = std::min(c->own1 / 2, c->nz / 2) - c->own0 / 2;
	if (nz2 > 0)
		launch_downsample(P.d + pl * (size_t)(c->own0 - P.zoff), P.nx, P.ny, d_dst, P.nx / 2, P.ny / 2, nz2, c->stream);
	if (sync) S3D_HIP(hipStreamSynchronize(c->stream));
	return SIFT3D_OK;
}
