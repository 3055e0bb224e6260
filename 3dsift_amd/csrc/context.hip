// context.hip -- life cycle of a context, the KpSiftAlgorithm pipeline and its accessors (C-ABI: include/sift3d_hip.h).
//
// One sift3d_ctx (ctx_internal.h) owns a device arena (both pyramids, scratch, keypoint lists), a set of HIP streams and
// the host-built constant tables (tables.hip).  sift3d_run enqueues the whole KpSiftAlgorithm pipeline
// (reference Src/cSIFT3D.cc:165-235) on those streams with no host synchronisation inside; the
// only sync is the final one that also brings the keypoint count back.  r06: the other entry points live in entry_free.hip (the
// reference's free functions), entry_slab.hip (seeded / z-slab contexts) and entry_test.hip (hooks, debug entry points).
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>

#include "ctx_internal.h"

#pragma clang fp contract(off)

namespace s3d {

static thread_local std::string g_last_error;
void set_last_error(const std::string &s) { g_last_error = s; }

int set_device(int device) {
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

int alloc_lists(sift3d_ctx *c, unsigned ext_cap) {
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

void plan_pyramid(sift3d_ctx *c, int noct_total) {
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

int upload_luts(sift3d_ctx *c, const std::vector<WinLut> &luts, std::vector<float> &pool) {
	if (pool.empty()) pool.push_back(-1.0f);
	if (c->d_luts) { S3D_HIP(hipFree(c->d_luts)); c->d_luts = nullptr; }
	if (c->d_lutpool) { S3D_HIP(hipFree(c->d_lutpool)); c->d_lutpool = nullptr; }
	S3D_HIP(hipMalloc(&c->d_luts, sizeof(WinLut) * luts.size()));
	S3D_HIP(hipMalloc(&c->d_lutpool, sizeof(float) * pool.size()));
	S3D_HIP(hipMemcpy(c->d_luts, luts.data(), sizeof(WinLut) * luts.size(), hipMemcpyHostToDevice));
	S3D_HIP(hipMemcpy(c->d_lutpool, pool.data(), sizeof(float) * pool.size(), hipMemcpyHostToDevice));
	return SIFT3D_OK;
}

std::vector<WinLut> blank_luts(const sift3d_ctx *c) {
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

static size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }

// floats needed for: input | per-octave scratch A,B | GSS levels | DoG levels
size_t arena_floats_of(const sift3d_ctx *c) {
	size_t total = c->in_own ? 0 : al64(c->in.n());
	if (!c->slab) for (int o = 0; o < c->noct; o++) total += 2 * al64(c->gss[(size_t)o * c->ng].n());  // slabs only run the fused kernel
	for (auto &L : c->gss) total += al64(L.n());
	for (auto &L : c->dog) total += al64(L.n());
	return total;
}

int create_common(sift3d_handle *out, const CreateCfg &cfg, const sift3d_params *params, int device) {
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

int run_impl(sift3d_ctx *c, int upto, bool part_orient) {
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

}  // namespace s3d
