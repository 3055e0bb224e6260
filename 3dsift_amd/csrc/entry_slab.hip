// entry_slab.hip -- seeded contexts and z-slab contexts: the C-ABI of the multi-GPU sharding (include/sift3d_hip.h).  The native
// driver over these entry points is sharded.hip, the python one 3dsift_amd/slab.py.
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

using namespace s3d;

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

// would the slab contexts take an octave of these GLOBAL dims?  Every level a slab builds goes through the z-march kernel (no separable fallback for
// slabs): half widths 2 .. 8, planes of one tile or >= 32 + hw, and a column of at least 2 hw + 2 planes (r06: a sharding plan whose deepest octave
// was thinner than that -- 32 planes, three sharded octaves -- was accepted and failed in its first run)
extern "C" int sift3d_slab_admits(const sift3d_params *params, int nx, int ny, int nz, int first_octave, int *ok) {
	if (!ok || nx <= 0 || ny <= 0 || nz <= 0) return SIFT3D_ERR_ARG;
	*ok = 0;
	sift3d_params p;
	if (params) p = *params; else sift3d_default_params(&p);
	if (p.num_kp_levels < 1 || p.num_kp_levels > 5) return SIFT3D_ERR_ARG;
	std::vector<float> sig;
	float base_sigma;
	level_sigmas(p, sig, base_sigma);
	const int ng = (int)sig.size();
	Taps t;
	for (int i = 1; i < ng; i++) {
		if (!build_taps(sig[(size_t)i], t)) return SIFT3D_OK;
		// the last Gaussian level is evaluated at parked candidates instead of being built while its kernel fits k_lazy_next (sift3d_slab_level)
		if (i == ng - 1 && 2 * (2 * t.hw + 1) <= kLazySlots) continue;
		if (!march_applicable(nx, ny, nz, t)) return SIFT3D_OK;
	}
	if (first_octave && (!build_taps(base_sigma, t) || !march_applicable(nx, ny, nz, t))) return SIFT3D_OK;  // the base blur of the input
	*ok = 1;
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
	// (r06: any integer boundaries -- DownSample_3D's plane 2k belongs to the slab that owns plane 2k, sift3d_slab_decimate; until then even starts)
	if (!d || d->nx <= 0 || d->ny <= 0 || d->nz <= 0 || d->z0 < 0 || d->z1 > d->nz || d->z1 <= d->z0 || d->halo < 1) {
		set_last_error("bad slab description (owned range must be non-empty and inside the volume)");
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

// ---- ghost zones (r06) ---------------------------------------------------------------------------------------------------------------------
// Instead of receiving, per level, the hw + 1 planes the next level's z-march reads beyond the owned range (an exchange the level chain WAITS for),
// a slab can produce every level on a range that shrinks level by level: level i on [own0 - g_i, own1 + g_i) with g_i = what its readers need --
// the next level's march (g_{i+1} + hw_{i+1} + 1; hw_{i+1} + 1 where that level is evaluated lazily), the orientation / descriptor windows of the
// keypoint levels, one plane for the DoG neighbours of the extremum test -- from an input that holds g_0 + hw_0 + 1 planes per side.  Every ghost
// plane holds exactly what its owner computes for it (same kernels, same inputs, boundary rule in global z), so nothing downstream changes; the
// abs-max of a DoG level then covers ghost planes too, which leaves the maximum over all slabs what it is.
static void slab_ghost_extents(const sift3d_ctx *c, bool last_lazy, int *g /* [ng] */) {
	for (int i = c->ng - 1; i >= 0; i--) {
		int need = 1;  // the z neighbours of the extremum test (every Gaussian level feeds a DoG level)
		if (i >= 1 && i <= c->p.num_kp_levels) {
			const float sc = c->dog[(size_t)i].scale, u = c->dog[(size_t)i].unit;
			const int desc_reach = (int)ceilf(__builtin_fabsf(2.0f * (sc * 7.071067812f)) / u) + 2, ori_reach = (int)floorf(4.5f * sc / u) + 1;
			need = std::max(need, c->desc_partial ? ori_reach : desc_reach);
		}
		if (i + 1 < c->ng) {
			const int hw1 = c->taps[i + 1].hw;
			const bool built = !(last_lazy && i + 1 == c->ng - 1);
			need = std::max(need, built ? g[i + 1] + hw1 + 1 : hw1 + 1);
		}
		g[i] = need;
	}
}
static bool slab_last_lazy(const sift3d_ctx *c) {  // (the rule of sift3d_slab_level)
	return !hook(SIFT3D_HOOK_DOG_EAGER) && c->nd >= 3 && !hook(SIFT3D_HOOK_GLAST_EAGER) && 2 * (2 * c->taps[c->ng - 1].hw + 1) <= kLazySlots;
}
extern "C" int sift3d_slab_set_ghost(sift3d_handle c, int on) {
	if (!c || !c->slab) return SIFT3D_ERR_ARG;
	if (on) {
		int g[16];
		for (int lazy = 0; lazy < 2; lazy++) {  // (whichever form the last level takes when the hooks change)
			slab_ghost_extents(c, lazy != 0, g);
			const int in_need = c->seeded ? 0 : g[0] + c->base_taps.hw + 1;
			for (int i = 0; i < c->ng; i++)
				if (g[i] > c->halo || in_need > c->halo) { set_last_error("ghost zones need a halo of sift3d_slab_min_halo_ghost planes"); return SIFT3D_ERR_ARG; }
		}
	}
	c->ghost = on != 0;
	return SIFT3D_OK;
}
// halo planes per side a slab context needs for ghost zones (partial: descriptor windows split along z; else whole windows)
extern "C" int sift3d_slab_min_halo_ghost(const sift3d_params *params, int partial, int *halo) {
	if (!halo) return SIFT3D_ERR_ARG;
	int win = 0;
	int rc = partial ? sift3d_slab_min_halo_partial(params, &win) : sift3d_slab_min_halo(params, &win);
	if (rc) return rc;
	sift3d_params p;
	if (params) p = *params; else sift3d_default_params(&p);
	std::vector<float> sig;
	float base_sigma;
	level_sigmas(p, sig, base_sigma);
	const int ng = (int)sig.size();
	std::vector<int> hw((size_t)ng, 0);
	Taps t;
	for (int i = 1; i < ng; i++) { if (!build_taps(sig[(size_t)i], t)) return SIFT3D_ERR_ARG; hw[(size_t)i] = t.hw; }
	if (!build_taps(base_sigma, t)) return SIFT3D_ERR_ARG;
	// (every keypoint level with the widest window's reach, the last level built: an upper bound of slab_ghost_extents)
	int g = 1;
	for (int i = ng - 2; i >= 0; i--) g = std::max(i >= 1 && i <= p.num_kp_levels ? win : 1, g + hw[(size_t)i + 1] + 1);
	*halo = g + t.hw + 1;
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
	int ext = 0;
	if (c->ghost) { int g[16]; slab_ghost_extents(c, slab_last_lazy(c), g); ext = g[i]; }
	const ZRange zr = L.zr(std::max(0, c->own0 - ext) - L.zoff, std::min(L.nz, c->own1 + ext) - L.zoff);
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
	// dst plane k = src global plane 2k for the planes 2k this slab owns: k in [ceil(own0 / 2), ceil(own1 / 2)) (Src/cSIFT3D.cc:321-344; r06: own0 may be odd)
	const int k0 = std::min((c->own0 + 1) / 2, c->nz / 2), k1 = std::min((c->own1 + 1) / 2, c->nz / 2);
	const int nz2 = k1 - k0;
	if (nz2 > 0)
		launch_downsample(P.d + pl * (size_t)(2 * k0 - P.zoff), P.nx, P.ny, d_dst, P.nx / 2, P.ny / 2, nz2, c->stream);
	if (sync) S3D_HIP(hipStreamSynchronize(c->stream));
	return SIFT3D_OK;
}
