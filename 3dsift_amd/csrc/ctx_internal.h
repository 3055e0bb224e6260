// ctx_internal.h -- the context of the HIP library and the host helpers its translation units share (not part of the C-ABI).
//   context.hip     life cycle of a sift3d_ctx, the KpSiftAlgorithm pipeline (Src/cSIFT3D.cc:165-235) and its accessors
//   tables.hip      host-built constants: Gaussian taps, icosahedron faces + symmetry table, window weight tables
//   entry_free.hip  the reference's free functions on caller-held host data (Include/cSIFT3D.h:208-239)
//   entry_slab.hip  seeded contexts and z-slab contexts of the multi-GPU sharding
//   entry_test.hip  test hooks, rare-path counters, unit-level debug entry points (include/sift3d_hip_test.h)
#pragma once

#include <string>
#include <thread>
#include <vector>

#include "sift3d_internal.h"

// (the context is the C-ABI's opaque `struct sift3d_ctx`: global scope; its members are the library's types)
using s3d::DescSplit; using s3d::DetectBufs; using s3d::DevKp; using s3d::Level; using s3d::LevelRef; using s3d::Taps; using s3d::WinLut;


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
	bool ghost = false;             // slab contexts (r06): every level is produced on ghost planes beyond the owned range too (sift3d_slab_set_ghost): no per-level halo exchange
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

namespace s3d {
// tables.hip
bool build_taps(float sigma, Taps &t);
void level_sigmas(const sift3d_params &p, std::vector<float> &sig, float &base_sigma);
void build_faces(FaceConst *F);
bool build_facesym(const FaceConst *F, FaceSym *S);
bool append_lut(std::vector<float> &pool, WinLut &L, int which, float sigma, float radius, float u, float scale);
// context.hip
int set_device(int device);
int alloc_lists(sift3d_ctx *c, unsigned ext_cap);
void plan_pyramid(sift3d_ctx *c, int noct_total);
int upload_luts(sift3d_ctx *c, const std::vector<WinLut> &luts, std::vector<float> &pool);
std::vector<WinLut> blank_luts(const sift3d_ctx *c);
size_t arena_floats_of(const sift3d_ctx *c);
int create_common(sift3d_handle *out, const CreateCfg &cfg, const sift3d_params *params, int device);
int run_impl(sift3d_ctx *c, int upto, bool part_orient = false);
}  // namespace s3d
