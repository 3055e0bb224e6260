// sift3d_internal.h -- shared declarations of the HIP library (not part of the public C-ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/sift3d_hip.h"
#include "../../include/sift3d_hip_test.h"

// All arithmetic that must reproduce the reference bit-for-bit is written as separate IEEE
// multiplies and adds.  The library is compiled with -ffp-contract=off; the pragma makes the
// translation units safe even if someone drops the flag.
#pragma clang fp contract(off)

namespace s3d {

#ifndef S3D_MAX_HW
#define S3D_MAX_HW 64
#endif
constexpr int kMaxHW = S3D_MAX_HW;              // largest Gaussian half width supported (129 taps: num_kp_levels = 1 with sigma_default 2.6 needs 54; the defaults 8)
constexpr int kMaxTaps = 2 * kMaxHW + 1;
constexpr int kDesc = SIFT3D_DESC_NUMEL;
constexpr int kFaces = 20;

// 1-D Gaussian kernel of one pyramid level, built on the host exactly like
// GaussianSmooth_3D (reference Src/cSIFT3D.cc:541-572) and passed to kernels by value.
struct Taps {
	int hw;
	float w[kMaxTaps];
};

// fp32 lerp fractions of the right-boundary rule per axis (x, y, z), see kernels_march.hip
struct EdgeFrac {
	float f[3][kMaxHW + 1];
};

// z geometry of a level buffer.  Whole volumes: {nz, 0, nz, 0, nz}.  A z-slab (multi-GPU sharding) holds only
// planes [zoff, zoff+nz) of a level that is nzg planes tall, and a kernel produces local planes [zo0, zo1);
// boundary rules and keypoint coordinates always use GLOBAL z = local z + zoff.
struct ZRange {
	int nz;    // planes held in the buffer
	int zoff;  // global z of local plane 0 (may be negative: planes below 0 are never touched)
	int nzg;   // global number of planes of the level
	int zo0, zo1;  // local range of planes to produce
};
inline ZRange whole(int nz) { return ZRange{nz, 0, nz, 0, nz}; }

struct Level {
	float *d = nullptr;
	int nx = 0, ny = 0, nz = 0;  // nz = GLOBAL number of planes of the level
	float unit = 1.f;   // 2^octave (Src/cUtil.cc:215-225)
	float scale = 0.f;  // scale-space location (Src/cUtil.cc:207-210)
	int bz = 0;         // planes held in the buffer (0 = all nz); z-slab contexts hold [zoff, zoff+bz)
	int zoff = 0;       // global z of buffer plane 0
	int planes() const { return bz ? bz : nz; }
	size_t n() const { return (size_t)nx * ny * planes(); }
	ZRange zr(int zo0, int zo1) const { return ZRange{planes(), zoff, nz, zo0, zo1}; }
	ZRange zr_all() const { return ZRange{planes(), zoff, nz, 0, planes()}; }
};

// One DoG extremum / keypoint while it lives on the device.
struct DevKp {
	int x, y, z;
	int octave, level;
	float scale;
	int code;   // Assign_Orientation_Imp result: 1 ok, -1 weak gradient, -2 eigen ratio, -3 corner; 0 = not run
	int slot;   // index among accepted keypoints (exclusive scan of code==1), -1 otherwise
	float win[3];
	float eigvalue[3];
	float eigvector[9];
	float rot[9];   // as written by orientation (not transposed)
	float st[9];
};

// Per-face constants of the icosahedron intersection test, hoisted from cart2bary
// (reference Src/cSIFT3D.cc:1592-1637): pure functions of the mesh.
struct FaceConst {
	float e1[3], e2[3], t[3], q[3];
	float qe2;  // dot(q, e2)
	int idx[3];
	float centre[3];  // face centroid (prediction table only)
};

// Gaussian window weight tables indexed by the INTEGER squared voxel offset
// n = dx^2+dy^2+dz^2 (the fp32 squared distance n*u^2 is exact), one per (octave, level 1..3):
// value = weight, or -1 outside the window sphere.  Built on the host with the same libm expf the
// reference uses, so the weights are bit-identical to the CPU path (no device exp on the path).
struct WinLut {
	int off;      // offset into the lut pool
	int len;      // entries; n >= len is outside
	int nin;      // largest n that is still inside the sphere (tables are monotone)
	float radius; // win_radius (fp32) for the box bounds
	float sigma;
	// descriptor tables only: the entries are weight * 0.5 / unit (exact: the unit is a power of two, so the reference's
	// ((0.5*d) * (1/u)) * w rounds like d * (w * 0.5/u)), and the histogram bins of k_describe are 32-bit fixed point in units of
	// 1/fix_scale (a power of two chosen so that no bin can overflow an int32 for this window size, see build_luts)
	float fix_scale;
	float wsum;  // sum of the (unscaled) Gaussian weights over the lattice points of the window sphere (gradient-mass estimate)
	// orientation tables only: the lattice points of the window sphere, plane by plane, as 32-bit words in the same pool
	// (-1: none).  At list_off: 2 list_R + 2 running starts (plane dz = -list_R .. list_R, then the end), then the entries
	// (dx + 128) | (dy + 128) << 8 | n << 16 in (dz, dy, dx) order.  k_orient walks this list instead of the window's box.
	int list_off, list_R;
};
constexpr int kMaxDescLut = 1536;  // descriptor window table entries staged in LDS (default params: 1293)

// Face lookup of the descriptor kernel by the SYMMETRY of the icosahedron (r03; see face_lookup in kernels_desc.hip).  The mesh
// (0, +-1, +-phi), (+-1, +-phi, 0), (+-phi, 0, +-1) is invariant under sign flips of the coordinates, so the direction |g| (componentwise)
// lies in the canonical octant face (0,1,phi) (1,phi,0) (phi,0,1) or in the half of one of its three neighbours that reaches into
// the positive octant; key = type * 8 + sign bits of g, type 0 = octant face, 1 + m = the neighbour across the edge opposite
// vertex m.  Per key: the mesh face, and for each of the three ROLES of the canonical solution (see face_lookup) the vertex whose
// histogram bins receive that weight (Tri::idx of the reference, Src/cUtil.cc:36-55, winding quirk included) and the position j of
// that weight in the reference's bary[0..2].
struct FaceSym {
	int face[32];
	int vert[32][3];
	int slot[32][3];
};

// ---- kernels_pyramid.hip -------------------------------------------------------------------
void launch_absmax(const float *src, size_t n, unsigned *d_max_bits, hipStream_t st);
void launch_scale_by_max(float *data, size_t n, const unsigned *d_max_bits, hipStream_t st);
void launch_dog_from_gss(const float *hi, const float *lo, float *dog, size_t n, hipStream_t st);  // dog = (hi - lo) * (-1)
// separable pass along AXIS (0 x, 1 y, 2 z); for AXIS==2 optionally also emits
// dog = -(dst - prev) and accumulates max|dog| (bits) into d_dogmax.
void launch_conv_axis(int axis, const float *src, float *dst, int nx, int ny, int nz, const Taps &t,
                      const float *prev, float *dog, unsigned *d_dogmax, hipStream_t st);
// fused single-pass level kernel (kernels_march.hip): descending z-march with scatter accumulators; false => not applicable (half
// width without an instantiation, planes smaller than a tile): the caller takes the separable kernels above
// level 0 of the next octave, written by the march kernel together with the seed level (DownSample_3D fused into the producer)
struct MarchHalf { float *d = nullptr; int nx = 0, ny = 0, nz = 0; };
bool march_half_ok(int nx, int ny, const ZRange &zr);
bool march_applicable(int nx, int ny, int nzg, const Taps &t);
bool launch_march_level(const float *src, float *dst, float *dog, unsigned *dogmax, int nx, int ny, const ZRange &zr, const Taps &t,
                        hipStream_t st, int plan_slots = 0, int prio = 0, const MarchHalf *half = nullptr);
void launch_copy16(const float *src, float *dst, size_t nfloats, hipStream_t st);  // float4 copy (bandwidth ceiling probe)
// simulated transport of the z-slab driver (sharded.hip): one exchange step's plane ranges in one launch; MAX over np arrays of n <= 64 floats, in place in all of them
constexpr int kCopySegs = 32, kMaxMergePtrs = 16;
struct CopySegs { const float *src[kCopySegs]; float *dst[kCopySegs]; size_t floats[kCopySegs]; int n = 0; };
struct MaxMerge { float *p[kMaxMergePtrs]; int np = 0, n = 0; };
void launch_copy_segments(const CopySegs &a, hipStream_t st);
void launch_max_merge(const MaxMerge &a, hipStream_t st);
// ---- kernels_small.hip: every level of the SMALL octaves (16^3-class and below) in one launch of one workgroup ----
constexpr int kSmallMaxOct = 4, kSmallMaxLv = 8;
struct SmallOct { float *g[kSmallMaxLv]; float *dog[kSmallMaxLv]; int nx, ny, nz; };  // Gaussian / DoG level buffers of one octave
struct SmallArgs {
	SmallOct oct[kSmallMaxOct];
	int noct, ng, nd, seed;          // octaves in the launch; Gaussian / DoG levels per octave; the seed level (num_kp_levels)
	unsigned build_mask, dog_mask;   // bit i: Gaussian level i is built; bit j: DoG level j is written (and its abs-max taken)
	unsigned *dogmax;                // [octave of the launch][nd] max|DoG| bits
	int hwp;                         // half width the launch runs every level with (6 or 8): narrower kernels are padded with zero taps
	float w[kSmallMaxLv][9];         // w[i][k], k = hwp - |d|: tap of level i at distance |d| from the centre (symmetric kernel), 0 beyond its own half width
	const float *parent;             // != null: level 0 of the first octave = every second voxel of this level (pnx x pny planes), formed by the launch
	int pnx, pny;
};
bool small_octave_fits(int nx, int ny, int nz, int max_hw);
int small_padded_hw(int max_hw);
void launch_small_octaves(const SmallArgs &a, hipStream_t st);
void launch_downsample(const float *src, int snx, int sny, float *dst, int nx, int ny, int nz, hipStream_t st);

// ---- kernels_detect.hip --------------------------------------------------------------------
struct DetectBufs {
	unsigned long long *masks;  // one ballot word per wave
	unsigned *block_counts;     // per block
	unsigned *block_offsets;    // exclusive scan (+ running base)
	unsigned *total;            // [0] running total over all levels, [1] overflow flag
	// lazy last Gaussian level (see DetectLevels::lazy_src): candidates that passed seven of the eight neighbour tests and wait
	// for the value of the level that is not materialised: entry = local voxel index | (tested as maximum) << 31
	unsigned *prov = nullptr;
	unsigned *prov_count = nullptr;
	unsigned prov_cap = 0;
};
// the keypoint levels of one octave (DoG levels 1..num_kp_levels), handled by one launch of each detect kernel
constexpr int kMaxKpLevels = 5;
struct DetectLevels {
	const float *cur[kMaxKpLevels], *prev[kMaxKpLevels], *next[kMaxKpLevels];
	// Elided DoG levels: the first and the last DoG level of an octave are read ONLY as the centre-voxel neighbour of extremum
	// candidates (Src/cSIFT3D.cc:889-896), so the single-volume path does not materialise them; the candidate test forms the
	// value from the two Gaussian levels exactly like Sub does, (hi - lo) * (-1) (Src/cSIFT3D.cc:875).  Null = materialised.
	const float *prev0_hi, *prev0_lo;        // prev of level slot 0 = DoG[0] = (G[1] - G[0]) * (-1)
	const float *nextl_hi, *nextl_lo;        // next of level slot nextl_slot = DoG[nd-1] = (G[nd] - G[nd-1]) * (-1)
	int nextl_slot;
	// The LAST Gaussian level of an octave, G[nd], is read by nothing but that formula -- not by a keypoint window (levels 1..nd-2),
	// not by the next octave (seeded by G[num_kp_levels]) -- i.e. only at the few thousand voxels of level slot nextl_slot that
	// pass the other seven tests.  lazy_src != null: the level was NOT built; those voxels are parked (DetectBufs::prov) and
	// k_lazy_next evaluates G[nd] = gauss_z(gauss_y(gauss_x(lazy_src))) at each of them with the arithmetic of the level kernels
	// (same tap chain, same boundary terms), then finishes the test.  lazy_src = G[nd-1], which is also nextl_lo.
	const float *lazy_src;
	const unsigned *absmax_bits[kMaxKpLevels];
	int level_id[kMaxKpLevels];
	float scale[kMaxKpLevels];
};
constexpr int kLazySlots = 52;  // k_lazy_next: 2 * (2 hw + 1) source slots per axis (boundary voxels) fit for hw <= 12 (r06; 36 / hw <= 8 before)
void launch_detect_mark(const DetectLevels &L, int nlevels, int nx, int ny, const ZRange &zr, float peak_thresh, int octave,
                        const DetectBufs &b, hipStream_t st, const Taps *lazy_taps);
void launch_detect_emit(const DetectLevels &L, int nlevels, int nx, int ny, const ZRange &zr, int octave, const DetectBufs &b, DevKp *out,
                        unsigned cap, hipStream_t st);
struct DetectEmitItem { const DetectLevels *L; int nlevels, nx, ny; ZRange zr; int octave; const DetectBufs *b; };
void launch_detect_emit_multi(const DetectEmitItem *items, int n, DevKp *out, unsigned cap, unsigned *total, hipStream_t st);
void launch_detect_octave(const DetectLevels &L, int nlevels, int nx, int ny, const ZRange &zr, float peak_thresh, int octave,
                          const DetectBufs &b, DevKp *out, unsigned cap, hipStream_t st, const Taps *lazy_taps = nullptr);

// ---- kernels_orient.hip --------------------------------------------------------------------
// Level pointers reach the keypoint kernels through a table in device memory, so the compiler cannot prove their address
// space and would emit FLAT loads, which count on both vmcnt and lgkmcnt and force a full s_waitcnt after every load
// group (no software pipelining).  Casting to the global address space gives global_load + in-order vmcnt.
#ifdef __HIPCC__
typedef const float __attribute__((address_space(1))) *gfloat_p;
__device__ __forceinline__ gfloat_p as_global(const float *p) { return (gfloat_p)p; }
#endif

struct LevelRef {
	const float *d;   // local plane 0 of the buffer
	int nx, ny, nz;   // nz = GLOBAL number of planes (window clipping, Src/cSIFT3D.cc:951-955)
	float unit;
	int zoff;         // global z of local plane 0 (0 for whole volumes)
};
// part_rank / part_world: only extrema with index % part_world == part_rank are oriented (the others get code 0)
void launch_orient(DevKp *kps, int *codes, const unsigned *d_count, unsigned cap, const LevelRef *d_levels /*[noct*8]*/,
                   const WinLut *d_luts, const float *d_lutpool, float max_eig, float corner, int part_rank, int part_world,
                   int *redo_list /* cap ints: the extrema whose sums are redone in the reference's order */, unsigned *redo_count,
                   hipStream_t st);
constexpr int kOrientWords = 34;  // code + win[3] + eigvalue[3] + eigvector[9] + rot[9] + st[9]  (== SIFT3D_ORIENT_WORDS)
void launch_orient_pack(const DevKp *kps, const unsigned *d_count, unsigned cap, int *dst, int part_rank, int part_world, hipStream_t st);
void launch_orient_unpack(DevKp *kps, int *codes, const unsigned *d_count, unsigned cap, const int *src, hipStream_t st);
// slot = order-preserving index among the accepted keypoints; order[slot] = extremum index
size_t slots_scratch_words();  // unsigned words of per-context scratch launch_slots needs
void launch_slots(DevKp *kps, const int *codes, const unsigned *d_count, unsigned cap, unsigned *d_nkp, int *order, unsigned kp_cap,
                  unsigned *scratch, hipStream_t st);

// ---- kernels_desc.hip ----------------------------------------------------------------------
hipError_t upload_faces(const FaceConst *faces, const FaceSym *sym);  // into the current device's __constant__ memory
// part_rank / part_world: only keypoints with slot % part_world == part_rank are described (multi-GPU split of
// replicated octaves); 0 / 1 = all
// r04, runs with few keypoints: one keypoint's window is marched by S workgroups (S = 8 / 4 / 2 by the keypoint count, which lives
// on the device: every workgroup derives the same S).  Each part adds its integer histogram into gacc (global atomics: exact, order
// free), the part that arrives last (gdone) normalises -- the same integers an unsplit run sums in LDS, so the descriptors are
// bit-identical.  A keypoint whose first fixed-point unit fails (k_describe's second pass) is repeated by its finisher alone.
// cap = positions (of the processing order) the scratch holds; runs with more keypoints than that are never split.
struct DescSplit {
	int *gacc = nullptr;            // [cap][768], zero between runs (the finisher clears what it read)
	float *gmass = nullptr;         // [cap][8] gradient mass per part
	unsigned *gdone = nullptr;      // [cap] parts arrived
	unsigned cap = 0;
};
// r05, the z-slab sharding: this rank's z PART of descriptor windows.  Records come in up to kDescSegs lists -- the rank's own keypoints and
// those of the z-neighbours whose windows reach into it.  [zc0, zc1): the planes this rank owns (global z of the level); a list's
// [o0, o1): the planes its OWNER owns.  The owner marches [o0 - H[level], o1 + H[level]) -- what its level buffer holds --, every other
// rank its owned planes outside that range.  hist[n][768] int32 (descriptor element order, the fixed-point sums of this part), mass[n]
// (its gradient mass); units: per-record fixed-point unit of a second round (entries <= 0 and a null pointer: the first-pass rule, a
// function of the record alone).
constexpr int kDescSegs = 6;
struct DescSeg {
	const DevKp *recs = nullptr;
	const float *units = nullptr;
	int *hist = nullptr;
	float *mass = nullptr;
	unsigned first = 0, n = 0;  // first = records of the lists in front of this one
	int o0 = 0, o1 = 0;
};
struct DescPartial {
	int zc0 = 0, zc1 = 0, nseg = 0;
	int H[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // per keypoint level
	DescSeg seg[kDescSegs];
};
void launch_export_records(const DevKp *ext, const int *order, unsigned n, DevKp *dst, hipStream_t st);
void launch_describe_partial(const LevelRef *d_levels, const WinLut *d_luts, const float *d_lutpool, const DescPartial &pp,
                             unsigned *d_work /* two device words, zeroed by the launch */, hipStream_t st, bool lut_in_lds = true);
// the owner's finish: nparts <= kDescSegs partial results of its n records (its own part and its neighbours', ascending rank: the masses
// are added in that order); rows go to d_desc[recs[k].slot]; records whose unit failed get redo[k] = 1, units_next[k] = the exact unit,
// and are counted in d_counters[0] (unless final_round)
void launch_describe_finish(const DevKp *recs, unsigned n, const LevelRef *d_levels, const WinLut *d_luts, int nparts, const int *const *d_hist, const float *const *d_mass,
                            const float *d_units, bool final_round, float *d_desc, int *d_redo, float *d_units_next, unsigned *d_counters,
                            hipStream_t st);
void launch_describe(const DevKp *kps, const unsigned *d_count, unsigned cap, const LevelRef *d_levels,
                     const WinLut *d_luts, const float *d_lutpool, float *d_desc, unsigned kp_cap, int part_rank, int part_world,
                     const int *order, const unsigned *d_nkp, unsigned *d_work /* device counter, zeroed by the launch */,
                     hipStream_t st, bool lut_in_lds = true, const DescSplit *split = nullptr);
void launch_face_lookup(const float *d_g3, int n, int route, int *d_face, float *d_bary3, hipStream_t st);  // sift3d_debug_face_lookup
void launch_finalize(const DevKp *kps, const unsigned *d_count, unsigned cap, int transposed,
                     sift3d_keypoint *d_out, float *d_xyz, unsigned kp_cap, hipStream_t st);

// ---- kernels_match.hip ---------------------------------------------------------------------
// best / second-best dot of every listed row of A against all m rows of B (calMatches);
// d_row_ids == nullptr means rows 0..nrows-1; outputs are indexed by the ORIGINAL row id.
// d_part: scratch for the per-column-split partial top-4 lists, 8 B * 4 * 16 * nrows
// near-tie guard scratch (see kernels_match.hip): s4[nrows], squared norms of the A rows (indexed by ORIGINAL row id), the maximal
// squared norm of the B rows (bits), redo list [1 + nrows]
struct MatchGuard {
	float *s4; const float *a_n2; const unsigned *b_n2max; int *redo;
	bool small_offsets = false;  // both descriptor matrices are smaller than 4 GB: k_scores_top4 may stage them by LDS-DMA (32-bit offsets)
};
int match_rows_device(const float *d_a, const int *d_row_ids, int nrows, const float *d_b, int m, int *d_cand /*nrows*4*/,
                      void *d_part, float *d_gd, float *d_sd, int *d_gi, int *d_si, const MatchGuard &g, hipStream_t st);
int match_redo_rows();  // rows the last sift3d_match re-scored exactly (near-tie guard)

// code-object preload of the translation units whose kernels would otherwise be loaded by the first KpSiftAlgorithm of a process
void preload_march_kernels();
void preload_small_kernels();
void preload_detect_kernels();
void preload_orient_kernels();
void preload_desc_kernels();
void preload_match_kernels();

// ---- test hooks and development switches --------------------------------------------------------
// Test hooks (include/sift3d_hip.h, sift3d_test_hook): process-wide integers that force code paths ordinary inputs rarely
// reach, so that the parity tests execute every branch of the product.  hook(SIFT3D_HOOK_x) reads the current value.
int hook(int which);
// Tuning values of the launch planning: compile-time defaults.  Only a -DS3D_DEV_SWITCHES build (scripts/build_variant.sh, the
// A/B measurement scripts) lets an environment variable override them; the product library never reads the environment.
int dev_tune_i(const char *env_name, int dflt);
double dev_tune_d(const char *env_name, double dflt);

// ---- staging.hip: pageable host memory <-> device through pinned, double-buffered chunks --------
int staged_h2d(void *d_dst, const void *h_src, size_t bytes, int device, hipStream_t st);  // stream-ordered on return
int staged_d2h(void *h_dst, const void *d_src, size_t bytes, int device, hipStream_t st);  // complete on return
struct D2HSeg { void *h_dst; const void *d_src; size_t bytes; };
int staged_d2h_v(const D2HSeg *segs, int nseg, int device, hipStream_t st);                   // several pairs through one pipeline

// error plumbing
void set_last_error(const std::string &s);
#define S3D_HIP(call)                                                                                   \
	do {                                                                                                \
		hipError_t e_ = (call);                                                                         \
		if (e_ != hipSuccess) {                                                                         \
			s3d::set_last_error(std::string(#call) + ": " + hipGetErrorString(e_));                     \
			return SIFT3D_ERR_HIP;                                                                      \
		}                                                                                               \
	} while (0)

}  // namespace s3d
