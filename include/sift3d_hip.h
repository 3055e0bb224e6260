/*
 * sift3d_hip.h -- C-ABI of the MI355X-native 3D SIFT library (lib: 3dsift_amd/libsift3d_hip.so).
 *
 * This is the drop-in boundary: plain C types, opaque handle, int error codes, no exceptions, no
 * torch types.  The C++ shell in 3dsift_amd/host/ (namespace CPUSIFT, same class / method names as
 * the reference) and the python ctypes binding in 3dsift_amd/capi.py are both thin layers over
 * exactly these entry points.  Each entry point cites the reference interface it replaces
 * (paths relative to the reference repo, 3DSIFT/...).
 *
 * Pointers are HOST pointers unless the name says otherwise (d_ prefix / "device" flag).
 * All volumes are fp32, x fastest: idx = x + nx*(y + ny*z)   (Include/Util/cTexImage.h:5,35).
 * A handle serialises its calls on one HIP stream; different handles are independent.
 */
#ifndef SIFT3D_HIP_H
#define SIFT3D_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIFT3D_DESC_NUMEL 768 /* Include/cSIFT3D.h:27 DESC_NUMEL = 4*4*4*12 */

/* error codes (the reference has none: it prints and carries on, Include/Util/cMemManager.h:35-39) */
enum {
	SIFT3D_OK = 0,
	SIFT3D_ERR_ARG = 1,      /* bad argument */
	SIFT3D_ERR_NO_DEVICE = 2,/* no usable HIP device: the library never falls back to the CPU */
	SIFT3D_ERR_HIP = 3,      /* a HIP runtime call failed (see sift3d_last_error) */
	SIFT3D_ERR_STATE = 4,    /* call out of order (e.g. results requested before run) */
	SIFT3D_ERR_CAPACITY = 5  /* an internal device list overflowed even after regrowing */
};

typedef struct sift3d_ctx *sift3d_handle;

/* Constructor parameters; defaults = Include/cSIFT3D.h:13-20 (factory default args :187-202). */
typedef struct sift3d_params {
	int num_kp_levels;      /* NUM_KP_LEVELS 3 */
	float sigma_default;    /* SIGMA_DEFAULT 1.6 */
	float sigma_n_default;  /* SIGMA_N_DEFAULT 1.15 */
	float peak_thresh;      /* PEAK_THRESH 0.1 */
	float max_eig_thres;    /* EIG_THRES 0.9 */
	float corner_thresh;    /* CORNER_THRESH 0.4 */
} sift3d_params;

/* POD mirror of CPUSIFT::Keypoint without the desc pointer (Include/cSIFT3D.h:54-70): 168 bytes. */
typedef struct sift3d_keypoint {
	float x, y, z;
	float scale;
	int octave, level;
	float rx, ry, rz;
	float win[3];
	float eigvalue[3];
	float eigvector[9];
	float Rotation[9];   /* returned TRANSPOSED after the descriptor stage, like Src/cSIFT3D.cc:1214 */
	float str_tensor[9];
} sift3d_keypoint;
/* layout guard of the record that crosses the boundary (SURVEY 8a-1 lists the offsets of CPUSIFT::Keypoint; this POD is that record
 * without its trailing desc pointer): a compiler / packing change breaks the build, not the results */
#if defined(__cplusplus)
#define SIFT3D_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define SIFT3D_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif
SIFT3D_STATIC_ASSERT(sizeof(sift3d_keypoint) == 168, "sift3d_keypoint must be 168 bytes");
SIFT3D_STATIC_ASSERT(offsetof(sift3d_keypoint, scale) == 12 && offsetof(sift3d_keypoint, octave) == 16 && offsetof(sift3d_keypoint, rx) == 24 &&
                     offsetof(sift3d_keypoint, win) == 36 && offsetof(sift3d_keypoint, eigvalue) == 48 && offsetof(sift3d_keypoint, eigvector) == 60 &&
                     offsetof(sift3d_keypoint, Rotation) == 96 && offsetof(sift3d_keypoint, str_tensor) == 132, "sift3d_keypoint field offsets");
SIFT3D_STATIC_ASSERT(sizeof(sift3d_params) == 24, "sift3d_params must be 24 bytes");

void sift3d_default_params(sift3d_params *p);

/* Replaces CSIFT3DFactory::CreateCSIFT3D(float*, nx, ny, nz, ...) + CSIFT3D::CSIFT3D
 * (Src/cSIFT3D.cc:103-110, 146-163): copies the caller's volume (caller keeps ownership), uploads it
 * to `device` and max-abs normalises it there (data_scale, Src/cUtil.cc:536-564).  Also reserves the
 * whole device arena (both pyramids, scratch, keypoint lists) so that sift3d_run allocates nothing.
 * volume_on_device != 0: `volume` is a device pointer on `device` (copied D2D). */
int sift3d_create(sift3d_handle *out, const float *volume, int nx, int ny, int nz,
                  const sift3d_params *params, int device, int volume_on_device);

/* Replaces CSIFT3D::~CSIFT3D (Src/cSIFT3D.cc:140-144). */
int sift3d_destroy(sift3d_handle h);

/* Replaces CSIFT3D::KpSiftAlgorithm (Src/cSIFT3D.cc:165-235): whole pipeline, results stay on the
 * device until sift3d_get_keypoints.  Returns after the stream has drained. */
int sift3d_run(sift3d_handle h);

/* KpSiftAlgorithm split in two (no reference counterpart: the reference's call blocks, Src/cSIFT3D.cc:165-235): sift3d_run_async
 * enqueues the whole pipeline on the handle's own streams and returns without waiting; sift3d_wait completes it (results, stage
 * times, the rare list regrow + rerun).  One host thread can keep several handles in flight on one GPU -- BASELINE configs[2] / [4]
 * extract several volumes: the pyramid of one is bound by memory while the descriptors of another are bound by instruction issue.
 * Every accessor of a handle with a run in flight completes it first; sift3d_wait without a run in flight returns SIFT3D_OK. */
int sift3d_run_async(sift3d_handle h);
int sift3d_wait(sift3d_handle h);
/* sift3d_run_async whose pipeline starts when the orientation stage of `after` (a handle with a run in flight on the same GPU) has ended:
 * the memory-bound front of this volume (pyramid, extrema, orientation) runs beside the descriptor stage of the volume before it, which is
 * bound by instruction issue and the LDS (Example.cpp:21-44 extracts two volumes back to back).  after == NULL / nothing in flight: plain
 * sift3d_run_async. */
int sift3d_run_async_after(sift3d_handle h, sift3d_handle after);

/* Replaces calling the public stage methods one by one (Include/cSIFT3D.h:157-165); `upto`:
 * 1 Initialize+Build_Gaussian_Scale_Space(+fused DoG), 2 Build_DOG_Scale_Space, 3 Detect_KeyPoints,
 * 4 Assign_Orientation, 5 Extract_Description.  Used by the parity tests. */
int sift3d_run_stages(sift3d_handle h, int upto);

/* Replaces SIFT_TimerPara m_timer (Include/Util/common.h:22-41; filled Src/cSIFT3D.cc:228-233).
 * Seconds, from HIP events on the handle's stream:
 * t[0] total, t[1] allocation(=0, arena is reserved at create), t[2] GSS(+fused DoG), t[3] DoG(=0 when
 * fused), t[4] detect, t[5] orientation, t[6] description, t[7] release(=0). */
int sift3d_stage_times(sift3d_handle h, double t[8]);

/* Replaces CSIFT3D::GetKeypoints (Src/cSIFT3D.cc:1686-1688).  Order = reference order:
 * (octave, level, z, y, x) scan order (Src/cSIFT3D.cc:373-416, 459-466). */
int sift3d_num_keypoints(sift3d_handle h, int *n);
int sift3d_get_keypoints(sift3d_handle h, sift3d_keypoint *out, float *desc /* n*768, may be NULL */);

/* Device-resident results for a matcher that never leaves the GPU: row-major n*768
 * descriptors and n*3 (rx,ry,rz); valid until the next run / destroy. */
int sift3d_device_results(sift3d_handle h, const float **d_desc, const float **d_xyz, int *n);

/* Checking accessors, replace GET_GSS / GET_DOG / GET_LEVEL (Include/cSIFT3D.h:167-177). */
int sift3d_num_octaves(sift3d_handle h, int *n);
int sift3d_level_info(sift3d_handle h, int is_dog, int idx, int dims3[3], float units3[3], float *scale);
int sift3d_copy_level(sift3d_handle h, int is_dog, int idx, float *out);
int sift3d_copy_input(sift3d_handle h, float *out);
int sift3d_num_extrema(sift3d_handle h, int *n);
int sift3d_get_extrema(sift3d_handle h, sift3d_keypoint *out);
/* per-extremum result code of Assign_Orientation_Imp (1 / -1 / -2 / -3), Src/cSIFT3D.cc:913-1138 */
int sift3d_get_orientation_codes(sift3d_handle h, int *codes);

/* Replaces the free function GaussianSmooth_3D (Include/cSIFT3D.h:212; Src/cSIFT3D.cc:535-622) on a
 * host volume (unit-level parity tests). */
int sift3d_gaussian_smooth(const float *src, int nx, int ny, int nz, float sigma, float *dst, int device);
/* Replaces the free function DownSample_3D (Include/cSIFT3D.h:210; Src/cSIFT3D.cc:506-533): dst(n, m, k) = src(2n, 2m, 2k) for every
 * voxel of the caller-sized dst (2 (nx - 1) < snx etc.), host volumes. */
int sift3d_downsample(const float *src, int snx, int sny, int snz, float *dst, int nx, int ny, int nz, int device);
/* Replaces the free function Sub (Include/cSIFT3D.h:218; Src/cSIFT3D.cc:849-882): dog = (cur - prev) * (-1), n voxels, host volumes. */
int sift3d_dog_sub(const float *prev, const float *cur, size_t n, float *dog, int device);
/* Replaces the free function GaussianSmooth_3D_Imp (Include/cSIFT3D.h:214; Src/cSIFT3D.cc:624-788): ONE pass along `dim` (0 x, 1 y, 2 z)
 * with the caller's taps weight[0 .. width) (width odd, <= 129), interior and mirror-boundary rule as in the pipeline; host volumes. */
int sift3d_conv_axis(const float *src, int nx, int ny, int nz, int dim, const float *weight, int width, float *dst, int device);
/* Replace the free functions Assign_Orientation_Imp / Extract_Descriptor_Imp (Include/cSIFT3D.h:224, 228; Src/cSIFT3D.cc:913-1138,
 * 1152-1381) for ONE keypoint on a caller-provided HOST level (nx x ny x nz, isotropic unit = 2^octave): the pipeline's own kernels run on
 * the box of the level the window reaches.  The keypoint sits on a voxel (integral x, y, z), as every keypoint of the pipeline does;
 * anything else is refused.  orient: in x, y, z, scale; out win, eigvalue, eigvector, Rotation (not transposed), str_tensor and *code =
 * the reference's return value (1 / -1 / -2 / -3).  describe: in x, y, z, scale, Rotation as orientation left it (+ str_tensor: first
 * guess of the fixed-point unit only); out desc768 (normalised) and Rotation TRANSPOSED, like Src/cSIFT3D.cc:1214 leaves it. */
int sift3d_orient_keypoint(const float *level, int nx, int ny, int nz, float unit, sift3d_keypoint *kp, float sigma, float max_eig_ratio,
                           float corner_thresh, int device, int *code);
int sift3d_describe_keypoint(const float *level, int nx, int ny, int nz, float unit, sift3d_keypoint *kp, float *desc768, int device);

/* Replaces muBruteMatcher::injectMatch / bijectMatch / enhancedMatch (Src/cMatcher.cc:146-228).
 * mode 1 inject, 2 biject, 3 enhanced.  desc: n*768 / m*768, xyz: n*3 / m*3 (rx,ry,rz).
 * on_device != 0: the four input pointers are device pointers on `device`.
 * Outputs (host, any may be NULL): gIdx/sIdx/gDist/sDist sized n = getGlodenIdx / getSilverIdx /
 * getGlodenDistSquare / getSilverDistSquare (Include/cMatcher.h:69-73); pairs6: up to n rows of
 * (ref rx,ry,rz, tar rx,ry,rz) in ascending ref index (toCvec, Src/cMatcher.cc:99-112). */
int sift3d_match(const float *ref_desc, const float *ref_xyz, int n, const float *tar_desc,
                 const float *tar_xyz, int m, double thresHold, int mode, int on_device, int device,
                 int *gIdx, int *sIdx, float *gDist, float *sDist, float *pairs6, int *npairs,
                 double *seconds /* device time of the call, may be NULL */);
/* The same match on the device-resident results of two extractors (sift3d_device_results), wherever they live: handles on one GPU
 * are matched in place; with `tar` on another GPU of the node its descriptors and coordinates are first copied peer to peer (xGMI)
 * into a scratch of `ref` on ref's device.  This is the building block of BASELINE configs[4] for a single-process C++ caller -- N
 * extractors, one per GPU, all ordered pairs (CPUSIFT::CSIFT3D::AllPairsMatch in the C++ shell; 3dsift_amd/dist.py does the same
 * over torch.distributed with an RCCL all-gather, one process per GPU).  Completes runs in flight on both handles first. */
int sift3d_match_handles(sift3d_handle ref, sift3d_handle tar, double thresHold, int mode, int *gIdx, int *sIdx, float *gDist,
                         float *sDist, float *pairs6, int *npairs, double *seconds);
/* times of the calling thread's last sift3d_match: device_seconds = HIP events around the device work on the matcher's stream
 * (what *seconds returned), wall_seconds = host clock around the whole call (scratch reuse, H2D of host inputs, the O(N) host
 * bookkeeping of Src/cMatcher.cc:81-144 and the D2H of the results included) */
int sift3d_match_times(double *device_seconds, double *wall_seconds);
/* muBruteMatcher's constructor (Include/cMatcher.h:30): the matcher's kernels, stream and first scratch exist before the first call */
int sift3d_match_warmup(int device);

int sift3d_device_count(int *n);

/* ------------------------------------------------------------------------------------------------------------
 * Multi-GPU sharding of ONE large volume (no reference counterpart: the reference is single process).
 * Octave 0 is split into z-slabs, one per rank; every level buffer of a slab context holds the owned global planes
 * [z0, z1) plus `halo` planes on each side, which the CALLER fills by exchanging planes with the z-neighbours
 * (3dsift_amd/slab.py does it with torch.distributed P2P over RCCL).  Boundary rules and keypoint coordinates use
 * global z.  Octaves >= 1 run replicated from the all-gathered G[1][0] in a SEEDED context, with the descriptor
 * work split by keypoint slot.  Results equal the single-GPU results bit for bit (pyramid, extrema) / to the
 * descriptor tolerance.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct sift3d_slab_desc {
	int nx, ny, nz;      /* GLOBAL dims of the volume */
	int z0, z1;          /* owned global planes [z0, z1) of this octave; any integers since r06 (the slab of the octave below then owns [ceil(z0/2), ceil(z1/2))) */
	int halo;            /* margin planes per side; >= 38 for default parameters (descriptor window reach) */
	int noct_total;      /* octaves of the ORIGINAL volume: (int)log2f(min dim) - 2 */
	int octave;          /* absolute octave this context holds (0 = the input octave).  octave > 0: nx,ny,nz,z0,z1 are in
	                      * that octave's voxels and G[octave][0] is provided by the caller (owned planes written by
	                      * sift3d_slab_decimate of the octave above, halo planes exchanged); sift3d_slab_level(h, 0) is a no-op */
} sift3d_slab_desc;

/* smallest admissible halo for `params`: the z reach of a descriptor window in octave 0 (38 for the defaults) */
int sift3d_slab_min_halo(const sift3d_params *params, int *halo);
/* r06: *ok = 1 if slab contexts can hold an octave of these GLOBAL dims with these parameters (every level a slab builds takes the z-march kernel:
 * half widths 2 .. 8, planes of 32 or >= 32 + hw voxels per side, at least 2 hw + 2 planes); first_octave != 0: the base blur of the input too */
int sift3d_slab_admits(const sift3d_params *params, int nx, int ny, int nz, int first_octave, int *ok);
/* floats the caller must provide for the level buffers (input, GSS and DoG levels of octave 0) */
int sift3d_slab_arena_floats(const sift3d_slab_desc *d, const sift3d_params *params, size_t *n);
int sift3d_slab_create(sift3d_handle *out, const sift3d_slab_desc *d, const sift3d_params *params, int device,
                       float *d_arena, size_t arena_floats);
/* kind 0 input, 1 GSS level idx, 2 DoG level idx: offset of the buffer inside the arena (floats), planes held and
 * the global z of its plane 0 (= z0 - halo); plane k of the buffer is global plane zoff + k, planes are nx*ny floats */
int sift3d_slab_buffer(sift3d_handle h, int kind, int idx, size_t *offset_floats, int *planes, int *zoff);
/* copy global planes [zg0, zg1) of the RAW volume into the input buffer (host or device source) */
int sift3d_slab_upload(sift3d_handle h, const float *planes, int zg0, int zg1, int on_device);
int sift3d_slab_input_absmax(sift3d_handle h, float *local_max);   /* over the OWNED planes (data_scale pass 1) */
int sift3d_slab_input_scale(sift3d_handle h, float global_max);    /* v /= max on every held plane (pass 2) */
/* GSS level i (and DoG i-1, local max|DoG i-1|) on the owned planes; needs level i-1 (input for i = 0) valid on
 * [z0-hw_i-1, z1+hw_i] -- i.e. after the caller exchanged that many halo planes.  Asynchronous on the handle's stream;
 * sift3d_slab_sync waits. */
int sift3d_slab_level(sift3d_handle h, int i);
int sift3d_slab_level_hw(sift3d_handle h, int i, int *hw);   /* half width of the Gaussian that produces GSS level i */
int sift3d_slab_halo_planes(sift3d_handle h, int gss_level, int *planes); /* planes of GSS level i its consumers need per side */
int sift3d_slab_sync(sift3d_handle h);
/* Stream-ordered driving (no host synchronisation between the levels): every later call of the handle enqueues on the caller's
 * stream (a hipStream_t of the handle's device; NULL = the handle's own stream again); the DoG maxima travel as nd floats in
 * device memory around the caller's MAX all-reduce; the decimation does not wait for completion. */
int sift3d_set_stream(sift3d_handle h, void *hip_stream);
int sift3d_slab_export_dogmax_device(sift3d_handle h, float *d_dst);
int sift3d_slab_import_dogmax_device(sift3d_handle h, const float *d_src);
int sift3d_slab_decimate_async(sift3d_handle h, float *d_dst);
int sift3d_slab_get_dogmax(sift3d_handle h, float *max5);          /* local maxima of the DoG levels (host) */
int sift3d_slab_set_dogmax(sift3d_handle h, const float *max5);    /* global maxima after the all-reduce */
int sift3d_slab_detect(sift3d_handle h);                            /* extrema of the owned planes (DoG halos of 1 plane exchanged) */
int sift3d_slab_describe(sift3d_handle h);                          /* orientation + descriptors; results via sift3d_get_keypoints */
/* r05 -- descriptor windows split along z over the ranks (no reference counterpart: Src/cSIFT3D.cc:484-502 walks whole windows in one
 * process).  Instead of the 24 / 30 / 38-plane halos of G[1..3] that whole windows reach, the ranks exchange keypoint RECORDS
 * (sift3d_slab_record_bytes each) with the z-neighbours within sift3d_slab_desc_reach planes; every rank marches, for its own and for
 * the foreign records, its part of the window planes (sift3d_slab_describe_partial: 768 int32 sums + the part's gradient mass per record);
 * the owner adds the parts' integers -- the sums the single-volume run forms -- and the masses in rank order, and finishes
 * (sift3d_slab_describe_finish).  A record whose fixed-point unit fails is flagged and repeated once by all parts with the exact unit.
 * sift3d_slab_set_desc_partial makes sift3d_slab_halo_planes answer with the orientation window's reach for G[1..levels]. */
int sift3d_slab_set_desc_partial(sift3d_handle h, int on);
/* r06 -- ghost zones: sift3d_slab_level(h, i) then produces level i on [z0 - g_i, z1 + g_i) with g_i shrinking level by level down to what the
 * windows and the extremum test read, from an input (octave 0) or a level 0 (octave > 0) that holds sift3d_slab_min_halo_ghost planes per side --
 * and NO halo of any level has to be exchanged (the exchange between consecutive levels is the one a slab's level chain waits for).  Costs the
 * levels' work on the ghost planes (defaults, 64-plane slabs: + 60 %); every plane holds what its owner computes for it, results unchanged. */
int sift3d_slab_set_ghost(sift3d_handle h, int on);
int sift3d_slab_min_halo_ghost(const sift3d_params *params, int partial_windows, int *halo);
int sift3d_slab_min_halo_partial(const sift3d_params *params, int *halo);  /* planes per side a level buffer needs in that mode */
int sift3d_slab_record_bytes(int *bytes);
int sift3d_slab_desc_reach(sift3d_handle h, int *planes);
int sift3d_slab_orient(sift3d_handle h);                            /* orientation of the owned extrema; then sift3d_num_keypoints */
int sift3d_slab_export_records(sift3d_handle h, void *d_dst);       /* accepted keypoints, processing order, device memory */
/* nlists record lists in one launch: the rank's own keypoints and those of its z-neighbours.  owner_z0/1[i]: the planes list i's owner owns
 * (the owner marches the window planes its level buffers hold, every other rank its owned planes outside that range). */
int sift3d_slab_describe_partial(sift3d_handle h, int nlists, const void *const *d_records, const int *n,
                                 const float *const *d_units /* NULL, or per list NULL / the second round's units */,
                                 int *const *d_hist /* [n[i]][768] each */, float *const *d_mass /* [n[i]] each */, const int *owner_z0,
                                 const int *owner_z1);
int sift3d_slab_describe_finish(sift3d_handle h, const void *d_records, int n, int nparts /* <= 6 */,
                                const int *const *d_hist /* the parts of the n records: the owner's and its neighbours', ascending rank */,
                                const float *const *d_mass, const float *d_units, int final_round, int *d_redo /* [n] out */,
                                float *d_units_next /* [n] out */, int *n_redo);
int sift3d_slab_orient_launch(sift3d_handle h);                     /* sift3d_slab_orient as two calls (several ranks in one process) */
int sift3d_slab_orient_count(sift3d_handle h, int *n_kp);
/* r06 -- the same stages without a host read-back in between (the native driver's critical path): sift3d_slab_keypoints_launch enqueues
 * Detect_KeyPoints + Assign_Orientation of the owned planes and the read-back of their counts; sift3d_slab_keypoints_count waits for it
 * (a list that overflowed is regrown and both stages repeated, blocking).  sift3d_slab_describe_finish_launch is the first round's finish
 * with the count of flagged records read back asynchronously; sift3d_slab_describe_finish_count waits for it (0: results complete). */
int sift3d_slab_keypoints_launch(sift3d_handle h);
int sift3d_slab_keypoints_count(sift3d_handle h, int *n_kp);
int sift3d_slab_describe_finish_launch(sift3d_handle h, const void *d_records, int n, int nparts, const int *const *d_hist,
                                       const float *const *d_mass, int *d_redo /* [n] out */, float *d_units_next /* [n] out */);
int sift3d_slab_describe_finish_count(sift3d_handle h, int *n_redo);
/* whole descriptor windows of the slab's own keypoints from its own level buffers (halo >= sift3d_slab_min_halo), enqueued behind
 * sift3d_slab_keypoints_launch / _count: complete when the handle's stream has drained */
int sift3d_slab_describe_launch(sift3d_handle h);
/* DownSample_3D of the owned planes of G[octave][num_kp_levels] -> d_dst = (nx/2) x (ny/2) x ((z1-z0)/2) floats (device):
 * the owned planes of level 0 of the next octave (a sharded slab context of octave+1, or the all-gather buffer of the tail) */
int sift3d_slab_decimate(sift3d_handle h, float *d_dst);

/* Seeded context: octaves octave_base.. of a volume whose G[octave_base][0] (dims nx,ny,nz) the caller provides */
int sift3d_create_seeded(sift3d_handle *out, int nx, int ny, int nz, int octave_base, int noct_total,
                         const sift3d_params *params, int device);
int sift3d_seed_upload(sift3d_handle h, const float *level0, int on_device);
/* device address of that level 0 (nx*ny*nz floats): a driver that gathers the seed level writes it in place on the stream it gave the
 * handle (sift3d_set_stream) and follows with sift3d_run_async -- no staging copy, no host synchronisation */
int sift3d_seed_buffer(sift3d_handle h, float **d_level0, size_t *floats);
/* only keypoints with slot % world == rank are described by this handle (rows of the others stay zero) */
int sift3d_set_describe_partition(sift3d_handle h, int rank, int world);
/* Partitioned orientation of a replicated context: sift3d_run_partial_orientation runs the pyramid, the extrema scan and
 * Assign_Orientation for the extrema k with k % world == rank only; sift3d_export_orientation_device packs the results as
 * SIFT3D_ORIENT_WORDS int32 words per extremum (zero rows for the extrema of other ranks), so that an integer
 * all-reduce(SUM) over the ranks yields every row exactly; after sift3d_import_orientation_device, sift3d_run_describe
 * runs Extract_Description for this handle's share of the keypoint slots. */
#define SIFT3D_ORIENT_WORDS 34
int sift3d_run_partial_orientation(sift3d_handle h);
int sift3d_export_orientation_device(sift3d_handle h, int *d_dst /* num_extrema * SIFT3D_ORIENT_WORDS */);
int sift3d_import_orientation_device(sift3d_handle h, const int *d_src);
int sift3d_run_describe(sift3d_handle h);
/* D2D copies between the handle's results and caller-owned device buffers (n*768, n*3 floats): lets a communication
 * layer that only addresses its own allocations all-reduce the partitioned descriptor rows and hand them back */
int sift3d_export_device(sift3d_handle h, float *d_desc_dst, float *d_xyz_dst);
int sift3d_import_descriptors_device(sift3d_handle h, const float *d_desc_src);

/* ------------------------------------------------------------------------------------------------------------
 * Native driver of the z-slab sharding (3dsift_amd/csrc/sharded.hip): ONE host volume over the GPUs of a node, with the call
 * shape of the single-GPU path -- create (copy + normalise, Src/cSIFT3D.cc:146-163), run (KpSiftAlgorithm, :165-235), read
 * back (GetKeypoints, :1686-1688; reference order).  devices[ndev]: one rank per listed GPU, halo exchange over RCCL (ncclSend /
 * ncclRecv between z-neighbours over xGMI, one host thread per GPU; librccl is opened at run time).  sim_ranks > 0 (ndev == 1):
 * that many ranks simulated on the one device -- device copies instead of sends -- which is how 1-GPU boxes test the driver.
 * sharded_octaves: octaves split into slabs (0 = every octave of at least 2^22 voxels and 16 planes per rank, at least two); the remaining
 * octaves run ONCE, on the last rank, from a seed level gathered there.
 * Results equal the single-GPU results: pyramid / extrema / orientation bit for bit, descriptors bit for bit as well (integer
 * histograms).  The C++ shell reaches it through CreateCSIFT3D when SIFT3D_DEVICES lists several GPUs.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct sift3d_sharded *sift3d_sharded_handle;
int sift3d_sharded_create(sift3d_sharded_handle *out, const float *volume, int nx, int ny, int nz, const sift3d_params *params,
                          const int *devices, int ndev, int sim_ranks, int sharded_octaves);
/* the same with option bits.  By default (r06) the descriptor windows are split along z over the ranks (records to the z-neighbours, partial
 * integer histograms back: the sift3d_slab_describe_partial / _finish protocol above; level halos of 13 planes), and whole windows on 39-plane
 * halos are what remains for slabs so thin that a window would span more than 6 ranks.  SIFT3D_SHARDED_WHOLE_WINDOWS asks for whole windows
 * always; SIFT3D_SHARDED_PARTIAL_WINDOWS for partial windows or a refusal (no silent change of form).  Same results bit for bit. */
#define SIFT3D_SHARDED_PARTIAL_WINDOWS 1u
#define SIFT3D_SHARDED_WHOLE_WINDOWS 2u
/* r06: the COPY transport instead of RCCL: the same rank threads, streams and plan, but a neighbour's planes / records / histograms are fetched by
 * device copies behind an event the sender recorded (one copy launch per exchange step on one device, hipMemcpyPeerAsync between devices; no
 * communicator, no librccl).  `devices` may then name a device several times -- N rank threads on ONE GPU, which is how the multi-threaded driver
 * is tested on a one-GPU box -- and at most 16 ranks are taken.  Same results bit for bit.  Ignored with sim_ranks > 0. */
#define SIFT3D_SHARDED_COPY_TRANSPORT 4u
/* r06: octave 0 on ghost zones (sift3d_slab_set_ghost): every rank uploads its planes + 33 (defaults) per side of the INPUT and recomputes what it
 * would otherwise receive -- no exchange at all for octave 0 (0.19 of the 0.28 GB a rank receives per side and step at 1024 x 1024 x 512 over 8, and
 * 6 of the 18 exchanges its level chains wait for), for + 0.5 ms of pyramid work per rank.  For nodes whose links, not whose GPUs, bound the step. */
#define SIFT3D_SHARDED_GHOST_OCTAVE0 8u
int sift3d_sharded_create_ex(sift3d_sharded_handle *out, const float *volume, int nx, int ny, int nz, const sift3d_params *params,
                             const int *devices, int ndev, int sim_ranks, int sharded_octaves, unsigned flags);
int sift3d_sharded_run(sift3d_sharded_handle h);
int sift3d_sharded_num_keypoints(sift3d_sharded_handle h, int *n);
int sift3d_sharded_get_keypoints(sift3d_sharded_handle h, sift3d_keypoint *out, float *desc /* n*768, may be NULL */);
/* ranks, sharded octaves, halo planes; seconds[0] = wall time of the last run (KpSiftAlgorithm: the results are complete on the devices),
 * [1] = the same + the read-back of every rank's results and their merge, once sift3d_sharded_get_keypoints has run (r05 counted both in [0]) */
int sift3d_sharded_info(sift3d_sharded_handle h, int *world, int *sharded_octaves, int *halo, double seconds[2]);
/* the plan: descriptor windows split along z in every sharded octave (1) or not (0), and per sharded octave (stage_partial: an octave whose
 * slabs are too thin for the split carries whole windows on wide halos); the rank that also runs the octaves behind the sharded ones, once
 * for the node (-1: the volume has none), and the planes of octave 0 every rank owns (that rank owns fewer) */
int sift3d_sharded_plan(sift3d_sharded_handle h, int *partial_windows, int *tail_rank, int *planes /* [world] or NULL */,
                        int *stage_partial /* [sharded octaves] or NULL */);
/* bytes every rank receives per step: plane halos (from the plan) and -- from the keypoint counts of the last run -- the records and partial
 * histograms of the octaves whose windows are split along z */
int sift3d_sharded_traffic(sift3d_sharded_handle h, double *halo_bytes /* [world] */, double *window_bytes /* [world] */);
const char *sift3d_sharded_error(sift3d_sharded_handle h);
int sift3d_sharded_destroy(sift3d_sharded_handle h);

/* Test hooks, rare-path counters and the unit-level debug entry points live in include/sift3d_hip_test.h: this header is the
 * product boundary only. */
const char *sift3d_error_string(int code);
const char *sift3d_last_error(void); /* thread-local detail of the last failure */

#ifdef __cplusplus
}
#endif
#endif
