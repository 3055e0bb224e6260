/*
 * sift3d_hip_test.h -- TEST-ONLY entry points of libsift3d_hip.so (not part of the drop-in boundary, include/sift3d_hip.h).
 *
 * The parity tests (tests/, through 3dsift_amd/capi.py) and bench.py use these to force rarely taken branches of the product, to
 * prove that they ran, and to check two device helpers at unit level.  No product caller (3dsift_amd/host/) includes this file.
 */
#ifndef SIFT3D_HIP_TEST_H
#define SIFT3D_HIP_TEST_H

#include "sift3d_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------------
 * Test hooks (no reference counterpart).  The product has branches that ordinary inputs rarely reach (list
 * overflow -> regrow -> rerun, the second descriptor pass with the exact fixed-point unit, the register-staged
 * matcher of >= 4 GB matrices, ...).  A hook forces such a branch so that the parity tests execute it; results must
 * not change.  Process-wide, read by the next create / run / match; returns the previous value (-1: unknown hook).
 * The library never reads the environment.
 * ------------------------------------------------------------------------------------------------------------ */
enum {
	SIFT3D_HOOK_DOG_EAGER = 0,      /* 1: write every DoG level (default: first / last level of an octave formed on request) */
	SIFT3D_HOOK_GLAST_EAGER = 1,    /* 1: build the last Gaussian level of every octave (default: evaluated at parked candidates) */
	SIFT3D_HOOK_DET_SERIAL = 2,     /* 1: extremum masks of all octaves on one stream with one scratch */
	SIFT3D_HOOK_SEPARABLE = 3,      /* 1: every Gaussian level by the generic three-pass kernels */
	SIFT3D_HOOK_DESC_NOCACHE = 4,   /* 1: k_describe recomputes the column chords (the path of windows > 255 planes) */
	SIFT3D_HOOK_MATCH_NODMA = 5,    /* 1: matcher tiles staged through registers (the path of matrices >= 4 GB) */
	SIFT3D_HOOK_ONE_STREAM = 6,     /* 1: all octaves on the handle's stream (isolated kernel durations in a trace) */
	SIFT3D_HOOK_DESC_MASS_SHIFT = 7,/* s: k_describe's first gradient-mass estimate is divided by 2^s -> the exact-unit second pass runs */
	SIFT3D_HOOK_LIST_CAP = 8,       /* n > 0: initial capacity of the extrema / keypoint lists -> overflow, regrow, rerun */
	SIFT3D_HOOK_PEER_COPY = 9,      /* 1: sift3d_match_handles stages the target's results through its peer-copy scratch even on one device */
	SIFT3D_HOOK_DESC_NOSPLIT = 10,  /* 1: a descriptor window is never split over several workgroups (the form of runs with many keypoints) */
	SIFT3D_HOOK_MARCH_TILES = 11,   /* 1: the 64 x 32 tiles of the pyramid kernel wherever a level's geometry allows them (default: big levels only); 2: never */
	SIFT3D_HOOK_DESC_EXACT_CELLS = 12, /* 1: k_describe forms the cell coordinates of EVERY voxel with the reference's arithmetic (default: only next to a discontinuity) */
	SIFT3D_HOOK_LAZY_GENERIC = 13,  /* 1: every parked candidate of the lazy last level takes the one-workgroup form (default: interior ones one wave each) */
	SIFT3D_HOOK_SHARDED_FAIL_RANK = 14, /* r + 1: rank r of the native z-slab driver gives up behind its pyramid -> the failure protocol (the other ranks are released, the handle is dead) */
	SIFT3D_HOOK_COUNT = 15
};
int sift3d_test_hook(int which, int value);
/* how often the rare paths ran: c[0] list regrows of the last run, c[1] keypoints whose descriptor took the second pass in
 * the last run, c[2] rows the calling thread's last sift3d_match re-scored exactly (near-tie guard), c[3] reserved */
int sift3d_debug_counters(sift3d_handle h, int c[4]);
/* GB/s (read + write, best of `iters`) of a float4 device-to-device copy of `bytes` bytes on `device`: the measured copy ceiling
 * reported beside the 8 TB/s spec peak */
int sift3d_debug_copy_bandwidth(size_t bytes, int iters, int device, double *gbs);
/* Check_intersect_faces + cart2bary (Src/cSIFT3D.cc:1542-1573, 1592-1637) of k_describe on n gradient vectors (host, n*3):
 * face index (-1: none) and the three barycentric weights as the kernel forms them, through both of its routes:
 * route 0 = predicted face verified with the margin (falls back to route 1 when the margin fails), route 1 = the literal
 * ordered 20-face scan.  Unit-level parity against golden g7. */
int sift3d_debug_face_lookup(const float *grad3, int n, int route, int *face, float *bary3, int device);

/* the byte range [*o, *e) of an n-byte staging chunk that copy thread t of nt moves (csrc/staging.hip): host arithmetic only, no GPU.
 * tests/test_cabi_cpu.py checks that the nt ranges tile [0, n) for the sizes that r05's floor division left short. */
int sift3d_test_staging_slice(size_t n, int nt, int t, size_t *o, size_t *e);
/* r06: the native z-slab driver's plan, host only: the owned plane ranges [z0[r], z1[r]) of `world` slabs of an octave of nz planes dealt by weight
 * (a z-neighbour side costs side_w planes, the rank tail_rank -- or none: -1 -- tail_w more, every rank owns at least min_planes), and of the
 * `halvings` octaves below (a rank owns the planes k whose plane 2k it owns above): z0 / z1 hold (halvings + 1) * world entries, octave major */
int sift3d_test_slab_plan(int nz, int world, int min_planes, double side_w, double tail_w, int tail_rank, int halvings, int *z0, int *z1);

/* Simulated ranks of the native z-slab driver (sift3d_sharded_create with sim_ranks > 0) after a run: the whole step of ONE rank enqueued again
 * on the buffers the run left behind (what it receives is copied from its neighbours' buffers) and timed alone on the GPU -- the GPU time of
 * that rank on a node of `world` GPUs, short of what its transfers wait for.  Results are not changed. */
int sift3d_test_sharded_time_rank(sift3d_sharded_handle h, int rank, double *seconds);

#ifdef __cplusplus
}
#endif
#endif
