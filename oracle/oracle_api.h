/*
 * oracle_api.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Common C interface exported, with two different prefixes, by
 *   - oracle/liboracle3dsift.so      (prefix orc_) : our own CPU restatement of the
 *                                      reference algorithm (oracle/sift3d_oracle.c)
 *   - oracle/_ref/libref3dsift.so    (prefix ref_) : the untouched reference sources
 *                                      under /root/reference compiled where they lie
 *                                      (oracle/Makefile, target `ref`), driven by
 *                                      oracle/ref_harness.cpp
 * so that tests can run exactly the same python driver against both.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * these libraries.  The product (3dsift_amd/) never links or calls them.
 *
 * ORC_API(name) expands to orc_name or ref_name depending on ORACLE_PREFIX_REF.
 */
#ifndef ORACLE_API_H
#define ORACLE_API_H

#ifdef __cplusplus
extern "C" {
#endif

#ifdef ORACLE_PREFIX_REF
#define ORC_API(n) ref_##n
#else
#define ORC_API(n) orc_##n
#endif

/* Mirror of CPUSIFT::Keypoint (reference 3DSIFT/Include/cSIFT3D.h:54-70) minus the
 * desc pointer; 42 four-byte fields = 168 bytes. */
typedef struct orc_kp {
	float x, y, z;
	float scale;
	int octave, level;
	float rx, ry, rz;
	float win[3];
	float eigvalue[3];
	float eigvector[9];
	float Rotation[9];
	float str_tensor[9];
} orc_kp;

/* extractor: construct (copy + max-abs normalise, reference cSIFT3D.cc:146-163) */
void *ORC_API(create)(const float *volume, int nx, int ny, int nz, int num_kp_levels,
                      float sigma_default, float sigma_n_default, float peak_thresh,
                      float max_eig_thres, float corner_thresh);
void ORC_API(destroy)(void *h);
void ORC_API(set_threads)(int n);

/* run the pipeline stage by stage (reference cSIFT3D.cc:165-235); `upto`:
 * 1 Initialize+GSS, 2 +DoG, 3 +Detect, 4 +Orientation, 5 +Description.
 * Pyramids are kept alive until destroy. times[6] (seconds, may be NULL):
 * init, gss, dog, detect, orient, descr. */
void ORC_API(run)(void *h, int upto, double *times);

int ORC_API(num_octaves)(void *h);
/* GSS level index = octave*(num_kp_levels+3)+i ; DoG index = octave*(num_kp_levels+2)+i */
void ORC_API(level_info)(void *h, int is_dog, int idx, int *dims3, float *units3, float *scale);
void ORC_API(copy_level)(void *h, int is_dog, int idx, float *out);
void ORC_API(copy_input)(void *h, float *out); /* normalised input volume */

int ORC_API(num_extrema)(void *h);
void ORC_API(copy_extrema)(void *h, orc_kp *out); /* after stage 3 (before orientation mutates them) or later */
int ORC_API(num_keypoints)(void *h);
void ORC_API(copy_keypoints)(void *h, orc_kp *out, float *desc /* N*768 or NULL */);

/* unit-level entry points (reference free functions, cSIFT3D.h:208-239) */
void ORC_API(gaussian_smooth)(const float *src, int nx, int ny, int nz, float sigma, float *dst);
int ORC_API(gaussian_taps)(float sigma, float *taps /* >= 64 */); /* returns width */
int ORC_API(mesh)(float *verts /*20*3*3*/, int *idx /*20*3*/);
/* returns face index or -1; bary[3], k */
int ORC_API(intersect)(const float *grad3, float *bary3);
/* orientation of one extremum on a given level: returns reference code 1/-1/-2/-3 */
int ORC_API(orient_one)(orc_kp *kp, const float *level, int nx, int ny, int nz, float unit,
                        float sigma, float max_eig_ratio, float corner_thresh);
/* descriptor of one oriented keypoint (kp->Rotation is transposed in place like the
 * reference, cSIFT3D.cc:1214) */
void ORC_API(describe_one)(orc_kp *kp, const float *level, int nx, int ny, int nz, float unit,
                           float *desc768);

/* matcher (reference cMatcher.cc:146-228). mode: 1 inject, 2 biject, 3 enhanced.
 * ref_desc: N*768, ref_xyz: N*3 (rx,ry,rz).  Outputs: gIdx/sIdx/gDist/sDist sized N
 * (may be NULL), pairs: up to N (ref_xyz, tar_xyz) rows of 6 floats; returns #pairs. */
int ORC_API(match)(const float *ref_desc, const float *ref_xyz, int n, const float *tar_desc,
                   const float *tar_xyz, int m, double thresh, int mode, int *gIdx, int *sIdx,
                   float *gDist, float *sDist, float *pairs6);

#ifdef __cplusplus
}
#endif
#endif
