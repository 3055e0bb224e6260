// cSIFT3D.h -- public C++ API of the MI355X-native 3D SIFT extractor.
//
// Drop-in for the reference header 3DSIFT/Include/cSIFT3D.h: same namespace, class, method, field
// and macro names (factory :184-204, CSIFT3D public part :142-177, Keypoint :54-70, Cvec :38-52), so
// user code such as the reference's Example.cpp compiles unchanged.  Everything behind it is
// different: a CSIFT3D object is a thin shell around an opaque sift3d_handle (include/sift3d_hip.h);
// volumes, both pyramids, keypoints and descriptors live on the GPU.
#ifndef S3D_HOST_CSIFT3D_H
#define S3D_HOST_CSIFT3D_H

#include <cstddef>
#include <string>
#include <vector>

#include "Util/cTexImage.h"
#include "Util/common.h"

namespace CPUSIFT {

// defaults of the factory arguments (reference values)
#define SIGMA_DEFAULT 1.6
#define SIGMA_N_DEFAULT 1.15
#define NUM_KP_LEVELS 3
#define PEAK_THRESH 0.1
#define EIG_THRES 0.9
#define CORNER_THRESH 0.4
// fixed algorithm constants
#define IMG_BORDER 1
#define NHIST_PER_DIM 4
#define ICOS_NFACES 20
#define ICOS_NVERT 12
#define DESC_NUMEL (NHIST_PER_DIM * NHIST_PER_DIM * NHIST_PER_DIM * ICOS_NVERT)

// kept for source compatibility: thread count of the reference's OpenMP stages; unused on the GPU
extern int sift_thread_num;

typedef struct _cCvec {
	float x, y, z;
	_cCvec(float x_ = 0, float y_ = 0, float z_ = 0) : x(x_), y(y_), z(z_) {}
} SIFT_LIBRARY_API Cvec;

// Field order and types match the reference record, so binaries that copy Keypoints around keep working.
typedef struct _cKeypoint {
	float x, y, z;       // voxel coordinates inside the keypoint's octave
	float scale;         // scale-space location of its DoG level
	int octave, level;
	float rx, ry, rz;    // coordinates in the original volume (x * 2^octave)
	Cvec win;            // weighted mean gradient of the orientation window
	float eigvalue[3];   // ascending
	float eigvector[9];
	float Rotation[9];   // returned transposed (inverse), as the reference leaves it after describing
	float str_tensor[9];
	float *desc = nullptr;  // DESC_NUMEL floats owned by the extractor: valid while the CSIFT3D lives
} SIFT_LIBRARY_API Keypoint;

// layout guard (the reference's record on LP64): binaries that copy Keypoints around, and the conversion from the
// C-ABI's 168-byte POD in fetch_results, rely on exactly this layout
static_assert(sizeof(Cvec) == 12, "Cvec is three packed floats");
static_assert(sizeof(void *) != 8 || sizeof(Keypoint) == 176, "Keypoint must be 176 bytes on LP64");
static_assert(offsetof(Keypoint, scale) == 12 && offsetof(Keypoint, octave) == 16 && offsetof(Keypoint, level) == 20 && offsetof(Keypoint, rx) == 24 &&
              offsetof(Keypoint, win) == 36 && offsetof(Keypoint, eigvalue) == 48 && offsetof(Keypoint, eigvector) == 60 &&
              offsetof(Keypoint, Rotation) == 96 && offsetof(Keypoint, str_tensor) == 132 && (sizeof(void *) != 8 || offsetof(Keypoint, desc) == 168),
              "Keypoint field offsets");

bool cmp_kp(const Keypoint &a, const Keypoint &b);
bool cmp_kp_orig(const Keypoint &a, const Keypoint &b);

// The remaining public records of the reference's header (Include/cSIFT3D.h:72-116): same fields and order, so user code that
// names them compiles.  The extractor itself does not use them (its mesh is a device constant table, its levels device buffers).
typedef struct _cTri {
	Cvec v[3];   // vertices
	int idx[3];  // index of each vertex in the solid
} Tri;
typedef struct _Mesh {
	Tri *tri;
	int num;
} Mesh;
typedef struct _cImage {
	float *data;
	int nx, ny, nz;
	size_t xs, ys, zs;  // strides: xs = 1, ys = nx, zs = nx * ny
	float ux, uy, uz;
	size_t size;        // voxels
	float s;            // scale-space location
} Image;
typedef struct _cEigenVal {
	float val;
	float vec[3];
} EigenVal;

class CSIFT3D {
protected:
	struct Impl;
	Impl *impl = nullptr;
	std::vector<Keypoint> filter;
	std::vector<TexImage> Gss_Pyramid, DoG_Pyramid;  // host copies, filled by GET_GSS()/GET_DOG()
	std::vector<std::vector<Keypoint>> level_extrema;
	float *global_descriptor = nullptr;
	void fetch_results();

public:
	SIFT_TimerPara m_timer;

	SIFT_LIBRARY_API CSIFT3D();
	SIFT_LIBRARY_API CSIFT3D(float *volume, int x_dim, int y_dim, int z_dim, int num_kp_levels_, float sigma_default_,
	                         float sigma_n_default_, float peak_thresh_, float max_eig_thres_, float corner_thresh_);
	SIFT_LIBRARY_API ~CSIFT3D();
	CSIFT3D(const CSIFT3D &) = delete;
	CSIFT3D &operator=(const CSIFT3D &) = delete;

	SIFT_LIBRARY_API void KpSiftAlgorithm();
	// extension (no reference counterpart): KpSiftAlgorithm split in two.  KpSiftAlgorithmAsync() enqueues the whole pipeline on the GPU
	// and returns; Wait() -- or GetKeypoints() -- completes it.  One host thread keeps several extractors (volumes) in flight on one GPU.
	SIFT_LIBRARY_API void KpSiftAlgorithmAsync();
	SIFT_LIBRARY_API void Wait();
	SIFT_LIBRARY_API void SetNumThreads(int t_num);
	SIFT_LIBRARY_API std::vector<Keypoint> GetKeypoints();

	// the reference exposes its stages publicly; here each call runs the device pipeline up to that stage
	void Initialize();
	void Build_Gaussian_Scale_Space();
	void Build_DOG_Scale_Space();
	void Detect_KeyPoints();
	void Assign_Orientation();
	void Extract_Description();
	void Release_SIFT();
	void SetHostImNull() {}

	// checking accessors: copy the pyramids / per-level extrema back from the device
	SIFT_LIBRARY_API std::vector<TexImage> *GET_GSS();
	SIFT_LIBRARY_API std::vector<TexImage> *GET_DOG();
	SIFT_LIBRARY_API std::vector<std::vector<Keypoint>> *GET_LEVEL();

	// extension: device-resident descriptors / coordinates for a matcher that never leaves the GPU
	SIFT_LIBRARY_API bool GetDeviceResults(const float **d_desc, const float **d_xyz, int *n, int *device);
	// extension: the live extractor whose GetKeypoints() produced `kp` unchanged (same count, descriptor
	// pointers and coordinates), or nullptr.  muBruteMatcher uses it to match straight from the device-resident results.
	SIFT_LIBRARY_API static CSIFT3D *OwnerOf(const std::vector<Keypoint> &kp);

	// extension (BASELINE configs[4] for a single-process C++ caller; no reference counterpart): every ORDERED pair (i, j), i != j, of
	// extractors that have run -- typically one volume per GPU of the node (SetDevice before each CreateCSIFT3D) -- matched from their
	// device-resident descriptors: mode 1 injectMatch, 2 bijectMatch, 3 enhancedMatch (Include/cMatcher.h:76-87).  A target on another
	// GPU is copied peer to peer (xGMI) to the reference's GPU; the pairs of different reference GPUs run on one host thread per GPU.
	struct PairMatch {
		int ref = 0, tar = 0;
		std::vector<Cvec> refMatch, tarMatch;   // as muBruteMatcher returns them
		std::vector<int> glodenIdx;             // best target index per reference keypoint (getGlodenIdx)
		double seconds = 0;                     // device time of the match
	};
	SIFT_LIBRARY_API static std::vector<PairMatch> AllPairsMatch(const std::vector<CSIFT3D *> &extractors, double thresHold = 0.85, int mode = 3);
};

class SIFT_LIBRARY_API CSIFT3DFactory {
public:
	static CSIFT3D *CreateCSIFT3D(float *volume, int x_dim, int y_dim, int z_dim, int num_kp_levels = NUM_KP_LEVELS,
	                              float sigma_default = SIGMA_DEFAULT, float sigma_n_default = SIGMA_N_DEFAULT,
	                              float peak_thresh = PEAK_THRESH, float max_eigo_thres = EIG_THRES,
	                              float corner_thresh = CORNER_THRESH);

	// raw-matrix file overload (12-byte int32 header m,n,p + fp32 payload)
	static CSIFT3D *CreateCSIFT3D(std::string path_, int num_kp_levels = NUM_KP_LEVELS, float sigma_default = SIGMA_DEFAULT,
	                              float sigma_n_default = SIGMA_N_DEFAULT, float peak_thresh = PEAK_THRESH,
	                              float max_eigo_thres = EIG_THRES, float corner_thresh = CORNER_THRESH);
};

// The volume-level free functions of the reference's header (Include/cSIFT3D.h:208-218), on host TexImages: each call moves its
// operands to the GPU, runs the pipeline's own kernel and copies the result back.  dst / dog are (re)sized like the reference's
// callers size them: GaussianSmooth_3D and Sub give dst the dimensions, units and scale of src / prev; DownSample_3D fills the
// caller-sized dst (dst(n, m, k) = src(2n, 2m, 2k)).
SIFT_LIBRARY_API void DownSample_3D(TexImage *src, TexImage *dst);
SIFT_LIBRARY_API void GaussianSmooth_3D(TexImage *src, TexImage *dst, float sigma);
SIFT_LIBRARY_API void Sub(TexImage *prev, TexImage *cur, TexImage *dog);
// the per-axis pass and the per-keypoint stages (Include/cSIFT3D.h:214, 224, 228 of the reference) on HOST data, run by the pipeline's
// device kernels: GaussianSmooth_3D_Imp one pass along `dim` with the caller's taps (odd width; `unit` is unused, as in the reference);
// Assign_Orientation_Imp / Extract_Descriptor_Imp one keypoint on a caller-held level (kp.x, y, z integral, one power-of-two unit, kp.desc
// -> DESC_NUMEL floats; the mesh argument is not read: the kernels carry the icosahedron of Initialize_geometry)
SIFT_LIBRARY_API void GaussianSmooth_3D_Imp(TexImage *src, TexImage *dst, int dim, float unit, float *weight, int width);
SIFT_LIBRARY_API int Assign_Orientation_Imp(Keypoint &kp, TexImage *gaussian, const float sigma, const float max_eig_ratio, const float corner_thresh);
SIFT_LIBRARY_API void Extract_Descriptor_Imp(Keypoint &kp, TexImage *gaussian, Mesh *mesh);

// The small per-voxel / per-vector helpers of the same header (Include/cSIFT3D.h:216-238), as HOST utilities on HOST data
// (host/src/helpers.cpp).  They are not part of the extraction path -- KpSiftAlgorithm never calls them, its kernels carry their own
// forms -- and exist so that user code which calls them directly keeps compiling and gets the reference's answers.
SIFT_LIBRARY_API void Im_permute(TexImage *src, TexImage *dst, int dim1, int dim2);                                   // Src/cSIFT3D.cc:790-847
SIFT_LIBRARY_API bool IsExtrema_neighbor(TexImage *prev, TexImage *cur, TexImage *next, int x, int y, int z);         // :884-911
SIFT_LIBRARY_API bool DistinctEig(float a, float b, float c);                                                          // :1140-1150
SIFT_LIBRARY_API int Check_intersect_faces(Mesh *mesh, Cvec *grad, Cvec *bary);                                        // :1542-1573
SIFT_LIBRARY_API void Transpose_Matrix(float *Rot);                                                                    // :1575-1582
SIFT_LIBRARY_API void Swap_Element(float &a, float &b);                                                                // :1584-1590
SIFT_LIBRARY_API int cart2bary(Cvec *cart, const Tri *const tri, Cvec *const bary, float *const k);                    // :1592-1637
SIFT_LIBRARY_API void normailize_desc(float *desc);                                                                    // :1639-1656
SIFT_LIBRARY_API void Trilinear_interpolation_over_desc(Mesh *mesh, Keypoint &kp, Cvec &vbins, Cvec &grad, int loop_idx);              // :1383-1448
SIFT_LIBRARY_API void Trilinear_interpolation_over_desc_debug(Mesh *mesh, Keypoint &kp, Cvec &vbins, Cvec &grad, int loop_idx,        // :1450-1540
                                                              float *host_dvbins, int *host_intersect_id, float *host_bary, int *host_offset,
                                                              float *host_desc_accum, int debug);
// the icosahedron the descriptor histograms are binned on (Include/cUtil.h:60, Src/cUtil.cc:113-175; winding quirk included).
// mesh->tri is malloc()ed (20 triangles); the caller free()s it.  Returns 0.
SIFT_LIBRARY_API int Initialize_geometry(Mesh *mesh);

// device selection for subsequently created extractors / matchers (default 0, or env SIFT3D_DEVICE)
SIFT_LIBRARY_API void SetDevice(int device);
SIFT_LIBRARY_API int GetDevice();

}  // namespace CPUSIFT
#endif
