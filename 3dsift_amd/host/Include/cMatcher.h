// cMatcher.h -- brute-force descriptor matcher, drop-in for the reference 3DSIFT/Include/cMatcher.h
// (muBruteMatcher :12-88).  The O(N*M*768) best / second-best search runs on the GPU matrix cores;
// the ratio / bijection bookkeeping replays the reference's rules, quirks included.
#ifndef S3D_HOST_CMATCHER_H
#define S3D_HOST_CMATCHER_H

#include <vector>

#include "Util/common.h"
#include "cSIFT3D.h"

#define DESC_LENGTH 768

namespace CPUSIFT {

class muBruteMatcher {
private:
	std::vector<float> glodenDistSquare, silverDistSquare;
	std::vector<int> glodenIdx, silverIdx;
	void run(std::vector<Cvec> &refMatch, std::vector<Cvec> &tarMatch, const std::vector<Keypoint> &ref_kp,
	         const std::vector<Keypoint> &tar_kp, double thresHold, int mode);

public:
	// the reference's public timing fields; only matchTime / totalTime are meaningful here (device seconds)
	float matchTime = 0.0, filterTime = 0.0, countMatchedTime = 0.0, revMatchTime = 0.0, revFilterTime = 0.0,
	      bijectFilterTime = 0.0, converseTime = 0.0, totalTime = 0.0;
	// extension: true when the last call matched straight from two live extractors' device-resident results
	bool usedDeviceResults = false;

	SIFT_LIBRARY_API muBruteMatcher();
	SIFT_LIBRARY_API float getCalculationTime();
	SIFT_LIBRARY_API std::vector<float> getGlodenDistSquare();
	SIFT_LIBRARY_API std::vector<float> getSilverDistSquare();
	SIFT_LIBRARY_API std::vector<int> getGlodenIdx();
	SIFT_LIBRARY_API std::vector<int> getSilverIdx();

	SIFT_LIBRARY_API void injectMatch(std::vector<Cvec> &refMatch, std::vector<Cvec> &tarMatch, const std::vector<Keypoint> &ref_kp,
	                                  const std::vector<Keypoint> &tar_kp, const double thresHold = 0.85);
	SIFT_LIBRARY_API void bijectMatch(std::vector<Cvec> &refMatch, std::vector<Cvec> &tarMatch, const std::vector<Keypoint> &ref_kp,
	                                  const std::vector<Keypoint> &tar_kp, const double thresHold = 0.85);
	SIFT_LIBRARY_API void enhancedMatch(std::vector<Cvec> &refMatch, std::vector<Cvec> &tarMatch, const std::vector<Keypoint> &ref_kp,
	                                    const std::vector<Keypoint> &tar_kp, const double thresHold = 0.85);
};

}  // namespace CPUSIFT
#endif
