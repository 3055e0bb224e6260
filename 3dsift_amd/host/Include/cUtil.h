// cUtil.h -- the two exported helpers of the reference's Include/cUtil.h:66-68 that downstream users call:
// matched-keypoint coordinate lists as text, one "x,y,z" line per point with 5 decimals
// (reference Src/cUtil.cc:938-954, 1002-1016).  Everything else in the reference's cUtil.h is internal to its
// CPU pipeline and has no counterpart here.
#pragma once
#include <vector>

#include "Util/common.h"
#include "cSIFT3D.h"

namespace CPUSIFT {
SIFT_LIBRARY_API void write_sift_kp(std::vector<Cvec> &kp, const char *file_name);
SIFT_LIBRARY_API void read_sift_kp(const char *file_name, std::vector<Cvec> &kp);
}  // namespace CPUSIFT
