// common.h -- export macro + stage timers of the drop-in C++ shell (namespace-free, like the
// reference's Include/Util/common.h:4-8, 22-41 whose public names these are).
#pragma once

#include <iostream>
#include <vector>

#ifndef SIFT_LIBRARY_API
#define SIFT_LIBRARY_API /* every symbol has default visibility in libsift3d.so */
#endif

// Seconds per stage of one KpSiftAlgorithm(); filled from HIP events on the extractor's stream.
// d_BuildDOG is ~0 because the DoG subtraction is fused into the Gaussian level kernel, d_Allocation
// and d_release are 0 because the device arena is reserved by the constructor.
struct SIFT_LIBRARY_API SIFT_TimerPara {
	double d_TotalTime = 0;
	double d_Allocation = 0;
	double d_BuildGSS = 0;
	double d_BuildDOG = 0;
	double d_Detect = 0;
	double d_AssignOrientation = 0;
	double d_Extraction = 0;
	double d_release = 0;
	double d_memoryOverhead = 0;
	std::vector<double> vD_octaveTime;
	std::vector<double> vD_octaveCompute;

	double getAllComputeTime() { return d_BuildGSS + d_BuildDOG + d_Detect + d_AssignOrientation + d_Extraction; }
};

SIFT_LIBRARY_API std::ostream &operator<<(std::ostream &os, const SIFT_TimerPara &st);
