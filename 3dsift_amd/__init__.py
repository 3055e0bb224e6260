"""MI355X-native 3D SIFT (drop-in for the SCUT-CCNL/3DSIFT OpenMP path).

The product is the HIP library (csrc/ -> libsift3d_hip.so, C-ABI in include/sift3d_hip.h) and the
C++ shell in host/ (namespace CPUSIFT).  `capi` is the ctypes binding used by tests and bench.py;
`synth` generates the deterministic synthetic volumes.  The package name starts with a digit, so
import it with importlib.import_module("3dsift_amd").
"""
