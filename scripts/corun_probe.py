"""two 512^3 volumes on one GPU (BASELINE configs[2]): one after the other / both in flight / the second gated behind the first one's
orientation stage (sift3d_run_async_after).  With a -DS3D_DEV_SWITCHES library (S3D_LIB=variants/libsift3d_hip_dev.so) S3D_DESC_GRID caps
k_describe's persistent grid."""
import importlib, os, sys, time, numpy as np
sys.path.insert(0, '.')
capi = importlib.import_module("3dsift_amd.capi")
synth = importlib.import_module("3dsift_amd.synth")
import torch
n = 512
va = synth.blobs_torch((n, n, n), "cuda", seed=1234); vb = synth.blobs_torch((n, n, n), "cuda", seed=1234, shift=(1.0, 0.0, 0.0)); torch.cuda.synchronize()
A = capi.CSIFT3D(None, device_ptr=va.data_ptr(), shape=(n, n, n)); B = capi.CSIFT3D(None, device_ptr=vb.data_ptr(), shape=(n, n, n))
for _ in range(3):
    A.KpSiftAlgorithm(); B.KpSiftAlgorithm()
ka, da = A.GetKeypoints(); kb, db = B.GetKeypoints()
res = {}
for mode in ("serial", "both", "gated", "serial", "both", "gated"):
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == "serial": A.KpSiftAlgorithm(); B.KpSiftAlgorithm()
        elif mode == "both": A.KpSiftAlgorithmAsync(); B.KpSiftAlgorithmAsync(); A.Wait(); B.Wait()
        else: A.KpSiftAlgorithmAsync(); B.KpSiftAlgorithmAsync(after=A); A.Wait(); B.Wait()
        ts.append(time.perf_counter() - t0)
    res.setdefault(mode, []).append(float(np.median(ts[1:])) * 1e3)
    ta, tb = A.m_timer, B.m_timer
    print(mode, "%.3f ms" % res[mode][-1], "A:", {k[2:6]: round(v * 1e3, 2) for k, v in ta.items() if v}, "B:", {k[2:6]: round(v * 1e3, 2) for k, v in tb.items() if v}, flush=True)
k2, d2 = A.GetKeypoints(); k3, d3 = B.GetKeypoints()
assert np.array_equal(k2, ka) and np.array_equal(d2, da) and np.array_equal(k3, kb) and np.array_equal(d3, db)
print("S3D_DESC_GRID", os.environ.get("S3D_DESC_GRID"), {k: [round(x, 3) for x in v] for k, v in res.items()}, "aggregate Mvoxel/s gated %.0f" % (2 * n ** 3 / min(res["gated"]) / 1e3))
