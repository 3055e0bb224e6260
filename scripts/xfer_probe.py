"""host <-> device transfer times of the boundary (r06): CreateCSIFT3D(float*) incl. H2D and GetKeypoints D2H, repeated in one process."""
import importlib, sys, time, numpy as np
sys.path.insert(0, '.')
capi = importlib.import_module("3dsift_amd.capi")
synth = importlib.import_module("3dsift_amd.synth")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
vol = synth.blobs_torch((n, n, n), "cuda", seed=1234); torch.cuda.synchronize()
host = vol.cpu().numpy()
ex0 = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(n, n, n)).KpSiftAlgorithm()
kp, ds = ex0.GetKeypoints()
bufs = (np.zeros(len(kp), capi.KP_DTYPE), np.ones((len(kp), 768), np.float32))
for rep in range(6):
    t3 = time.perf_counter(); ex = capi.CSIFT3D(host); t4 = time.perf_counter(); ex.close()
    t5 = time.perf_counter(); ex0.GetKeypoints(out=bufs); t6 = time.perf_counter()
    assert np.array_equal(bufs[1], ds) and np.array_equal(bufs[0], kp)
    t7 = time.perf_counter(); ex0.GetKeypoints(); t8 = time.perf_counter()
    print("create(host vol) %.2f ms   GetKeypoints %.3f ms (%.1f GB/s)  into new arrays %.3f ms" % ((t4 - t3) * 1e3, (t6 - t5) * 1e3, (ds.nbytes + kp.nbytes) / (t6 - t5) / 1e9, (t8 - t7) * 1e3), flush=True)
g = capi.CSIFT3D(host).KpSiftAlgorithm()
k2, d2 = g.GetKeypoints()
assert np.array_equal(k2, kp) and np.array_equal(d2, ds)
print("host-volume extractor == device-volume extractor")
