import importlib, sys, time, numpy as np
sys.path.insert(0, '.')
capi = importlib.import_module("3dsift_amd.capi")
synth = importlib.import_module("3dsift_amd.synth")
import torch
n = 512
vol = synth.blobs_torch((n, n, n), "cuda", seed=1234); torch.cuda.synchronize()
host = vol.cpu().numpy()
for rep in range(8):
    t0 = time.perf_counter(); ex = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(n, n, n)); t1 = time.perf_counter(); ex.close(); t2 = time.perf_counter()
    t3 = time.perf_counter(); ex = capi.CSIFT3D(host); t4 = time.perf_counter(); ex.close()
    print("create(device vol) %.2f ms  close %.2f ms  create(host vol) %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t4 - t3) * 1e3), flush=True)
