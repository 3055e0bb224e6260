#!/usr/bin/env python3
"""how many octaves to shard?  sum over 8 simulated ranks of one 1024x1024x512 KpSiftAlgorithm for sharded_octaves = 1, 2, 3 (python driver)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
capi = importlib.import_module("3dsift_amd.capi"); synth = importlib.import_module("3dsift_amd.synth"); slab = importlib.import_module("3dsift_amd.slab")
dims = (1024, 1024, 512); shape = (512, 1024, 1024)
for S in (2, 3, 1):
    comm = slab.SimComm(8)
    ex = slab.SlabExtractor(dims, comm, device=0, sharded_octaves=S)
    slabs = {r: synth.blobs_torch(shape, "cuda", seed=4321, zrange=ex.bounds[r]) for r in range(8)}
    torch.cuda.synchronize(); ex.load(device_slabs=slabs); del slabs
    ex.KpSiftAlgorithm(); ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); ex.KpSiftAlgorithm(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("sharded_octaves asked", S, "got", ex.S, "sum over 8 simulated ranks %.1f ms" % (1e3 * float(np.median(ts))), {k: round(v * 1e3, 1) for k, v in ex.times.items()}, flush=True)
    ex.close(); torch.cuda.empty_cache()
