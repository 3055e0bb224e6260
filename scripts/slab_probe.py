"""Where does the time of the simulated-rank slab run go?  (development probe, not a benchmark)"""
import importlib, sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
capi = importlib.import_module("3dsift_amd.capi")
slab = importlib.import_module("3dsift_amd.slab")
synth = importlib.import_module("3dsift_amd.synth")
dims = (1024, 1024, 512)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ex = slab.SlabExtractor(dims, slab.SimComm(R), sharded_octaves=S)
shape = (dims[2], dims[1], dims[0])
slabs = {r: synth.blobs_torch(shape, dev, seed=4321, zrange=ex.bounds[r]) for r in range(R)}
ex.load(device_slabs=slabs)
for _ in range(2):
    ex.KpSiftAlgorithm()
print("times", {k: round(v * 1e3, 2) for k, v in ex.times.items()})
w = ex._wl()[0]
print("tail timer", {k: round(v * 1e3, 3) for k, v in w.tail.m_timer.items()})
kp, _ = w.tail.GetKeypoints(with_desc=False)
print("tail keypoints", len(kp), "per octave", np.bincount(kp["octave"]).tolist(), "extrema", len(w.tail.extrema()))
print("sharded octaves", ex.S, "keypoints per rank", [[int(st.ctx.device_results()[2]) for st in x.stages] for x in ex._wl()])
# per-call timings of one rank
for name, fn in (("detect", w.ctx.detect), ("describe", w.ctx.describe), ("tail.run", w.tail.KpSiftAlgorithm)):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); print(name, round((time.perf_counter() - t) * 1e3, 3), "ms")
