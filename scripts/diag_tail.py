"""dev diagnostic: which keypoints of the sharded extractor (world of one, real RCCL rank) differ from the single-volume ones"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
capi = importlib.import_module("3dsift_amd.capi")
synth = importlib.import_module("3dsift_amd.synth")
vol = synth.blobs((160, 96, 128), seed=77, noise=0.01)
ex = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
kp, ds = ex.GetKeypoints()
sim = int(os.environ.get("DIAG_SIM", "0"))
sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=sim, sharded_octaves=2)
for rep in range(3):
    k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
    print("rep", rep, "n", len(kp), len(k2), "per octave", np.bincount(kp["octave"], minlength=4), np.bincount(k2["octave"], minlength=4), flush=True)
    if len(kp) == len(k2):
        bad = [i for i in range(len(kp)) if kp[i] != k2[i] or not np.array_equal(ds[i], d2[i])]
        print("  differing rows", len(bad), "octaves", sorted(set(int(kp["octave"][i]) for i in bad)))
        for i in bad[:3]:
            for f in kp.dtype.names:
                if not np.array_equal(kp[i][f], k2[i][f]): print("   row", i, f, kp[i][f], k2[i][f])
            print("   desc equal", np.array_equal(ds[i], d2[i]), np.abs(ds[i] - d2[i]).max())
sh.close()
