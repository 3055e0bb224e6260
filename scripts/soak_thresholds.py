#!/usr/bin/env python3
"""One-off (GPU box): keypoint COUNTS at and next to the values where k_describe changes its form (window split over 8 / 4 workgroups below
320 / 700 keypoints, eight waves per keypoint below 1 400) -- a white-noise volume whose count is steered by the peak threshold; the whole
result against the oracle at every count found.   python3 scripts/soak_thresholds.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
capi = importlib.import_module("3dsift_amd.capi")
import oracle_lib as ol
from hipcheck import compare_keypoints, extrema_table
orc = ol.load("orc")
vol = np.random.default_rng(11).random((112, 120, 128)).astype(np.float32)
cache = {}
def count(th):
    if th not in cache:
        g = capi.CreateCSIFT3D(vol, peak_thresh=th).KpSiftAlgorithm()
        cache[th] = len(g.GetKeypoints()[0]); g.close()
    return cache[th]
lo, hi = 0.02, 0.6
print("counts at the ends:", count(lo), count(hi), flush=True)
done = set()
for target in (319, 320, 321, 699, 700, 701, 1399, 1400, 1401):
    a, b = lo, hi   # count decreases with the threshold
    for _ in range(40):
        m = 0.5 * (a + b)
        c = count(m)
        if c == target: break
        if c > target: a = m
        else: b = m
    best = min(cache, key=lambda t: (abs(cache[t] - target), t))
    if cache[best] in done: continue
    done.add(cache[best])
    g = capi.CreateCSIFT3D(vol, peak_thresh=best).KpSiftAlgorithm()
    o = orc.extractor(vol, peak_thresh=best).run(5)
    kp, desc = g.GetKeypoints(); okp, odesc = o.keypoints()
    assert np.array_equal(extrema_table(g.extrema()), extrema_table(o.extrema()))
    rms = compare_keypoints(kp, desc, okp, odesc)
    with capi.hook("desc_nosplit", 1):
        g2 = capi.CreateCSIFT3D(vol, peak_thresh=best).KpSiftAlgorithm()
        d2 = g2.GetKeypoints()[1]; g2.close()
    assert np.array_equal(d2, desc), "split and unsplit windows differ"
    print("target %4d: %4d keypoints at peak_thresh %.6f  descriptor rms %.2e  == oracle, split == unsplit" % (target, len(kp), best, rms), flush=True)
    g.close()
print("thresholds: all equal")
