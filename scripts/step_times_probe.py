#!/usr/bin/env python3
"""First-steps ramp: per-step wall and stage times of the first steps of a process, the shader clock read from
sysfs after every step, and the same for a SECOND handle created later in the same process (is the ramp per process -- clocks, code
objects -- or per arena -- first touch of the level buffers?).   python3 scripts/step_times_probe.py [n=512]"""
import glob, importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
capi = importlib.import_module("3dsift_amd.capi"); synth = importlib.import_module("3dsift_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512

def sclk():
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for ln in open(f):
                if "*" in ln:
                    return ln.split(":")[1].strip().rstrip(" *")
        except OSError:
            pass
    return "?"

def run(tag, ex, steps):
    rows = []
    for i in range(steps):
        t0 = time.perf_counter(); ex.KpSiftAlgorithm(); w = (time.perf_counter() - t0) * 1e3
        t = ex.m_timer
        rows.append((w, t["d_BuildGSS"] * 1e3, t["d_Detect"] * 1e3, t["d_AssignOrientation"] * 1e3, t["d_Extraction"] * 1e3, sclk()))
    print(tag, "wall / pyramid / detect / orient / describe ms, sclk")
    for i, r in enumerate(rows):
        if i < 10 or i % 10 == 0:
            print("  step %2d  %.2f  %.2f %.2f %.2f %.2f  %s" % ((i,) + r))

vol = synth.blobs_torch((n, n, n), "cuda", seed=1234); torch.cuda.synchronize()
ex = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(n, n, n))
run("first handle", ex, 31)
ex2 = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(n, n, n))
run("second handle (same process, fresh arena)", ex2, 11)
time.sleep(2.0)
run("first handle again after 2 s idle", ex, 6)
