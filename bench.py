#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mvoxels/s of end-to-end KpSiftAlgorithm on a 512^3 fp32 volume.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" is one full CSIFT3D::KpSiftAlgorithm (Gaussian/DoG pyramid, extrema, orientation,
descriptors) on one synthetic 512^3 volume that is ALREADY resident and normalised in HBM (the
reference does copy + normalise in the constructor too, outside KpSiftAlgorithm).  With N ranks every
GPU processes its own 512^3 volume (BASELINE.json configs[4], independent volumes => weak scaling, no
data-path collective); value = N * 512^3 * K / max-over-ranks time.

    python bench.py --workload slab [--slab-dims 1024x1024x512] [--sim-ranks R]

times BASELINE.json configs[3] instead: ONE volume sharded as z-slabs over the N ranks (3dsift_amd/slab.py: halo
exchange + all-reduce over RCCL; strong scaling, N=1 is the plain single-GPU extractor on the whole volume).
--sim-ranks R runs R simulated ranks on one GPU (functional check / redundancy accounting, not a speed claim).
A default N>1 run appends the slab measurement as "slab": {...} to its JSON line: "native" (the library's C++ driver, first class),
"python" (3dsift_amd/slab.py over torch.distributed), each behind a watchdog and with speedup_vs_n1 against the committed N = 1 record.

The JSON line also carries
  roofline      pyramid build (the HBM-bound part north_star sets a target for): the bytes its kernels MOVE
                (52 B x pyramid voxels: two DoG levels per octave and the last Gaussian level are not built)
                / HIP-event time of that stage on the library's own stream, against the 8 TB/s HBM3E peak;
                frac_every_level_built is SURVEY.md 8d's work as defined (every level built and written, 68 B per
                pyramid voxel), timed in its own runs
  descriptor    the descriptor stage (half of the step): keypoints/s, window voxels/s, and the VALU-issue roofline
                of k_describe (wave-instructions from the committed PMC profile of the same kernel sources)
  cpu_baseline  the CPU oracle (our restatement of the reference, OpenMP) timed on this host on a
                bounded sample of the same workload
  parity        the GPU result on that sample checked against the oracle
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # same guide: dense f32-input MFMA peak (= the f32 vector peak; no xf32/TF32 on gfx950)
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 2  # same guide: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles at 2.4 GHz = 1228.8 G wave-instructions/s


def pyramid_voxels(shape, levels=3):
    nz, ny, nx = shape
    noct = max(0, int(np.log2(np.float32(min(shape)))) - 3 + 1)
    tot = 0
    for _ in range(noct):
        tot += nx * ny * nz
        nx, ny, nz = nx // 2, ny // 2, nz // 2
    return tot, noct


def sphere_lattice_points(r2_over_u2):
    """number of integer offsets (dx, dy, dz) with dx^2 + dy^2 + dz^2 <= r2_over_u2"""
    R = int(np.floor(np.sqrt(r2_over_u2)))
    a = np.arange(-R, R + 1, dtype=np.int64) ** 2
    return int((a[:, None, None] + a[None, :, None] + a[None, None, :] <= r2_over_u2).sum())


def descriptor_window_voxels(kp):
    """voxels of the descriptor windows of a keypoint set: lattice points of the sphere r = 2 * 7.0711 * scale at the keypoint's
    octave (Src/cSIFT3D.cc:1155-1156, 1276-1296), not clipped at the volume border"""
    tot = 0
    for o in np.unique(kp["octave"]):
        for lv in np.unique(kp["level"][kp["octave"] == o]):
            sel = kp[(kp["octave"] == o) & (kp["level"] == lv)]
            r = np.float32(2.0) * (sel["scale"][0] * np.float32(7.071067812))
            u = np.float32(2.0 ** int(o))
            tot += len(sel) * sphere_lattice_points(float(r * r) / float(u * u))
    return tot


def parse_dims(txt):
    nx, ny, nz = (int(v) for v in txt.lower().split("x"))
    return nx, ny, nz


def run_slab_native(dims, devices, steps, warmup, sim_ranks=0, seed=4321, partial_windows=None, rank_times=True, transport="rccl", ghost=False):
    """The same workload through the library's NATIVE driver (csrc/sharded.hip: one process, one host thread per GPU, RCCL halo
    exchange; or sim_ranks simulated on devices[0]).  The volume starts on the host, like CreateCSIFT3D(float*) gets it.
    partial_windows: None = the driver's rule (descriptor windows split along z), False = whole windows on the wide halos."""
    import torch
    capi = importlib.import_module("3dsift_amd.capi")
    synth = importlib.import_module("3dsift_amd.synth")
    nx, ny, nz = dims
    vol = synth.blobs_torch((nz, ny, nx), torch.device("cuda", devices[0]), seed=seed).cpu().numpy()
    t0 = time.perf_counter()
    sh = capi.ShardedCSIFT3D(vol, devices=tuple(devices), sim_ranks=sim_ranks, partial_windows=partial_windows, transport=transport, ghost_octave0=ghost)
    t_ctor = time.perf_counter() - t0
    del vol
    for _ in range(warmup):
        sh.KpSiftAlgorithm()
    ts = []
    for _ in range(steps):
        sh.KpSiftAlgorithm()
        ts.append(sh.info()["seconds"])
    t0 = time.perf_counter()
    kp, _ = sh.GetKeypoints()
    t_get = time.perf_counter() - t0
    info = sh.info()
    hb, wb = sh.traffic()
    per_rank = None
    if sim_ranks and rank_times:
        # what ONE rank does in a step, re-run alone on the GPU on the buffers the last run left behind (sift3d_test_sharded_time_rank): the
        # GPU time of that rank on a node of sim_ranks GPUs, short of what its transfers wait for
        try:
            per_rank = [round(min(sh.time_rank(r) for _ in range(3)) * 1e3, 3) for r in range(info["world"])]
        except Exception as e:  # noqa: BLE001 -- a side measurement
            per_rank = f"{type(e).__name__}: {e}"
    sh.close()
    dt = float(np.median(ts))
    return {"workload": f"{nx}x{ny}x{nz} fp32 synthetic blob volume, z-slabs over {info['world']} rank(s), native C++ driver"
                        + (" SIMULATED on one GPU" if sim_ranks else (" (RCCL)" if transport == "rccl" else
                           (" (copy transport: events + peer copies, no RCCL)" if len(set(devices)) > 1 else " (copy transport: %d rank THREADS on one GPU)" % len(devices)))),
            "value": nx * ny * nz / dt / 1e6, "unit": "Mvoxels/s", "ms_per_step": dt * 1e3, "ms_per_step_all": [round(t * 1e3, 3) for t in ts], "keypoints": int(len(kp)),
            "octave0_on_ghost_zones": bool(ghost),
            "sharded_octaves": info["sharded_octaves"], "halo_planes": info["halo"], "slab_planes": info["planes"], "tail_rank": info["tail_rank"],
            "descriptor_windows": "partial integer histograms" if info["partial_windows"] else ("whole windows on plane halos" if not any(info["stage_partial"]) else
                                  "per sharded octave: " + ", ".join("partial integer histograms" if p else "whole windows on plane halos" for p in info["stage_partial"])),
            "ctor_s_incl_H2D_of_the_slabs": round(t_ctor, 3),
            "get_keypoints_ms_D2H_of_every_rank_and_merge": round(t_get * 1e3, 3),
            "GB_received_per_rank_per_step": {"max": round(max(a + b for a, b in zip(hb, wb)) / 1e9, 3), "per_side_of_an_inner_rank": round(max(a + b for a, b in zip(hb, wb)) / 2e9, 3),
                                              "plane_halos_max": round(max(hb) / 1e9, 3), "window_records_and_histograms_max": round(max(wb) / 1e9, 3)},
            **({"sim_rank_alone_ms": per_rank, "sim_slowest_rank_alone_ms": max(per_rank)} if isinstance(per_rank, list) else ({"sim_rank_alone_ms": per_rank} if per_rank else {})),
            "note": "ms_per_step = host wall time of sift3d_sharded_run: KpSiftAlgorithm, results complete on the devices (like the single-GPU extractor's step; "
                    "r05 counted the D2H of every rank's results and the merge in it: now get_keypoints_ms)"}


def run_slab_native_child(dims_txt, gpus, steps, warmup, partial_windows, timeout, transport="rccl", ghost=False):
    """run_slab_native in a CHILD process (N > 1: rank 0's process starts it while the other ranks idle).  The native driver's RCCL
    point-to-point transport meets its first second GPU in the driver's run: a hard fault there (one host thread per GPU inside librccl)
    must not take the process that holds the headline measurement with it.  Returns (slab dict, None) or (None, reason)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "slab", "--native", "--gpus", str(gpus), "--slab-dims", dims_txt,
           "--steps", str(steps), "--warmup", str(warmup), "--transport", transport] + (["--ghost"] if ghost else []) + ([] if partial_windows is None else ["--partial-windows"] if partial_windows else ["--whole-windows"])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                                              "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "GROUP_WORLD_SIZE", "ROLE_NAME")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    except subprocess.TimeoutExpired:
        return None, "timed out after %d s" % timeout
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)["slab"], None
            except Exception:
                break
    return None, "exit code %d: %s" % (r.returncode, (r.stderr.strip().splitlines() or ["no output"])[-1][:300])


def run_slab(dims, world, rank, local, dev, steps, warmup, sim_ranks=0, seed=4321):
    """Strong-scaling workload: one nx x ny x nz volume, z-slabs over the ranks.  Returns a dict (same on all ranks)."""
    import torch
    import torch.distributed as dist

    capi = importlib.import_module("3dsift_amd.capi")
    synth = importlib.import_module("3dsift_amd.synth")
    slab = importlib.import_module("3dsift_amd.slab")
    s3d_dist = importlib.import_module("3dsift_amd.dist")
    nx, ny, nz = dims
    shape = (nz, ny, nx)
    torch.cuda.set_device(local)  # the current device is per thread, and the N>1 leg runs in a watchdog thread

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    nranks = sim_ranks if sim_ranks else world
    if nranks == 1:
        vol = synth.blobs_torch(shape, dev, seed=seed)
        torch.cuda.synchronize()
        ex = capi.CSIFT3D(None, device=local, device_ptr=vol.data_ptr(), shape=shape)
        del vol
        step = ex.KpSiftAlgorithm
        count = lambda: len(ex.GetKeypoints(with_desc=False)[0])
        detail = lambda: {k: round(v * 1e3, 3) for k, v in ex.m_timer.items()}
    else:
        comm = slab.SimComm(sim_ranks) if sim_ranks else slab.DistComm()
        ex = slab.SlabExtractor(dims, comm, device=local)
        slabs = {r: synth.blobs_torch(shape, dev, seed=seed, zrange=ex.bounds[r]) for r in comm.local_ranks()}
        torch.cuda.synchronize()
        ex.load(device_slabs=slabs)
        del slabs
        step = ex.KpSiftAlgorithm

        def count():
            n0 = sum(int(st.ctx.device_results()[2]) for w in ex._wl() for st in w.stages)
            t = torch.tensor([n0], dtype=torch.int64, device=dev)
            if world > 1 and not sim_ranks:
                dist.all_reduce(t)
            nt = int(ex._wl()[0].tail.device_results()[2]) if ex.noct > ex.S else 0
            return int(t.item()) + nt

        detail = lambda: {k: round(v * 1e3, 3) for k, v in ex.times.items()}
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = s3d_dist.max_over_ranks(time.perf_counter() - t0, device=dev)
    res = {"workload": f"{nx}x{ny}x{nz} fp32 synthetic blob volume, z-slabs over {nranks} rank(s)"
                       + (" SIMULATED on one GPU" if sim_ranks else ""),
           "value": nx * ny * nz * steps / dt / 1e6, "unit": "Mvoxels/s", "ms_per_step": dt / steps * 1e3,
           "keypoints": count(), "last_step_ms": detail()}
    if nranks > 1:
        res["halo_planes"] = ex.halo
        hb, wb = ex.halo_bytes(), ex.window_bytes()   # plane halos (from the posted plan) + records / partial histograms of the descriptor windows (r05, last step's counts)
        tot = [a + b for a, b in zip(hb, wb)]
        res["halo_GB_received_per_rank_per_step"] = {"max": round(max(tot) / 1e9, 3), "per_side_of_an_inner_rank": round(max(tot) / 2e9, 3),
                                                     "plane_halos_max": round(max(hb) / 1e9, 3), "window_records_and_histograms_max": round(max(wb) / 1e9, 3),
                                                     "descriptor_windows": "partial integer histograms" if ex.desc_partial else "whole windows on plane halos"}
        res["sharded_octaves"] = ex.S
        res["slab_planes"] = [b[1] - b[0] for b in ex.bounds]
    ex.close()
    return res


def guarded(fn, seconds):
    """run fn() in a thread; (result, None) or (None, reason) when it raised or did not finish in time"""
    import threading
    box = {}

    def body():
        try:
            box["ok"] = fn()
        except BaseException as e:  # noqa: BLE001 -- reported in the JSON line
            box["err"] = f"{type(e).__name__}: {e}"

    th = threading.Thread(target=body, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return None, f"timeout after {seconds} s"
    if "err" in box:
        return None, box["err"]
    return box["ok"], None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=512, help="cubic volume edge (BASELINE metric: 512)")
    ap.add_argument("--cpu-sample", type=int, default=512, help="edge of the CPU-baseline sample crop (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-match", action="store_true", help="skip the configs[2] matcher leg")
    ap.add_argument("--no-nonaligned", action="store_true", help="skip the non-tile-aligned volume leg")
    ap.add_argument("--nonaligned-dims", default="480x500x300", help="nx x ny x nz of the non-aligned leg")
    ap.add_argument("--thin-dims", default="512x512x32", help="nx x ny x nz of the thin-volume leg")
    ap.add_argument("--allpairs", action="store_true", help="N>1: all-gather descriptors + all-pairs enhancedMatch (configs[4])")
    ap.add_argument("--workload", choices=["volumes", "slab"], default="volumes")
    ap.add_argument("--slab-dims", default="1024x1024x512", help="nx x ny x nz of the sharded volume (configs[3])")
    ap.add_argument("--sim-ranks", type=int, default=0, help="slab workload: simulate R ranks on one GPU")
    ap.add_argument("--no-slab-leg", action="store_true", help="do not append the configs[3] measurement (N=1: the single-GPU run of the 1024x1024x512 volume; N>1: z-slabs over the ranks)")
    ap.add_argument("--strict-legs", action="store_true", help="exit non-zero when a side leg (slab / slab_native) failed; the JSON line is printed either way")
    ap.add_argument("--native", action="store_true", help="slab workload on one process: the library's native C++ driver (RCCL over --gpus devices, or --sim-ranks)")
    ap.add_argument("--partial-windows", action="store_true", help="with --native: descriptor windows split along z over the ranks or a refusal (the default is the driver's rule: split unless a slab is too thin)")
    ap.add_argument("--no-rank-times", action="store_true", help="with --native --sim-ranks: skip the solo re-run of every rank (sim_rank_alone_ms)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "copies"], help="with --native: RCCL point-to-point (one rank per GPU), or the copy transport (events + device / peer copies)")
    ap.add_argument("--rank-threads", type=int, default=0, help="with --native on ONE GPU: that many rank threads sharing the GPU over the copy transport (the multi-threaded driver without a second GPU)")
    ap.add_argument("--ghost", action="store_true", help="with --native: octave 0 on ghost zones (SIFT3D_SHARDED_GHOST_OCTAVE0): recomputed instead of exchanged level by level")
    ap.add_argument("--whole-windows", action="store_true", help="with --native: whole descriptor windows on the wide plane halos (r05's default)")
    args = ap.parse_args()

    import torch

    import torch.distributed as dist

    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    capi = importlib.import_module("3dsift_amd.capi")
    synth = importlib.import_module("3dsift_amd.synth")
    s3d_dist = importlib.import_module("3dsift_amd.dist")
    rank, world = s3d_dist.init_from_env(backend="nccl", device=dev)  # "nccl" is RCCL on ROCm

    if args.workload == "slab" and args.native and world == 1:
        dims = parse_dims(args.slab_dims)
        devs = [local] * args.rank_threads if args.rank_threads else (list(range(max(1, args.gpus))) if not args.sim_ranks else [local])
        r = run_slab_native(dims, devs, args.steps, args.warmup, sim_ranks=args.sim_ranks,
                            partial_windows=True if args.partial_windows else (False if args.whole_windows else None), rank_times=not args.no_rank_times,
                            transport="copies" if args.rank_threads else args.transport, ghost=args.ghost)
        print(json.dumps({"metric": "Mvoxels/s end-to-end KpSiftAlgorithm, one volume sharded as z-slabs (native driver)", "value": r["value"],
                          "unit": "Mvoxels/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": r["workload"], "parallelism": f"z-slabs x{args.sim_ranks or args.rank_threads or args.gpus}"}, "slab": r}))
        return
    if args.workload == "slab":
        dims = parse_dims(args.slab_dims)
        r = run_slab(dims, world, rank, local, dev, args.steps, args.warmup, sim_ranks=args.sim_ranks)
        out = {"metric": "Mvoxels/s end-to-end KpSiftAlgorithm, one volume sharded as z-slabs", "value": r["value"],
               "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": r["workload"], "parallelism": f"z-slabs x{args.sim_ranks or world}"}, "slab": r}
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    n = args.size
    shape = (n, n, n)
    vol = synth.blobs_torch(shape, dev, seed=1234 + rank)  # one volume per GPU
    torch.cuda.synchronize()
    ex = capi.CSIFT3D(None, device=local, device_ptr=vol.data_ptr(), shape=shape)  # ctor: D2D copy + normalise

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ex.KpSiftAlgorithm()
    barrier()
    t0 = time.perf_counter()
    stage = {}
    for _ in range(args.steps):
        ex.KpSiftAlgorithm()  # returns after its stream drained
        for k, v in ex.m_timer.items():
            stage[k] = stage.get(k, 0.0) + v
    barrier()
    dt = time.perf_counter() - t0
    dt = s3d_dist.max_over_ranks(dt, device=dev)
    kp, _ = ex.GetKeypoints(with_desc=False)
    nkp = len(kp)
    next_ = len(ex.extrema())
    stage = {k: v / args.steps for k, v in stage.items()}

    pv, noct = pyramid_voxels(shape)
    # HBM traffic of the pyramid stage from the PMC counters (scripts/measure_traffic.py, separate --pmc passes,
    # gfx950 FETCH_SIZE correction); measured offline on the same workload and committed under profiles/
    traffic, traffic_note = None, None
    tfile = os.path.join(ROOT, "profiles", f"pyramid_traffic_{n}.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            # the file is only valid for the kernel sources it was measured on (rocprofv3 cannot run inside this process)
            if tj.get("kernel_source_sha") == capi.kernel_source_sha():
                traffic = float(tj["total_bytes"])
            else:
                traffic_note = "profiles/pyramid_traffic file was measured on other kernel sources: not reported"
        except Exception:
            traffic = None
    t_pyr = stage["d_BuildGSS"] + stage["d_BuildDOG"]
    # What the pyramid kernels MOVE: per octave 5 Gaussian levels read + written (40 B per voxel) + 3 DoG levels written (12 B) = 52 B.
    # Not built at all: DoG[0] and DoG[nd-1] (only candidate voxels ever read them; the extrema test forms those values from the two
    # Gaussian levels) and the last Gaussian level G[nd] (k_lazy_next evaluates it at the few thousand voxels that pass seven of the
    # eight tests of the last keypoint level -- that kernel's time is in d_Detect, its reads are 17^3 samples per parked voxel out
    # of the L2).  SURVEY 8d's accounting of an unfused build is 68 B: `frac_every_level_built` below TIMES that work.
    moved_bytes = 52.0 * pv
    achieved = moved_bytes / t_pyr / 1e9
    out = {
        "metric": "Mvoxels/s end-to-end KpSiftAlgorithm on 512^3 fp32",
        "value": world * n ** 3 * args.steps / dt / 1e6,
        "unit": "Mvoxels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{n}^3 fp32 synthetic blob volume per GPU, default SIFT params, full KpSiftAlgorithm "
                               f"({noct} octaves, {next_} DoG extrema -> {nkp} keypoints)",
                   "volumes_per_gpu": 1, "parallelism": f"independent volumes x{world}"},
        "stage_ms": {k: round(v * 1e3, 4) for k, v in stage.items()},
        "descriptor_keypoints_per_s": (nkp / stage["d_Extraction"]) if stage["d_Extraction"] > 0 else None,
        "roofline": {"bound": "hbm", "kernel": "pyramid build (all Gaussian/DoG level kernels of one KpSiftAlgorithm: k_march_level / k_conv_axis / k_downsample)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "algorithmic_bytes": moved_bytes, "bytes_per_pyramid_voxel": 52, "seconds": t_pyr,
                     "traffic": traffic, "traffic_note": traffic_note},
    }
    # ... and SURVEY 8d's work AS DEFINED, timed ("work moved out of the timed region earns no credit" -- the default
    # build leaves G[nd], DoG[0] and DoG[nd-1] to the detection stage): the same extractor with every Gaussian and DoG level built and
    # written (hooks glast_eager + dog_eager), its pyramid stage against the 68 B per pyramid voxel of that accounting.
    try:
        if not (rank == 0 and world == 1):
            raise RuntimeError("measured at N = 1 only")
        with capi.hook("glast_eager", 1), capi.hook("dog_eager", 1):
            exe = capi.CSIFT3D(None, device=local, device_ptr=vol.data_ptr(), shape=(n, n, n))
            te = []
            for _ in range(2 + 5):
                exe.run_stages(1)
                te.append(exe.m_timer["d_BuildGSS"] + exe.m_timer["d_BuildDOG"])
            exe.close()
        t_eager = float(np.median(te[2:]))
        # scalars of the roofline object itself, so that the as-defined figure travels wherever the line's roofline is quoted
        out["roofline"]["frac_every_level_built"] = 68.0 * pv / t_eager / 1e9 / HBM_PEAK_GBS
        out["roofline"]["seconds_every_level_built"] = t_eager
        out["roofline"]["bytes_every_level_built"] = 68.0 * pv
        out["roofline"]["every_level_built_note"] = ("all 6 Gaussian + 5 DoG levels of every octave built and written (hooks glast_eager, dog_eager): "
                                                     "SURVEY 8d's 68 B per pyramid voxel with nothing deferred to detection")
    except Exception as e:  # noqa: BLE001 -- a side measurement
        out["roofline"]["every_level_built_note"] = f"not measured: {type(e).__name__}: {e}"
    # ---- descriptor stage (keypoints/s and window-voxels/s, not an HBM fraction) + the VALU-issue roofline of k_describe
    if stage["d_Extraction"] > 0 and nkp:
        wv = descriptor_window_voxels(kp)
        dsc = {"keypoints": nkp, "seconds": stage["d_Extraction"], "keypoints_per_s": nkp / stage["d_Extraction"],
               "window_voxels": wv, "window_voxels_per_s": wv / stage["d_Extraction"]}
        # wave-instructions of k_describe from the committed SQ-counter profile of the same kernel sources (rocprofv3 cannot run inside
        # this process); issue peak = the guide's rate: 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction
        pfile = os.path.join(ROOT, "profiles", f"pmc_k_describe_{n}.json")
        if os.path.exists(pfile):
            try:
                pj = json.load(open(pfile))
                if pj.get("kernel_source_sha") == capi.kernel_source_sha():
                    valu = float(pj["SQ_INSTS_VALU"])
                    peak = VALU_ISSUE_PEAK
                    dsc["roofline"] = {"bound": "valu-issue", "kernel": "k_describe", "achieved": valu / stage["d_Extraction"] / 1e9, "peak": peak / 1e9,
                                       "unit": "G wave-instructions/s", "frac": valu / stage["d_Extraction"] / peak,
                                       "valu_wave_instructions": valu, "per_keypoint": valu / nkp,
                                       "lds_bank_conflict_share": (pj["SQ_LDS_BANK_CONFLICT"] / pj["SQ_LDS_IDX_ACTIVE"]) if pj.get("SQ_LDS_IDX_ACTIVE") else None}
                else:
                    dsc["roofline_note"] = "profiles/pmc_k_describe file was measured on other kernel sources: not reported"
            except Exception:
                pass
        out["descriptor"] = dsc
    out["debug_counters"] = ex.debug_counters()

    if rank == 0 and world == 1:
        # ---- measured device-copy ceiling beside the spec peak: the library's float4 copy kernel on 1 GiB, read + write counted
        # (the MI355X guide measures 6.29 TB/s with such a kernel)
        copy_gbs = capi.copy_bandwidth(1 << 30, 5, local)
        out["roofline"]["copy_ceiling_GBs"] = copy_gbs
        out["roofline"]["frac_of_copy_ceiling"] = achieved / copy_gbs
        # ---- constructor (never part of `value`): D2D copy + max-abs normalise from a device-resident volume, and the same from
        # a host volume (PCIe H2D included)
        tcs = []
        for _ in range(3):
            tc0 = time.perf_counter(); exc = capi.CSIFT3D(None, device=local, device_ptr=vol.data_ptr(), shape=shape); torch.cuda.synchronize()
            tcs.append(time.perf_counter() - tc0); exc.close()
        host_vol = vol.cpu().numpy()
        tc0 = time.perf_counter(); exc = capi.CSIFT3D(host_vol, device=local); torch.cuda.synchronize()
        t_h2d = time.perf_counter() - tc0
        exc.close(); del host_vol
        out["ctor_ms"] = {"device_resident_volume": round(min(tcs) * 1e3, 3), "host_volume_incl_H2D": round(t_h2d * 1e3, 3),
                          "note": "arena allocation + copy + data_scale; outside KpSiftAlgorithm in the reference too (Src/cSIFT3D.cc:146-163); "
                                  "the host volume is pageable memory, staged through pinned chunks (csrc/staging.hip)"}
        tg, tg2 = [], []
        bufs = (np.zeros(nkp, capi.KP_DTYPE), np.ones((nkp, 768), np.float32))  # touched: the caller's vectors exist already
        for _ in range(4):
            tc0 = time.perf_counter(); ex.GetKeypoints(out=bufs); tg.append(time.perf_counter() - tc0)
            tc0 = time.perf_counter(); ex.GetKeypoints(); tg2.append(time.perf_counter() - tc0)
        out["get_keypoints_ms"] = {"with_descriptors_D2H": round(min(tg) * 1e3, 3), "bytes": int(nkp * (768 * 4 + 168)),
                                   "GBs": round(nkp * (768 * 4 + 168) / min(tg) / 1e9, 1),
                                   "into_freshly_allocated_arrays": round(min(tg2) * 1e3, 3),
                                   "note": "sift3d_get_keypoints into arrays the caller has touched; the second figure adds the page faults of two new numpy arrays"}
    if rank == 0 and world == 1 and not args.no_nonaligned:
        # ---- a volume whose width and height are NOT multiples of the 32 x 32 tile of the level kernel (not part of `value`): the
        # pyramid stage priced like the headline's, per pyramid voxel
        def leg(dims, **params):
            nx2, ny2, nz2 = dims
            shape2 = (nz2, ny2, nx2)
            v2 = synth.blobs_torch(shape2, dev, seed=4242)
            torch.cuda.synchronize()
            e2 = capi.CSIFT3D(None, device=local, device_ptr=v2.data_ptr(), shape=shape2, **params)
            e2.KpSiftAlgorithm()
            st2, t2 = {}, []
            for _ in range(5):
                tc0 = time.perf_counter(); e2.KpSiftAlgorithm(); t2.append(time.perf_counter() - tc0)
                for k, v in e2.m_timer.items():
                    st2[k] = st2.get(k, 0.0) + v / 5
            pv2, noct2 = pyramid_voxels(shape2)
            tp2 = st2["d_BuildGSS"] + st2["d_BuildDOG"]
            e2.close(); del v2
            return {"workload": f"{nx2}x{ny2}x{nz2} fp32 synthetic blob volume ({noct2} octaves)", "ms_per_step": float(np.median(t2)) * 1e3,
                    "Mvoxels_per_s": nx2 * ny2 * nz2 / float(np.median(t2)) / 1e6, "stage_ms": {k: round(v * 1e3, 4) for k, v in st2.items()},
                    "pyramid_frac": 52.0 * pv2 / tp2 / 1e9 / HBM_PEAK_GBS, "pyramid_ns_per_pyramid_voxel": tp2 / pv2 * 1e9}

        nx2, ny2, nz2 = parse_dims(args.nonaligned_dims)
        out["nonaligned"] = leg((nx2, ny2, nz2))
        # the same volume with width and height rounded up to whole 32 x 32 tiles: what the shifted last tile column / row costs
        twin = leg(((nx2 + 31) // 32 * 32, (ny2 + 31) // 32 * 32, nz2))
        out["nonaligned"]["aligned_twin"] = twin
        out["nonaligned"]["pyramid_ns_per_voxel_vs_twin"] = out["nonaligned"]["pyramid_ns_per_pyramid_voxel"] / twin["pyramid_ns_per_pyramid_voxel"]
        # ---- a THIN volume (r04, shape cliff #2): planes of many tiles, few of them; priced per pyramid voxel against the headline volume
        tx, ty, tz = parse_dims(args.thin_dims)
        out["thin"] = leg((tx, ty, tz))
        out["thin"]["pyramid_ns_per_voxel_vs_headline"] = out["thin"]["pyramid_ns_per_pyramid_voxel"] / (t_pyr / pv * 1e9)
        # a volume of this few voxels is bound by the launch chain of its small octaves whatever its shape: the CUBE of the same voxel
        # count (edge rounded to 16) is the like-for-like comparison of what the thin shape costs
        ce = max(16, int(round((tx * ty * tz) ** (1.0 / 3.0) / 16.0)) * 16)
        cube = leg((ce, ce, ce))
        out["thin"]["equal_voxel_cube"] = {k: cube[k] for k in ("workload", "ms_per_step", "stage_ms", "pyramid_ns_per_pyramid_voxel")}
        out["thin"]["pyramid_ns_per_voxel_vs_equal_voxel_cube"] = out["thin"]["pyramid_ns_per_pyramid_voxel"] / cube["pyramid_ns_per_pyramid_voxel"]
        # ---- other constructor arguments (Include/cSIFT3D.h:187-194 makes them ordinary arguments): the pyramid stage per pyramid
        # voxel against the headline's.  sigma_default 2.0: half widths 4, 4, 5, 6, 8 fused and 10 for the last level (r06: evaluated at the
        # parked candidates like the default's, it used to be BUILT by three separable passes); 1.7: half width 7 (r06: fused); two keypoint levels
        out["other_parameters"] = {}
        for tag, prm in (("sigma_default_2.0", dict(sigma_default=2.0)), ("sigma_default_1.7", dict(sigma_default=1.7)), ("num_kp_levels_2", dict(num_kp_levels=2))):
            try:
                lg = leg((n, n, n), **prm)
                out["other_parameters"][tag] = {"ms_per_step": lg["ms_per_step"], "stage_ms": lg["stage_ms"], "pyramid_ns_per_pyramid_voxel": lg["pyramid_ns_per_pyramid_voxel"],
                                                "pyramid_ns_per_voxel_vs_headline": lg["pyramid_ns_per_pyramid_voxel"] / (t_pyr / pv * 1e9)}
            except Exception as e:  # noqa: BLE001 -- a side leg
                out["other_parameters"][tag] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu and args.cpu_sample > 0:
        # ---- CPU baseline + parity on a bounded sample: the [0:s]^3 crop of the same volume ----
        import oracle_lib as ol  # test infrastructure, used here ONLY as the timed baseline / checker

        s = min(args.cpu_sample, n)
        crop = vol[:s, :s, :s].contiguous().cpu().numpy()
        orc = ol.load("orc")
        # the GPU boxes expose 256 hardware threads shared with other tenants; OpenMP barriers collapse
        # beyond ~64 threads there (measured), so the baseline uses min(64, half the logical CPUs)
        cores = max(1, min(64, (os.cpu_count() or 2) // 2))
        orc.set_threads(cores)
        tcs = []
        for _ in range(3):  # median of three
            o = orc.extractor(crop)
            tc = time.perf_counter()
            o.run(5)
            tcs.append(time.perf_counter() - tc)
        tcpu = float(np.median(tcs))
        okp, odesc = o.keypoints()
        # one thread on a bounded crop (a scalar 512^3 run would take minutes)
        s1 = min(160, s)
        orc.set_threads(1)
        o1 = orc.extractor(np.ascontiguousarray(crop[:s1, :s1, :s1]))
        tc = time.perf_counter(); o1.run(5); t1 = time.perf_counter() - tc
        orc.set_threads(cores)
        out["cpu_baseline"] = {"value": s ** 3 / tcpu / 1e6, "unit": "Mvoxels/s", "cores": cores, "kind": "port",
                               "sample": f"[0:{s}]^3 crop of the benchmark volume, full KpSiftAlgorithm, median of 3 runs {tcpu:.2f} s "
                                         f"(runs {[round(t, 2) for t in tcs]}), {len(okp)} keypoints; stages of the last run "
                                         f"{json.dumps({k: round(v, 3) for k, v in o.times.items()})}",
                               "one_thread": {"value": s1 ** 3 / t1 / 1e6, "unit": "Mvoxels/s", "cores": 1,
                                              "sample": f"[0:{s1}]^3 crop, one run, {t1:.2f} s"}}
        # ---- the REAL reference (oracle/_ref/libref3dsift.so: the untouched sources of /root/reference compiled by `make -C oracle
        # ref`, a binary that travels with the tree) on a bounded crop with the same threads: its OpenMP path as it is -- transposes,
        # serial boundary sweep (Src/cSIFT3D.cc:609-617, 722-788) -- next to the restatement above (which is ~20x faster per core)
        ref_block = None
        if ol.available("ref"):
            try:
                ref = ol.load("ref")
                ref.set_threads(cores)
                sr = min(256, s)
                rcrop = np.ascontiguousarray(crop[:sr, :sr, :sr])
                ro = ref.extractor(rcrop)
                tc = time.perf_counter(); ro.run(5); tr = time.perf_counter() - tc
                rkp, rdesc = ro.keypoints()
                # the restatement on the same crop: pins it to the reference in this very run (same keypoints, same descriptor bits)
                oo = orc.extractor(rcrop).run(5)
                okp2, odesc2 = oo.keypoints()
                ref_block = {"value": sr ** 3 / tr / 1e6, "unit": "Mvoxels/s", "cores": cores, "kind": "reference",
                             "sample": f"[0:{sr}]^3 crop of the benchmark volume, full KpSiftAlgorithm, one run {tr:.2f} s, {len(rkp)} keypoints",
                             "stages_s": {k: round(v, 3) for k, v in ro.times.items()},
                             # (every field but the eigenvectors, which carry their solver's sign: Eigen there, Jacobi here)
                             "port_equals_reference_on_this_crop": bool(len(rkp) == len(okp2) and all(
                                 np.array_equal(np.ascontiguousarray(rkp[f]).view(np.uint32), np.ascontiguousarray(okp2[f]).view(np.uint32))
                                 for f in ("x", "y", "z", "scale", "octave", "level", "rx", "ry", "rz", "win", "eigvalue", "Rotation", "str_tensor"))
                                 and np.array_equal(rdesc.view(np.uint32), odesc2.view(np.uint32)))}
                ro.close(); oo.close()
            except Exception as e:  # noqa: BLE001 -- reported, never fatal for the headline
                ref_block = {"error": f"{type(e).__name__}: {e}"}
        out["cpu_baseline"]["reference"] = ref_block if ref_block is not None else None
        if ref_block is None:
            out["cpu_baseline"]["reference_note"] = "oracle/_ref/libref3dsift.so is not in this tree (it is built only where /root/reference exists)"
        g = capi.CSIFT3D(crop, device=local).KpSiftAlgorithm()
        gkp, gdesc = g.GetKeypoints()
        same = len(gkp) == len(okp) and all(np.array_equal(gkp[f], okp[f]) for f in ("x", "y", "z", "octave", "level"))
        from hipcheck import descriptor_errors
        rms, worst_kp, worst_abs = descriptor_errors(gdesc, odesc) if same and len(okp) else (None, None, None)
        out["parity"] = {"sample_keypoints_gpu": len(gkp), "sample_keypoints_cpu": len(okp), "same_keypoint_set": bool(same),
                         "descriptor_rms": rms, "worst_keypoint_rms": worst_kp, "worst_element_abs": worst_abs,
                         "bars": "2e-5 RMS per keypoint, 1e-4 per element (tests/hipcheck.py)"}
    if rank == 0 and world == 1 and not args.no_match:
        # ---- BASELINE configs[2] leg (not part of `value`): second volume = the same blobs shifted by one voxel in x,
        # extract, then muBruteMatcher::enhancedMatch on the device-resident descriptors; the score GEMM is the one MFMA
        # kernel of the path (v_mfma_f32_32x32x2_f32), priced against the dense f32 matrix peak
        vol2 = synth.blobs_torch(shape, dev, seed=1234 + rank, shift=(1.0, 0.0, 0.0))
        torch.cuda.synchronize()
        ex2 = capi.CSIFT3D(None, device=local, device_ptr=vol2.data_ptr(), shape=shape)
        del vol2
        ex2.KpSiftAlgorithm()
        (da, xa, na), (db, xb, nb) = ex.device_results(), ex2.device_results()
        mt = capi.muBruteMatcher(device=local)
        secs, secs_e, wall, wall_e, exact = [], [], [], [], []
        for _ in range(6):   # (the first launches of the MFMA kernel run ~8 % below the settled rate: min of six)
            mt.injectMatch(da, xa, db, xb, 0.85, on_device=True, n=na, m=nb)   # one full N x M pass: exact flop count
            secs.append(mt.totalTime); wall.append(mt.wallTime); exact.append(mt.exact_rows)
            r = mt.enhancedMatch(da, xa, db, xb, 0.85, on_device=True, n=na, m=nb)
            secs_e.append(mt.totalTime); wall_e.append(mt.wallTime)
        tm = min(secs)
        flop = 2.0 * na * nb * 768
        tmed = float(np.median(secs))
        out["matcher"] = {"workload": f"injectMatch (one full pass) of {na} x {nb} descriptors (two {n}^3 volumes, second shifted 1 voxel)",
                          "seconds": tm, "seconds_median": tmed, "seconds_all": [round(v, 6) for v in secs],
                          "wall_seconds": min(wall), "wall_seconds_median": float(np.median(wall)),
                          "enhancedMatch_seconds": min(secs_e), "enhancedMatch_seconds_median": float(np.median(secs_e)),
                          "enhancedMatch_wall_seconds": min(wall_e), "enhancedMatch_wall_seconds_median": float(np.median(wall_e)),
                          "rows_rescored_exactly": int(exact[-1]), "matched_pairs": int(len(r["pairs"])),
                          "roofline": {"bound": "mfma", "kernel": "k_scores_topk2 (A.B^T on v_mfma_f32_32x32x2_f32, fused top-K) + k_row_norm2 + k_merge_top4 + k_rescore + k_exact_rows: device time of the whole pass",
                                       "achieved": flop / tmed / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                       "frac": flop / tmed / 1e12 / MFMA_F32_PEAK_TF,
                                       "achieved_best": flop / tm / 1e12, "frac_best": flop / tm / 1e12 / MFMA_F32_PEAK_TF,
                                       "note": "achieved / frac: the MEDIAN of six repetitions; *_best: the best of the six (the first launches of a process run ~8 % below the settled rate)",
                                       "traffic": None}}
        if not args.no_cpu:
            # parity gate of the measured match: the oracle's matcher (restatement of Src/cMatcher.cc, OpenMP) on the SAME
            # two descriptor sets -- device-resident inputs on the GPU side
            import oracle_lib as ol
            orc = ol.load("orc")
            orc.set_threads(max(1, min(64, (os.cpu_count() or 2) // 2)))
            kpa, dsa = ex.GetKeypoints(); kpb, dsb = ex2.GetKeypoints()
            xa_h = np.stack([kpa["rx"], kpa["ry"], kpa["rz"]], 1); xb_h = np.stack([kpb["rx"], kpb["ry"], kpb["rz"]], 1)
            tmc = time.perf_counter()
            want = orc.match(dsa, xa_h, dsb, xb_h, 0.85, 3)
            tmc = time.perf_counter() - tmc
            out["matcher"]["parity"] = {"pairs_equal": bool(np.array_equal(r["pairs"], want["pairs"])),
                                        "gIdx_equal": bool(np.array_equal(r["gIdx"], want["gIdx"])),
                                        "sIdx_equal": bool(np.array_equal(r["sIdx"], want["sIdx"])),
                                        "gDist_equal": bool(np.array_equal(r["gDist"], want["gDist"])),
                                        "oracle_enhancedMatch_seconds": round(tmc, 3)}
        # ---- BASELINE configs[2] as a pipeline (r04; not part of `value`): both volumes' KpSiftAlgorithm enqueued by ONE host thread
        # (sift3d_run_async on each handle's own streams), waited for, then enhancedMatch straight from the device-resident results
        blk, par, tot = [], [], []
        for _ in range(6):
            torch.cuda.synchronize()
            t0p = time.perf_counter(); ex.KpSiftAlgorithm(); ex2.KpSiftAlgorithm(); blk.append(time.perf_counter() - t0p)
            t0p = time.perf_counter(); ex.KpSiftAlgorithmAsync(); ex2.KpSiftAlgorithmAsync(); ex.Wait(); ex2.Wait()
            t1p = time.perf_counter()
            (da, xa, na), (db, xb, nb) = ex.device_results(), ex2.device_results()
            r2 = mt.enhancedMatch(da, xa, db, xb, 0.85, on_device=True, n=na, m=nb)
            t2p = time.perf_counter()
            par.append(t1p - t0p); tot.append(t2p - t0p)
        mp, mb = float(np.median(par[1:])), float(np.median(blk[1:]))
        out["pipeline2"] = {"workload": f"two {n}^3 volumes: KpSiftAlgorithm of both in flight on one GPU (sift3d_run_async x2 + sift3d_wait x2 from one host thread), then enhancedMatch on the device-resident descriptors",
                            "extract2_ms_in_flight": mp * 1e3, "extract2_ms_one_after_the_other": mb * 1e3,
                            "aggregate_Mvoxels_per_s": 2 * n ** 3 / mp / 1e6, "aggregate_Mvoxels_per_s_one_after_the_other": 2 * n ** 3 / mb / 1e6,
                            "extract2_plus_enhancedMatch_wall_ms": float(np.median(tot[1:])) * 1e3,
                            "same_pairs_as_the_blocking_path": bool(np.array_equal(r2["pairs"], r["pairs"]))}
        ex2.close()
    if args.allpairs and world > 1:
        # BASELINE configs[4] matching leg (not part of `value`): all-gather the device-resident descriptors
        # over RCCL, then every rank runs enhancedMatch on its share of the ordered volume pairs
        # the rank's contribution to the all-gather: D2D export of the device-resident results into buffers the communication
        # layer owns (sift3d_export_device) -- no host hop
        _, _, nk = ex.device_results()
        desc_t = torch.empty((nk, 768), dtype=torch.float32, device=dev)
        xyz_t = torch.empty((nk, 3), dtype=torch.float32, device=dev)
        ex.export_device(desc_t.data_ptr(), xyz_t.data_ptr())
        torch.cuda.synchronize()
        mt = capi.muBruteMatcher(device=local)
        acc = {"pairs": 0, "tm": 0.0}

        def match_fn(da_, xa_, db_, xb_):
            r_ = mt.enhancedMatch(da_.data_ptr(), xa_.data_ptr(), db_.data_ptr(), xb_.data_ptr(), 0.85, on_device=True,
                                  n=da_.shape[0], m=db_.shape[0])
            acc["pairs"] += len(r_["pairs"]); acc["tm"] += mt.totalTime
            return len(r_["pairs"])

        ta = time.perf_counter()
        s3d_dist.allpairs_match(desc_t, xyz_t, match_fn)   # all-gather (two ragged collectives) + this rank's 7 of the 56 pairs
        torch.cuda.synchronize()
        t_gather = time.perf_counter() - ta - acc["tm"]
        npairs, tm = acc["pairs"], acc["tm"]
        tm = s3d_dist.max_over_ranks(tm, device=dev)
        if rank == 0:
            out["allpairs"] = {"allgather_s": t_gather, "match_s_max_rank": tm, "ordered_pairs": len(s3d_dist.ordered_pairs(world)),
                               "rank0_matched": npairs}
    slab_attempted, slab_err = False, None
    SLAB_STEPS, SLAB_WARMUP = 10, 3
    if world == 1 and rank == 0 and not args.no_slab_leg:
        # BASELINE configs[3] at N = 1: the same 1024x1024x512 volume on ONE GPU through the plain extractor -- the denominator of the
        # 1 -> N speed-up the north star asks for, in the default driver record (r04; VERDICT r03 missing #2)
        ex.close(); del vol
        torch.cuda.empty_cache()
        res, slab_err = guarded(lambda: run_slab(parse_dims(args.slab_dims), 1, 0, local, dev, SLAB_STEPS, SLAB_WARMUP), 240)
        out["slab"] = dict(res, n=1, steps=SLAB_STEPS, warmup=SLAB_WARMUP) if slab_err is None else {"error": slab_err}
    if world > 1 and not args.no_slab_leg:
        # BASELINE configs[3] next to the headline number: one 1024x1024x512 volume over the same GPUs.  No multi-GPU box exists in development:
        # every leg runs behind a watchdog and, whatever happens, the headline line is printed.
        #   slab.native (first class since r06)  the library's own driver (csrc/sharded.hip): one process, one host thread per GPU, RCCL
        #                         point-to-point halos, descriptor windows split along z, the tail once on the last rank -- what a C++ user
        #                         of SIFT3D_DEVICES=0-7 gets.  Runs in a CHILD process of rank 0 while the other ranks wait on the host-side store.
        #   slab.native_ghost_octave0   the native driver with octave 0 recomputed on ghost zones instead of exchanged (r06): the plan for nodes whose
        #                         links bound the step
        #   slab.python           3dsift_amd/slab.py over torch.distributed (one process per GPU): the driver the protocol is tested with over gloo
        #   slab.native_copy_transport  the native driver without RCCL: a neighbour's data fetched by peer copies behind the sender's event (r06; the
        #                         transport the multi-threaded driver is TESTED with, as rank threads sharing one GPU)
        slab_attempted = True
        ex.close(); del vol
        torch.cuda.empty_cache()

        def speedup(block):
            # against the single-GPU run of the SAME volume: a committed record of a `--gpus 1` run (profiles/slab_1gpu.json, written by
            # scripts/collect_profile.sh), valid only for the kernel sources it was measured on
            try:
                one = json.load(open(os.path.join(ROOT, "profiles", "slab_1gpu.json")))
                if one.get("kernel_source_sha") == capi.kernel_source_sha() and one.get("dims") == args.slab_dims:
                    block["ms_per_step_n1"] = one["ms_per_step"]
                    block["speedup_vs_n1"] = one["ms_per_step"] / block["ms_per_step"]
                else:
                    block["speedup_note"] = "profiles/slab_1gpu.json was measured on other kernel sources / dims: no speed-up reported"
            except Exception:
                block["speedup_note"] = "no profiles/slab_1gpu.json: no speed-up reported"

        store = None
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:
            store = None
        slab = {"n": world, "steps": SLAB_STEPS, "warmup": SLAB_WARMUP,
                "development_note": "the RCCL transports of both drivers have only met simulated ranks (one GPU) and gloo (CPU) in development"}
        if rank == 0:
            nres, nat_err = run_slab_native_child(args.slab_dims, world, SLAB_STEPS, SLAB_WARMUP, None, 180)
            slab["native"] = nres if nat_err is None else {"error": nat_err}
            if nat_err is None:
                speedup(slab["native"])
            # the same driver over its COPY transport (events + hipMemcpyPeerAsync, no communicator): whatever the RCCL leg did
            cres, c_err = run_slab_native_child(args.slab_dims, world, SLAB_STEPS, SLAB_WARMUP, None, 150, transport="copies")
            slab["native_copy_transport"] = cres if c_err is None else {"error": c_err}
            if c_err is None:
                speedup(slab["native_copy_transport"])
            # ... and with octave 0 on ghost zones (3.5x fewer bytes, none of octave 0's level-by-level exchanges, + 0.5 ms of work per rank): over
            # whichever transport ran
            if nat_err is None or c_err is None:
                gres, g_err = run_slab_native_child(args.slab_dims, world, SLAB_STEPS, SLAB_WARMUP, None, 150, transport="rccl" if nat_err is None else "copies", ghost=True)
                slab["native_ghost_octave0"] = gres if g_err is None else {"error": g_err}
                if g_err is None:
                    speedup(slab["native_ghost_octave0"])
            if store is not None:
                try:
                    store.set("s3d_native_done", "1")
                except Exception:
                    pass
        elif store is not None:
            # (the other ranks wait on the rendezvous STORE, host side: a NCCL barrier would park a spinning kernel on their GPUs while the
            # child's threads use them)
            try:
                import datetime
                store.wait(["s3d_native_done"], datetime.timedelta(seconds=500))
            except Exception:
                pass
        res, slab_err = guarded(lambda: run_slab(parse_dims(args.slab_dims), world, rank, local, dev, SLAB_STEPS, SLAB_WARMUP), 240)
        slab["python"] = res if slab_err is None else {"error": slab_err}
        if slab_err is None:
            speedup(slab["python"])
        # the block's own headline: the native driver's figures where it ran
        okleg = lambda k: rank == 0 and "error" not in slab.get(k, {"error": 1})
        head = slab["native"] if okleg("native") else (slab["native_copy_transport"] if okleg("native_copy_transport") else (slab["python"] if slab_err is None else None))
        if head is not None:
            for k in ("value", "unit", "ms_per_step", "keypoints", "speedup_vs_n1", "ms_per_step_n1"):
                if k in head:
                    slab[k] = head[k]
            slab["driver"] = "native" if head is slab.get("native") else ("native_copy_transport" if head is slab.get("native_copy_transport") else "python")
        else:
            slab["error"] = "both drivers failed: native: %s; python: %s" % (slab.get("native", {}).get("error"), slab_err)
        out["slab"] = slab
    failed = []
    if rank == 0:
        # the line's contract is the headline metric, measured above; a side leg that failed says so IN the line (and on stderr)
        legs = {"slab": out.get("slab")}
        if isinstance(out.get("slab"), dict):
            legs.update({"slab." + k: out["slab"].get(k) for k in ("native", "python", "native_copy_transport", "native_ghost_octave0")})
        failed = [k for k, v in legs.items() if isinstance(v, dict) and "error" in v]
        if failed:
            out["legs_failed"] = failed
            print("bench.py: side leg(s) failed: %s" % ", ".join("%s (%s)" % (k, legs[k]["error"]) for k in failed), file=sys.stderr, flush=True)
        print(json.dumps(out), flush=True)
    if slab_attempted:
        sys.stdout.flush()
        # a wedged collective must not keep the job alive (the process has touched the GPU: no re-exec, no in-process retry).  The
        # exit code is that of the HEADLINE measurement: a failed side leg is reported in the line (`legs_failed`) and on stderr and
        # does not turn a valid N-GPU measurement of the headline metric into a failed run -- unless the caller asked for it
        # (--strict-legs: CI that wants a broken multi-GPU path to fail the job)
        os._exit(1 if (args.strict_legs and failed) else 0)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if args.strict_legs and failed:
        sys.exit(1)


if __name__ == "__main__":
    main()
