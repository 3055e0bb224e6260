"""Z-slab sharding of ONE large volume across the GPUs of a node (BASELINE.json configs[3]).

No reference counterpart: the reference is a single process (Src/cSIFT3D.cc:165-235).  What is sharded is exactly that
pipeline, with results equal to the single-GPU run (pyramid / extrema bit for bit, descriptors to the stated tolerance):

  octave 0   rank r owns the global planes [z0_r, z1_r) of every level; level buffers carry `halo` extra planes per
             side.  Per Gaussian level the ranks exchange the planes the NEXT consumer reaches:
               G[i] -> level i+1 : hw_{i+1}+1 planes per side         (urgent, awaited before level i+1 starts)
               G[1..3]           : up to `halo` planes (orientation + descriptor windows reach +-37 planes; deferred)
               DoG[1..3]         : 1 plane (extrema test reads z+-1; deferred)
             as point-to-point sends between z-neighbours (RCCL over xGMI), posted while the next level computes.
             The normalisation max and the 5 DoG maxima are all-reduced (MAX); x/y passes need no communication.
  octave 1.. the first `sharded_octaves` octaves are sharded the same way: G[o+1][0] = DownSample_3D(G[o][3]) is
             decimated slab-wise straight into the next octave's slab buffers (slab starts are multiples of
             2^sharded_octaves, so plane 2k never leaves its slab); thin slabs simply take halo planes from several ranks.
  tail       the level 0 of the first replicated octave is all-gathered (1/8^S of one level); the remaining octaves run
             replicated in a SEEDED context; their orientation work is dealt by extremum index (integer all-reduce(SUM)
             of zero-padded rows restores it exactly), their descriptor work by keypoint and stays distributed.

Descriptor windows (r05, `desc_partial`, the default): a window reaches up to 38 planes beyond its keypoint, so whole windows need
24 / 30 / 38-plane halos of G[1..3] -- 92 of the 108 planes a rank received per side and octave.  Instead the ranks exchange keypoint
RECORDS (164 bytes) with the z-neighbours a window reaches into, every rank marches the window planes it OWNS for its own and for
the foreign records and returns 768 int32 sums + the part's gradient mass, and the owner adds the parts: the same integers the
single-volume run sums, so the descriptors stay bit-identical.  (The owner itself takes the window planes its level buffers hold,
its halo included; the other ranks their owned planes beyond that.)  The halos of G[1..3] shrink to the orientation window's reach
(8 / 10 / 12 planes).  A record whose fixed-point unit fails is repeated once, by every part, with the exact unit.

Ordering (r02): all device work of a rank's sharded octaves is enqueued on ONE torch stream, which the slab contexts adopt
(sift3d_set_stream); the exchanges are posted in that stream's order (RCCL work is ordered behind the current stream and
`Work.wait()` makes the stream -- not the host -- wait), the DoG maxima are all-reduced as a device tensor, and the host only
synchronises before the extrema counts are read back.  The replicated tail (octaves >= S: small, launch-latency bound, with its
own collectives) runs on a second host thread on the tail context's own stream, beside the sharded detection / descriptors.

The same driver runs over a communicator:
  DistComm  one worker per process, torch.distributed (backend "nccl" == RCCL); works with gloo on CPU tensors for tests
  SimComm   all workers in ONE process on one GPU (device copies instead of sends) -- the bit-exact equality test against
            the single-volume result runs this way on a 1-GPU box (tests/test_gpu_slab.py)
"""
import math
from collections import namedtuple

import numpy as np

# global planes [zg0, zg1) of buffer (kind, idx) of sharded octave `stage`: src -> dst
Transfer = namedtuple("Transfer", "src dst kind idx zg0 zg1 stage", defaults=(0,))

KIND_INPUT, KIND_GSS, KIND_DOG = 0, 1, 2


# --------------------------------------------------------------------------------------------------------------------
# planning (pure python, covered by the CPU tests)
# --------------------------------------------------------------------------------------------------------------------
def octaves_total(nx, ny, nz):
    """Src/cSIFT3D.cc:254-255: (int)log2f(min dim) - 3 + 1"""
    return max(0, int(np.log2(np.float32(min(nx, ny, nz)))) - 2)


def slab_bounds(nz, world, align=2):
    """Owned plane ranges [z0, z1) per rank: contiguous, starts on multiples of `align` (DownSample_3D keeps plane 2k, so
    an even start keeps the decimated planes of a slab inside it; 2^S when S octaves are sharded), as equal as possible;
    the remainder planes go to the last rank."""
    units = nz // align
    if units < world:
        raise ValueError(f"{nz} planes cannot be split into {world} slabs aligned to {align}")
    base, rem = divmod(units, world)
    out, z = [], 0
    for r in range(world):
        n = align * (base + (1 if r < rem else 0))
        out.append((z, z + n))
        z += n
    out[-1] = (out[-1][0], nz)
    return out


def halve_bounds(bounds, nz):
    """owned ranges of the next octave: plane k of octave o+1 is plane 2k of octave o (Src/cSIFT3D.cc:321-344)"""
    return [(z0 // 2, min(z1 // 2, nz // 2)) for z0, z1 in bounds]


def halo_transfers(bounds, nz, kind, idx, lo, hi, stage=0):
    """Transfers that fill, for every rank, the global planes at distance (lo, hi] outside its owned range:
    [z0-hi, z0-lo) and [z1+lo, z1+hi), clipped to the volume, from whichever ranks own them.  Deterministic order
    (destination-major), identical on every rank, so matching sends and receives are posted in the same order."""
    out = []
    if hi <= lo:
        return out
    for r, (z0, z1) in enumerate(bounds):
        for a, b in ((max(0, z0 - hi), max(0, z0 - lo)), (min(nz, z1 + lo), min(nz, z1 + hi))):
            if b <= a:
                continue
            for q, (q0, q1) in enumerate(bounds):
                if q == r:
                    continue
                s, e = max(a, q0), min(b, q1)
                if e > s:
                    out.append(Transfer(q, r, kind, idx, s, e, stage))
    return out


def window_neighbours(bounds, reach):
    """neigh[r] = the ranks q != r whose owned planes a descriptor window of a keypoint of rank r can reach into: the keypoints of r sit
    in [z0_r, z1_r), their windows cover at most `reach` planes either side.  Symmetric; ranks with empty ranges take no part."""
    out = []
    for r, (z0, z1) in enumerate(bounds):
        lo, hi = z0 - reach, z1 - 1 + reach
        out.append([q for q, (q0, q1) in enumerate(bounds) if q != r and z1 > z0 and q1 > q0 and q0 <= hi and q1 - 1 >= lo])
    return out


def capi_words():
    from . import capi
    return capi.ORIENT_WORDS


def described_rows(levels, rank, world):
    """Rows (keypoint slots, reference order) whose descriptor a partitioned handle computes: the library deals the
    accepted keypoints in its processing order -- keypoint level descending, stable (longest windows first,
    kernels_orient.hip k_slots) -- position p goes to rank p % world."""
    levels = np.asarray(levels)
    order = np.argsort(-levels, kind="stable")
    return np.sort(order[rank::world])


def merge_keypoints(parts_oct0, tail):
    """Reference order (octave, level, z, y, x) (Src/cSIFT3D.cc:373-416) from per-slab octave-0 lists plus the tail.
    parts_oct0: list of (kp, desc) per rank; tail: (kp, desc) of octaves >= 1.  Returns (kp, desc)."""
    kps = [p[0] for p in parts_oct0]
    dss = [p[1] for p in parts_oct0]
    kp = np.concatenate(kps) if kps else tail[0][:0]
    ds = np.concatenate(dss) if dss else tail[1][:0]
    if len(kp):
        order = np.lexsort((kp["x"], kp["y"], kp["z"], kp["level"]))
        kp, ds = kp[order], ds[order]
    return np.concatenate([kp, tail[0]]), np.concatenate([ds, tail[1]])


# --------------------------------------------------------------------------------------------------------------------
# communicators
# --------------------------------------------------------------------------------------------------------------------
class SimComm:
    """All ranks live in this process (one GPU): sends become device copies on the caller's current stream, reductions are
    computed directly."""

    stream_ordered = True

    def __init__(self, world):
        self.world = world

    def local_ranks(self):
        return list(range(self.world))

    def exchange(self, workers, transfers):
        for t in transfers:
            workers[t.dst].view(t.kind, t.idx, t.zg0, t.zg1, t.stage).copy_(workers[t.src].view(t.kind, t.idx, t.zg0, t.zg1, t.stage))
        return None

    def wait(self, handle):
        pass  # the copies are ordered on the stream they were issued on

    def exchange_tensors(self, items):
        """items: (src rank, dst rank, tensor at the source, tensor at the destination), the same list in the same order on every
        rank (tensors of ranks living elsewhere are None)."""
        for _src, _dst, a, b in items:
            b.copy_(a)
        return None

    def allgather_ints(self, per_worker):
        """per_worker: one int per local worker (rank order) -> the ints of all ranks"""
        return [int(v) for v in per_worker]

    def sync(self):
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def allreduce_max(self, per_worker):
        m = np.maximum.reduce([np.asarray(a, np.float32) for a in per_worker])
        return [m.copy() for _ in per_worker]

    def allreduce_max_(self, per_worker_tensor):
        """in-place MAX over the workers' device tensors (stream ordered)"""
        import torch
        m = per_worker_tensor[0].clone()
        for t in per_worker_tensor[1:]:
            torch.maximum(m, t, out=m)
        for t in per_worker_tensor:
            t.copy_(m)

    def allgather_planes(self, per_worker_tensor, outs, counts):
        """outs[w][off_r : off_r + counts[r]] = tensor of rank r, for every worker w"""
        off = 0
        for r, t in enumerate(per_worker_tensor):
            for o in outs:
                o[off:off + counts[r]].copy_(t[:counts[r]])
            off += counts[r]

    def allreduce_sum_(self, per_worker_tensor):
        if not per_worker_tensor:
            return
        total = per_worker_tensor[0].clone()
        for t in per_worker_tensor[1:]:
            total += t
        for t in per_worker_tensor:
            t.copy_(total)

    def gather_objects(self, per_worker):
        return list(per_worker)


class DistComm:
    """One worker per process over torch.distributed (nccl == RCCL on ROCm; gloo with CPU tensors in the tests).  With RCCL the
    collectives and point-to-point operations are ordered behind the CURRENT torch stream and `Work.wait()` blocks that stream,
    not the host: the driver issues them under the worker's stream and never synchronises the host between the levels."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.stream_ordered = dist.get_backend() == "nccl"
        # the replicated tail runs on a second host thread and has its own collective (the integer SUM all-reduce of the orientation
        # rows) while the main thread exchanges the records / partial histograms of the sharded octaves' descriptor windows (r05):
        # operations of ONE communicator must be issued in the same order on every rank, so the tail gets its own
        self.tail_group = dist.new_group(ranks=list(range(self.world)))
        # (its communicator is created by its first collective: here, on the constructing thread, not later inside the tail's thread
        # beside the main thread's point-to-point traffic)
        import torch
        warm = torch.zeros(1, dtype=torch.int32, device="cuda" if self.stream_ordered else "cpu")
        dist.all_reduce(warm, op=dist.ReduceOp.SUM, group=self.tail_group)

    def local_ranks(self):
        return [self.rank]

    def sync(self):
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def exchange(self, workers, transfers):
        """workers: [the local worker].  Posts every send / receive this rank takes part in as ONE batched group."""
        dist = self.dist
        w = workers[0]
        ops = []
        for t in transfers:
            if t.src == self.rank:
                ops.append(dist.P2POp(dist.isend, w.view(t.kind, t.idx, t.zg0, t.zg1, t.stage), t.dst))
            elif t.dst == self.rank:
                ops.append(dist.P2POp(dist.irecv, w.view(t.kind, t.idx, t.zg0, t.zg1, t.stage), t.src))
        if not ops:
            return []
        return dist.batch_isend_irecv(ops)

    def wait(self, handle):
        for r in handle or []:
            r.wait()  # RCCL: the current stream waits; gloo: the host waits

    def exchange_tensors(self, items):
        """see SimComm.exchange_tensors: this rank posts its sends and receives as ONE batched group, in list order"""
        dist = self.dist
        ops = []
        for src, dst, a, b in items:
            if src == self.rank and dst != self.rank:
                ops.append(dist.P2POp(dist.isend, a, dst))
            elif dst == self.rank and src != self.rank:
                ops.append(dist.P2POp(dist.irecv, b, src))
        if not ops:
            return []
        return dist.batch_isend_irecv(ops)

    def allgather_ints(self, per_worker):
        import torch
        dev = "cuda" if (self.dist.get_backend() == "nccl") else "cpu"
        mine = torch.tensor([int(per_worker[0])], dtype=torch.int64, device=dev)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [int(t.item()) for t in out]

    def allreduce_max(self, per_worker):
        import torch
        a = np.asarray(per_worker[0], np.float32)
        t = torch.from_numpy(a.copy())
        dev = "cuda" if (self.dist.get_backend() == "nccl") else "cpu"
        t = t.to(dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [t.cpu().numpy()]

    def allreduce_max_(self, per_worker_tensor):
        self.dist.all_reduce(per_worker_tensor[0], op=self.dist.ReduceOp.MAX)

    def allgather_planes(self, per_worker_tensor, outs, counts):
        import torch
        t, out = per_worker_tensor[0], outs[0]
        if len(set(counts)) == 1 and t.shape[0] == counts[0]:
            self.dist.all_gather_into_tensor(out, t.contiguous())
        else:  # uneven slabs: pad to the thickest
            mx = max(counts)
            pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            pad[: counts[self.rank]] = t[: counts[self.rank]]
            bufs = [torch.empty_like(pad) for _ in range(self.world)]
            self.dist.all_gather(bufs, pad)
            off = 0
            for r in range(self.world):
                out[off:off + counts[r]].copy_(bufs[r][: counts[r]])
                off += counts[r]

    def allreduce_sum_(self, per_worker_tensor):
        t = per_worker_tensor[0]
        if t.numel():
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.tail_group)   # (the tail's thread: see __init__)

    def gather_objects(self, per_worker):
        out = [None] * self.world
        self.dist.all_gather_object(out, per_worker[0])
        return out


# --------------------------------------------------------------------------------------------------------------------
# one rank's device state
# --------------------------------------------------------------------------------------------------------------------
class SlabStage:
    """one sharded octave of one rank: slab context + the torch-owned arena its level buffers live in"""

    def __init__(self, rank, octave, dims, bounds, halo, noct, device, params, desc_partial=False):
        import torch
        from . import capi
        nx, ny, nz = dims
        self.octave, self.dims, self.bounds = octave, dims, bounds
        self.z0, self.z1 = bounds[rank]
        self.plane = nx * ny
        dev = torch.device("cuda", device)
        n = capi.SlabCSIFT3D.arena_floats(nx, ny, nz, self.z0, self.z1, halo, noct, octave=octave, **params)
        self.arena = torch.zeros(n, dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        self.ctx = capi.SlabCSIFT3D(nx, ny, nz, self.z0, self.z1, halo, noct, self.arena.data_ptr(), n, device=device,
                                    octave=octave, **params)
        if desc_partial:
            self.ctx.set_desc_partial(True)
        self._buf = {}

    def view(self, kind, idx, zg0, zg1):
        """arena view of the global planes [zg0, zg1) of a level buffer (contiguous: buffers are plane-major)"""
        key = (kind, idx)
        if key not in self._buf:
            self._buf[key] = self.ctx.buffer(kind, idx)
        off, planes, zoff = self._buf[key]
        assert zoff <= zg0 < zg1 <= zoff + planes, (kind, idx, zg0, zg1, zoff, planes)
        return self.arena[off + (zg0 - zoff) * self.plane: off + (zg1 - zoff) * self.plane]


class SlabWorker:
    """The sharded octaves (SlabStage each) + the seeded, replicated tail context of one rank."""

    def __init__(self, rank, world, dims, device=0, halo=None, sharded_octaves=1, stream=None, desc_partial=True, **params):
        import torch
        from . import capi
        nx, ny, nz = dims
        self.rank, self.world, self.dims, self.device = rank, world, dims, device
        self.params = params
        self.desc_partial = bool(desc_partial)
        self.levels = params.get("num_kp_levels", 3)
        self.halo = int(halo) if halo is not None else (capi.slab_min_halo_partial(**params) if desc_partial else capi.slab_min_halo(**params))
        self.noct = octaves_total(nx, ny, nz)
        # octaves sharded as slabs: at most all but none below the fused kernel's minimum extent (40 voxels in x, y: one 32 x 32 tile + the widest half width)
        S = max(1, min(sharded_octaves, self.noct))
        fits = lambda n: n == 32 or n >= 40   # planes the level kernel tiles: one 32 x 32 tile, or room for a shifted last tile behind the widest mirror zone
        while S > 1 and (not fits(nx >> (S - 1)) or not fits(ny >> (S - 1)) or (nz >> S) < world):
            S -= 1
        # ... and none the slab contexts cannot hold (r06, sift3d_slab_admits: a level thinner than its kernel's column -- 2 hw + 2 planes -- has no
        # separable fallback in a slab; such a plan used to be accepted and failed in its first run)
        while not capi.slab_admits(nx >> (S - 1), ny >> (S - 1), nz >> (S - 1), S == 1, **params):
            if S == 1:
                raise ValueError("this volume / these parameters do not fit the slab kernels (half widths 2 .. 8, planes of 32 or >= 32 + hw voxels per side, at least 2 hw + 2 planes)")
            S -= 1
        self.S = S
        self.stages = []
        b = slab_bounds(nz, world, align=1 << S)
        d = (nx, ny, nz)
        for o in range(S):
            self.stages.append(SlabStage(rank, o, d, b, self.halo, self.noct, device, params, desc_partial=self.desc_partial))
            b = halve_bounds(b, d[2])
            d = (d[0] // 2, d[1] // 2, d[2] // 2)
        self.bounds = self.stages[0].bounds
        self.z0, self.z1 = self.bounds[rank]
        self.ctx = self.stages[0].ctx
        self.arena = self.stages[0].arena
        dev = torch.device("cuda", device)
        # one stream for everything the sharded octaves enqueue (kernels, exchanges, reductions); see the module docstring
        self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        for st in self.stages:
            st.ctx.set_stream(self.stream.cuda_stream)
        self.dogmax = [torch.zeros(8, dtype=torch.float32, device=dev) for _ in self.stages]
        # replicated tail (octaves >= S): seeded with the all-gathered level 0 of octave S
        self.tail = None
        self.seed = self.seed_mine = None
        if self.noct > S:
            n2 = (d[2], d[1], d[0])
            self.tail = capi.SeededCSIFT3D(n2, S, self.noct, device=device, **params)
            self.tail.set_partition(rank, world)
            self.seed = torch.empty(n2, dtype=torch.float32, device=dev)
            self.counts2 = [z1 - z0 for z0, z1 in b]
            self.seed_mine = torch.empty((max(max(self.counts2), 1), d[1], d[0]), dtype=torch.float32, device=dev)

    def view(self, kind, idx, zg0, zg1, stage=0):
        return self.stages[stage].view(kind, idx, zg0, zg1)

    def close(self):
        if self.tail is not None:
            self.tail.close()
        for st in self.stages:
            st.ctx.set_stream(0)  # the torch stream may die before the context
            st.ctx.close()


# --------------------------------------------------------------------------------------------------------------------
# the lock-step driver
# --------------------------------------------------------------------------------------------------------------------
class SlabExtractor:
    """CSIFT3D for a volume sharded over `world` ranks.  `comm.local_ranks()` are the ranks living in this process."""

    def __init__(self, dims, comm, device=0, halo=None, sharded_octaves=2, desc_partial=True, **params):
        self.dims, self.comm = dims, comm
        self.desc_partial = bool(desc_partial)
        self.kp_counts = {}   # stage -> accepted keypoints per rank of the last run (sizes of the record / histogram messages)
        self.world = comm.world
        import torch
        devs = device if isinstance(device, (list, tuple)) else [device] * len(comm.local_ranks())
        # simulated ranks share ONE stream (their "sends" are copies between the workers' buffers, ordered on that stream)
        shared = torch.cuda.Stream(device=torch.device("cuda", devs[0])) if isinstance(comm, SimComm) else None
        self.workers = {r: SlabWorker(r, self.world, dims, device=d, halo=halo, sharded_octaves=sharded_octaves, stream=shared,
                                      desc_partial=desc_partial, **params)
                        for r, d in zip(comm.local_ranks(), devs)}
        w0 = next(iter(self.workers.values()))
        self.bounds, self.halo, self.levels, self.noct, self.S = w0.bounds, w0.halo, w0.levels, w0.noct, w0.S
        self.ng = self.levels + 3
        self.need = [w0.ctx.halo_planes(i) for i in range(self.ng)]   # planes of G[i] its consumers reach
        self.hws = [w0.ctx.level_hw(i) for i in range(self.ng)]       # half width of the Gaussian producing G[i]
        self.times = {}

    def halo_bytes(self):
        """bytes every rank RECEIVES per KpSiftAlgorithm from its z-neighbours (the same plan KpSiftAlgorithm posts: per level the planes
        the next level marches first, the keypoint-window halo of G[1..levels] and one DoG plane; the all-gather of the tail's seed level
        is not a halo).  -> list per rank"""
        w0 = next(iter(self.workers.values()))
        recv = [0] * self.world
        for s in range(self.S):
            st = w0.stages[s]
            bounds, nzs = st.bounds, st.dims[2]
            plane = st.dims[0] * st.dims[1] * 4
            for i in range(self.ng):
                urgent_h = self.hws[i + 1] + 1 if i + 1 < self.ng else 0
                ts = halo_transfers(bounds, nzs, KIND_GSS, i, 0, urgent_h, s) + halo_transfers(bounds, nzs, KIND_GSS, i, urgent_h, self.need[i], s)
                if 1 <= i - 1 <= self.levels:
                    ts += halo_transfers(bounds, nzs, KIND_DOG, i - 1, 0, 1, s)
                for t in ts:
                    recv[t.dst] += (t.zg1 - t.zg0) * plane
        return recv

    # workers as the list the communicator expects ([mine] for DistComm, all ranks for SimComm)
    def _wl(self):
        if isinstance(self.comm, SimComm):
            return [self.workers[r] for r in range(self.world)]
        return list(self.workers.values())

    def load(self, volume=None, device_slabs=None):
        """CSIFT3D constructor work (Src/cSIFT3D.cc:146-163): copy the owned planes, max-abs normalise over the WHOLE
        volume (all-reduce of the maxima), exchange the input halo of the base blur.
        volume: host array [nz, ny, nx] (every rank reads its own planes) or device_slabs: {rank: torch [z1-z0, ny, nx]}."""
        ws = self._wl()
        for w in ws:
            if device_slabs is not None:
                t = device_slabs[w.rank]
                assert tuple(t.shape) == (w.z1 - w.z0, self.dims[1], self.dims[0]) and t.is_contiguous()
                w.ctx.upload(None, w.z0, w.z1, device_ptr=t.data_ptr())
            else:
                w.ctx.upload(volume[w.z0:w.z1], w.z0, w.z1)
        mx = self.comm.allreduce_max([[w.ctx.input_absmax()] for w in ws])
        for w, m in zip(ws, mx):
            w.ctx.input_scale(float(m[0]))
        nz = self.dims[2]
        hw0 = self.hws[0] + 1
        import torch
        with torch.cuda.stream(ws[0].stream):
            self.comm.wait(self.comm.exchange(ws, halo_transfers(self.bounds, nz, KIND_INPUT, 0, 0, hw0)))
        self.comm.sync()

    def KpSiftAlgorithm(self):
        """CSIFT3D::KpSiftAlgorithm (Src/cSIFT3D.cc:165-235) over the slabs."""
        import threading
        import time
        import torch
        comm, ws = self.comm, self._wl()
        cur = ws[0].stream   # DistComm: the local worker's stream; SimComm: the stream all simulated ranks share
        deferred = []
        comm.sync()
        t0 = time.perf_counter()
        with torch.cuda.stream(cur):
            for s in range(self.S):
                sts = [w.stages[s] for w in ws]
                bounds, nzs = sts[0].bounds, sts[0].dims[2]
                for i in range(self.ng):
                    for st in sts:
                        st.ctx.level_async(i)          # level 0 of an octave > 0 was written by the decimation below
                    urgent_h = self.hws[i + 1] + 1 if i + 1 < self.ng else 0   # planes p-hw-1 .. p+hw of the next level's z-march
                    # the planes level i+1 needs first, then the wider keypoint-window halo and the DoG plane behind it; both are
                    # ordered behind the level kernel on the stream, only the first is awaited (by the stream) before level i+1
                    h_urgent = comm.exchange(ws, halo_transfers(bounds, nzs, KIND_GSS, i, 0, urgent_h, s))
                    late = halo_transfers(bounds, nzs, KIND_GSS, i, urgent_h, self.need[i], s)
                    if 1 <= i - 1 <= self.levels:
                        late += halo_transfers(bounds, nzs, KIND_DOG, i - 1, 0, 1, s)
                    deferred.append(comm.exchange(ws, late))
                    if i == self.levels and s + 1 < self.noct:
                        # G[s+1][0] = DownSample_3D(G[s][levels]) (Src/cSIFT3D.cc:293-296, 321-344), owned planes only:
                        # straight into the next sharded octave's level-0 buffer, or into the all-gather piece of the tail
                        for w in ws:
                            if s + 1 < self.S:
                                nst = w.stages[s + 1]
                                if nst.z1 > nst.z0:
                                    w.stages[s].ctx.decimate(nst.view(KIND_GSS, 0, nst.z0, nst.z1).data_ptr(), wait=False)
                            else:
                                w.stages[s].ctx.decimate(w.seed_mine.data_ptr(), wait=False)
                    comm.wait(h_urgent)
                    if not comm.stream_ordered:
                        comm.sync()   # a backend whose transfers are not ordered on the stream (gloo): host-side wait per level
                # DoG maxima -> global (threshold of Detect_KeyPoints, Src/cSIFT3D.cc:379-384): MAX all-reduce of a device tensor
                for w, st in zip(ws, sts):
                    st.ctx.export_dogmax(w.dogmax[s].data_ptr())
                comm.allreduce_max_([w.dogmax[s] for w in ws])
                if not comm.stream_ordered:
                    comm.sync()
                for w, st in zip(ws, sts):
                    st.ctx.import_dogmax(w.dogmax[s].data_ptr())
            for h in deferred:
                comm.wait(h)
            if self.noct > self.S:
                comm.allgather_planes([w.seed_mine for w in ws], [w.seed for w in ws], ws[0].counts2)
            ev_seed = torch.cuda.Event()
            ev_seed.record(cur)
        self.times["pyramid_enqueue"] = time.perf_counter() - t0

        # ---- replicated tail on its own thread / stream, beside the sharded detection and descriptors -------------------------
        err = []

        def tail_body():
            try:
                torch.cuda.set_device(ws[0].device)   # the current device is per thread
                ev_seed.synchronize()   # the all-gathered level 0 of octave S exists
                for w in ws:
                    w.tail.seed(w.seed.data_ptr())
                # replicated pyramid + extrema of the remaining octaves; orientation dealt by extremum index, results restored
                # on every rank by an integer all-reduce(SUM) of zero-padded rows (exact); descriptors dealt by keypoint and
                # LEFT distributed, like the sharded keypoints (GetKeypoints / the matcher's all-gather collect them)
                for w in ws:
                    w.tail.run_partial_orientation()
                bufs = []
                for w in ws:
                    b = torch.empty(w.tail.num_extrema() * capi_words(), dtype=torch.int32, device=w.arena.device)
                    if b.numel():
                        w.tail.export_orientation(b.data_ptr())
                    bufs.append(b)
                comm.allreduce_sum_(bufs)
                torch.cuda.current_stream().synchronize()
                for w, b in zip(ws, bufs):
                    if b.numel():
                        w.tail.import_orientation(b.data_ptr())
                    w.tail.run_describe()
            except BaseException as e:  # noqa: BLE001 -- re-raised on the calling thread
                err.append(e)

        th = None
        t2 = time.perf_counter()
        if self.noct > self.S:
            th = threading.Thread(target=tail_body, daemon=True)
            th.start()
        # whatever happens on this thread (a list-capacity error on one rank, say), the tail thread is joined before the method is
        # left: it issues a collective and calls into contexts that close() would otherwise destroy under it
        try:
            cur.synchronize()
            self.times["pyramid"] = time.perf_counter() - t0
            t1 = time.perf_counter()
            for s in range(self.S):
                for w in ws:
                    w.stages[s].ctx.detect()
                if self.desc_partial:
                    with torch.cuda.stream(cur):
                        self._describe_partial(s, ws)
                else:
                    for w in ws:
                        w.stages[s].ctx.describe()
            self.times["keypoints_sharded"] = time.perf_counter() - t1
        finally:
            if th is not None:
                th.join()
        if err:
            raise err[0]
        self.times["tail"] = time.perf_counter() - t2
        self.times["total"] = time.perf_counter() - t0
        return self

    # ---- r05: descriptor windows split along z over the ranks ------------------------------------------------------------------------
    def _describe_partial(self, s, ws):
        """Orientation of the owned extrema, then the descriptors of sharded octave s from partial integer histograms (module docstring)."""
        import torch
        from . import capi
        comm = self.comm
        sts = {w.rank: w.stages[s] for w in ws}
        for w in ws:
            sts[w.rank].ctx.orient_launch()
        counts = comm.allgather_ints([sts[w.rank].ctx.orient_count() for w in ws])
        self.kp_counts[s] = list(counts)
        RW = capi.slab_record_words()
        neigh = window_neighbours(ws[0].stages[s].bounds, ws[0].stages[s].ctx.desc_reach())
        recs = {}
        for w in ws:
            recs[w.rank] = torch.empty((counts[w.rank], RW), dtype=torch.int32, device=w.arena.device)
            if counts[w.rank]:
                sts[w.rank].ctx.export_records(recs[w.rank].data_ptr())
        n_redo, redo, units_next = self._partial_round(ws, sts, neigh, counts, recs, None, False)
        tot = comm.allgather_ints([n_redo[w.rank] for w in ws])
        if any(tot):   # rare: records whose first fixed-point unit failed are repeated, by every part, with the exact unit
            recs2, units2 = {}, {}
            for w in ws:
                idx = torch.nonzero(redo[w.rank], as_tuple=False).flatten() if tot[w.rank] else torch.zeros(0, dtype=torch.int64, device=w.arena.device)
                recs2[w.rank] = recs[w.rank].index_select(0, idx).contiguous()
                units2[w.rank] = units_next[w.rank].index_select(0, idx).contiguous()
            self._partial_round(ws, sts, neigh, tot, recs2, units2, True)

    def _partial_round(self, ws, sts, neigh, counts, recs, units, final):
        import torch
        from . import capi
        comm = self.comm
        RW = capi.slab_record_words()
        local = {w.rank: w for w in ws}
        dev = {w.rank: w.arena.device for w in ws}
        # 1. records (and the second round's units) to the ranks their windows reach into
        inbox = {q: {} for q in local}
        items = []
        for r in range(self.world):
            for q in neigh[r]:
                if not counts[r]:
                    continue
                a = recs[r] if r in local else None
                b = torch.empty((counts[r], RW), dtype=torch.int32, device=dev[q]) if q in local else None
                items.append((r, q, a, b))
                ub = None
                if units is not None:
                    ub = torch.empty(counts[r], dtype=torch.float32, device=dev[q]) if q in local else None
                    items.append((r, q, units[r] if r in local else None, ub))
                if q in local:
                    inbox[q][r] = (b, ub)
        comm.wait(comm.exchange_tensors(items))
        # 2. every rank marches its part of the windows -- its own records and the foreign ones -- in one launch
        parts = {q: {} for q in local}
        bounds = next(iter(sts.values())).bounds
        for q in local:
            lists = []
            for r in [q] + sorted(neigh[q]):
                if not counts[r]:
                    continue
                rec, un = (recs[q], units[q] if units is not None else None) if r == q else inbox[q][r]
                h = torch.empty((counts[r], 768), dtype=torch.int32, device=dev[q])
                m = torch.empty(counts[r], dtype=torch.float32, device=dev[q])
                lists.append((rec.data_ptr(), counts[r], un.data_ptr() if un is not None else None, h.data_ptr(), m.data_ptr(), bounds[r][0], bounds[r][1]))
                parts[q][r] = (h, m)
            sts[q].ctx.describe_partial(lists)
        # 3. the parts back to their owners
        got = {r: {} for r in local}
        items = []
        for q in range(self.world):
            for r in neigh[q]:
                if not counts[r]:
                    continue
                ha, ma = parts[q][r] if q in local else (None, None)
                hb = torch.empty((counts[r], 768), dtype=torch.int32, device=dev[r]) if r in local else None
                mb = torch.empty(counts[r], dtype=torch.float32, device=dev[r]) if r in local else None
                items.append((q, r, ha, hb))
                items.append((q, r, ma, mb))
                if r in local:
                    got[r][q] = (hb, mb)
        comm.wait(comm.exchange_tensors(items))
        if not comm.stream_ordered:
            comm.sync()
        # 4. the owner's finish: the parts' integers are added, the masses in rank order (sift3d_slab_describe_finish)
        n_redo, redo, units_next = {}, {}, {}
        for r in local:
            n = counts[r]
            redo[r] = torch.zeros(n, dtype=torch.int32, device=dev[r])
            units_next[r] = torch.zeros(n, dtype=torch.float32, device=dev[r])
            ps = [parts[r][r] if q == r else got[r][q] for q in sorted(neigh[r] + [r])] if n else []
            if len(ps) > 6:   # slabs much thinner than a window's reach: more parts than one finish launch takes -- added here, in the same order
                th, tm = ps[0][0].clone(), ps[0][1].clone()
                for h, m in ps[1:]:
                    th += h
                    tm += m
                ps = [(th, tm)]
            n_redo[r] = sts[r].ctx.describe_finish(recs[r].data_ptr() if n else 0, n, [(h.data_ptr(), m.data_ptr()) for h, m in ps],
                                                   units[r].data_ptr() if (units is not None and n) else None, final,
                                                   redo[r].data_ptr() if n else None, units_next[r].data_ptr() if n else None)
        return n_redo, redo, units_next

    def window_bytes(self):
        """bytes every rank RECEIVED for the partial descriptor windows of the last run (records in, histograms + masses back) -> list per
        rank; zeros before the first run or without desc_partial"""
        recv = [0] * self.world
        if not self.desc_partial:
            return recv
        from . import capi
        rb = capi.slab_record_words() * 4
        w0 = next(iter(self.workers.values()))
        for s, counts in self.kp_counts.items():
            neigh = window_neighbours(w0.stages[s].bounds, w0.stages[s].ctx.desc_reach())
            for r in range(self.world):
                for q in neigh[r]:
                    recv[q] += counts[r] * rb               # r's records arrive at q
                    recv[r] += counts[r] * (768 * 4 + 4)    # q's part of r's windows comes back
        return recv

    def num_local_keypoints(self):
        """keypoints this process holds records for: the sharded octaves of its slabs (+ the replicated tail records once)"""
        ws = self._wl()
        n = sum(int(st.ctx.device_results()[2]) for w in ws for st in w.stages)
        if self.noct > self.S:
            n += int(ws[0].tail.device_results()[2])
        return n

    def GetKeypoints(self):
        """Global result in reference order on every rank (gathers through the host; not part of the timed path)."""
        ws = self._wl()
        kps, dss = [], []
        for s in range(self.S):
            parts = self.comm.gather_objects([w.stages[s].ctx.GetKeypoints() for w in ws])
            empty = (parts[0][0][:0], parts[0][1][:0])
            k, d = merge_keypoints(parts, empty)
            kps.append(k); dss.append(d)
        if self.noct > self.S:
            # tail keypoint records are complete everywhere; the descriptor rows are dealt (see described_rows)
            mine = []
            for w in ws:
                wkp, wds = w.tail.GetKeypoints()
                mine.append(wds[described_rows(wkp["level"], w.rank, self.world)])
            rows = self.comm.gather_objects(mine)
            tkp, tds = ws[0].tail.GetKeypoints()
            for r, part in enumerate(rows):
                tds[described_rows(tkp["level"], r, self.world)] = part
            kps.append(tkp); dss.append(tds)
        return np.concatenate(kps), np.concatenate(dss)

    def close(self):
        for w in self.workers.values():
            w.close()
