"""Host-side drivers above the C ABI: the benchmark configurations of BASELINE.json, the panorama chain of
ImageProcess::matching (ImageProcess.cpp:159-268, hot-path calls only), and the sharding of independent pairs
across the GPUs of a node (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI).

Nothing here computes pixels: every per-pixel operation is a HIP kernel reached through capi.
"""
import math

from . import capi

SEED_MAP = (1.0, 0.002, 1e-6, -2048.0, -0.001, 1.0, 5e-7, 1.5)  # config-2 backward map, SURVEY.md 8(d)


def config_map(pair_index=0, frame=4096):
    """Backward map of synthetic pair `pair_index`: p[3] = -(frame/2) - 8*i so that pairs differ (config 4)."""
    p = list(SEED_MAP)
    p[3] = -(frame / 2.0) - 8.0 * pair_index
    return p


def config_canvas(frame=4096):
    """Canvas of the synthetic pair configs: fixed by definition at 1.5*frame x frame (6144x4096 for config 2)."""
    return frame * 3 // 2, frame


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank` (config 4: 32 pairs -> 4 per GPU on 8 ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def algorithmic_bytes(frame_px_a, frame_px_b, level_w, level_h, bytes_per_sample=4, fused_sweep_levels=0, fused_decimate=True,
                      source_fused=False, implicit_mask=False, coarse_from=0):
    """SURVEY.md 8(d) byte accounting for one pair: every distinct input array read once + every output array
    written once, pyramid planes f32.  Returns (per_kernel, stages): `stages` are the canonical totals S1..S3 the
    headline fraction uses (independent of how the kernels are fused); `per_kernel` gives each of THIS implementation's
    kernels the bytes it actually owns -- its own inputs read once and its own outputs written once, given the fusion
    the plan runs with -- so that no kernel is credited traffic another kernel (or nobody) moves:
      fused_decimate   k_vv_y_bwd_dec reads level l and writes only level l+1 (the blurred level is never stored)
      fused_sweep_levels  levels whose anticausal-x and causal-y sweeps are one kernel (k_vv_xbyf: one R + one W)
      source_fused     level 0 is never materialised: k_src_index writes one 4-byte index plane; the causal x sweep and the
                       level-0 collapse read the frames (through that plane) instead of six level-0 planes
      implicit_mask    the level-0 mask plane is never read (a step function of x); the fused sweep still writes its blur
      coarse_from      > 0: levels coarse_from .. L-1 run in one launch (k_coarse): it reads G of level coarse_from, writes E of
                       that level and G, E of the levels above it; the per-level kernels are credited the finer levels only"""
    n = [w * h for w, h in zip(level_w, level_h)]
    L = len(n)
    P = n[0]
    inputs = (frame_px_a + frame_px_b) * 3 * bytes_per_sample
    s1 = inputs + 2 * P * 3 * 4
    s2 = sum(7 * 4 * (3 * n[l] + n[l + 1]) for l in range(L - 1))
    s3 = sum(4 * (10 * n[l] + 9 * n[l + 1]) for l in range(L - 1)) + 4 * 10 * n[L - 1]
    F = min(fused_sweep_levels, L - 1)
    m0 = 6 if implicit_mask else 7  # level-0 planes that exist as inputs of the blur
    x_fwd = x_fwd_src = x_bwd = y_fwd = y_bwd = xbyf = 0
    Lr = coarse_from if coarse_from > 0 else L - 1  # levels [0, Lr) are reduced / collapsed by per-level launches
    for l in range(Lr):
        rd = 4 * n[l] * (m0 if l == 0 else 7)
        wr = 4 * n[l] * 7
        if l == 0 and source_fused:
            x_fwd_src += inputs + 4 * P + 4 * n[0] * m0  # frames + index plane in, six (or seven) x-swept planes out
        else:
            x_fwd += rd + (4 * n[l] * m0 if l == 0 else wr)
        if l < F:
            xbyf += rd + wr
        else:
            x_bwd += rd + (4 * n[l] * m0 if l == 0 else wr)
            y_fwd += wr + wr
        y_bwd += wr + (7 * 4 * n[l + 1] if fused_decimate else wr)
    decimate = 0 if fused_decimate else sum(7 * 4 * (n[l] + n[l + 1]) for l in range(Lr))
    if L > 1:
        g0_in = (inputs + 4 * P) if source_fused else 4 * 6 * P
        collapse_l0 = g0_in + (0 if implicit_mask else 4 * P) + 4 * 9 * n[1] + 3 * P * bytes_per_sample
    else:
        collapse_l0 = 0
    per_kernel = {
        "compose": 4 * P if source_fused else s1,  # source-fused: the index plane
        "seam": 2 * bytes_per_sample * level_w[0],  # two mid rows
        "mask": 0 if implicit_mask else 4 * P,
        "vv_x_fwd": x_fwd, "vv_x_fwd_src": x_fwd_src, "vv_x_bwd": x_bwd, "vv_y_fwd": y_fwd, "vv_y_bwd": y_bwd, "vv_xbyf": xbyf,
        "decimate": decimate,
        "collapse_top": 0 if coarse_from > 0 else 4 * 10 * n[L - 1],
        "collapse": sum(4 * (10 * n[l] + 9 * n[l + 1]) for l in range(1, Lr)),
        "collapse_l0": collapse_l0,
        "coarse": 4 * 10 * sum(n[l] for l in range(coarse_from, L)) if coarse_from > 0 else 0,
    }
    return per_kernel, {"S1": s1, "S2": s2, "S3": s3, "total": s1 + s2 + s3}


def batches_of(n_pairs, rank, world, batch):
    """Config 4 as a schedule: this rank's contiguous shard of an n_pairs batch, cut into launch sequences of at most
    `batch` pairs -> list of (first, last+1) global pair indices."""
    lo, hi = shard_range(n_pairs, rank, world)
    return [(b, min(b + batch, hi)) for b in range(lo, hi, batch)]


def steps_per_sequence(n_local, batch, n_seqs_per_step, steps):
    """How many consecutive steps' shares a rank puts into ONE launch sequence: a shard smaller than a launch sequence (8 GPUs:
    4 pairs per rank and step, sequences of up to 16) is launched together with the shares of the following steps -- steps run
    back to back without host synchronisation anyway, every step keeps its own outputs and its own gather."""
    if n_seqs_per_step != 1 or n_local <= 0:
        return 1
    return max(1, min(batch // n_local, max(steps, 1)))


def sequence_sizes(count, per_seq, lanes):
    """`count` steps cut, in step order, into launch sequences of at most `per_seq` steps each; when there are more sequences than
    lanes their number is rounded up to a multiple of the lanes and the sizes are balanced, so that the lanes finish together
    instead of one sequence running on alone at the end (20 steps, 4 per sequence, 4 lanes: 3+3+3+3+2+2+2+2, not 4+4+4+4+4)."""
    if count <= 0:
        return []
    m = -(-count // per_seq)
    if per_seq > 1 and m > lanes and m % lanes:
        m = min(-(-m // lanes) * lanes, count)
    return [count // m + (1 if i < count % m else 0) for i in range(m)]


def stitch_chain(frames, steps, opts=None, finish=True, num=19.0, den=20.0, plans=None):
    """The hot-path calls of ImageProcess::matching for a recorded stitch order, device resident.

    frames: list of (3,H,W) uint8 device tensors (unprojected).  steps: list of dicts with keys
    src (index of the frame to warp), p (8 doubles), offx, offy, ox, oy, cw, ch -- what the reference passes at
    ImageProcess.cpp:218-230; the first mosaic is projection(frames[start]).
    plans: optional dict kept by the caller across calls; the workspace of every canvas size met is created once and
    reused (a camera rig stitches every frame with the same geometry: without it half of a small panorama's time is
    the creation of the three workspaces).  Close them with close_plans(plans).
    Returns the final uint8 mosaic tensor (after equalisation + luminance mix when finish=True, :237-268)."""
    proj = {}

    def projected(i):
        if i not in proj:
            proj[i] = capi.dev_project(frames[i])
        return proj[i]

    result = projected(steps[0]["start"])
    used = []
    for st in steps:
        key = (st["cw"], st["ch"])
        plan = plans.get(key) if plans is not None else None
        if plan is None:
            plan = capi.Plan(st["cw"], st["ch"], opts)
            if plans is not None:
                plans[key] = plan
        elif plan in used:  # the same workspace twice in one chain: its seam record must be read before it is overwritten
            plan.status()
        result = plan.pair(projected(st["src"]), st["p"], st["offx"], st["offy"], result, st["ox"], st["oy"])
        used.append(plan)
    if finish:
        capi.dev_finish(result, num, den)
    # the steps only depend on each other on the device; their seam scans are checked once, at the end (raises StitchError)
    try:
        for plan in used:
            plan.status()
    finally:
        if plans is None:
            for plan in used:
                plan.close()
    return result


def close_plans(plans):
    for plan in plans.values():
        plan.close()
    plans.clear()


def levels_of(cw, ch, level_rule=0):
    length = min(cw, ch) if level_rule else max(cw, ch)
    return int(math.floor(math.log2(length)))


class MosaicGather:
    """Assembles a sharded batch on every rank: each rank contributes one block per step (`shape`: one finished uint8
    mosaic, or its whole shard of the step's batch, (n, 3, H, W)) and the step's blocks of all ranks are all-gathered (torch.distributed: backend "nccl" = RCCL over xGMI on the GPUs;
    "gloo" on CPU tensors in the tests).  The collective is asynchronous, so on the GPU it overlaps the next pair's
    kernels; a ring of `slots` input/output buffers bounds memory.  With keep=True every step's gathered block is
    retained (tests / small batches): result()[k, r] is the mosaic rank r produced at step k.

    Pairs are independent, so this end-of-pair exchange is the only communication of the batch configs."""

    def __init__(self, shape, device, world, rank, slots=2, keep=False, steps=0, group=None, force_collective=False, out_slots=None,
                 staged=False):
        import torch
        self.torch = torch
        self.world, self.rank, self.slots, self.keep, self.group = world, rank, slots, keep, group
        self.force_collective = force_collective  # issue the collective even with one rank (rehearsal of the N>1 path)
        # staged (bench.py's gloo rehearsal: several ranks drive ONE GPU, the collective runs on host tensors): submit() waits for
        # the block, moves it through host memory and returns with the gathered batch back on the device -- same buffers,
        # same step order, same slots as the RCCL form, only synchronous.
        self.staged = staged
        self.inp = [torch.empty(shape, dtype=torch.uint8, device=device) for _ in range(slots)]
        # input blocks are small (a rank's share of a step), gathered blocks are `world` times that: the input ring bounds how many
        # launch sequences may be in flight ahead of the exchange, the output ring only has to outlive one collective
        self.out_slots = (steps if keep else (out_slots or slots))
        self.out = [torch.empty((world,) + tuple(shape), dtype=torch.uint8, device=device) for _ in range(self.out_slots)]
        self.works = [None] * slots

    def input_slot(self, k):
        """Buffer to fill with step k's local mosaic(s); first waits for the gather that last used the slot.  May be
        called from several streams for the same step (each launch sequence of a step fills its own part of the block):
        every caller's stream waits, the handle is only replaced by submit()."""
        s = k % self.slots
        if self.works[s] is not None:
            self.works[s].wait()
        return self.inp[s]

    def gathered(self, k):
        """The gathered batch of step k, [rank][...] (valid until out_slots later steps have been submitted)."""
        return self.out[k if self.keep else k % self.out_slots]

    def submit(self, k):
        import torch.distributed as dist
        s = k % self.slots
        out = self.gathered(k)
        if self.world == 1 and not self.force_collective:
            out[0].copy_(self.inp[s])
            return
        if self.staged:
            torch = self.torch
            torch.cuda.current_stream(self.inp[s].device).synchronize()
            host_in = self.inp[s].cpu()
            host_out = torch.empty((self.world,) + tuple(host_in.shape), dtype=torch.uint8)
            dist.all_gather_into_tensor(host_out.flatten(0, 1), host_in, group=self.group)
            out.copy_(host_out)
            return
        # output viewed as the concatenation along dim 0 (the layout every backend accepts)
        self.works[s] = dist.all_gather_into_tensor(out.flatten(0, 1), self.inp[s], group=self.group, async_op=True)

    def drain(self):
        for i, wk in enumerate(self.works):
            if wk is not None:
                wk.wait()
                self.works[i] = None

    def result(self):
        assert self.keep
        return self.torch.stack(self.out)


def batch_order(n_pairs, world):
    """(step, rank) -> global pair index for a batch sharded with shard_range (contiguous per rank), or None
    where a rank has run out of pairs (ragged batches)."""
    table = {}
    for r in range(world):
        lo, hi = shard_range(n_pairs, r, world)
        for k in range(math.ceil(n_pairs / world)):
            table[(k, r)] = lo + k if lo + k < hi else None
    return table


class RankTransport:
    """What a band-split pair moves between ranks, over torch.distributed.  backend "nccl" (= RCCL over xGMI): device tensors
    go straight into send/recv/all_gather, ordered on the streams like any other kernel.  staged=True (gloo: the one-GPU
    test, where every rank drives the same device): tensors are staged through host memory."""

    def __init__(self, group=None, staged=False):
        import torch.distributed as dist
        self.dist, self.group, self.staged = dist, group, staged
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def send(self, t, dst):
        self.dist.send(t.cpu() if self.staged else t, dst, group=self.group)

    def recv(self, t, src):
        if self.staged:
            c = t.new_empty(t.shape, device="cpu")
            self.dist.recv(c, src, group=self.group)
            t.copy_(c)
        else:
            self.dist.recv(t, src, group=self.group)
        return t

    def all_gather(self, t):
        import torch
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device="cpu" if self.staged else t.device)
        self.dist.all_gather_into_tensor(out.flatten(0, 1), t.cpu() if self.staged else t, group=self.group)
        return out.to(t.device) if self.staged else out

    def swap(self, to_prev, to_next, from_prev, from_next):
        """Neighbour exchange: to_prev goes to rank-1 (None on rank 0), to_next to rank+1 (None on the last rank); the
        neighbours' counterparts arrive in from_prev / from_next."""
        ops, keep = [], []
        P = self.dist.P2POp
        for t, peer, fn in ((to_prev, self.rank - 1, self.dist.isend), (to_next, self.rank + 1, self.dist.isend)):
            if t is not None:
                c = t.cpu() if self.staged else t
                keep.append(c)
                ops.append(P(fn, c, peer, group=self.group))
        stage = []
        for t, peer in ((from_prev, self.rank - 1), (from_next, self.rank + 1)):
            if t is not None:
                c = t.new_empty(t.shape, device="cpu") if self.staged else t
                stage.append((t, c))
                ops.append(P(self.dist.irecv, c, peer, group=self.group))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        if self.staged:
            for t, c in stage:
                t.copy_(c)


class LocalTransport:
    """RankTransport's interface for ranks that are THREADS of one process driving ONE device, each on its own HIP stream (the
    one-GPU form of the band split: tests, scripts/bench_band.py).  What crosses ranks never leaves the device and nobody waits
    on the host for the GPU: a hand-off is a device copy made on the sender's stream plus an event; the receiving thread takes
    the handle from a queue (a host-side hand-over of POINTERS, in enqueue order) and makes ITS stream wait for the event, so the
    bands' kernels are ordered on the device exactly as RCCL's stream-ordered send / recv orders them across GPUs."""

    def __init__(self, rank, world, queues):
        self.rank, self.world, self.qs = rank, world, queues

    @staticmethod
    def make_queues(world):
        import queue
        return {(a, b): queue.Queue() for a in range(world) for b in range(world) if a != b}

    def _post(self, t, dst):
        import torch
        c = t.clone()  # on this rank's stream: the source buffer may be rewritten by this rank's next launch
        ev = torch.cuda.Event()
        ev.record()
        self.qs[(self.rank, dst)].put((c, ev))

    def _take(self, src):
        import torch
        item = self.qs[(src, self.rank)].get()
        if item is None:
            raise RuntimeError(f"rank {src} failed")
        c, ev = item
        torch.cuda.current_stream().wait_event(ev)
        c.record_stream(torch.cuda.current_stream())  # allocated on the sender's stream, consumed on this one
        return c

    def send(self, t, dst):
        self._post(t, dst)

    def recv(self, t, src):
        t.copy_(self._take(src))
        return t

    def all_gather(self, t):
        import torch
        for d in range(self.world):
            if d != self.rank:
                self._post(t, d)
        return torch.stack([t if s_ == self.rank else self._take(s_) for s_ in range(self.world)])

    def swap(self, to_prev, to_next, from_prev, from_next):
        if to_prev is not None:
            self._post(to_prev, self.rank - 1)
        if to_next is not None:
            self._post(to_next, self.rank + 1)
        if from_prev is not None:
            from_prev.copy_(self._take(self.rank - 1))
        if from_next is not None:
            from_next.copy_(self._take(self.rank + 1))


PLANE_PIPELINE_MIN = None  # samples per plane of a band (rows x pitch) from which a level hands its state over plane by plane; None: never


class BandStitcher:
    """One pair split into row bands over the ranks of a node (BASELINE.json configs[4]; SURVEY.md 8(e)(ii): recurrence
    state hand-off, not a transpose).  Every rank holds the input frames and calls run() with the same arguments; it gets
    back ITS band of the mosaic, rows [rank*ch/world, (rank+1)*ch/world).  The per-band computation is capi.Band (HIP
    kernels); this class only sequences it and moves what crosses ranks through `transport`:
      * per split level, the causal y sweep waits for 3 doubles per column and plane from the rank above and passes its own
        on, then the anticausal sweep (+ decimation) the other way round;
      * the bands of the first replicated level are all-gathered and the coarse levels run on every rank;
      * before a split level is collapsed, two halo rows of G and E of the level above come from either neighbour."""

    def __init__(self, cw, ch, split_levels, transport, device, opts=None, fuse_sweeps=None, plane_pipeline_min=None, col_chunks=1):
        import torch
        self.t, self.dev = transport, device
        # the anticausal x sweep fused with the causal y sweep on the two finest levels (stitch_band_reduce_xy_fwd: one pass over a
        # level less).  The fused sweep finds its parallelism in the 64-row bands of a plane, and a rank's share of the rows has few
        # of them: at 24576 x 16384 one band gains, two lose and eight take twice as long (profiles/r03_config5_band.json).
        # Default: only when there is no split.
        self.fuse_sweeps = (transport.world == 1) if fuse_sweeps is None else bool(fuse_sweeps)
        self.rank, self.world = transport.rank, transport.world
        self.band = capi.Band(cw, ch, self.rank, self.world, split_levels, opts)
        self.Ls, self.geom = split_levels, self.band.geom
        f64 = dict(dtype=torch.float64, device=device)
        # recurrence states of all seven planes, [k][7][pitch] doubles per split level
        self.st_f = [torch.zeros(4 * 7 * g["pitch"], **f64) for g in self.geom[:-1]]  # causal state out (+ the band's last row)
        self.st_b = [torch.zeros(3 * 7 * g["pitch"], **f64) for g in self.geom[:-1]]  # anticausal state out
        self.res = [torch.zeros(3 * 7 * g["pitch"], **f64) for g in self.geom[:-1]]   # what a neighbour left
        # Optional (plane_pipeline_min = samples per plane of a band from which a level does it): the state crosses ranks PLANE BY
        # PLANE, so that rank r sweeps plane p while rank r+1 sweeps plane p-1 and the chain from rank to rank is 7 + N - 1
        # plane-steps long instead of 7 N.  Off by default: a one-plane launch has 192-384 wavefronts and streams at a quarter of
        # the seven-plane launch's rate, so on ONE GPU the split gets slower (24576 x 16384: 2 bands 31.4 -> 41.8 ms, 8 bands
        # 38.7 -> 41.6 from one host thread), and what it would gain across 8 GPUs (a 14-step chain of slower steps) is not
        # measurable on this pool.
        pmin = PLANE_PIPELINE_MIN if plane_pipeline_min is None else plane_pipeline_min
        self.per_plane = [pmin is not None and self.world > 1 and g["rows"] * g["pitch"] >= pmin for g in self.geom[:-1]]
        if self.per_plane[0]:
            self.band.set_level0(False)  # one plane at a time: level 0 must exist as planes
        # col_chunks = C > 1: the state crosses ranks in C column chunks (stitch_band_reduce_y_*_cols): columns are independent in a y
        # sweep, so rank r+1 starts on chunk c while rank r sweeps chunk c+1 and the chain of a level is N + C - 1 chunk-sweeps long
        # instead of N sweeps.  Levels with an odd width, the plane-by-plane form and the fused sweep keep one chunk.
        self.chunks = []
        for l, g in enumerate(self.geom[:-1]):
            C = 1 if (col_chunks <= 1 or g["w"] % 2 or self.per_plane[l] or (self.fuse_sweeps and l < 2)) else int(col_chunks)
            step = -(-g["pitch"] // (128 * C)) * 128  # columns per chunk: a multiple of 128
            self.chunks.append([(x0, min(x0 + step, g["pitch"])) for x0 in range(0, g["pitch"], step)] if C > 1 else None)
        pad = lambda n: -(-n // 128) * 128  # the chunk's state arrays have the launch's width: 128 columns per workgroup
        self.cst_f = [[torch.zeros(4 * 7 * pad(x1 - x0), **f64) for x0, x1 in ch_] if ch_ else None for ch_ in self.chunks]
        self.cst_b = [[torch.zeros(3 * 7 * pad(x1 - x0), **f64) for x0, x1 in ch_] if ch_ else None for ch_ in self.chunks]
        self.cres = [[torch.zeros(3 * 7 * pad(x1 - x0), **f64) for x0, x1 in ch_] if ch_ else None for ch_ in self.chunks]
        self.st_fp = [torch.zeros((7, 4 * g["pitch"]), **f64) if pp else None for pp, g in zip(self.per_plane, self.geom[:-1])]
        self.st_bp = [torch.zeros((7, 3 * g["pitch"]), **f64) if pp else None for pp, g in zip(self.per_plane, self.geom[:-1])]
        self.res_p = [torch.zeros((7, 3 * g["pitch"]), **f64) if pp else None for pp, g in zip(self.per_plane, self.geom[:-1])]

    def close(self):
        self.band.close()

    def run(self, frame, p, offx, offy, mosaic, ox, oy, out=None):
        """This rank's band of the mosaic.  Blocks only where the transport does (RCCL: nowhere on the host)."""
        gen = self.steps(frame, p, offx, offy, mosaic, ox, oy, out)
        T, val = self.t, None
        try:
            while True:
                req = gen.send(val)
                kind = req[0]
                val = (T.send(*req[1:]) if kind == "send" else T.recv(*req[1:]) if kind == "recv" else
                       T.all_gather(*req[1:]) if kind == "all_gather" else T.swap(*req[1:]))
        except StopIteration as e:
            out = e.value
        self.seam = self.band.status()
        return out

    def steps(self, frame, p, offx, offy, mosaic, ox, oy, out=None):
        """The band's launch sequence as a generator: it enqueues kernels and YIELDS what must cross ranks -- ("send", t, dst),
        ("recv", t, src) -> t, ("all_gather", t) -> stacked, ("swap", to_prev, to_next, from_prev, from_next) -- so that one host
        thread can interleave the sequences of several ranks (LocalBandGroup) and a rank of its own simply executes them (run)."""
        import torch
        B, r, N, Ls = self.band, self.rank, self.world, self.Ls
        g0 = self.geom[0]
        if out is None:
            out = torch.empty((3, g0["rows"], g0["w"]), dtype=frame.dtype, device=self.dev)
        B.compose(frame, p, offx, offy, mosaic, ox, oy)
        for l in range(Ls):
            n3 = 3 * 7 * self.geom[l]["pitch"]
            if self.per_plane[l]:
                n1 = 3 * self.geom[l]["pitch"]
                B.reduce_x(l)
                for pl in range(7):
                    res = (yield ("recv", self.res_p[l][pl], r - 1)) if r > 0 else None
                    B.reduce_y_fwd(l, pl, res, self.st_fp[l][pl])
                    if r < N - 1:
                        yield ("send", self.st_fp[l][pl][:n1], r + 1)
                for pl in range(7):
                    res = (yield ("recv", self.res_p[l][pl], r + 1)) if r < N - 1 else None
                    B.reduce_y_bwd(l, pl, self.st_fp[l][pl], res, self.st_bp[l][pl])
                    if r > 0:
                        yield ("send", self.st_bp[l][pl], r - 1)
                continue
            fuse = self.fuse_sweeps and l < 2  # as in a plan: the fused sweep on the two finest levels only (it loses on small levels)
            if self.chunks[l]:
                B.reduce_x(l)
                for c, (x0, x1) in enumerate(self.chunks[l]):  # causal sweep chunk by chunk, rank 0 first
                    res = (yield ("recv", self.cres[l][c], r - 1)) if r > 0 else None
                    B.reduce_y_fwd_cols(l, x0, x1, res, self.cst_f[l][c])
                    if r < N - 1:
                        yield ("send", self.cst_f[l][c][:self.cres[l][c].numel()], r + 1)
                for c, (x0, x1) in enumerate(self.chunks[l]):  # anticausal sweep + decimation chunk by chunk, last rank first
                    res = (yield ("recv", self.cres[l][c], r + 1)) if r < N - 1 else None
                    B.reduce_y_bwd_cols(l, x0, x1, self.cst_f[l][c], res, self.cst_b[l][c])
                    if r > 0:
                        yield ("send", self.cst_b[l][c], r - 1)
                continue
            if not fuse:
                B.reduce_x(l)
            # causal sweep, rank 0 first; then the anticausal sweep + decimation, last rank first.  A band's sweep is one chain
            # of dependent rows -- as long for seven planes as for one -- so all planes go in one launch and one message
            res = (yield ("recv", self.res[l], r - 1)) if r > 0 else None
            if fuse:
                B.reduce_xy_fwd(l, res, self.st_f[l])
            else:
                B.reduce_y_fwd(l, -1, res, self.st_f[l])
            if r < N - 1:
                yield ("send", self.st_f[l][:n3], r + 1)
            res = (yield ("recv", self.res[l], r + 1)) if r < N - 1 else None
            B.reduce_y_bwd(l, -1, self.st_f[l], res, self.st_b[l])
            if r > 0:
                yield ("send", self.st_b[l], r - 1)
        # the first replicated level: gather the bands, run the coarse levels on every rank
        gt = self.geom[Ls]
        mine = torch.empty((7, gt["rows"], gt["w"]), dtype=torch.float32, device=self.dev)
        B.rows(Ls, 2, 0, gt["rows"], mine, True)
        gathered = yield ("all_gather", mine)
        full = gathered.permute(1, 0, 2, 3).reshape(7, N * gt["rows"], gt["w"]).contiguous()
        B.top(full)
        for l in range(Ls - 1, -1, -1):
            if l + 1 < Ls and N > 1:  # halo rows of the level above: G (a, b) and E, from both neighbours (none without neighbours)
                gs = self.geom[l + 1]
                H, rows, w = gs["halo"], gs["rows"], gs["w"]
                for kind, planes in ((0, 6), (1, 3)):
                    mk = lambda: torch.empty((planes, H, w), dtype=torch.float32, device=self.dev)
                    first, last, above, below = mk(), mk(), mk(), mk()
                    B.rows(l + 1, kind, 0, H, first, True)
                    B.rows(l + 1, kind, rows - H, H, last, True)
                    yield ("swap", first if r > 0 else None, last if r < N - 1 else None, above if r > 0 else None, below if r < N - 1 else None)
                    if r > 0:
                        B.rows(l + 1, kind, -H, H, above, False)
                    if r < N - 1:
                        B.rows(l + 1, kind, rows, H, below, False)
            B.collapse(l, out if l == 0 else None)
        return out


class _Addr:
    """rank / world only: the BandStitchers of a LocalBandGroup never call a transport themselves."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world


class LocalBandGroup:
    """All N row bands of one pair on ONE device, driven by ONE host thread: every band has its own HIP stream, the bands'
    launch sequences (BandStitcher.steps) are advanced round-robin, and what crosses bands is a device copy on the sender's stream
    plus an event the receiver's stream waits for.  The host never waits for the GPU and never switches threads, so the device sees
    the pure dependency graph of the split -- the one-GPU rehearsal of what RCCL's stream-ordered send / recv gives across GPUs
    (8 bands at 24576 x 16384: 57 ms with host-staged hand-offs from 8 threads, 38.5 ms with device hand-offs from 8 threads,
    this class: see profiles/r03_config5_band.json)."""

    def __init__(self, cw, ch, split_levels, world, device, opts=None, fuse_sweeps=None, plane_pipeline_min=None, col_chunks=1):
        import collections
        import torch
        self.world, self.dev = world, device
        self.streams = [torch.cuda.Stream(device=device) for _ in range(world)]
        self.bands = [BandStitcher(cw, ch, split_levels, _Addr(r, world), device, opts, fuse_sweeps, plane_pipeline_min, col_chunks) for r in range(world)]
        self.box = collections.defaultdict(collections.deque)  # (src, dst) -> posted (tensor, event), FIFO

    def close(self):
        for b in self.bands:
            b.close()

    def _post(self, t, src, dst):
        import torch
        c = t.clone()
        ev = torch.cuda.Event()
        ev.record()
        self.box[(src, dst)].append((c, ev))

    def _take(self, src, dst):
        import torch
        if not self.box[(src, dst)]:
            return None
        c, ev = self.box[(src, dst)].popleft()
        torch.cuda.current_stream().wait_event(ev)
        c.record_stream(torch.cuda.current_stream())
        return c

    def _try(self, r, st):
        """Advance rank r's pending request as far as the posted messages allow; True when it is complete (st["val"] set)."""
        import torch
        req, N = st["req"], self.world
        kind = req[0]
        if kind == "send":
            self._post(req[1], r, req[2])
            st["val"] = None
            return True
        if kind == "recv":
            c = self._take(req[2], r)
            if c is None:
                return False
            req[1].copy_(c)
            st["val"] = req[1]
            return True
        if kind == "all_gather":
            if not st["posted"]:
                for d in range(N):
                    if d != r:
                        self._post(req[1], r, d)
                st["posted"], st["got"] = True, {r: req[1]}
            for s_ in range(N):
                if s_ not in st["got"]:
                    c = self._take(s_, r)
                    if c is None:
                        return False
                    st["got"][s_] = c
            st["val"] = torch.stack([st["got"][s_] for s_ in range(N)])
            return True
        _, to_prev, to_next, from_prev, from_next = req  # swap
        if not st["posted"]:
            if to_prev is not None:
                self._post(to_prev, r, r - 1)
            if to_next is not None:
                self._post(to_next, r, r + 1)
            st["posted"], st["got"] = True, {}
        for key, buf, src in (("p", from_prev, r - 1), ("n", from_next, r + 1)):
            if buf is not None and key not in st["got"]:
                c = self._take(src, r)
                if c is None:
                    return False
                buf.copy_(c)
                st["got"][key] = True
        st["val"] = None
        return True

    def run(self, frame, p, offx, offy, mosaic, ox, oy, outs=None):
        """-> the N bands of the mosaic (list, top to bottom)."""
        import torch
        N = self.world
        ready = torch.cuda.Event()
        ready.record()  # the inputs were produced on the caller's stream
        gens = [b.steps(frame, p, offx, offy, mosaic, ox, oy, None if outs is None else outs[r]) for r, b in enumerate(self.bands)]
        state = [{"req": None, "val": None, "posted": False, "got": None, "done": False, "out": None} for _ in range(N)]
        for s_ in self.streams:
            s_.wait_event(ready)
        left = N
        while left:
            progressed = False
            for r in range(N):
                st = state[r]
                if st["done"]:
                    continue
                with torch.cuda.stream(self.streams[r]):
                    while True:
                        if st["req"] is not None:
                            if not self._try(r, st):
                                break  # parked: the message has not been posted yet
                            st["req"] = None
                            progressed = True
                        try:
                            st["req"] = gens[r].send(st["val"])
                            st["val"], st["posted"], st["got"] = None, False, None
                        except StopIteration as e:
                            st["done"], st["out"] = True, e.value
                            left -= 1
                            progressed = True
                            break
            if not progressed:
                raise RuntimeError("band group: every rank waits for a message nobody has posted")
        done = torch.cuda.Event()
        for s_ in self.streams:  # the caller's stream continues behind every band
            done.record(s_)
            torch.cuda.current_stream().wait_event(done)
        for b in self.bands:
            b.seam = b.band.status()
        return [st["out"] for st in state]
