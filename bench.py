#!/usr/bin/env python3
"""bench.py -- warp+blend MPix/s at 4096x4096x3 f32 (BASELINE.json), one process per GPU.

Workload (the same at every N, so the N = 1, 2, 4, 8 lines are one strong-scaling curve): a STEP is one batch of
--pairs-per-step (default 32) independent config-2 pairs -- BASELINE.json configs[3]: pair i = synthetic frames 2i+1
(warped through the map with p[3] = -2048 - 8i) and 2i (the running mosaic), two 4096x4096x3 f32 frames ->
6144x4096x3 f32 mosaic, warp + move + 12-level multi-band blend.  The batch is sharded contiguously over the ranks
(pipeline.shard_range: 4 pairs per GPU on 8 ranks); a rank runs its shard as launch sequences of at most --batch (16)
pairs on batched plans, up to --streams (4) sequences in flight on separate HIP streams, steps back to back with no
host synchronisation between them.  Frames are generated on the device before the timed region: every input is
resident in HBM when timing starts.

N > 1: every finished mosaic also leaves the level-0 collapse as unsigned char (the reference's own output type,
CImg<unsigned char>; `out_u8` of the pair descriptor) and the
step's 32 mosaics are all-gathered (RCCL over xGMI, asynchronously, overlapping the next step's kernels) so that every
rank ends up holding the whole batch -- the one exchange the path has, INSIDE the timed region; `value` includes it and
`config.no_exchange_mpix_s` gives the same run without it.  N = 1: the rank already holds the batch, no exchange.

Every timed region is checked: after it, status() of every plan and pair (seam scan + the fused sweep's sticky
time-out count) and a bit-for-bit comparison of EVERY output buffer with the result of a separately created
single-pair plan (the unfused launch sequence that tests/test_gpu_golden.py compares with the oracle at this size);
N > 1 also checks the gathered blocks of all ranks by checksum.  `outputs_verified` reports it (all ranks).

Prints ONE JSON line (rank 0).  `value` comes from timed region 1 (all sequences in flight).  `roofline` describes
the dominant kernel (largest share of device time) in timed region 2, where one sequence is in flight so that a
launch's duration is the kernel's own, timed with HIP events on the launch stream; `pipeline` gives the byte
accounting for the whole pair at the `value` rate.  `cpu_baseline` (N=1, rank 0) times the oracle's CPU restatement
on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime multiplexes streams onto 4 hardware queues by default; sequences that share a queue serialise behind
# each other (4 streams on 4 queues, one of them shared with the default stream: 1.50 ms/pair; on 8 queues: 1.31).  Must
# be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E nominal, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs-per-step", type=int, default=32, help="pairs of one step over ALL ranks (32 = config 4's batch); "
                    "--pairs-per-step 4 on one GPU rehearses one rank's share of the 8-GPU run")
    ap.add_argument("--batch", type=int, default=16, help="pairs per launch sequence (batched plan), at most (the ABI's limit is 16)")
    ap.add_argument("--streams", type=int, default=4, help="launch sequences in flight per GPU, each on its own HIP stream")
    ap.add_argument("--frame", type=int, default=4096, help="frame edge (4096 = the metric's configuration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-frame", type=int, default=4096, help="frame edge of the CPU baseline's bounded sample")
    ap.add_argument("--no-kernel-events", action="store_true", help="time without per-launch HIP events")
    ap.add_argument("--pixel", choices=["f32", "u8"], default="f32", help="frame pixel type (f32 = the metric; u8 = the reference's own contract)")
    ap.add_argument("--no-single", action="store_true", help="skip the single-pair-in-flight latency measurement")
    ap.add_argument("--no-gather", action="store_true", help="N>1: leave the finished mosaics on their ranks (no exchange at all)")
    ap.add_argument("--no-verify", action="store_true", help="skip the output comparison (status() checks stay)")
    ap.add_argument("--no-coalesce", action="store_true", help="one launch sequence per step even when a rank's share of a step is "
                    "smaller than --batch (default: consecutive steps' shares are launched together, up to --batch pairs)")
    ap.add_argument("--stagger-ms", type=float, default=0.0, help="delay (host sleep) in front of the first launch sequence of lanes 1, 2, ... "
                    "of every timed region, inside the region: the lanes then run out of phase (A/B switch, see DESIGN.md 5)")
    ap.add_argument("--verbose", action="store_true")
    return ap.parse_args()


def cpu_baseline(sample_frame, verbose=False):
    """The oracle (CPU restatement proven bit-identical to the reference, kind 'port') on one f32 pair of
    sample_frame^2 frames -> 1.5*sample_frame x sample_frame canvas, same map family; single thread and all cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from oracle_lib import Oracle
    from computervisionimagestich2_amd import pipeline
    O = Oracle()
    S = sample_frame
    cw, ch = pipeline.config_canvas(S)
    A, B = O.synth(S, S, 0, np.float32), O.synth(S, S, 1, np.float32)
    p = pipeline.config_map(0, S)
    # threads this process may actually run on (the GPU box grants a share of the host's cores)
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, int(os.environ.get("STITCH_CPU_THREADS", "16"))))
    res = {}
    for thr in sorted({1, ncores}):
        O.set_threads(thr)
        t = time.time()
        rc, _ = O.pair(B, p, 0.0, 0.0, A, 0, 0, cw, ch)
        dt = time.time() - t
        assert rc == 0
        res[thr] = cw * ch / dt / 1e6
        if verbose:
            print(f"[cpu_baseline] {thr} thread(s): {dt:.2f} s -> {res[thr]:.3f} MPix/s", file=sys.stderr)
    best = max(res, key=lambda k: res[k])
    return {"value": round(res[best], 4), "unit": "MPix/s", "cores": best, "kind": "port",
            "sample": f"one {S}x{S}x3 f32 pair -> {cw}x{ch} canvas (oracle/stitch_oracle.c, OpenMP over lines)",
            "single_thread_value": round(res[1], 4), "host_cores": ncores}


class _StdoutToStderr:
    """RCCL prints a version banner on stdout when a communicator is created; keep stdout for the ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def self_launch(n, argv, worker=None, extra_env=None):
    """`python3 bench.py --gpus N` without a launcher around it: start the N ranks as FRESH child processes (one per GPU,
    `python -m torch.distributed.run`, rendezvous on 127.0.0.1) and hand back the launcher's exit status.  Rank 0's single
    JSON line reaches stdout because the children inherit it (every other rank prints to stderr only).  Runs before
    anything imports torch or touches HIP: this process never initialises the GPU, so nothing is exec'ed or forked from a
    process that holds a device.  `worker` (tests): the script the ranks run instead of this file."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:  # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), worker or os.path.abspath(__file__)] + list(argv)
    print(f"[bench] --gpus {n} without WORLD_SIZE: launching {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        assert "torch" not in sys.modules, "the launcher must run before torch is imported"
        sys.exit(self_launch(args.gpus, sys.argv[1:], worker=os.environ.get("STITCH_BENCH_WORKER")))
    import torch
    import torch.distributed as dist
    from computervisionimagestich2_amd import capi, pipeline

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    # STITCH_BENCH_BACKEND=gloo: a REHEARSAL of the N > 1 code of this file on a one-GPU box -- every rank drives cuda:0, the
    # collectives run over gloo on host tensors and the mosaics are staged through host memory (MosaicGather(staged=True)).
    # Everything else (shards, ragged blocks, coalesced sequences, verify(), the reductions) is the code the RCCL run executes.
    # Never set by the driver; the line it produces says so (`config.backend`) and is not a scaling measurement.
    backend = os.environ.get("STITCH_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit(f"STITCH_BENCH_BACKEND={backend}: nccl or gloo")
    rehearsal = backend == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearsal else dev  # where the small collectives' tensors live
    # STITCH_FORCE_DIST=1 runs the N>1 code path (RCCL init, quantise, asynchronous all-gather, all-reduces) with a single
    # rank -- a rehearsal of the multi-GPU path on a one-GPU box; it is never set by the driver
    force_dist = world == 1 and os.environ.get("STITCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    use_gather = use_dist and not args.no_gather
    n_ranks_seen = 1
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with _StdoutToStderr():
            if force_dist:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29517")
                dist.init_process_group(backend, rank=0, world_size=1, **({} if rehearsal else {"device_id": dev}))
            elif rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            # a real collective over the communicator (creates it now, banner included, not inside the timed region):
            # every rank contributes 1, so the sum is the number of ranks RCCL actually connected
            ones = torch.ones(1, dtype=torch.int32, device=cdev)
            dist.all_reduce(ones)
            torch.cuda.synchronize()
            n_ranks_seen = int(ones.item())
        if n_ranks_seen != dist.get_world_size():
            raise SystemExit(f"{backend} saw {n_ranks_seen} ranks, world size is {dist.get_world_size()}")

    F = args.frame
    cw, ch = pipeline.config_canvas(F)
    K, W, P = args.steps, args.warmup, args.pairs_per_step
    if P < world:
        raise SystemExit(f"--pairs-per-step {P} < {world} ranks")
    tdt = torch.float32 if args.pixel == "f32" else torch.uint8
    px_bytes = 4 if args.pixel == "f32" else 1

    # ---- this rank's shard of the step, as launch sequences ------------------------------------------------------
    lo, hi = pipeline.shard_range(P, rank, world)
    n_local = hi - lo
    n_max = pipeline.shard_range(P, 0, world)[1]  # the largest shard (rank 0's): equal blocks for the all-gather
    B = min(args.batch, n_local)
    seqs = pipeline.batches_of(P, rank, world, B)  # [(first, last+1)] global pair indices, one launch sequence each
    nb = len(seqs)
    # A shard smaller than a launch sequence (8 GPUs: 4 pairs per rank and step): the shares of G consecutive steps go into ONE
    # launch sequence of G * n_local pairs -- steps run back to back without host synchronisation anyway, every step keeps its own
    # output buffers and its own gather -- so that a rank's launches are as long as the single-GPU run's (the fused sweep's fill
    # and the coarse levels' launches are paid once per sequence, not once per 4 pairs).  --no-coalesce: one sequence per step.
    G = 1 if args.no_coalesce else pipeline.steps_per_sequence(n_local, args.batch, nb, K)
    S = max(1, min(args.streams, -(-nb * max(K, 1) // G)))
    pair_in = {}
    for i in range(lo, hi):
        pair_in[i] = (capi.dev_synth(F, F, 2 * i + 1, tdt, dev), pipeline.config_map(i, F), 0.0, 0.0,
                      capi.dev_synth(F, F, 2 * i, tdt, dev), 0, 0)
    lanes = []
    for ln in range(S):
        lanes.append({"plan": capi.Plan(cw, ch, max_pairs=B * G), "stream": torch.cuda.Stream(device=dev),
                      "outs": [[torch.empty((3, ch, cw), dtype=tdt, device=dev) for _ in range(B * G)] for _ in range(2)],
                      "holds": [None, None], "last": None})
    plan = lanes[0]["plan"]
    gather = None
    if use_gather:
        # finished mosaics travel as unsigned char through pipeline.MosaicGather -- the class the gloo tests cover; one block
        # of n_max mosaics per rank and step, two steps of buffers so that the gather of step k overlaps the kernels of k+1
        # Input ring: (S + 1) * G blocks -- a launch sequence fills the blocks of G steps and waits for the gathers that last used
        # them, so with S sequences in flight none of them waits for an exchange that has not even been submitted (2 * G blocks
        # held the sequences in flight to two).  The gathered batches (world times larger) rotate through two buffers.
        gather = pipeline.MosaicGather((n_max, 3, ch, cw), dev, world, rank, slots=(S + 1) * G, out_slots=2, force_collective=True,
                                       staged=rehearsal)
        gstream = torch.cuda.Stream(device=dev)

    def run_seq(c, seq, gather_steps=None, n=None, copies=1):
        """Launch sequence number c (a global counter picks lane and output slot) over global pairs seq = (first, last+1),
        `copies` times over (the shares of that many consecutive steps in one sequence); gather_steps: the step numbers whose
        gather blocks receive the unsigned char mosaics, one per copy."""
        ln, slot = c % S, (c // S) % 2
        L = lanes[ln]
        idx = list(range(seq[0], seq[1]))[:n] * copies
        per = len(idx) // copies
        with torch.cuda.stream(L["stream"]):
            outs = L["outs"][slot]
            if gather_steps is None:
                L["plan"].pairs([pair_in[i] + (outs[q],) for q, i in enumerate(idx)])
            else:
                blks = [gather.input_slot(st_) for st_ in gather_steps]  # waits (on this stream) for the gathers that last used the buffers
                if tdt == torch.float32:  # the unsigned char copy that travels is written by the level-0 collapse itself (out_u8)
                    L["plan"].pairs([pair_in[i] + (outs[q], blks[q // per][i - lo]) for q, i in enumerate(idx)])
                else:
                    L["plan"].pairs([pair_in[i] + (outs[q],) for q, i in enumerate(idx)])
                    for q, i in enumerate(idx):
                        blks[q // per][i - lo].copy_(outs[q])
            L["holds"][slot] = idx
            L["last"] = slot
            if gather_steps is not None:
                ev = torch.cuda.Event()
                ev.record(L["stream"])
                return ev
        return None

    def run_steps(k0, count, with_gather):
        """Steps k0 .. k0+count-1: launch sequences of the rank's shard; when the shard is small, up to G consecutive steps'
        shares per sequence.  The sequences are cut in step order (the gathers are submitted in step order on every rank) and
        balanced: their number is rounded up to a multiple of the lanes, so that the lanes finish together instead of one
        sequence running on alone at the end (20 steps, G = 4, 4 lanes: 3+3+3+3+2+2+2+2, not 4+4+4+4+4)."""
        sizes = pipeline.sequence_sizes(count, G, S)
        k = k0
        launched = 0
        for i, g in enumerate(sizes):
            steps_ = list(range(k, k + g))
            evs = []
            for j in range(nb):
                if args.stagger_ms > 0 and 0 < launched < S:  # the first sequence of lanes 1 .. S-1
                    time.sleep(args.stagger_ms / 1e3)
                launched += 1
                evs.append(run_seq((k0 // G + i) * nb + j, seqs[j], steps_ if with_gather else None, copies=g))
            if with_gather:
                with torch.cuda.stream(gstream):
                    for ev in evs:
                        gstream.wait_event(ev)
                    for st_ in steps_:
                        gather.submit(st_)  # asynchronous: runs on RCCL's stream behind gstream
            k += g

    def drain():
        if gather is not None:
            with torch.cuda.stream(gstream):
                gather.drain()
        torch.cuda.synchronize()

    def timed(fn, steps):
        """fn(steps) = exactly `steps` steps of work, bracketed by barrier + synchronize on both sides; max over ranks."""
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(steps)
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # ---- reference results for the output check: every local pair through a single-pair plan (unfused sweeps, plain
    # launch sequence; compared with the oracle at this very size by tests/test_gpu_golden.py) ------------------------
    ref_out, ref_q = {}, {}
    if not args.no_verify:
        one = capi.Plan(cw, ch)
        for i in range(lo, hi):
            ref_out[i] = one.pair(pair_in[i][0], pair_in[i][1], 0.0, 0.0, pair_in[i][4], 0, 0)
            one.status()
            if use_gather:
                q8 = capi.dev_quantize(ref_out[i]) if tdt == torch.float32 else ref_out[i]
                ref_q[i] = int(q8.view(torch.int32).sum(dtype=torch.int64).item()) if q8.numel() % 4 == 0 else int(q8.sum(dtype=torch.int64).item())
        one.close()
    checks = {"regions": 0, "buffers": 0, "bad": []}

    def verify(tag, gather_step=None):
        """After a timed region: status of every plan and pair, every output buffer against its reference."""
        torch.cuda.synchronize()
        for ln, L in enumerate(lanes):
            if L["last"] is None:
                continue
            try:
                for q in range(len(L["holds"][L["last"]])):
                    L["plan"].status(q)  # raises on a failed seam scan or a timed-out hand-off in ANY queued call
            except capi.StitchError as e:
                checks["bad"].append(f"{tag}: lane {ln}: {e}")
                L["plan"].clear_fault()
            if args.no_verify:
                continue
            for slot in range(2):
                for q, i in enumerate(L["holds"][slot] or []):
                    checks["buffers"] += 1
                    if not torch.equal(L["outs"][slot][q], ref_out[i]):
                        checks["bad"].append(f"{tag}: lane {ln} slot {slot} pair {i} differs from the single-pair plan")
        if gather_step is not None and not args.no_verify:
            # the gathered batch of the last step: this rank's own checksums travel by all-gather, every block is checked
            mine = torch.tensor([ref_q.get(lo + j, 0) for j in range(n_max)], dtype=torch.int64, device=cdev)
            allq = torch.empty(world * n_max, dtype=torch.int64, device=cdev)
            dist.all_gather_into_tensor(allq, mine)
            got = gather.gathered(gather_step)
            for r in range(world):
                rlo, rhi = pipeline.shard_range(P, r, world)
                for j in range(rhi - rlo):
                    blk = got[r][j]
                    s_ = int(blk.view(torch.int32).sum(dtype=torch.int64).item()) if blk.numel() % 4 == 0 else int(blk.sum(dtype=torch.int64).item())
                    checks["buffers"] += 1
                    if s_ != int(allq[r * n_max + j].item()):
                        checks["bad"].append(f"{tag}: gathered mosaic of rank {r}, pair {rlo + j}: checksum differs")
        checks["regions"] += 1

    def log(msg):
        if args.verbose:
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    # ---- warm-up ---------------------------------------------------------------------------------------------------
    log(f"warm-up: {W} step(s), {n_local} pairs per step, {G} step(s) per launch sequence, {S} lane(s)")
    run_steps(0, W, use_gather)
    drain()
    log("warm-up done")
    verify("warm-up", (W - 1) if (use_gather and W > 0) else None)
    # A hand-off time-out in the WARM-UP (seen once in several hundred runs: the first launch sequence on a fresh plan, every later
    # one on the same plan correct) is reported in the line (`warmup_handoff_faults`) and the warm-up is run once more; the timed
    # regions are never retried -- a fault there fails the run.
    warmup_faults = [m for m in checks["bad"] if "hand-off wait" in m]
    if warmup_faults and not use_gather:
        print(f"[bench] rank {rank}: hand-off time-out during the warm-up, warm-up repeated once: {warmup_faults[0]}", file=sys.stderr, flush=True)
        checks["bad"] = []
        run_steps(0, W, use_gather)
        drain()
        verify("warm-up (second attempt)", None)

    # pilot (lane 0, every launch bracketed by HIP events; ~10 % overhead, so never the timed region): per-kernel
    # table and the choice of the dominant kernel
    plan.set_profiling(True)
    plan.read_profile()
    PILOT = 3
    for k in range(PILOT):
        run_seq(k * S, seqs[0], copies=G)
    torch.cuda.synchronize()
    pilot = plan.read_profile()
    dom = max(pilot, key=lambda k_: pilot[k_][0])
    plan.set_profiling(False)

    log("pilot done")
    # timed region 1 -> `value`: every launch sequence of K steps in flight over S streams (+ the exchange, N > 1)
    elapsed = timed(lambda n_: run_steps(0, n_, use_gather), K)
    verify("region 1", (K - 1) if use_gather else None)
    elapsed_noex = None
    if use_gather:
        elapsed_noex = timed(lambda n_: run_steps(0, n_, False), K)
        verify("region 1 (no exchange)")

    log("region 1 done")
    # timed region 2 -> `roofline`: ONE sequence in flight (kernels do not overlap, so a launch's duration is the kernel's
    # own), HIP events around the dominant kernel's launches only, on the launch stream
    if not args.no_kernel_events:
        plan.set_profiling_kernel(dom)
    plan.read_profile()
    def one_lane(n_):
        for k in range(n_):
            run_seq(k * S, seqs[0], copies=G)

    elapsed_one = timed(one_lane, K)
    prof = plan.read_profile()
    plan.set_profiling(False)
    verify("region 2")
    seam = plan.status(0)
    n_seq0 = (seqs[0][1] - seqs[0][0]) * G

    log("region 2 done")
    # single pair in flight (config 2 as a latency figure)
    single_ms = None
    if world == 1 and not args.no_single:
        for k in range(2):
            run_seq(k * S, seqs[0], n=1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(5):
            run_seq(k * S, seqs[0], n=1)
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t1) / 5 * 1e3

    # ONE step -- the whole batch of P pairs, start to finish, nothing else in flight: this rank's share as its own launch
    # sequence(s) (no coalescing with other steps), the step's exchange included when N > 1; max over ranks, best of three
    def one_step(k):
        evs = [run_seq(k * nb + j, seqs[j], [k] if use_gather else None) for j in range(nb)]
        if use_gather:
            with torch.cuda.stream(gstream):
                for ev in evs:
                    gstream.wait_event(ev)
                gather.submit(k)

    single_batch_ms = min(timed(lambda n_, k=k: one_step(k), 1) for k in range(3)) * 1e3
    verify("single batch", 2 if use_gather else None)

    ok = torch.tensor([0 if checks["bad"] else 1], dtype=torch.int32, device=cdev)
    per_rank = [[n_local, nb, G, S]]
    if use_dist:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        # every rank's own schedule (ragged shards cut differently: the line reports all of them, not rank 0's)
        mine_ = torch.tensor(per_rank[0], dtype=torch.int64, device=cdev)
        all_ = torch.empty(dist.get_world_size() * 4, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(all_, mine_)
        per_rank = all_.view(-1, 4).tolist()
    verified = bool(ok.item()) and not args.no_verify
    for msg in checks["bad"]:
        print(f"[bench] rank {rank}: OUTPUT CHECK FAILED: {msg}", file=sys.stderr)

    if rank == 0:
        mpix_pair = cw * ch / 1e6
        value = mpix_pair * K * P / elapsed
        # the forms the timed launch sequences ran with, from the plan and the library's per-call rule (not re-derived from the environment)
        forms = plan.call_forms(n_seq0)
        src_fused = "source_fused" in forms
        # levels whose anticausal-x and causal-y sweeps are one launch: the plan's (a batch), or -- one pair per sequence, the
        # five-wavefront sweep -- the first levels of 20 MPix and more (the library's rule, lone_fused_level)
        if "fused_sweep" not in forms:
            fused_levels = 0
        elif n_seq0 == 1:
            fused_levels = sum(1 for l_ in range(min(4, len(plan.level_w) - 1)) if plan.level_w[l_] * plan.level_h[l_] >= 20_000_000)
        else:
            fused_levels = plan.fused_sweep_levels
        per_kernel, stages = pipeline.algorithmic_bytes(F * F, F * F, plan.level_w, plan.level_h, px_bytes, fused_levels,
                                                        fused_decimate="fused_decimate" in plan.fast_paths, source_fused=src_fused,
                                                        implicit_mask="implicit_mask" in plan.fast_paths, coarse_from=plan.coarse_from)
        line = {
            "metric": "warp+blend MPix/s at 4096x4096x3 f32" if args.pixel == "f32" else "warp+blend MPix/s at 4096x4096x3 u8", "value": round(value, 2), "unit": "MPix/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (f64 accumulators)" if args.pixel == "f32" else "u8 frames, f32 pyramids (f64 accumulators)",
            "data": "synthetic", "outputs_verified": verified, "n_ranks_seen": n_ranks_seen,
            "config": {"workload": f"config 2 pairs ({F}x{F}x3 {args.pixel} frames -> {cw}x{ch}x3 {args.pixel} mosaic: warp + move + "
                                   f"{plan.levels}-level multi-band blend), {P} independent pairs per step (config 4's batch) sharded contiguously "
                                   f"over {world} rank(s): {n_local} pairs per rank per step as {nb} launch sequence(s) of {B}"
                                   + (f" (the shares of {G} consecutive steps launched together: {B * G} pairs per sequence)" if G > 1 else "") + " on batched plans, "
                                   f"{S} sequences in flight per GPU on separate HIP streams; canvas pixels counted",
                       "frame": [F, F, 3], "canvas": [cw, ch, 3], "levels": plan.levels, "pairs_per_step": P, "pairs_per_rank_per_step": n_local,
                       "pairs_per_sequence": B * G, "steps_per_sequence": G, "sequences_in_flight": S, "fused_sweep_levels": fused_levels,
                       "forms": sorted(forms), "backend": ("gloo REHEARSAL on one GPU: not a scaling measurement" if rehearsal else "nccl (RCCL)") if use_dist else None,
                       "per_rank": [{"rank": r_, "pairs_per_step": v[0], "sequences_per_step": v[1], "steps_per_sequence": v[2], "lanes": v[3],
                                     "gather_input_blocks": (v[3] + 1) * v[2] if use_gather else 0} for r_, v in enumerate(per_rank)],
                       "single_batch_ms": round(single_batch_ms, 4),
                       "single_batch_note": f"ONE step of {P} pairs start to finish with nothing else in flight (no coalescing with other steps"
                                            + (", its all-gather included" if use_gather else "") + "), max over ranks, best of 3",
                       "mpix_per_pair": round(mpix_pair, 3),
                       "ms_per_pair_per_gpu": round(elapsed / K / n_local * 1e3, 4),
                       "one_sequence_in_flight_ms_per_pair": round(elapsed_one / K / n_seq0 * 1e3, 4),
                       "one_sequence_in_flight_mpix_s": round(mpix_pair * K * n_seq0 / elapsed_one, 1),
                       "single_pair_in_flight_ms": round(single_ms, 4) if single_ms else None,
                       "single_pair_in_flight_mpix_s": round(mpix_pair / single_ms * 1e3, 1) if single_ms else None,
                       "input_frame_mpix_per_s": round(2 * F * F / 1e6 * K * P / elapsed, 2),
                       "exchange": (f"every step's {P} finished mosaics cast to uint8 and all-gathered (RCCL, asynchronous, overlapping the "
                                    f"next step) inside the timed region: {n_max * 3 * ch * cw * (world - 1) / 1e9:.2f} GB into every rank per step")
                                   if use_gather else "none (one rank holds the whole batch)" if world == 1 else "none (--no-gather)",
                       "no_exchange_mpix_s": round(mpix_pair * K * P / elapsed_noex, 2) if elapsed_noex else None,
                       "warmup_handoff_faults": len(warmup_faults),
                       "output_check": {"timed_regions_checked": checks["regions"], "buffers_compared": checks["buffers"],
                                        "against": "single-pair plan (separate launch sequence), bit for bit; status() of every plan and pair",
                                        "failures": len(checks["bad"])},
                       "seam": list(seam.as_tuple())},
        }
        if prof[dom][1] > 0:
            ms, launches, _ = prof[dom]
            bytes_per_launch = per_kernel[dom] * n_seq0 * K / launches  # region 2: K launch sequences of n_seq0 pairs
            avg_s = ms / launches / 1e3
            achieved = bytes_per_launch / avg_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    te = json.load(open(tpath)).get(dom, {})
                    traffic = te.get("hbm_bytes_per_launch") if te.get("batch") == n_seq0 and F == 4096 else None
                except Exception:
                    traffic = None
            pilot_tot = sum(v[0] for v in pilot.values())
            # the box's device-to-device copy rate (bytes read + bytes written), for orientation next to the nominal peak
            csrc = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            cdst = torch.empty_like(csrc)
            cdst.copy_(csrc)
            torch.cuda.synchronize()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(5):
                cdst.copy_(csrc)
            c1.record()
            torch.cuda.synchronize()
            copy_gbs = 5 * 2 * csrc.numel() / (c0.elapsed_time(c1) / 1e3) / 1e9
            del csrc, cdst
            dom_sym = capi.KERNEL_SYMBOLS[dom].replace("<T,", "<float,")
            if dom == "vv_xbyf" and n_seq0 == 1:  # one pair per sequence: the five-wavefront sweep, not the batch's
                dom_sym = "k_vv_xby_m"
            line["roofline"] = {"bound": "hbm", "kernel": dom_sym, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                                "traffic_source": "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command run by the builder "
                                                  "(scripts/pmc.sh); NOT measured in this run" if traffic is not None else None,
                                "avg_launch_ms": round(ms / launches, 5), "launches": launches,
                                "algorithmic_bytes_per_launch": int(bytes_per_launch),
                                "share_of_device_time": round(pilot[dom][0] / pilot_tot, 4),
                                "device_copy_GBps": round(copy_gbs, 1),
                                "note": "timed region 2 (one launch sequence in flight, so launches do not overlap): average over every launch of "
                                        "this kernel symbol (all pyramid levels); HIP events on the launch stream; bytes = what this kernel "
                                        "itself reads and writes once (pipeline.algorithmic_bytes, fusion taken into account)"}
            kern = {}
            for k_, v in pilot.items():
                gbps = per_kernel[k_] / (v[0] / PILOT / n_seq0 / 1e3) / 1e9 if v[0] > 0 else None
                kern[k_] = {"ms_per_pair": round(v[0] / PILOT / n_seq0, 4), "launches_per_sequence": v[1] // PILOT,
                            "level0_ms_per_pair": round(v[2] / PILOT / n_seq0, 4), "owned_bytes_per_pair": per_kernel[k_],
                            "algorithmic_GBps": round(gbps, 1) if gbps is not None else None}
                if gbps is not None and gbps > copy_gbs * 1.6:  # sanity: nothing streams far above the box's own copy rate
                    kern[k_]["suspect"] = "above the device copy rate: byte accounting or timing is off"
            line["kernels"] = kern
        pair_s = elapsed / K / P
        line["pipeline"] = {"algorithmic_bytes_per_pair": stages["total"], "S1": stages["S1"], "S2": stages["S2"], "S3": stages["S3"],
                            "achieved_GBps_per_gpu": round(stages["total"] / pair_s / 1e9 / world, 1),
                            "frac_of_hbm_peak": round(stages["total"] / pair_s / 1e9 / world / HBM_PEAK_GBS, 4),
                            "owned_bytes_per_pair_all_kernels": sum(per_kernel.values())}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample_frame, args.verbose)
            # the reference's OWN loop on the same pair (unsigned char frames, one thread: ~2 minutes, so not inside this run):
            # timed once on a GPU box's host by scripts/bench_reference_config2.py, which also compares the MI355X result with it
            rpath = os.path.join(ROOT, "profiles", "r04_reference_config2.json")
            if os.path.exists(rpath) and F == 4096:
                try:
                    rj = json.load(open(rpath))
                    line["cpu_baseline_reference"] = {"value": rj["value"], "unit": rj["unit"], "cores": rj["cores"], "kind": "reference", "measured_offline": True,
                                                      "sample": f"the reference's warpingImageByHomography + movingImageByOffset + blendTwoImages ({rj['reference_lines']}) "
                                                                f"compiled in place, config 2's pair with unsigned char frames, {rj['total_s']} s on one thread of a GPU box's "
                                                                "host (profiles/r04_reference_config2.json, scripts/bench_reference_config2.py)",
                                                      "mi355x_bit_identical": rj.get("mi355x_same_pair", {}).get("bit_identical_to_the_reference")}
                except Exception:
                    pass
        print(json.dumps(line), flush=True)
    for L in lanes:
        L["plan"].close()
    if use_dist:
        dist.destroy_process_group()
    if checks["bad"] and not args.no_verify:
        sys.exit(3)


if __name__ == "__main__":
    main()
