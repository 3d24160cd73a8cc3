#!/usr/bin/env python3
"""bench.py -- warp+blend MPix/s at 4096x4096x3 f32 (BASELINE.json), one process per GPU.

A step = one pass of the hot path (warp + move + multi-band blend) over synthetic input: --streams (default 4)
batches of --batch (default 8) independent config-2 pairs per rank (two 4096x4096x3 f32 frames -> 6144x4096x3 f32
mosaic each; pair i of the config-4 family has p[3] = -2048 - 8i); a batch is ONE launch sequence of a batched plan
on its own HIP stream.  Frames are generated on the device before the timed
region, so every input is resident in HBM when timing starts.  Pairs are independent: each rank works on its own
shard and there is NO data-path collective (weak scaling; RCCL carries only the barriers and the max-over-ranks of
the step time).  --gather adds the optional assembly of SURVEY.md 8(e): each finished mosaic is cast to unsigned char
(the reference's own output type) and all-gathered (RCCL over xGMI) on the communicator's stream while the next
batch computes, so that every rank ends up holding the whole batch -- that exchange is then inside the timed region
(at 8 GPUs it moves 5.4 GB per step into every rank and is link-bound, which is why it is not the default).

Prints ONE JSON line (rank 0).  `value` comes from timed region 1 (all batches in flight).  `roofline` describes
the dominant kernel (largest share of device time) in timed region 2, where one batch is in flight so that a
launch's duration is the kernel's own, timed with HIP events on the launch stream; `pipeline` gives the byte
accounting for the whole pair at the `value` rate.  `cpu_baseline` (N=1, rank 0) times the oracle's CPU restatement on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime multiplexes streams onto 4 hardware queues by default; batches that share a queue serialise behind each
# other (4 streams on 4 queues, one of them shared with the default stream: 1.50 ms/pair; on 8 queues: 1.31).  Must be set
# before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E nominal, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="independent pairs per batch (one launch sequence)")
    ap.add_argument("--streams", type=int, default=4, help="batches in flight per GPU, each on its own HIP stream")
    ap.add_argument("--frame", type=int, default=4096, help="frame edge (4096 = the metric's configuration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-frame", type=int, default=4096, help="frame edge of the CPU baseline's bounded sample")
    ap.add_argument("--no-kernel-events", action="store_true", help="time without per-launch HIP events")
    ap.add_argument("--pixel", choices=["f32", "u8"], default="f32", help="frame pixel type (f32 = the metric; u8 = the reference's own contract)")
    ap.add_argument("--no-single", action="store_true", help="skip the single-pair-in-flight latency measurement")
    ap.add_argument("--gather", action="store_true", help="N>1: all-gather the finished uchar mosaics to every rank each step (inside the timed region)")
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


def main():
    args = parse()
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # STITCH_FORCE_DIST=1 runs the N>1 code path (RCCL init, quantise, asynchronous all-gather) with a single rank --
    # a rehearsal of the multi-GPU path on a one-GPU box; it is never set by the driver
    force_dist = world == 1 and os.environ.get("STITCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    use_gather = force_dist or (world > 1 and args.gather)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with _StdoutToStderr():
            if force_dist:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29517")
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            else:
                dist.init_process_group("nccl", device_id=dev)
            dist.barrier()  # creates the communicator now (and its banner), not inside the timed region
            torch.cuda.synchronize()

    F = args.frame
    cw, ch = pipeline.config_canvas(F)
    K, W, B, S = args.steps, args.warmup, args.batch, args.streams
    tdt = torch.float32 if args.pixel == "f32" else torch.uint8
    px_bytes = 4 if args.pixel == "f32" else 1

    # S "lanes": each lane = one batched plan (B pairs per launch sequence) on its own HIP stream, so that the
    # latency-bound small pyramid levels of one batch overlap the bandwidth-bound sweeps of the other.
    # Inputs: this rank's pairs come from the config-4 family (frames 2i, 2i+1; map p[3] = -F/2 - 8i); a few distinct
    # batches per lane are cycled; everything is resident in HBM before timing starts.
    n_distinct = 2
    first = rank * K * B * S
    lanes = []
    for ln in range(S):
        batches = []
        for j in range(n_distinct):
            items = []
            for q in range(B):
                i = (first + (ln * n_distinct + j) * B + q) % 32
                items.append((capi.dev_synth(F, F, 2 * i + 1, tdt, dev), pipeline.config_map(i, F), 0.0, 0.0,
                              capi.dev_synth(F, F, 2 * i, tdt, dev), 0, 0))
            batches.append(items)
        lanes.append({
            "plan": capi.Plan(cw, ch, max_pairs=B),
            "stream": torch.cuda.Stream(device=dev),
            "batches": batches,
            "outs": [[torch.empty((3, ch, cw), dtype=tdt, device=dev) for _ in range(B)] for _ in range(2)],
            # N>1: finished mosaics travel as unsigned char (the reference's output type) through pipeline.MosaicGather
            # -- the class the gloo tests cover -- asynchronously: the gather of step k overlaps the kernels of step k+1
            "gather": pipeline.MosaicGather((B, 3, ch, cw), dev, world, rank, slots=2, force_collective=force_dist) if use_gather else None,
        })
    plan = lanes[0]["plan"]

    def lane_step(ln, k, n=B):
        L = lanes[ln]
        with torch.cuda.stream(L["stream"]):
            outs = L["outs"][k % 2]
            L["plan"].pairs([it + (outs[q],) for q, it in enumerate(L["batches"][k % n_distinct][:n])])
            if L["gather"] is not None:
                slot = L["gather"].input_slot(k)
                for q in range(n):
                    if tdt == torch.float32:
                        capi.dev_quantize(outs[q], slot[q])
                    else:
                        slot[q].copy_(outs[q])
                L["gather"].submit(k)

    def drain():
        for L in lanes:
            if L["gather"] is not None:
                with torch.cuda.stream(L["stream"]):
                    L["gather"].drain()
        torch.cuda.synchronize()

    def timed(n_lanes, steps, k0):
        """steps x (one batch on each of n_lanes lanes), bracketed by barrier + synchronize; max over ranks."""
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            for ln in range(n_lanes):
                lane_step(ln, k0 + k)
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for k in range(W):
        for ln in range(S):
            lane_step(ln, k)
    drain()
    for L in lanes:
        for q in range(B):
            L["plan"].status(q)  # raises if a seam scan failed

    # pilot (one lane, every launch bracketed by HIP events; ~10 % overhead, so never the timed region): per-kernel
    # table and the choice of the dominant kernel
    plan.set_profiling(True)
    plan.read_profile()
    PILOT = 3
    for k in range(PILOT):
        lane_step(0, W + k)
    drain()
    pilot = plan.read_profile()
    dom = max(pilot, key=lambda k_: pilot[k_][0])
    plan.set_profiling(False)

    # timed region 1 -> `value`: all S lanes in flight
    elapsed = timed(S, K, W + PILOT)

    # timed region 2 -> `roofline`: ONE lane in flight (kernels do not overlap, so a launch's duration is the kernel's
    # own), HIP events around the dominant kernel's launches only, on the launch stream
    if not args.no_kernel_events:
        plan.set_profiling_kernel(dom)
    plan.read_profile()
    elapsed_one = timed(1, K, W + PILOT + K)
    prof = plan.read_profile()
    plan.set_profiling(False)
    seam = plan.status(0)

    # single pair in flight (config 2 as a latency figure)
    single_ms = None
    if world == 1 and not args.no_single:
        for k in range(2):
            lane_step(0, k, 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(5):
            lane_step(0, k, 1)
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t1) / 5 * 1e3

    if rank == 0:
        mpix_pair = cw * ch / 1e6
        value = mpix_pair * K * B * S * world / elapsed
        per_kernel, stages = pipeline.algorithmic_bytes(F * F, F * F, plan.level_w, plan.level_h, px_bytes, plan.fused_sweep_levels)
        line = {
            "metric": "warp+blend MPix/s at 4096x4096x3 f32" if args.pixel == "f32" else "warp+blend MPix/s at 4096x4096x3 u8", "value": round(value, 2), "unit": "MPix/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (f64 accumulators)" if args.pixel == "f32" else "u8 frames, f32 pyramids (f64 accumulators)",
            "data": "synthetic",
            "config": {"workload": f"config 2 pairs ({F}x{F}x3 {args.pixel} frames -> {cw}x{ch}x3 {args.pixel} mosaic: warp + move + "
                                   f"{plan.levels}-level multi-band blend); per GPU per step {S} batches of {B} independent pairs, "
                                   f"each batch one launch sequence on its own HIP stream (config 4's per-GPU shard); canvas pixels counted",
                       "frame": [F, F, 3], "canvas": [cw, ch, 3], "levels": plan.levels, "pairs_per_batch": B,
                       "batches_in_flight": S, "fused_sweep_levels": plan.fused_sweep_levels, "pairs_per_step": B * S * world, "mpix_per_pair": round(mpix_pair, 3),
                       "ms_per_pair_per_gpu": round(elapsed / K / B / S * 1e3, 4),
                       "one_batch_in_flight_ms_per_pair": round(elapsed_one / K / B * 1e3, 4),
                       "one_batch_in_flight_mpix_s": round(mpix_pair * K * B * world / elapsed_one, 1),
                       "single_pair_in_flight_ms": round(single_ms, 4) if single_ms else None,
                       "single_pair_in_flight_mpix_s": round(mpix_pair / single_ms * 1e3, 1) if single_ms else None,
                       "input_frame_mpix_per_s": round(2 * F * F / 1e6 * K * B * S * world / elapsed, 2),
                       "exchange": "uint8 mosaics all-gathered (RCCL) overlapped with compute" if use_gather else "none (independent shards; RCCL for barriers and timing only)",
                       "seam": list(seam.as_tuple())},
        }
        if prof[dom][1] > 0:
            ms, launches, _ = prof[dom]
            bytes_per_launch = per_kernel[dom] * B * K / launches  # region 2: K steps of one batch of B pairs
            avg_s = ms / launches / 1e3
            achieved = bytes_per_launch / avg_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    te = json.load(open(tpath)).get(dom, {})
                    traffic = te.get("hbm_bytes_per_launch") if te.get("batch") == B and F == 4096 else None
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
            line["roofline"] = {"bound": "hbm", "kernel": capi.KERNEL_SYMBOLS[dom].replace("<T,", "<float,"), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                                "avg_launch_ms": round(ms / launches, 5), "launches": launches,
                                "algorithmic_bytes_per_launch": int(bytes_per_launch),
                                "share_of_device_time": round(pilot[dom][0] / pilot_tot, 4),
                                "device_copy_GBps": round(copy_gbs, 1),
                                "note": "timed region 2 (one batch in flight, so launches of different batches do not overlap): average over "
                                        "every launch of this kernel symbol (all pyramid levels); HIP events on the launch stream"}
        line["kernels"] = {k_: {"ms_per_pair": round(v[0] / PILOT / B, 4), "launches_per_step": v[1] // PILOT,
                                "level0_ms_per_pair": round(v[2] / PILOT / B, 4),
                                "algorithmic_GBps": round(per_kernel[k_] / (v[0] / PILOT / B / 1e3) / 1e9, 1) if v[0] > 0 else None}
                           for k_, v in pilot.items()}
        pair_s = elapsed / K / B / S
        line["pipeline"] = {"algorithmic_bytes_per_pair": stages["total"], "S1": stages["S1"], "S2": stages["S2"], "S3": stages["S3"],
                            "achieved_GBps_per_gpu": round(stages["total"] / pair_s / 1e9, 1),
                            "frac_of_hbm_peak": round(stages["total"] / pair_s / 1e9 / HBM_PEAK_GBS, 4)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample_frame, args.verbose)
        print(json.dumps(line), flush=True)
    if use_gather and rank == 0 and os.environ.get("STITCH_CHECK_GATHER") == "1":
        # rehearsal check: the last gathered block of lane 0 holds this rank's own quantised mosaics
        L = lanes[0]
        last = W + PILOT + 2 * K - 1
        own = L["gather"].out[last % 2][rank]
        ok = all(torch.equal(own[q], capi.dev_quantize(L["outs"][last % 2][q]) if tdt == torch.float32 else L["outs"][last % 2][q]) for q in range(B))
        print(f"[gather check] own mosaics in the gathered block: {'ok' if ok else 'MISMATCH'}", file=sys.stderr)
    for L in lanes:
        L["plan"].close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
