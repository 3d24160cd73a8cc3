"""CPU: host-side logic of the package (sharding, byte accounting, BMP I/O, config recipes)."""
import os

import numpy as np
import pytest

from computervisionimagestich2_amd import bmp, pipeline


def test_shard_range_covers_everything_once():
    for n, world in [(32, 8), (32, 1), (5, 2), (3, 4), (0, 2), (33, 8)]:
        seen = []
        for r in range(world):
            lo, hi = pipeline.shard_range(n, r, world)
            seen += list(range(lo, hi))
        assert seen == list(range(n))
    assert pipeline.shard_range(32, 3, 8) == (12, 16)  # config 4: 4 pairs per GPU, contiguous


def test_algorithmic_bytes_match_survey():
    """config-2 canvas 6144x4096 (L=12) -> 1.007 + 3.053 + 1.644 = 5.704 GB per pair."""
    lw = lw0 = [6144 >> i for i in range(12)]
    lh = [4096 >> i for i in range(12)]
    per_kernel, st = pipeline.algorithmic_bytes(4096 * 4096, 4096 * 4096, lw, lh, 4, fused_decimate=False)
    assert round(st["S1"] / 1e9, 3) == 1.007 and round(st["S2"] / 1e9, 3) == 3.053 and round(st["S3"] / 1e9, 3) == 1.644
    assert round(st["total"] / 1e9, 3) == 5.704
    lw = [4096 >> i for i in range(12)]
    _, st2 = pipeline.algorithmic_bytes(4096 * 4096, 4096 * 4096, lw, lw, 4)
    assert round(st2["total"] / 1e9, 3) == 3.937  # the 4096x4096 canvas figure of the same section
    assert per_kernel["collapse"] + per_kernel["collapse_l0"] + per_kernel["collapse_top"] == st["S3"] and per_kernel["compose"] == st["S1"]
    # with the fusion the benchmarked plan runs, every kernel is credited only what it moves itself: the fused anticausal-y
    # + decimation reads level l and writes level l+1; the fused sweep reads 6 and writes 7 planes at level 0 (the mask is
    # implicit); source-fused, S1 is one 4-byte index plane and level 0 is gathered from the frames
    pk, st3 = pipeline.algorithmic_bytes(4096 * 4096, 4096 * 4096, lw0, lh, 4, 2, True, True, True)
    n = [w * h for w, h in zip(lw0, lh)]
    assert st3 == st
    assert pk["vv_y_bwd"] == sum(7 * 4 * (n[l] + n[l + 1]) for l in range(11)) and pk["decimate"] == 0
    assert pk["vv_xbyf"] == 4 * n[0] * (6 + 7) + 4 * n[1] * 14
    assert pk["compose"] == 4 * n[0] and pk["mask"] == 0
    # the source-fused causal sweep of level 0 is a kernel symbol (and id) of its own: frames + index plane in, six planes out
    assert pk["vv_x_fwd_src"] == 2 * 4096 * 4096 * 3 * 4 + 4 * n[0] + 4 * n[0] * 6
    assert pk["vv_x_fwd"] == sum(4 * n[l] * 14 for l in range(1, 11))
    inputs = 2 * 4096 * 4096 * 3 * 4
    assert pk["collapse_l0"] == inputs + 4 * n[0] + 4 * 9 * n[1] + 3 * 4 * n[0]
    assert sum(pk.values()) < 1.1 * st["total"]  # the fused design moves about what the canonical accounting counts


def test_batches_of_is_the_config4_schedule():
    assert pipeline.batches_of(32, 3, 8, 8) == [(12, 16)]  # 8 GPUs: one launch sequence of 4 pairs per rank and step
    assert pipeline.batches_of(32, 0, 1, 8) == [(0, 8), (8, 16), (16, 24), (24, 32)]
    assert pipeline.batches_of(32, 1, 2, 8) == [(16, 24), (24, 32)]
    assert pipeline.batches_of(5, 0, 2, 2) == [(0, 2), (2, 3)] and pipeline.batches_of(5, 1, 2, 2) == [(3, 5)]
    assert pipeline.batches_of(3, 3, 4, 8) == []


def test_small_shares_are_launched_together():
    """bench.py's schedule for a rank whose share of a step is smaller than a launch sequence (8 GPUs: 4 of 32 pairs)."""
    assert pipeline.steps_per_sequence(4, 16, 1, 20) == 4    # 8 ranks: four steps' shares = 16 pairs per sequence
    assert pipeline.steps_per_sequence(8, 16, 1, 20) == 2    # 4 ranks
    assert pipeline.steps_per_sequence(16, 16, 1, 20) == 1   # 2 ranks: a share is a whole sequence
    assert pipeline.steps_per_sequence(16, 16, 2, 20) == 1   # 1 rank: two sequences per step, nothing to merge
    assert pipeline.steps_per_sequence(4, 16, 1, 3) == 3     # never more than the steps there are
    assert pipeline.steps_per_sequence(5, 16, 1, 20) == 3    # ragged shard: 15 pairs per sequence
    for count, g, lanes in [(20, 4, 4), (5, 4, 4), (3, 4, 4), (21, 4, 4), (20, 1, 4), (7, 2, 4), (1, 4, 4), (0, 4, 4), (64, 4, 3)]:
        sizes = pipeline.sequence_sizes(count, g, lanes)
        assert sum(sizes) == count and all(1 <= s_ <= g for s_ in sizes), (count, g, lanes, sizes)  # every step exactly once, in order
        assert max(sizes, default=0) - min(sizes, default=0) <= 1
        if g > 1 and len(sizes) > lanes:
            assert len(sizes) % lanes == 0 or len(sizes) == count, (count, g, lanes, sizes)  # the lanes finish together
    assert pipeline.sequence_sizes(20, 4, 4) == [3, 3, 3, 3, 2, 2, 2, 2]
    assert pipeline.sequence_sizes(20, 1, 4) == [1] * 20


def test_config_recipes():
    assert pipeline.config_canvas(4096) == (6144, 4096)
    assert pipeline.config_map(0) == [1.0, 0.002, 1e-6, -2048.0, -0.001, 1.0, 5e-7, 1.5]
    assert pipeline.config_map(3)[3] == -2048.0 - 24.0
    assert pipeline.levels_of(6144, 4096) == 12 and pipeline.levels_of(600, 800, 1) == 9


def test_bmp_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    for (w, h) in [(7, 11), (8, 3), (1, 1), (33, 2)]:
        img = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
        p = os.path.join(tmp_path, f"{w}x{h}.bmp")
        bmp.save_bmp(p, img)
        assert np.array_equal(bmp.load_bmp(p), img)
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "input", "1.bmp")
    assert bmp.load_bmp(g).shape == (3, 512, 384)


def test_canvas_bbox_and_feature_updates_are_host_arithmetic(st, oracle):
    """SURVEY.md 8(f) row 2 (ImageProcess.cpp:206-216, 622-640): computed on the host inside the library -- no
    device needed -- and equal to the oracle / the reference's recorded canvases."""
    import json
    J = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.json")))
    for e in J["canvas"]:
        got = st.capi.canvas_bbox(e["fw"], e["fh"], e["p"], e["rw"], e["rh"])
        assert list(got) == e["out"] == list(oracle.canvas_bbox(e["fw"], e["fh"], e["p"], e["rw"], e["rh"]))
    rng = np.random.default_rng(5)
    p = [0.9724, -0.0398, 0.000149, 206.67, 0.00141, 1.00076, -1.2e-06, 4.55]
    x, y = rng.uniform(0, 384, 300).astype(np.float32), rng.uniform(0, 512, 300).astype(np.float32)
    for a, b in zip(st.capi.map_points(x, y, p, -230.579239, -4.68064785), oracle.map_points(x, y, p, -230.579239, -4.68064785)):
        assert np.array_equal(a, b)
    for a, b in zip(st.capi.shift_points(x, y, -230, -4), oracle.shift_points(x, y, -230, -4)):
        assert np.array_equal(a, b)


def test_step_geometry_reproduces_the_recorded_canvases(st):
    """stitch_step_geometry (host arithmetic, no device): forward map + sizes -> canvas, warp offsets, move offsets of every
    step of the reference's recorded Input/ runs (golden.json, written by the reference build)."""
    import json
    J = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.json")))
    for n in ("2", "4"):
        for s_ in J["runs"][n]["steps"]:
            g = st.capi.step_geometry(s_["fw"], s_["fh"], s_["p_fwd"], s_["mw"], s_["mh"])
            assert (g.cw, g.ch, g.ox, g.oy) == (s_["cw"], s_["ch"], s_["ox"], s_["oy"])
            assert g.min_x == np.float32(s_["offx"]) and g.min_y == np.float32(s_["offy"])


def test_bmp_header_arithmetic_is_host_only(st, oracle):
    """stitch_bmp_parse / stitch_bmp_file_bytes are host arithmetic on the 54-byte header (CImg.h:48413-48441): they work
    without a GPU and agree with the oracle's parse field by field for every layout knob, and refuse what CImg's 24/32-bit
    branch does not cover."""
    import ctypes as C
    from computervisionimagestich2_amd import capi
    from oracle_lib import BmpInfo, make_bmp
    img = oracle.synth(37, 11, 2)
    for kw in [dict(), dict(bpp=32), dict(top_down=True), dict(header_size=108), dict(extra_gap=10), dict(extra_gap=1), dict(size_field=0),
               dict(size_field=60), dict(truncate=7), dict(truncate=200), dict(bpp=32, top_down=True, header_size=124, extra_gap=3)]:
        data = make_bmp(img, **kw)
        got = capi.bmp_parse(data[:54], len(data))
        buf = np.frombuffer(data, np.uint8)
        want = BmpInfo()
        oracle.lib.oracle_bmp_parse.restype = C.c_int
        assert oracle.lib.oracle_bmp_parse(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(want)) == 0
        for f, _ in BmpInfo._fields_:
            assert getattr(got, f) == getattr(want, f), (kw, f)
    assert capi.lib().stitch_bmp_file_bytes(37, 11) == 54 + 112 * 11 == len(oracle.bmp_encode(img))
    assert capi.lib().stitch_bmp_file_bytes(0, 5) == 0
    good = bytearray(make_bmp(img))
    for off, patch in [(0, b"XM"), (0x1C, bytes([8, 0])), (0x1E, bytes([1, 0, 0, 0])), (0x16, bytes([0, 0, 0, 0]))]:
        bad = bytearray(good)
        bad[off:off + len(patch)] = patch
        try:
            capi.bmp_parse(bytes(bad[:54]), len(bad))
            raise AssertionError("accepted a header CImg's 24/32-bit branch does not cover")
        except capi.StitchError as e:
            assert e.code == capi.ERR_ARG


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("mode", ["separate", "fused", "planes"])
def test_band_split_message_order_cannot_deadlock(monkeypatch, world, mode):
    """pipeline.BandStitcher.steps on every rank of a world, with the per-band computation stubbed out (no GPU): the requests the
    ranks yield -- send / recv of the recurrence state, the all-gather of the first replicated level, the neighbour swaps of
    halo rows -- are matched under the STRICTEST transport model (unbuffered: a send completes only when its receive is
    posted, a collective only when every rank has reached it).  Every rank must run to completion in all three forms of the
    reduce (separate sweeps, fused anticausal-x + causal-y sweep, state handed over plane by plane), and every message must be
    consumed: this is what RCCL's stream-ordered send / recv needs across GPUs, checked without one."""
    import torch

    Ls, cw, ch = 3, 768, 384 * 8

    class FakeBand:
        def __init__(self, cw_, ch_, rank, nranks, split, opts=None):
            self.calls = []
            self.geom = [{"rows": (ch_ >> l) // nranks, "w": cw_ >> l, "pitch": ((cw_ >> l) + 63) // 64 * 64, "halo": 2} for l in range(split + 1)]

        def __getattr__(self, name):  # compose, reduce_*, rows, top, collapse, close: recorded, nothing computed
            if name.startswith("__"):
                raise AttributeError(name)
            return lambda *a, **k: self.calls.append(name)

        def status(self):
            return "seam"

    monkeypatch.setattr(pipeline.capi, "Band", FakeBand)
    dev = torch.device("cpu")
    frame = torch.zeros((3, 8, 8), dtype=torch.float32)
    ranks = [pipeline.BandStitcher(cw, ch, Ls, pipeline._Addr(r, world), dev, fuse_sweeps=(mode == "fused"),
                                   plane_pipeline_min=0 if mode == "planes" else None) for r in range(world)]
    gens = [b.steps(frame, [0.0] * 8, 0.0, 0.0, frame, 0, 0) for b in ranks]
    pending, done, outs = [None] * world, [False] * world, [None] * world

    def advance(r, value=None):
        try:
            pending[r] = gens[r].send(value)
        except StopIteration as e:
            pending[r], done[r], outs[r] = None, True, e.value

    for r in range(world):
        advance(r)
    messages = 0
    while not all(done):
        progressed = False
        for r in range(world):
            q = pending[r]
            if q is None:
                continue
            if q[0] == "send":  # ("send", tensor, dst): completes together with the matching recv
                d = q[2]
                if pending[d] is not None and pending[d][0] == "recv" and pending[d][2] == r:
                    assert pending[d][1].shape == q[1].shape
                    buf = pending[d][1]
                    advance(r)
                    advance(d, buf)
                    messages += 1
                    progressed = True
            elif q[0] == "all_gather":
                if all(p is not None and p[0] == "all_gather" for p in pending):
                    stacked = torch.stack([p[1] for p in pending])
                    for k in range(world):
                        advance(k, stacked)
                    progressed = True
            elif q[0] == "swap":  # (to_prev, to_next, from_prev, from_next): both neighbours must be at their swap too
                nb = [k for k in (r - 1, r + 1) if 0 <= k < world]
                if all(pending[k] is not None and pending[k][0] == "swap" for k in nb):
                    _, to_prev, to_next, from_prev, from_next = q
                    assert (to_prev is None) == (r == 0) and (from_prev is None) == (r == 0)
                    assert (to_next is None) == (r == world - 1) and (from_next is None) == (r == world - 1)
                    if r > 0:
                        assert pending[r - 1][2].shape == from_prev.shape  # what the rank above sends down is what this one expects
                    # a rank leaves its swap only when the whole chain of neighbours is there: release all of them together
                    if all(p is not None and p[0] == "swap" for p in pending):
                        for k in range(world):
                            advance(k)
                        progressed = True
        assert progressed, ("deadlock", mode, world, [None if p is None else p[0] for p in pending])
    assert all(o is not None and tuple(o.shape) == (3, ch // world, cw) for o in outs)
    if world > 1:
        per_level = 14 if mode == "planes" else 2  # hand-offs per boundary and level: 7 planes down + 7 up, or one each way
        assert messages == Ls * per_level * (world - 1)
    kinds = {"separate": "reduce_y_fwd", "fused": "reduce_xy_fwd", "planes": "reduce_y_fwd"}
    assert all(kinds[mode] in b.band.calls for b in ranks)
    assert all(("reduce_xy_fwd" in b.band.calls) == (mode == "fused") for b in ranks)
