"""Where does the source-fused level 0 start to pay?  n pairs per launch sequence, 1 or 4 sequences in flight, at two real
canvas sizes, source-fused (default for n >= 2) against materialised (STITCH_NO_SRC_FUSE=1).  ms per pair."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from computervisionimagestich2_amd import capi
dev = torch.device("cuda:0")
tdt = torch.float32
res = []
for (cw, ch, fw, fh) in [(1081, 527, 384, 512), (4421, 2315, 1536, 2048)]:
    F, M = capi.dev_synth(fw, fh, 1, tdt, dev), capi.dev_synth(cw - fw // 2, ch - 7, 2, tdt, dev)
    P = [1.0, 0.002, 1e-6, -(cw - fw - 3.0), -0.001, 1.0, 5e-7, -3.5]
    for n in (1, 2, 4, 8, 16):
        for lanes_n in (1, 4):
            row = {"canvas": [cw, ch], "pairs_per_sequence": n, "sequences_in_flight": lanes_n}
            for label, env in (("fused", {"STITCH_SINGLE_FAST": "1"}), ("materialised", {"STITCH_NO_SRC_FUSE": "1"})):
                for k in ("STITCH_SINGLE_FAST", "STITCH_NO_SRC_FUSE"):
                    os.environ.pop(k, None)
                os.environ.update(env)
                lanes = [(capi.Plan(cw, ch, max_pairs=n), torch.cuda.Stream(device=dev), [torch.empty((3, ch, cw), dtype=tdt, device=dev) for _ in range(n)]) for _ in range(lanes_n)]

                def go():
                    for plan, st, outs in lanes:
                        with torch.cuda.stream(st):
                            plan.pairs([(F, P, -0.25, -1.5, M, 0, -2, o) for o in outs])
                for _ in range(3):
                    go()
                torch.cuda.synchronize()
                t = time.perf_counter()
                R = 10
                for _ in range(R):
                    go()
                torch.cuda.synchronize()
                row[label] = round((time.perf_counter() - t) / R / (n * lanes_n) * 1e3, 4)
                for plan, _, _ in lanes:
                    plan.close()
                del lanes
            row["fused_over_materialised"] = round(row["fused"] / row["materialised"], 3)
            res.append(row)
            print(row, file=sys.stderr, flush=True)
print(json.dumps(res, indent=1))
