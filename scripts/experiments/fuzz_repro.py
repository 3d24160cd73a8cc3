"""Re-runs ONE case of scripts/fuzz_pairs.py (same seed, same draws) under extra environment overrides applied right before the
plan is created, to find which kernel a mismatch belongs to.  usage: fuzz_repro.py seed case [KEY=VAL ...] (FUZZ_GENERAL etc. as for the campaign)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fuzz_pairs
from oracle_lib import Oracle
from computervisionimagestich2_amd import capi
seed, case = int(sys.argv[1]), int(sys.argv[2])
sets = [a for a in sys.argv[3:]]
variants = [[]] + [[s] for s in sets]
O = Oracle(); dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
for c in range(case):
    fuzz_pairs._case(c, rng, O, capi, torch, dev, 0, 0, False)
state = rng.bit_generator.state
RealPlan = capi.Plan
for ov in variants:
    rng.bit_generator.state = state
    def P(*a, **k):
        for kv in ov:
            key, val = kv.split("=")
            if val == "": os.environ.pop(key, None)
            else: os.environ[key] = val
        return RealPlan(*a, **k)
    capi.Plan = P
    bad, done = fuzz_pairs._case(case, rng, O, capi, torch, dev, 0, 0, True)
    print("override", ov, "-> mismatches", bad, "compared", done, flush=True)
    for kv in ov: os.environ.pop(kv.split("=")[0], None)
