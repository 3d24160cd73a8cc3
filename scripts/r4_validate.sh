#!/bin/bash
# round 4 validation in one GPU call: the whole GPU suite, a randomised parity campaign with every switch (scripts/fuzz_pairs.py), the bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4v; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { echo "pytest failed rc=$rc"; tail -80 $O/pytest.log; exit $rc; }
for seed in ${FUZZ_SEEDS:-401 402 403}; do
  FUZZ_GENERAL=1 timeout -k 10 900 python scripts/fuzz_pairs.py $seed ${FUZZ_N:-250} > $O/general_$seed.log 2>&1; tail -1 $O/general_$seed.log
done
FUZZ_GENERAL=1 FUZZ_BIG=1 timeout -k 10 900 python scripts/fuzz_pairs.py 501 ${FUZZ_NBIG:-40} > $O/big_501.log 2>&1; tail -1 $O/big_501.log
grep -h "MISMATCH" $O/*.log | head -20
timeout -k 10 600 python bench.py --verbose > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
print(d["value"], d["unit"], "verified", d["outputs_verified"], "| one seq", c["one_sequence_in_flight_ms_per_pair"], "| single pair", c["single_pair_in_flight_ms"], "| single batch ms", c["single_batch_ms"], "| roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "| cpu", d.get("cpu_baseline", {}).get("value"))
print({k: v["ms_per_pair"] for k, v in d["kernels"].items()})
PY
