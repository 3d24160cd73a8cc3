#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
for seed in 411 412 413 414; do
  FUZZ_GENERAL=1 timeout -k 10 900 python scripts/fuzz_pairs.py $seed 300 > $O/general_$seed.log 2>&1; tail -1 $O/general_$seed.log
done
FUZZ_GENERAL=1 FUZZ_BIG=1 timeout -k 10 900 python scripts/fuzz_pairs.py 511 50 > $O/big_511.log 2>&1; tail -1 $O/big_511.log
grep -h "MISMATCH" $O/*.log | head -20
bash scripts/r4_config5_profile.sh
