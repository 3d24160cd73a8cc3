#!/bin/bash
# randomised parity campaign (scripts/fuzz_pairs.py): general mode = any canvas size, every tuning switch drawn per case, pairs and
# dense-canvas blends, both pixel types, root and ex6 options; prints a progress line per chunk
cd $GRAFT_REPO_ROOT
o=gpurun_out/r3fuzz; mkdir -p $o
for seed in ${FUZZ_SEEDS:-101 102 103 104 105 106}; do
  FUZZ_GENERAL=1 timeout -k 10 900 python scripts/fuzz_pairs.py $seed ${FUZZ_N:-250} > $o/general_$seed.log 2>&1; tail -1 $o/general_$seed.log
done
FUZZ_GENERAL=1 FUZZ_BIG=1 timeout -k 10 900 python scripts/fuzz_pairs.py 201 ${FUZZ_NBIG:-40} > $o/big_201.log 2>&1; tail -1 $o/big_201.log
for seed in 301 302; do STITCH_WAVEFRONT=2 timeout -k 10 600 python scripts/fuzz_pairs.py $seed 150 > $o/fast_$seed.log 2>&1; tail -1 $o/fast_$seed.log; done
grep -h "MISMATCH" $o/*.log | head -20
