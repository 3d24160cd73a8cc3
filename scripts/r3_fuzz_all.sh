#!/bin/bash
# every randomised parity campaign of the repo in one gpurun call; one progress line per campaign (seeds from $FUZZ_BASE)
cd $GRAFT_REPO_ROOT
o=gpurun_out/r3fuzzall; mkdir -p $o
B=${FUZZ_BASE:-7000}
run() { name=$1; shift; "$@" > $o/$name.log 2>&1; echo "$name: $(grep -E 'passed|failed|mismatch' $o/$name.log | tail -1)"; }
for i in 1 2 3; do
  run pairs_$i env FUZZ_GENERAL=1 timeout -k 10 400 python scripts/fuzz_pairs.py $((B+i)) 1500
  run bands_$i env FUZZ_BANDS=500 FUZZ_BANDS_WORLD=8 FUZZ_BANDS_SEED=$((B+i)) timeout -k 10 300 python -m pytest tests/test_gpu_band.py -m gpu -x -q -k random_configs
  run rows_$i env FUZZ_ROWS=800 FUZZ_ROWS_SEED=$((B+i)) timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k other_rows_random
  run host_$i env FUZZ_HOST=800 FUZZ_HOST_SEED=$((B+i)) timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k host_entry_points_random
  run steps_$i env FUZZ_STEPS=500 FUZZ_STEPS_SEED=$((B+i)) timeout -k 10 300 python -m pytest tests/test_gpu_golden.py -m gpu -x -q -k random_maps
  run batch_$i env FUZZ_BATCH=100 FUZZ_BATCH_SEED=$((B+i)) timeout -k 10 300 python -m pytest tests/test_gpu_benchpath.py -m gpu -x -q -k random_pair_counts
  run project_$i timeout -k 10 300 python scripts/fuzz_project.py $((B+i)) 400 2600
done
run big env FUZZ_GENERAL=1 FUZZ_BIG=1 timeout -k 10 600 python scripts/fuzz_pairs.py $((B+50)) 150
grep -h "MISMATCH\|^E  " $o/*.log | head -20
