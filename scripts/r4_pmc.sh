#!/bin/bash
# round-4 counters: the PMC passes of the bench (defaults), FETCH_SIZE / WRITE_SIZE of the collapse's block orders, the full-size
# two-rank rehearsal of bench.py (8 pairs per step = the 8-GPU run's 4 pairs per rank) and k_coarse's phase stamps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/pmc.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log
o=gpurun_out/r4pmc; rm -rf $o; mkdir -p $o
for v in "STITCH_C4_SWIZZLE=0" "STITCH_C4_SWIZZLE=1 STITCH_C4_LOCKSTEP=1"; do
  n=$(echo $v | tr ' =' '__')
  for C in FETCH_SIZE WRITE_SIZE; do
    ( export $v; rocprofv3 --pmc $C --kernel-trace --output-format csv -d $o/${n}_$C -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-single --no-verify --streams 1 --batch 16 > $o/${n}_$C.json 2> $o/${n}_$C.err ) || echo "pass $n $C failed"
  done
done
STITCH_BENCH_BACKEND=gloo python bench.py --gpus 2 --pairs-per-step 8 --steps 10 --warmup 2 --no-cpu-baseline > $o/bench_2rank_rehearsal.json 2> $o/bench_2rank_rehearsal.err; tail -c 600 $o/bench_2rank_rehearsal.json
for c in "1081 527 384 512" "6144 4096 4096 4096"; do STITCH_COARSE_STAMP=1 python scripts/experiments/exp_single.py $c 5 pair f32 2>&1 | tail -3; done
ls $o | head -30
