#!/bin/bash
# round 4, GPU call 7: chain with compile-time tile bounds + LDS reads one group / one chunk ahead; mover with one or two tiles in registers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { echo "pytest failed rc=$rc"; tail -60 $O/pytest.log; exit $rc; }
echo "== single pair, ms per call"
for c in "1081 527 384 512" "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  for v in "STITCH_MOVER=0" "STITCH_MOVER=1" "STITCH_MOVER=2"; do
    echo -n "[$v] "; env $v timeout -k 10 120 python scripts/experiments/exp_single.py $c 20 pair f32 2>&1 | tail -1 | sed 's/; paths.*//'
  done
done | tee $O/single.txt
echo "== timelines"
for v in "STITCH_MOVER=1" "STITCH_MOVER=2"; do
for c in "4421 2315 1536 2048"; do
  set -- $c
  n=$(echo "$v" | tr -c 'A-Za-z0-9=' '_')
  ( export $v; rocprofv3 --kernel-trace --output-format csv -d $O/tl_$1_$n -- python3 scripts/experiments/exp_single.py $c 5 pair f32 > $O/tl_$1_$n.log 2>&1 )
  python scripts/experiments/timeline.py $O/tl_$1_$n > $O/tl_$1_$n.txt; echo "[$v]"; grep "x_m\|y_m" $O/tl_$1_$n.txt | head -9; rm -rf $O/tl_$1_$n
done
done
