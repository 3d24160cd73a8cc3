#!/bin/bash
# round 4, GPU call 3: chain + mover sweeps (STITCH_MOVER) against the one-wavefront forms; the whole GPU suite (with the two-rank bench rehearsal and the literal drop-in executable) first
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log
[ $rc -ne 0 ] && { echo "pytest failed rc=$rc"; tail -60 $O/pytest.log; exit $rc; }
echo "== single pair, ms per call"
for c in "1081 527 384 512" "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  for v in "STITCH_MOVER=0" "STITCH_MOVER=1" "STITCH_MOVER=1 STITCH_Y1S=2"; do
    echo -n "[$v] "; env $v timeout -k 10 120 python scripts/experiments/exp_single.py $c 20 pair f32 2>&1 | tail -1
  done
done | tee $O/single.txt
echo "== timelines"
for v in "STITCH_MOVER=1"; do
for c in "1081 527 384 512" "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  set -- $c
  ( export $v; rocprofv3 --kernel-trace --output-format csv -d $O/tl_$1 -- python3 scripts/experiments/exp_single.py $c 5 pair f32 > $O/tl_$1.log 2>&1 )
  python scripts/experiments/timeline.py $O/tl_$1 > $O/tl_$1.txt; tail -1 $O/tl_$1.txt; rm -rf $O/tl_$1
done
done
