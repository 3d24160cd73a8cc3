#!/bin/bash
# collapse, levels >= 1: the mask through LDS (STITCH_C4_LOCKSTEP=2): FETCH_SIZE, equality, batch A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4aj; rm -rf $O; mkdir -p $O
for v in 0 2; do
  ( export STITCH_C4_LOCKSTEP=$v; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f$v -- python3 scripts/experiments/exp_collapse_fetch.py default > $O/f$v.log 2>&1 )
  echo "== STITCH_C4_LOCKSTEP=$v"; python scripts/experiments/fetch_report.py $O/f$v 8 "k_collapse4<float, false"; rm -rf $O/f$v
done
timeout -k 10 200 python scripts/experiments/exp_env_ab.py STITCH_C4_LOCKSTEP - 2 6144 4096 4096 4096 20 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python scripts/experiments/exp_env_ab.py STITCH_C4_LOCKSTEP - 2 4421 2315 2048 2048 20 2>&1 | grep -v amdgpu.ids
AB_ARGS="--no-single" bash scripts/experiments/ab_env.sh $O "off:STITCH_C4_LOCKSTEP=0" "lds:STITCH_C4_LOCKSTEP=2" "off2:STITCH_C4_LOCKSTEP=0" "lds2:STITCH_C4_LOCKSTEP=2"
