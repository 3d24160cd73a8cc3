#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4al; rm -rf $O; mkdir -p $O
AB_ARGS="--no-single --pairs-per-step 4 --steps 20" bash scripts/experiments/ab_env.sh $O "p4old:STITCH_CROWS_WGS=0" "p4new:STITCH_X=0" "p4old2:STITCH_CROWS_WGS=0" "p4new2:STITCH_X=0" "p4old3:STITCH_CROWS_WGS=0" "p4new3:STITCH_X=0"
for c in "6144 4096 4096 4096" "4421 2315 2048 2048"; do
  timeout -k 10 200 python scripts/experiments/exp_env_ab.py STITCH_CROWS_WGS 0 - $c 40 2>&1 | grep -v amdgpu.ids | grep "float32 STITCH"
done
