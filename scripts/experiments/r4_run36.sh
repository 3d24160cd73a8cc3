#!/bin/bash
# collapse strip heights by workgroup count (STITCH_CROWS_WGS, default 8192) against round 3's rule (=0): lone pairs at three sizes, the batch, 4 pairs per step, batched 4421
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ak; rm -rf $O; mkdir -p $O
for c in "6144 4096 4096 4096" "4421 2315 2048 2048" "1081 527 384 512" "2048 1024 1408 1024"; do
  timeout -k 10 200 python scripts/experiments/exp_env_ab.py STITCH_CROWS_WGS 0 - $c 40 2>&1 | grep -v amdgpu.ids | grep -v "uint8"
done
timeout -k 10 200 python scripts/experiments/exp_env_ab.py STITCH_CROWS_WGS 0 2048 1081 527 384 512 40 2>&1 | grep float32
AB_ARGS="--no-single" bash scripts/experiments/ab_env.sh $O "old:STITCH_CROWS_WGS=0" "new:STITCH_X=0" "old2:STITCH_CROWS_WGS=0" "new2:STITCH_X=0"
AB_ARGS="--no-single --pairs-per-step 4" bash scripts/experiments/ab_env.sh $O "p4old:STITCH_CROWS_WGS=0" "p4new:STITCH_X=0" "p4old2:STITCH_CROWS_WGS=0" "p4new2:STITCH_X=0"
