#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 2 3; do STITCH_D7_STAMP_MODE=$m STITCH_D7_STAMP=1 timeout -k 10 120 python scripts/experiments/exp_single.py 6144 4096 4096 4096 3 pair f32 2>&1 | grep -v amdgpu.ids | grep -E "pair|chain 0" | cut -c1-260; done
