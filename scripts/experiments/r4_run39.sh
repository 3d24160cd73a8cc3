#!/bin/bash
# bench.py: the lanes out of phase (--stagger-ms: host sleep in front of the first sequence of lanes 1..3 of every timed region)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4an; rm -rf $O; mkdir -p $O
for v in 0 8 16 0 4 12 24 0 16; do
  timeout -k 10 240 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-single --no-kernel-events --stagger-ms $v > $O/b_$v.json 2> $O/b_$v.err
  python - $O/b_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print('stagger', sys.argv[2], d['value'], 'verified', d['outputs_verified'], 'ms/step', d['ms_per_step'], 'single batch', c['single_batch_ms'])
PY
done
