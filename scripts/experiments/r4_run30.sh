#!/bin/bash
# soak: N fresh bench processes (default sizes, 6 steps), looking for hand-off time-outs / failed output checks / slow warm-ups
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ae; rm -rf $O; mkdir -p $O
for i in $(seq 1 ${SOAK_N:-40}); do
  t0=$(date +%s.%N)
  timeout -k 10 180 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-single > $O/b_$i.json 2> $O/b_$i.err; rc=$?
  t1=$(date +%s.%N)
  python - $O/b_$i.json $i $rc 0 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
    print('run', sys.argv[2], 'rc', sys.argv[3], 'wall', sys.argv[4], 'value', d['value'], 'verified', d['outputs_verified'], 'warmup faults', c.get('warmup_handoff_faults'), flush=True)
except Exception as e:
    print('run', sys.argv[2], 'rc', sys.argv[3], 'no line', e, flush=True)
PY
done
grep -l "CHECK FAILED\|time-out" $O/*.err | head
