#!/bin/bash
# is it the rows' distance?  the 16-pair batch at config 2 with the row pitch of config 5 (STITCH_PITCH_PAD=18432: 24576 floats per row
# instead of 6144), one sequence in flight
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ad; rm -rf $O; mkdir -p $O
for pad in 0 18432 6144 0; do
  STITCH_PITCH_PAD=$pad timeout -k 10 300 python bench.py --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-single --pairs-per-step 16 > $O/b_$pad.json 2> $O/b_$pad.err
  python - $O/b_$pad.json $pad <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; k=d['kernels']
print('pitch pad', sys.argv[2], d['value'], 'verified', d['outputs_verified'], 'one-seq ms/pair', c['one_sequence_in_flight_ms_per_pair'], ' '.join(f"{n} {k[n]['ms_per_pair']:.3f}" for n in k if k[n]['ms_per_pair']>0.02))
PY
done
