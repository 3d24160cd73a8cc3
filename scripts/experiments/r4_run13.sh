#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m; rm -rf $O; mkdir -p $O
for c in "1081 527 384 512" "2000 1100 900 1000" "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  timeout -k 10 120 python scripts/experiments/exp_xbym_ab.py $c 10 2>&1 | grep -v amdgpu.ids; rc=${PIPESTATUS[0]}; [ $rc -ne 0 ] && { echo "A/B failed rc=$rc"; exit $rc; }
done
FUZZ_GENERAL=1 timeout -k 10 400 python scripts/fuzz_pairs.py 501 150 > $O/fuzz_501.log 2>&1; rc=$?; tail -2 $O/fuzz_501.log; [ $rc -ne 0 ] && { grep -m5 MISMATCH $O/fuzz_501.log; exit $rc; }
for c in "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  set -- $c
  rocprofv3 --kernel-trace --output-format csv -d $O/tl_$1 -- python3 scripts/experiments/exp_single.py $c 5 pair f32 > $O/tl_$1.log 2>&1
  python scripts/experiments/timeline.py $O/tl_$1 > $O/tl_$1.txt; tail -1 $O/tl_$1.log; grep -E "xby_m|x_m<true|dec7|dispatches" $O/tl_$1.txt | head -8; rm -rf $O/tl_$1
done
for v in "" "STITCH_C4_SWIZZLE=2" "STITCH_CROWS_L0=64" "STITCH_C4_SWIZZLE=2 STITCH_CROWS_L0=64"; do
  n=$(echo "default $v" | tr ' =' '__')
  ( [ -n "$v" ] && export $v; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single > $O/bench_$n.json 2> $O/bench_$n.err )
  python - $O/bench_$n.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['value'], d['outputs_verified'], {k: v['ms_per_pair'] for k,v in d['kernels'].items() if k.startswith('collapse')})
PY
done
