#!/bin/bash
# (1) dec7 stage placement A/B (build with -DSTITCH_D7_PLACE=0 under csrc/ab), (2) STITCH_PITCH_PAD: bit equality + one pair,
# config 5, the batch headline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; rm -rf $O; mkdir -p $O
for c in "6144 4096 4096 4096" "4421 2315 2048 2048" "1081 527 384 512"; do
  for v in default noplace default noplace; do
    if [ "$v" = default ]; then unset STITCH_LIB; else export STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_d7_$v.so; fi
    echo "$v: $(timeout -k 10 120 python scripts/experiments/exp_single.py $c 40 pair f32 2>&1 | grep -v amdgpu.ids | tail -1)"
  done
done
unset STITCH_LIB
echo "== pitch pad: equality and one pair"
timeout -k 10 200 python scripts/experiments/exp_env_ab.py STITCH_PITCH_PAD - 64 6144 4096 4096 4096 20 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python scripts/experiments/exp_env_ab.py STITCH_PITCH_PAD - 192 4421 2315 2048 2048 20 2>&1 | grep -v amdgpu.ids
echo "== pitch pad: batch headline"
AB_ARGS="--no-single" bash scripts/experiments/ab_env.sh $O "pad0:STITCH_X=0" "pad64:STITCH_PITCH_PAD=64" "pad192:STITCH_PITCH_PAD=192" "pad0b:STITCH_X=0" "pad320:STITCH_PITCH_PAD=320"
echo "== pitch pad: config 5"
for pad in 0 64 192; do
  STITCH_PITCH_PAD=$pad timeout -k 10 300 python bench.py --frame 16384 --pairs-per-step 1 --batch 1 --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-single > $O/c5_$pad.json 2> $O/c5_$pad.err
  python - $O/c5_$pad.json $pad <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; k=d['kernels']
print('config5 pad', sys.argv[2], 'ms/pair', c['ms_per_pair_per_gpu'], 'verified', d['outputs_verified'], ' '.join(f"{n} {k[n]['ms_per_pair']:.3f}" for n in k if k[n]['ms_per_pair']>0.3))
PY
done
