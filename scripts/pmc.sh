#!/bin/bash
# PMC passes (separate runs, per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc
rm -rf $OUT && mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-single --no-verify --streams 1 --batch ${BATCH:-16} > $OUT/$name.json 2> $OUT/$name.err || echo "pass $name failed"
done
ls -R $OUT | head -40
