#!/bin/bash
# SQ counters of the projection kernels (exp_project.py), one pass per group; output gpurun_out/pmc_pj/<group>/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_pj
rm -rf $OUT && mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/g$i -- python3 scripts/experiments/exp_project.py > $OUT/g$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/g$i.log)"
done
ls -R $OUT | head -30
