#!/bin/bash
# PMC passes over one short bench run (counters only: no trace flags besides --kernel-trace).
# usage: tools/pmc.sh <tag> <bench args...>
tag=$1; shift
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_${tag}_$i -- python bench.py "$@" > gpurun_out/pmc_${tag}_$i.json 2> gpurun_out/pmc_${tag}_$i.err || exit 1
done
python tools/pmc_summary.py gpurun_out/pmc_${tag}_ > gpurun_out/pmc_${tag}_summary.txt
cat gpurun_out/pmc_${tag}_summary.txt
