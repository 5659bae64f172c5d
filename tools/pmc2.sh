#!/bin/bash
# second PMC set: instruction cache, issue mix, TA/TCP stalls
tag=$1; shift
export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc2_${tag}_$i -- python bench.py "$@" > gpurun_out/pmc2_${tag}_$i.json 2> gpurun_out/pmc2_${tag}_$i.err || { tail -5 gpurun_out/pmc2_${tag}_$i.err; }
done
python tools/pmc_summary.py gpurun_out/pmc2_${tag}_ | head -40
