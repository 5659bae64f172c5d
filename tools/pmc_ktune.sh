#!/bin/bash
# SQ instruction-mix counters of the search kernels over tools/ktune.py (counters only, one group per pass).
# usage: tools/pmc_ktune.sh <tag> <ktune args...>
tag=$1; shift
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmck_${tag}_$i -- python3 tools/ktune.py "$@" > gpurun_out/pmck_${tag}_$i.json 2> gpurun_out/pmck_${tag}_$i.err || { tail -5 gpurun_out/pmck_${tag}_$i.err; exit 1; }
  echo "pass $i done" >> gpurun_out/pmck_${tag}_progress.txt
done
python tools/pmc_summary.py gpurun_out/pmck_${tag}_ > gpurun_out/pmck_${tag}_summary.txt
rm -rf gpurun_out/pmck_${tag}_[0-9]*/      # the raw per-dispatch csv is tens of MB per pass: only the summary travels back
