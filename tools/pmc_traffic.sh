#!/bin/bash
# HBM traffic of the kernels as the microarch guide prescribes: FETCH_SIZE and WRITE_SIZE in separate
# --pmc passes, counters only.  usage: tools/pmc_traffic.sh <tag> <bench args...>
tag=$1; shift
export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmct_${tag}_$i -- python bench.py "$@" > gpurun_out/pmct_${tag}_$i.json 2> gpurun_out/pmct_${tag}_$i.err || { tail -5 gpurun_out/pmct_${tag}_$i.err; exit 1; }
  echo "pass $i done" >> gpurun_out/pmct_${tag}_progress.txt
done
python tools/pmc_summary.py gpurun_out/pmct_${tag}_ > gpurun_out/pmct_${tag}_summary.txt
