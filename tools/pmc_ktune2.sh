#!/bin/bash
# Where the search kernel's waves wait: issue stalls, instruction fetch, vector-memory FIFOs, LDS conflicts, L1 TLB.
# Counters only, one group per pass, each pass under its own timeout.
# About the round-1 hang (tools/pmc2.sh, third group): that group was TA_BUSY / TA_ADDR_STALLED_BY_TC / TA_DATA_STALLED_BY_TC /
# TA_FLAT_*_WAVEFRONTS plus TCP_PENDING_STALL_CYCLES; its pass never wrote a counter file (the two SQ groups before it did).  The one
# counter it shares with this script, TCP_PENDING_STALL_CYCLES_sum, has since run to completion here together with the other TCP_*
# counters below (round 2: all four passes "done" in the progress file, profiles/r02_wait_counters_1gbp_4m.txt), so the TA_* block
# counters are what that pass hung on.  No TA_* counter is collected by any script that is still in use; pmc2.sh keeps its
# warning and is not to be run.
# usage: tools/pmc_ktune2.sh <tag> <ktune args...>
tag=$1; shift
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmcw_${tag}_$i -- python3 tools/ktune.py "$@" > gpurun_out/pmcw_${tag}_$i.json 2> gpurun_out/pmcw_${tag}_$i.err || { echo "pass $i failed rc=$?" >> gpurun_out/pmcw_${tag}_progress.txt; tail -5 gpurun_out/pmcw_${tag}_$i.err; exit 1; }
  echo "pass $i done" >> gpurun_out/pmcw_${tag}_progress.txt
done
python tools/pmc_summary.py gpurun_out/pmcw_${tag}_ > gpurun_out/pmcw_${tag}_summary.txt
rm -rf gpurun_out/pmcw_${tag}_[0-9]*/      # the raw per-dispatch csv is tens of MB per pass: only the summary travels back
