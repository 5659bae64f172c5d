#!/bin/bash
# HBM traffic of the search kernel as the microarch guide prescribes: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (counters only), L2 hit / miss in a third, over tools/ktune.py; and the calibration of FETCH_SIZE for this access pattern
# (per-lane 16-byte loads of random 64-byte blocks) on tools/microbench_gather2, whose byte count is known.
# usage: tools/pmc_traffic_ktune.sh <tag> <ktune args...>
tag=$1; shift
export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmct_${tag}_$i -- python3 tools/ktune.py "$@" > gpurun_out/pmct_${tag}_$i.json 2> gpurun_out/pmct_${tag}_$i.err || { echo "pass $i failed" >> gpurun_out/pmct_${tag}_progress.txt; tail -5 gpurun_out/pmct_${tag}_$i.err; exit 1; }
  echo "pass $i done" >> gpurun_out/pmct_${tag}_progress.txt
done
python tools/pmc_summary.py gpurun_out/pmct_${tag}_ > gpurun_out/pmct_${tag}_summary.txt
rm -rf gpurun_out/pmct_${tag}_[0-9]*/
# calibration: k_lane reads steps x lanes random 64-byte blocks of a table far larger than the caches
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcg_${tag} -- tools/microbench_gather2 > gpurun_out/pmcg_${tag}.txt 2> gpurun_out/pmcg_${tag}.err || { echo "gather pass failed"; exit 1; }
python - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
rows = []
for f in glob.glob("gpurun_out/pmcg_%s/**/*counter_collection.csv" % tag, recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
out = open("gpurun_out/pmcg_%s_summary.txt" % tag, "w")
out.write("# FETCH_SIZE (KB) per dispatch of tools/microbench_gather2; every dispatch reads 768 workgroups x 256 lanes x 300 steps x 64 B = 3774.9 MB of random 64-byte blocks\n")
for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
    kb = float(r["Counter_Value"])
    out.write("%s dispatch %s: FETCH_SIZE %.4g KB = %.3f x the 3774.9 MB asked for\n" % (r["Kernel_Name"].split("(")[0], r["Dispatch_Id"], kb, kb * 1024 / (768 * 256 * 300 * 64.0)))
PY
rm -rf gpurun_out/pmcg_${tag}/
