#!/bin/bash
# The bench lines kept under profiles/ for a round, in two gpurun calls (each below the 20-minute limit):
#   tools/final_profiles.sh a   -> default line, the driver's step counts, a rank's share of a 2 / 4 / 8-GPU strong-scaling job
#   tools/final_profiles.sh b   -> the other workloads (repeat-rich genome, stock costs, exact matching) and the ps_map timeline
set -o pipefail
out=gpurun_out/final; mkdir -p $out
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" > $out/$name.json 2> $out/$name.err && python - $out/$name.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], round(d["value"]), "reads/s,", round(d["ms_per_step"], 1), "ms/step, frac", round(d["roofline"]["frac"], 4), "kernel alone", round(d["roofline"]["avg_launch_ms"], 1), "e2e", d.get("t_e2e_s"), d.get("t_e2e_cold_s"), flush=True)
PY
}
if [ "$1" = a ]; then
  run bench_default &&
  run bench_steps20 --gpus 1 --steps 20 --warmup 5 &&
  run bench_share2 --reads 5000000 --steps 20 --warmup 4 --e2e 0 --cpu-sample 0 --drain 0 &&
  run bench_share4 --reads 2500000 --steps 20 --warmup 4 --e2e 0 --cpu-sample 0 --drain 0 &&
  run bench_share8 --reads 1250000 --steps 20 --warmup 4 --e2e 0 --cpu-sample 0 --drain 0
else
  run bench_repeats --genome-profile repeats --e2e 0 &&
  run bench_stock --penalty stock --e2e 0 &&
  run bench_exact_10m --workload exact --steps 5 --warmup 2 --e2e 0 &&
  run bench_exact_1m --workload exact --reads 1000000 --steps 5 --warmup 2 --e2e 0 &&
  timeout -k 10 400 python tools/e2e_time.py 10000000 3100 1 > $out/e2e_time.txt 2>&1; grep "ps_map\|ps_sam_to_bam" $out/e2e_time.txt | grep -v "^\[parasuite-hip\]   " | cut -c1-400
fi
