"""Sum rocprofv3 --pmc counter rows per kernel (csv output of tools/pmc.sh)."""
import csv
import glob
import sys
from collections import defaultdict

pref = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for f in sorted(glob.glob(pref + "*/**/*counter_collection.csv", recursive=True)):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (k, row["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    if "ps::" not in k:
        continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-28s %.4g" % (c, v))
