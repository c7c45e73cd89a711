#!/bin/bash
# Kernel-trace summary of one bench workload (run on the GPU box from the repo root).
# usage: tools/prof_stats.sh tag [bench.py arguments...]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --also "" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-clock --no-n68 --no-script-batch --no-full-variant --no-variants "$@" > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
f=$(ls $O/stats/*/*kernel_stats.csv | head -1)
cp $f $O/kernel_stats.csv
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over the profiled steps")
for r in rows[:32]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>5} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f} %")
PY
