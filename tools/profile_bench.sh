#!/bin/bash
# Kernel-trace summary and HBM-traffic counters of the default bench workload (run on the GPU box
# from the repo root): one --stats pass, then FETCH_SIZE and WRITE_SIZE in passes of their own.
R=${GRAFT_REPO_ROOT:-$PWD}
ARGS="--also  --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-clock --no-n68 --no-script-batch --no-full-variant"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python3 $R/bench.py --also "" --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-clock --no-n68 --no-script-batch --no-full-variant > $R/gpurun_out/prof_final.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_final_$c -- python3 $R/bench.py --also "" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-clock --no-n68 --no-script-batch --no-full-variant > $R/gpurun_out/pmc_final_$c.log 2>&1 || exit 1
done
python3 - $R/gpurun_out <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{root}/pmc_final_{c}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mdnvae" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    tot[c] = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(tot["FETCH_SIZE"]):
    rd = tot["FETCH_SIZE"][k] * 1024 * 2 / 1e6
    wr = tot["WRITE_SIZE"].get(k, 0.0) * 1024 / 1e6
    print(f"{k:70s} read {rd:9.1f} MB  write {wr:9.1f} MB  total {rd + wr:9.1f} MB")
PY
