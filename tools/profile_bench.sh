#!/bin/bash
# Kernel-trace summary and counters of the default bench workload (run on the GPU box from the repo
# root): one --stats pass, then FETCH_SIZE, WRITE_SIZE and the matrix-pipe counters in passes of their own.
# usage: tools/profile_bench.sh [tag] [extra bench.py arguments...]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-final}; shift
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-kernel-clock --no-n68 --no-script-batch --no-full-variant --no-variants $*"
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --also "" $ARGS > $O/stats.log 2>&1 || exit 1
i=0
for c in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc$i -- python3 $R/bench.py --also "" $ARGS > $O/pmc$i.log 2>&1 || exit 1
done
python3 - $O <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2, 3):
    f = glob.glob(f"{root}/pmc{i}/*/*counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        if "mdnvae" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{root}/counters.csv", "w") as out:
    names = ["FETCH_SIZE", "WRITE_SIZE", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY",
             "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"]
    out.write("kernel,launches,read_MB(FETCH_SIZE x2),write_MB," + ",".join(names[2:]) + "\n")
    for k in sorted(acc):
        c = {n: (sum(v) / len(v) if v else 0.0) for n, v in acc[k].items()}
        rd = c.get("FETCH_SIZE", 0.0) * 1024 * 2 / 1e6
        wr = c.get("WRITE_SIZE", 0.0) * 1024 / 1e6
        line = f"\"{k}\",{len(acc[k].get('GRBM_GUI_ACTIVE', []))},{rd:.1f},{wr:.1f}," + ",".join(f"{c.get(n, 0.0):.0f}" for n in names[2:])
        out.write(line + "\n")
        print(line)
PY
