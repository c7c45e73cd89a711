#!/bin/bash
# PMC passes over tools/outer_probe.py (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_outer$i -- python $R/tools/outer_probe.py 16384 87 > $R/gpurun_out/pmc_outer$i.log 2>&1 || exit 1
done
python - $R/gpurun_out/pmc_outer1 $R/gpurun_out/pmc_outer2 <<'PY'
import csv, glob, sys, collections
for d in sys.argv[1:]:
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "mdn_outer" not in name: continue
        acc["sym" if "true" in name or "Lb1" in name else "gen"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kind, c in acc.items():
        print(kind, {k: round(sum(v) / len(v)) for k, v in c.items()})
PY
