#!/bin/bash
# PMC passes over the chain-kernel harness (GPU box, repo root): where the waves' cycles go
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -s KILL 60 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_chain$i -- $R/tune_bin/chain_base 16384 87 1 > $R/gpurun_out/pmc_chain$i.log 2>&1 || exit 1
done
python3 - $R/gpurun_out/pmc_chain1 $R/gpurun_out/pmc_chain2 <<'PY'
import csv, glob, sys, collections
for d in sys.argv[1:]:
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "chain" in name: acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kind, c in acc.items():
        print(kind, {k: round(sum(v) / len(v)) for k, v in c.items()})
PY
