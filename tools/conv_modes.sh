#!/bin/bash
# The conv-encoder workload in each multiplication mode of its 3x3 convolutions (GPU box, repo root).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2
mkdir -p $O
for p in fp32_mfma fp32 bf16x2 bf16; do
  timeout -k 10 200 python $R/bench.py --model cnn_gcn --precision $p --also "" --steps 4 --no-cpu-baseline > $O/bench_conv_$p.json 2> $O/bench_conv_$p.err || exit 1
done
