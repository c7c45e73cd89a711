#!/bin/bash
# builds tuning variants of the loss kernels: name:flags
set -e
cd "$(dirname "$0")/.."
mkdir -p tune_bin
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $flags -o tune_bin/loss_$name tools/loss_tune.cpp mdn-vae_amd/csrc/loss.hip mdn-vae_amd/csrc/mdn.hip mdn-vae_amd/csrc/api.cpp 2>/dev/null &
done
wait
ls tune_bin
