#!/bin/bash
# builds tuning variants of the chain decoder kernels: name:flags
set -e
cd "$(dirname "$0")/.."
mkdir -p tune_bin
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $flags -o tune_bin/chain_$name tools/chain_tune.cpp mdn-vae_amd/csrc/decoder.hip mdn-vae_amd/csrc/api.cpp 2>/dev/null &
done
wait
ls tune_bin
