#!/bin/bash
# builds tuning variants of the mixture likelihood kernels: name:flags
set -e
cd "$(dirname "$0")/.."
mkdir -p tune_bin
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude $flags -o tune_bin/mdn_$name tools/mdn_tune.cpp mdn-vae_amd/csrc/mdn.hip mdn-vae_amd/csrc/api.cpp 2>/dev/null &
done
wait
ls tune_bin
