#!/bin/bash
# Dev helper: build an engine variant that differs in the tier-1 kernel only (extra -D flags for wave_kernel.hip; the other
# objects are the main build's).  usage: bash tools/build_wave_variant.sh <name> [-DFOO=1 ...]  ->  <pkg>/libsparse_rx_<name>.so
set -e
root=$(cd $(dirname $0)/.. && pwd)
pkg=$root/optimized-sparse-retrieval-for-high-performance-rag-pipelines_amd
name=$1; shift
mkdir -p $pkg/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -fvisibility=hidden -std=c++17 -I$root/include -I$pkg/csrc "$@" -c -o $pkg/csrc/build/wave_kernel_$name.o $pkg/csrc/wave_kernel.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o $pkg/libsparse_rx_$name.so $pkg/csrc/build/wave_kernel_$name.o $pkg/csrc/build/sparse_rx.o $pkg/csrc/build/dense.o
echo built libsparse_rx_$name.so
