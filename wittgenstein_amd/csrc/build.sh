#!/bin/bash
# Builds libwittgpu.so for gfx950 (MI355X) in-tree. hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-unused-result \
  -x hip engine.hip -x hip abi.cpp -x hip host_mirror.cpp -o ../libwittgpu.so "$@"
