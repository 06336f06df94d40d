#!/bin/bash
# builds tools/bin/gemm_lab (stand-alone GEMM A/B harness) with the product library's flags; the ISA goes to tools/bin/*.s
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops \
    $LAB_DEFS tools/gemm_lab.hip -o tools/bin/${LAB_OUT:-gemm_lab} -ldl -save-temps=obj 2>&1 | grep -v "not a recognized feature\|ignoring feature" || true
