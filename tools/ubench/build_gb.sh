#!/bin/bash
# build_gb.sh name [-D...]: one GEMM micro-benchmark binary per variant (gemm_bench.hip includes the production conv_igemm.hip)
n=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value "$@" gemm_bench.hip -o bin/gb_$n 2>&1 | grep -E "error|undefined"
true
