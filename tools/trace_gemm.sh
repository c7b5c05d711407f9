#!/bin/bash
# builds the trace variant of the GEMM next to the tool (never shipped) and prints the timeline
set -e
cd "$(dirname "$0")/../flex-dm_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMFP_GEMM_TRACE -shared gemm.hip gemm_fp8.hip block_fused.hip attention.hip layernorm.hip embed.hip loss.hip optim.hip masking.hip debug.hip error.cpp -o ../../tools/libmfp_trace.so
cd ../.. && python tools/${TRACE_TOOL:-trace_gemm.py}
