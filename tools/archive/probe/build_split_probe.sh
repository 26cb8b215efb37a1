#!/bin/bash
# builds tools/probe/conv_split_probe (optionally with extra -D flags: $1.., output name suffix via OUT=)
cd "$(dirname "$0")/../.."
OUT=${OUT:-tools/probe/conv_split_probe}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-inline-asm \
  -Wno-unused-function -Wno-pass-failed -I include "$@" -o $OUT tools/probe/conv_split_probe.hip videoyolo_amd/csrc/conv_small.hip
