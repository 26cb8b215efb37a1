#!/bin/bash
# builds tools/probe/wino_abl_probe (the product's Winograd kernel) and its ablation variants wino_abl_probe_<bits>
cd "$(dirname "$0")/../.."
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-inline-asm -Wno-unused-function -Wno-pass-failed -DVY_WINO_BM128 -I include"
for abl in "" ${ABLS:-1 2 3 4 8 16 24 7}; do
  out=tools/probe/wino_abl_probe${abl:+_$abl}
  /opt/rocm/bin/hipcc $F ${abl:+-DVY_WINO_ABL=$abl} "$@" -o $out tools/probe/wino_abl_probe.hip videoyolo_amd/csrc/conv_small.hip &
done
wait
ls -la tools/probe/wino_abl_probe*
