#!/bin/bash
# PROBE: K-chunked summation order at small batches (VY_CONV_KSPLIT, conv_igemm.hip): latency of one frame at 608 / 416 with
# the launches that leave most CUs empty split into S independent chains (S = 4 for K >= 4096, 2 for K >= 1024), the last
# chunk's block adding the sums in chunk order.  Not bit-equal to the oracle's single chain (that is the question being priced).
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for size in 608 416; do
  for ks in 0 1 2 4; do
    echo "== size $size VY_CONV_KSPLIT=$ks"
    VY_CONV_KSPLIT=$ks python3 $R/tools/small_batch_latency.py --size $size --batches 1,2,4 2>/dev/null
    VY_CONV_KSPLIT=$ks python3 $R/tools/small_batch_latency.py --size $size --batches 1 --graph 2>/dev/null | sed 's/^/graph /'
  done
done
for size in 608 416; do for ks in 0 1; do
  VY_CONV_KSPLIT=$ks python3 $R/tools/layer_profile.py --size $size --batch 1 --out $R/gpurun_out/r06_ks${ks}_layers_${size}_b1.txt > /dev/null 2>&1
done; done
python3 $R/tools/probe/ksplit_numerics.py
