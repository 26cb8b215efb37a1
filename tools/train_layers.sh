#!/bin/bash
# per-layer matrix-core table of one training step, weight gradients on the main stream (clean durations)
# usage: tools/train_layers.sh tag [extra bench args]
tag=${1:-tl}; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
export VY_TRAIN_LABELS=$R/gpurun_out/${tag}_labels.txt
export VY_TRAIN_SIDE_STREAM=0
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_tl -- python3 $R/bench.py --mode train --steps 3 --warmup 2 --no-roofline "$@" > $R/gpurun_out/${tag}_tl_bench.json 2> /dev/null
python3 $R/tools/train_layers.py $(ls $R/gpurun_out/${tag}_tl/*/*kernel_trace.csv | head -1) $VY_TRAIN_LABELS > $R/gpurun_out/${tag}_layers.txt
tail -3 $R/gpurun_out/${tag}_layers.txt
