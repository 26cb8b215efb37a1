#!/bin/bash
# kernel-trace stats of the training bench (per-kernel time; --no-pmc: bench.py must not start its own
# rocprofv3 --pmc children from inside a profiled process) -> gpurun_out/<tag>_trn/ ; usage: tools/train_trace.sh tag
tag=${1:-t}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trn -- python3 $R/bench.py --mode train --steps 10 --warmup 3 --no-pmc > $R/gpurun_out/${tag}_trn_bench.json 2> /dev/null
f=$(ls $R/gpurun_out/${tag}_trn/*/*kernel_stats.csv | head -1)
cut -c1-150 $f | head -30
