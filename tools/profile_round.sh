#!/bin/bash
# Collect the round's committed evidence on the GPU box: bench JSONs, rocprofv3 kernel stats (inference and
# training), HBM counters (separate FETCH_SIZE / WRITE_SIZE passes, no tracing combined), MFMA-busy counters,
# per-layer table.  usage: tools/profile_round.sh r01   -> gpurun_out/<tag>_*
tag=${1:-r01}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp
python3 $R/bench.py > $O/${tag}_infer608_b64_bench.json 2> /dev/null
python3 $R/bench.py --mode train > $O/${tag}_train416_b16_bench.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_p_inf -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-frames 0 > $O/${tag}_infer608_b64_bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_p_trn -- python3 $R/bench.py --mode train --steps 10 --warmup 3 > $O/${tag}_train416_b16_bench_under_rocprof.json 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_p_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-roofline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_p_write -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-roofline > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/${tag}_p_mfma -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-roofline > /dev/null 2>&1
python3 $R/tools/layer_profile.py --out $O/${tag}_layers_608_b64.txt > /dev/null 2>&1
ls $O | grep ${tag}_
