#!/bin/bash
# Collect the round's committed evidence on the GPU box: bench JSONs, rocprofv3 kernel stats, HBM counters (separate
# FETCH_SIZE / WRITE_SIZE passes, counters only — never combined with tracing) and MFMA-busy counters for BOTH the
# inference (608x608 batch 64) and the training (416x416 batch 16) workload, plus the per-layer tables.
# usage: tools/profile_round.sh r02   -> gpurun_out/<tag>_*   then: python tools/summarize_profiles.py r02 r02
tag=${1:-r02}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp
python3 $R/bench.py > $O/${tag}_infer608_b64_bench.json 2> /dev/null
python3 $R/bench.py --mode train > $O/${tag}_train416_b16_bench.json 2> /dev/null
Q="--cpu-frames 0 --no-roofline --no-pmc --no-latency --no-train-legs --no-split-leg --no-host-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_p_inf -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-pmc --no-latency --no-train-legs --no-split-leg --no-host-legs > $O/${tag}_infer608_b64_bench_under_rocprof.json 2> /dev/null
# the opt-in split-fp32 conv mode is FROZEN since round 5: its passes run only with VY_PROFILE_SPLIT=1;
# the one `also_infer608_split` leg of the default bench line stays
SPL=${VY_PROFILE_SPLIT:-0}
[ $SPL = 1 ] && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_p_spl -- python3 $R/bench.py --conv-mode split_bf16x3 --steps 5 --warmup 2 --cpu-frames 0 --no-pmc --no-latency --no-train-legs --no-host-legs > $O/${tag}_infer608_b64_split_bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_p_trn -- python3 $R/bench.py --mode train --steps 10 --warmup 3 --no-pmc > $O/${tag}_train416_b16_bench_under_rocprof.json 2> /dev/null
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
for m in inf trn $([ $SPL = 1 ] && echo spl); do
  if [ $m = inf ]; then A="--steps 1 --warmup 1 $Q"; elif [ $m = spl ]; then A="--conv-mode split_bf16x3 --steps 1 --warmup 1 $Q"; else A="--mode train --steps 1 --warmup 1 $Q"; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_${m}_fetch -- python3 $R/bench.py $A > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_${m}_write -- python3 $R/bench.py $A > /dev/null 2>&1
  rocprofv3 --pmc $MF --output-format csv -d $O/${tag}_${m}_mfma -- python3 $R/bench.py $A > /dev/null 2>&1
done
python3 $R/tools/layer_profile.py --out $O/${tag}_layers_608_b64.txt > /dev/null 2>&1
[ $SPL = 1 ] && python3 $R/tools/layer_profile.py --conv-mode split_bf16x3 --out $O/${tag}_layers_608_b64_split.txt > /dev/null 2>&1
python3 $R/tools/layer_profile.py --batch 1 --out $O/${tag}_layers_608_b1.txt > /dev/null 2>&1
[ $SPL = 1 ] && python3 $R/tools/layer_profile.py --batch 1 --conv-mode split_bf16x3 --out $O/${tag}_layers_608_b1_split.txt > /dev/null 2>&1
python3 $R/tools/layer_profile.py --size 416 --batch 1 --out $O/${tag}_layers_416_b1.txt > /dev/null 2>&1
[ $SPL = 1 ] && bash $R/tools/ab_split_train.sh > $O/${tag}_ab_split_train.txt 2>&1
[ $SPL = 1 ] && bash $R/tools/ab_split_train_prof.sh > $O/${tag}_train416_b16_split_kernel_totals.txt 2>&1
python3 $R/tools/nms_latency.py > $O/${tag}_nms_latency.txt 2>&1
python3 $R/tools/small_batch_latency.py > $O/${tag}_small_batch_latency.txt 2>&1
python3 $R/tools/small_batch_latency.py --size 416 > $O/${tag}_small_batch_latency_416.txt 2>&1
[ $SPL = 1 ] && python3 $R/tools/small_batch_latency.py --conv-mode split_bf16x3 > $O/${tag}_small_batch_latency_split.txt 2>&1
$R/tools/train_layers.sh ${tag}
ls $O | grep ${tag}_
