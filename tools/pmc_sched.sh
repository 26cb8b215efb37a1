#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for v in 0 1; do
  VY_CONV_SCHED=$v rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_s$v -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-roofline > /dev/null 2>&1
  VY_CONV_SCHED=$v rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --output-format csv -d $R/gpurun_out/pmc_t$v -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-roofline > /dev/null 2>&1
done
