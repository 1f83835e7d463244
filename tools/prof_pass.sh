#!/bin/bash
# usage (on the GPU box): bash tools/prof_pass.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/ + pmc_<tag>_*/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
# the SAME command the driver runs (default arguments), so that the average kernel durations can be compared
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py "$@" > $R/gpurun_out/prof_$tag.log 2>&1 || echo "kernel-trace failed"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE"; do
  t2=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$t2 -- python3 $R/bench.py --heliostats 100 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${tag}_$t2.log 2>&1 || echo "fail $t2"
done
