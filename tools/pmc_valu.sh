#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_BUSY_CU_CYCLES"; do
  t2=$(echo $set | cut -d' ' -f2)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$t2 -- python3 $R/bench.py --heliostats 100 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${tag}_$t2.log 2>&1 || echo "fail $t2"
done
