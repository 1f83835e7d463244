#!/bin/bash
# LDS replay split of the trace kernels: SQ_LDS_ADDR_CONFLICT (same-address collisions) beside SQ_LDS_BANK_CONFLICT
# (all conflict cycles).  GPU box only.  usage: bash tools/pmc_lds.sh <tag>   -> gpurun_out/pmc_<tag>_*/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN"; do
  t2=$(echo $set | cut -d' ' -f2)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$t2 -- python3 $R/bench.py --heliostats 100 --steps 2 --warmup 1 --no-cpu-baseline --no-check > $R/gpurun_out/pmc_${tag}_$t2.log 2>&1 || echo "fail $t2"
done
cd $R && python3 tools/pmc_summary.py $tag > gpurun_out/pmc_${tag}_summary.txt 2>&1
