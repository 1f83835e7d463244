cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --heliostats 100 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1 || echo "fail $tag"
done
ls $R/gpurun_out/pmc_*/*/ | head
