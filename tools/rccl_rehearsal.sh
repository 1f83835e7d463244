# one-rank RCCL rehearsal of a 125-heliostat share (what a rank of an 8-GPU run does), with the kernel sequence of a step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ARTIST_AMD_COLLECTIVES_AT_WORLD_1=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
rm -rf $R/gpurun_out/seq
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/seq -- python3 $R/bench.py --heliostats ${1:-125} --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/rehearsal.log 2>&1
tail -1 $R/gpurun_out/rehearsal.log | cut -c1-220
python3 $R/tools/step_sequence.py $R/gpurun_out/seq
rm -rf $R/gpurun_out/seq
