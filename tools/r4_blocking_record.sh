#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/blocking_bench.py 2>/dev/null | tail -1 > gpurun_out/r04_blocking_bench.json
timeout -k 10 400 python tools/blocking_bench.py --dense 2>/dev/null | tail -1 > gpurun_out/r04_blocking_bench_dense.json
cat gpurun_out/r04_blocking_bench.json gpurun_out/r04_blocking_bench_dense.json | cut -c1-600
