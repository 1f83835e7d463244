#!/usr/bin/env python3
"""gpurun_out/final_* (tools/final_pass.sh on the GPU box) -> profiles/<round>_* : the summaries the round's DESIGN.md cites.
usage: python tools/collect_final.py r04"""
import pathlib, shutil, sys
rnd = sys.argv[1]
root = pathlib.Path(__file__).resolve().parent.parent
src, dst = root / "gpurun_out", root / "profiles"
names = {"final_bench.json": "bench_line.json", "final_bench_h125.json": "bench_h125.json", "final_bench_config4.json": "bench_config4.json",
         "final_config_bench.jsonl": "config_bench.jsonl", "final_bench_2rank_gloo.log": "bench_2rank_gloo.log",
         "final_bench_rccl_world1.log": "bench_rccl_world1.log", "final_blocking_bench.json": "blocking_bench.json",
         "final_cylinder_bench.json": "cylinder_bench.json", "final_flux_bench_1000.json": "flux_bench.json",
         "final_flux_bench_125.json": "flux_bench_125.json", "final_flux_ab_r03.txt": "flux_ab_vs_r03.txt",
         "final_nurbs_bench.json": "nurbs_bench.json", "final_nurbs_mfma_bench.json": "nurbs_mfma_bench.json",
         "final_pipeline_probe.txt": "pipeline_probe.txt", "final_kstats.txt": "kernel_stats_top.txt",
         "final_kstats_h125.txt": "kernel_stats_top_h125.txt", "final_gap_h125.txt": "gap_report_h125.txt", "final_gap.txt": "gap_report.txt",
         "final_blocking_bench_dense.json": "blocking_bench_dense.json", "final_adam_bench.json": "adam_bench.json",
         "final_per_target_bench.json": "per_target_bench.json", "final_parity_margins.txt": "parity_margins.txt",
         "final_pmc_summary.txt": "pmc_summary.txt", "pmc_finallds_summary.txt": "pmc_lds_conflicts.txt"}
for a, b in names.items():
    if (src / a).exists() and (src / a).stat().st_size > 0:
        shutil.copy(src / a, dst / f"{rnd}_{b}")
        print(f"{a} -> profiles/{rnd}_{b}")
    else:
        print(f"MISSING {a}")
