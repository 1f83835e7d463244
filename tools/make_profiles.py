#!/usr/bin/env python3
"""gpurun_out/ -> profiles/: the rocprofv3 summaries the bench's roofline object cites.

usage: python tools/make_profiles.py <tag> [round]   (after `bash tools/prof_pass.sh <tag>` and `bash tools/pmc_hbm.sh <tag>`
                                                 ran on the GPU box and gpurun merged their output back; round = r02)
  gpurun_out/prof_<tag>/**/_kernel_stats.csv                 -> profiles/<round>_kernel_stats_bench_default.csv
  gpurun_out/hbm_<tag>_{FETCH_SIZE,WRITE_SIZE,...}/**/*.csv  -> profiles/<round>_hbm_traffic.json (bytes per launch)
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts TCC_EA0_RDREQ x 64 B while the requests are 128 B, so it
is doubled (MI355X_MICROARCH.md, HBM section) and cross-checked against TCC_EA0_RDREQ_sum x 128 B.
"""
import collections, csv, glob, json, pathlib, shutil, sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
root = pathlib.Path(__file__).resolve().parent.parent
stats = sorted(glob.glob(str(root / f"gpurun_out/prof_{tag}/*/*_kernel_stats.csv")))
if stats:
    shutil.copy(stats[-1], root / f"profiles/{rnd}_kernel_stats_bench_default.csv")
    print("kernel stats <-", stats[-1])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(str(root / f"gpurun_out/hbm_{tag}_*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for short in ("trace_fwd_lds_kernel", "trace_bwd_lds_kernel"):
            if short in k:
                agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"config": {"heliostats": 1000, "rays_per_point": 100, "points_per_heliostat": 10000},
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/pmc_hbm.sh; FETCH_SIZE doubled (gfx950: "
                 "RDREQ x 64 B, requests are 128 B; cross-checked with TCC_EA0_RDREQ_sum x 128 B)"}
for k, v in agg.items():
    mean = lambda c: sum(v[c]) / len(v[c]) if v.get(c) else None
    rd = mean("FETCH_SIZE") * 1024 * 2 if mean("FETCH_SIZE") is not None else None
    wr = mean("WRITE_SIZE") * 1024 if mean("WRITE_SIZE") is not None else None
    out[k] = {"read_bytes": rd, "write_bytes": wr, "rdreq": mean("TCC_EA0_RDREQ_sum"), "atomic_req": mean("TCC_EA0_ATOMIC_sum"),
              "total_bytes": (rd or 0) + (wr or 0)}
    if mean("TCC_EA0_RDREQ_sum") is not None and rd:
        out[k]["rdreq_x128_over_read_bytes"] = mean("TCC_EA0_RDREQ_sum") * 128 / rd
if agg:
    json.dump(out, open(root / f"profiles/{rnd}_hbm_traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))
