#!/usr/bin/env python3
"""bench.py - BASELINE.json's headline metric on MI355X.

metric   : traced rays/s fwd+bwd, 1000-heliostat NURBS field (rays = H*R*P per step)
workload : the metric config of BASELINE.md section 4 - H=1000 heliostats x R=100 rays/point x
           P=10^4 points (4 facets x 50x50), 10x10 degree-3 NURBS control nets, one 8 m x 8 m planar
           receiver, 256x256 bitmap, Gaussian sun; synthetic data, random (seeded) surface noise.
step     : one surface-reconstruction epoch over the whole field:
           NURBS points+normals (HIP) -> alignment (HIP) -> trace_rays (HIP) -> per-target sum ->
           [N>1: RCCL all_reduce of the [T,256,256] flux] -> crop around the centre of mass (HIP) -> PixelLoss
           vs fixed measured bitmaps (HIP) -> backward (loss, crop, trace_bwd, alignment, nurbs_bwd: all HIP)
           -> [N>1: RCCL exchange of the control-point gradients (surface_reconstructor.py:767-777; the shards are
           row-disjoint, so the reference's all_reduce(SUM) is issued as an all-gather)] -> Adam step (:779).
           Inputs (control points, orientations, distortions) are resident in HBM before timing.
N GPUs   : heliostats are sharded over ranks exactly like RestrictedDistributedSampler
           (heliostat i -> rank i mod N); total work is fixed  => "scaling": "strong".

Every rank draws ITS rows of the field-wide seed-7 sample, so an N-rank run traces exactly the rays of the 1-rank run; the line
carries a checksum of the reduced flux ("sharding_check") and, at N > 1, rank 0 re-traces the whole field alone and the run fails
when the reduced flux differs from it by more than 2e-6.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
HBM_PEAK_FILE = ROOT / "profiles" / "r02_hbm_peak.json"       # tools/hbm_peak.hip on this pool's MI355X (float4 copy / read / write)
HBM_TRAFFIC_FILE = ROOT / "profiles" / "r04_hbm_traffic.json"  # rocprofv3 --pmc passes of this command (tools/pmc_hbm.sh)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--heliostats", type=int, default=1000, help="total heliostats in the field (all ranks)")
    ap.add_argument("--rays", type=int, default=100, help="rays per surface point (Sun.number_of_rays)")
    ap.add_argument("--n-eval", type=int, default=50, help="evaluation points per facet per direction")
    ap.add_argument("--n-cp", type=int, default=10, help="NURBS control points per facet per direction")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle comparison of the timed rays")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def cpu_baseline(args, cp_host, canting, transl, uv, orientation, incident, planar, seconds):
    """Oracle (C restatement of the reference, kind='port') on the host cores: fwd+bwd of the same
    workload on a bounded sample of heliostats.  Checker code used as a yardstick only."""
    import numpy as np

    import oracle
    # the GPU box gives a one-GPU job a 16-core share of the host; stay inside it
    threads = max(1, min(oracle.max_threads(), int(os.environ.get("ARTIST_CPU_THREADS", "16"))))
    H1 = cp_host.shape[0]
    R, P = args.rays, 4 * args.n_eval * args.n_eval
    rng = np.random.default_rng(7)

    def run(h_count):
        sel = np.arange(h_count) % H1
        cp = cp_host[sel]
        pts, nrm = oracle.nurbs_fwd(cp, np.broadcast_to(uv, (h_count,) + uv.shape[1:]), [3, 3], canting[sel], transl[sel])
        ori = orientation[sel]
        ap = (pts.reshape(h_count, P, 4) @ ori.transpose(0, 2, 1)).astype(np.float32)
        an = (nrm.reshape(h_count, P, 4) @ ori.transpose(0, 2, 1)).astype(np.float32)
        both = (rng.standard_normal((h_count, R, P, 2), dtype=np.float32) * np.float32(np.sqrt(4.3681e-06)))
        du, de = both[..., 0], both[..., 1]
        tix = np.zeros(h_count, dtype=np.int32)
        t0 = time.perf_counter()
        pts, nrm = oracle.nurbs_fwd(cp, np.broadcast_to(uv, (h_count,) + uv.shape[1:]), [3, 3], canting[sel], transl[sel])
        flux, _ = oracle.trace_fwd(ap, an, incident[sel], du, de, tix, *planar, (256, 256), nthreads=threads)
        g = 2.0 * (flux - flux.mean()) / flux.size
        go, gn = oracle.trace_bwd(ap, an, incident[sel], du, de, tix, *planar, (256, 256), g.astype(np.float32),
                                  nthreads=threads)
        oracle.nurbs_bwd(cp, np.broadcast_to(uv, (h_count,) + uv.shape[1:]), [3, 3],
                         (go @ ori).reshape(pts.shape), (gn @ ori).reshape(nrm.shape), canting[sel])
        return time.perf_counter() - t0

    probe = max(threads, 1)
    t_probe = run(probe)
    scale = max(1, min(int(seconds / max(t_probe, 1e-3)), 64))      # <= 1024 heliostats = 8 GB of host distortions
    h_count = probe * scale
    t = run(h_count) if scale > 1 else t_probe
    return {"value": h_count * R * P / t, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"C oracle (OpenMP over heliostats), fwd+bwd of {h_count} heliostats x {R} rays x {P} points "
                      f"({h_count * R * P:.2e} rays) in {t:.2f} s"}


def torch_eager_baseline(args, ap, an, inc, planar, seconds):
    """The reference's own shape of the hot path - eager PyTorch-CPU ops with materialised per-ray intermediates
    (tools/torch_eager_baseline.py, pinned to the reference-generated fixtures) - fwd+bwd on a bounded sample of the
    metric workload, batch_size in {10, 100} best-of, torch's own thread pool."""
    sys.path.insert(0, str(ROOT / "tools"))
    import torch_eager_baseline as teb

    threads = max(1, min(os.cpu_count() or 1, int(os.environ.get("ARTIST_CPU_THREADS", "16"))))
    torch.set_num_threads(threads)
    R, P = args.rays, 4 * args.n_eval * args.n_eval
    tables = tuple(t.detach().cpu() for t in (planar.centers, planar.normals, planar.dimensions))
    gen = torch.Generator().manual_seed(7)

    def run(h, batch):
        both = torch.randn((h, R, P, 2), generator=gen) * (4.3681e-06 ** 0.5)
        return teb.time_epoch(ap[:h], an[:h], inc[:h], both[..., 0], both[..., 1], torch.zeros(h, dtype=torch.long), *tables,
                              batch_size=batch, backward=True)

    t1 = run(1, 10)                                        # 1e6 rays: sizes the sample
    h = max(2, min(ap.shape[0], 20, int(seconds / 2 / max(t1, 1e-3))))
    best = None
    for batch in (10, 100):
        t = run(h, batch)
        if best is None or t < best[0]:
            best = (t, batch)
    return {"value": h * R * P / best[0], "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"eager PyTorch-CPU restatement of heliostat_ray_tracer.py:316-506 + autograd, fwd+bwd of {h} heliostats x {R} rays "
                      f"x {P} points ({h * R * P:.2e} rays) in {best[0]:.2f} s, batch_size {best[1]} (best of 10/100), "
                      f"torch {torch.__version__}, {threads} threads"}


def correctness_stamp(ap, an, inc, dist_u, dist_e, tix, planar, ops, points_per_facet=0):
    """Four heliostats of the field that was just timed (first, one third, two thirds, last of this rank's list = near to far)
    through the HIP kernels once more and through the oracle: flux relative L2, ray counters, gradient relative L2."""
    import numpy as np

    import oracle
    H = ap.shape[0]
    sel = sorted({0, H // 3, (2 * H) // 3, H - 1})
    idx = torch.tensor(sel, device=ap.device)
    a, n_, i_, t_ = ap[idx].contiguous(), an[idx].contiguous(), inc[idx].contiguous(), tix[idx].contiguous()
    both = torch.stack((dist_u[idx], dist_e[idx]), dim=-1).contiguous()
    du, de = both[..., 0], both[..., 1]
    ag, ng = a.clone().requires_grad_(True), n_.clone().requires_grad_(True)
    flux, factors = ops.trace_rays(ag, ng, i_, du, de, t_, planar.centers, planar.normals, planar.dimensions, 1.0, 0.0, 0.935,
                                   (256, 256), points_per_facet=points_per_facet)
    gen = torch.Generator(device=ap.device).manual_seed(11)
    w = torch.rand(flux.shape, generator=gen, device=ap.device)
    go, gn = torch.autograd.grad(flux, (ag, ng), w)
    npf = lambda x: x.detach().cpu().numpy()  # noqa: E731
    tabs = (npf(planar.centers), npf(planar.normals), npf(planar.dimensions))
    o_flux, o_fac = oracle.trace_fwd(npf(a), npf(n_), npf(i_), npf(du), npf(de), npf(t_).astype(np.int32), *tabs, (256, 256))
    o_go, o_gn = oracle.trace_bwd(npf(a), npf(n_), npf(i_), npf(du), npf(de), npf(t_).astype(np.int32), *tabs, (256, 256), npf(w))
    rel = lambda x, y: float(np.linalg.norm(x.astype(np.float64) - y) / max(np.linalg.norm(y.astype(np.float64)), 1e-300))  # noqa: E731
    return {"heliostats": sel, "rays": int(len(sel) * du.shape[1] * du.shape[2]),
            "flux_rel_l2": rel(npf(flux), o_flux), "flux_rel_l2_per_heliostat": [rel(npf(flux[k]), o_flux[k]) for k in range(len(sel))],
            "ray_counters_equal": bool(np.array_equal(npf(factors[:2]), o_fac[:2])),
            "grad_origins_rel_l2": rel(npf(go), o_go), "grad_normals_rel_l2": rel(npf(gn), o_gn),
            "against": "oracle (C restatement of the reference, fp32 op for op) on the same inputs"}


def flux_checksum(per_target):
    """What an N-rank line and the 1-rank line can be compared by: sum, L2 norm and a 16x16 block-sum of the REDUCED
    per-target flux (fp64 on the host, printed to 9 digits)."""
    f = per_target.detach().double().cpu()
    t, hh, w = f.shape
    down = f.reshape(t, 16, hh // 16, 16, w // 16).sum(dim=(2, 4)) if hh % 16 == 0 and w % 16 == 0 else f.sum(dim=(1, 2), keepdim=True)
    return {"sum": float(f.sum()), "l2": float(f.norm()), "down16": [float(f"{v:.9g}") for v in down.flatten().tolist()]}


def launch_ranks(n: int) -> int:
    """``python bench.py --gpus N`` without a launcher: start N ranks of this script under ``torch.distributed.run``
    (one process per GPU, rendezvous on 127.0.0.1) as a CHILD process and hand back its exit code.  The parent never
    touches the GPU (tutorials/02_heliostat_raytracing_distributed_tutorial.py:60-75 leaves the launch to torchrun too)."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(pathlib.Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # (ARTIST_AMD_COLLECTIVES_AT_WORLD_1=1 under a one-rank torch.distributed.run: the RCCL calls of the N>1 path on one GPU)
    rehearse = world == 1 and "RANK" in os.environ and os.environ.get("ARTIST_AMD_COLLECTIVES_AT_WORLD_1", "0") == "1"
    if world > 1 or rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm (xGMI inside a node).  ARTIST_BENCH_BACKEND=gloo exists only to rehearse the
        # N>1 code path with several ranks sharing one GPU (RCCL refuses duplicate devices).
        backend = os.environ.get("ARTIST_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        print(f"[bench] rank {rank}/{world} on {dev} ({torch.cuda.get_device_name(dev)}), backend {backend}", file=sys.stderr, flush=True)

    from artist_amd import HeliostatRayTracer, NURBSSurfaces
    from artist_amd.distributed import all_reduce_sum_async, gather_owned_rows, owned_heliostats
    from artist_amd.scene import build_synthetic_scenario

    H_total, R, n_eval = args.heliostats, args.rays, args.n_eval
    P = 4 * n_eval * n_eval
    own = owned_heliostats(H_total, world, rank)            # heliostat i -> rank i mod N
    H = len(own)

    # ---- synthetic field; this rank keeps only its own rows (positions from the global fan) ----------
    from artist_amd import scene
    scenario, uv = build_synthetic_scenario(H_total, n_rays=R, n_cp=(args.n_cp, args.n_cp), n_eval=n_eval, device=dev)
    group = scenario.heliostat_field.heliostat_groups[0]
    mask = torch.ones(H_total, dtype=torch.int32, device=dev)
    group.activate_heliostats(mask)
    tix_all = torch.zeros(H_total, dtype=torch.long, device=dev)
    inc_all = torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=dev).repeat(H_total, 1)
    aim = scenario.solar_tower.get_centers_of_target_areas(tix_all)
    orientation_all = scene.ideal_orientations(group.active_positions, aim, inc_all)
    own_t = torch.tensor(own, dtype=torch.long, device=dev)
    orientation = orientation_all[own_t].contiguous()
    inc, tix = inc_all[own_t].contiguous(), tix_all[own_t].contiguous()
    cp_all = group.active_nurbs_control_points
    cp = cp_all[own_t].clone().requires_grad_(True)
    canting, transl = group.active_canting[own_t].contiguous(), group.active_facet_translations[own_t].contiguous()
    uv_local = uv[:1].expand(H, -1, -1, -1)
    degrees = group.nurbs_degrees
    planar = scenario.solar_tower.target_areas[0]
    T = planar.centers.shape[0]
    # Distortions: this rank's ROWS of the field-wide seed-7 sample (artist/raytracing/sampling.py:49-53 samples [H,R,P] for the
    # whole field with random_seed=7; the product's DistortionsDataset(rows=...) / Sun.get_distortions_rows draw row i from a
    # stream keyed by (seed, i)): an N-rank run traces EXACTLY the rays of the 1-rank run, whatever N is - which is what lets
    # the reduced flux be compared with the single-rank flux below.  ONE interleaved [H,R,P,2] buffer, (u,e) = stride-2 views.
    sun = scenario.light_sources.light_source_list[0]
    dist_u, dist_e = sun.get_distortions_rows(own, number_of_points=P, number_of_active_heliostats=H_total, random_seed=7)

    from artist_amd import ops
    from artist_amd.flux import FluxCrop, FluxCropPixelLoss
    target = None

    def forward():
        # evaluation + alignment in one kernel (the kinematics is fixed during a surface reconstruction)
        ap, an = NURBSSurfaces(degrees, cp, device=dev).calculate_surface_points_and_normals(uv_local, canting, transl,
                                                                                             orientations=orientation)
        flux, factors = ops.trace_rays(ap.reshape(H, P, 4), an.reshape(H, P, 4), inc, dist_u, dist_e, tix, planar.centers,
                                       planar.normals, planar.dimensions, 1.0, 0.0, 0.935, (256, 256),
                                       points_per_facet=n_eval * n_eval)   # the surfaces' [H, F, M] layout (heliostat_group.py:26-63)
        return flux, factors

    def step(backward=True):
        optimizer.zero_grad(set_to_none=True)
        with torch.set_grad_enabled(backward):
            flux, _ = forward()
            per_target = ops.per_target_sum(flux.detach(), tix, T)
            pending = all_reduce_sum_async(per_target)           # RCCL reduce of the receiver flux bitmap, overlapped
            if backward:
                # the epoch's epilogue (surface_reconstructor.py:575-590, 664-676): crop around the centre of mass,
                # pixel loss against the (cropped) measured flux
                loss = FluxCropPixelLoss.apply(flux, crop_dims, target, 6.0, 6.0).sum()      # one fused pass per direction
                loss.backward()
                # surface_reconstructor.py:767-777: every rank ends with the field's gradient.  The shards are
                # row-disjoint, so the reference's all_reduce(SUM) is an all-gather of the own rows (half the bytes)
                pending_grad = gather_owned_rows(cp.grad, H_total, async_op=True)    # travels during the optimiser step
                optimizer.step()                                 # :779 - Adam on the rows this rank owns
                field_grad = pending_grad.wait()
                del field_grad, pending_grad
            if pending is not None:
                pending.wait()
        return per_target

    # the reconstructor's optimiser (surface_reconstructor.py:452-455); a small rate keeps the workload stationary
    # (artist_amd.optim.Adam: torch.optim.Adam's update rule as one HIP kernel - torch's fused multi-tensor launch costs 41-48 us
    #  at any size, tests/test_gpu_optim.py compares the two)
    from artist_amd.optim import Adam
    optimizer = Adam([cp], lr=1e-6)
    crop_dims = planar.dimensions.index_select(0, tix.long()).contiguous()
    with torch.no_grad():
        f0, _ = forward()
        target = (FluxCrop.apply(f0, crop_dims, 6.0, 6.0) * 1.05 + 1e-3).detach()     # stands in for the measured flux
        del f0

    def sync():
        if world > 1 or rehearse:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, k):
        sync()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        sync()
        dt = time.perf_counter() - t0
        if world > 1 or rehearse:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    for _ in range(args.warmup):
        step(True)
    # ---- region 1, the headline: EXACTLY K steps between barrier + synchronize on both sides, nothing else inside ----
    dt = timed(lambda: step(True), args.steps)
    # ---- region 2, the same K steps once more with HIP events: one pair per STEP (the median step) and one pair around each
    # of the two trace calls, on the stream they are launched on (ops.record_launch_events: two event records per call, no
    # synchronisation) - the roofline's launch durations.  Outside the headline region (round 3 recorded them inside it).
    launches = ops.record_launch_events(True)
    step_events = []

    def step_with_events():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        step(True)
        b.record()
        step_events.append((a, b))

    dt_events = timed(step_with_events, args.steps)
    ops.record_launch_events(False)
    dt_fwd = timed(lambda: step(False), args.steps)
    step_ms = sorted(a.elapsed_time(b) for a, b in step_events)
    median_step_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])

    def mean_ms(pairs):
        return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)

    # art_trace_bwd at this size = ONE launch of trace_bwd_lds_kernel (small fields: + reduce_chunks_kernel);
    # art_trace_fwd = trace_fwd_lds_kernel + the accumulator conversion + the factors
    assert len(launches.get("art_trace_bwd", ())) == args.steps and len(launches.get("art_trace_fwd", ())) == args.steps, \
        {k: len(v) for k, v in launches.items()}
    ms_bwd_timed = mean_ms(launches["art_trace_bwd"])
    ms_fwd_call_timed = mean_ms(launches["art_trace_fwd"])

    # ---- the same calls back to back (five of each), HIP events on the launch stream: reported beside the timed region's ----
    def kernel_ms(fn, k=5):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / k

    with torch.no_grad():
        pts, nrm = NURBSSurfaces(degrees, cp, device=dev).calculate_surface_points_and_normals(uv_local, canting, transl)
        ap, an = ops.align_surfaces(pts.reshape(H, P, 4), nrm.reshape(H, P, 4), orientation)
    ms_fwd = kernel_ms(lambda: ops.trace_rays(ap, an, inc, dist_u, dist_e, tix, planar.centers, planar.normals,
                                              planar.dimensions, 1.0, 0.0, 0.935, (256, 256), points_per_facet=n_eval * n_eval))
    apg, ang = ap.clone().requires_grad_(True), an.clone().requires_grad_(True)
    flux, _ = ops.trace_rays(apg, ang, inc, dist_u, dist_e, tix, planar.centers, planar.normals, planar.dimensions,
                             1.0, 0.0, 0.935, (256, 256), points_per_facet=n_eval * n_eval)
    gflux = torch.ones_like(flux)
    ms_bwd = kernel_ms(lambda: torch.autograd.grad(flux, (apg, ang), gflux, retain_graph=True))
    del flux, gflux, apg, ang

    rays_local = H * R * P
    # algorithmic bytes per launch: distortions 8 B/ray + origin/normal 32 B/point
    # + one bitmap write (fwd) / one grad-bitmap read + 32 B/point grad write (bwd)
    bytes_fwd = rays_local * 8 + H * P * 32 + H * 256 * 256 * 4
    bytes_bwd = rays_local * 8 + H * P * 32 + H * P * 32 + H * 256 * 256 * 4
    # dominant kernel: the longer of the two trace CALLS as measured in region 2.  art_trace_bwd is ONE launch of
    # trace_bwd_lds_kernel at this size; the forward call is trace_fwd_lds_kernel + the accumulator conversion pass (+ two
    # 5 us kernels), priced as a whole with the call's algorithmic bytes (the conversion's traffic is overhead, not algorithm).
    if ms_bwd_timed >= ms_fwd_call_timed:
        dom = dict(kernel="trace_bwd_lds_kernel", ms=ms_bwd_timed, bytes=bytes_bwd, what="one launch = all of art_trace_bwd")
    else:
        dom = dict(kernel="trace_fwd_lds_kernel", ms=ms_fwd_call_timed, bytes=bytes_fwd,
                   what="the art_trace_fwd call: trace_fwd_lds_kernel + accum_to_flux_kernel + factor kernels")
    achieved = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
    # HBM bytes per launch: REPLAYED from the rocprofv3 --pmc passes of this same command that are committed under
    # profiles/ (separate passes, FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes) - not measured by this run,
    # hence "traffic_source"; null when the profiled workload is not the one timed here.
    traffic, traffic_source = None, None
    try:
        prof = json.load(open(HBM_TRAFFIC_FILE))
        c = prof["config"]
        if (c["heliostats"], c["rays_per_point"], c["points_per_heliostat"]) == (H, R, P) and world == 1:
            traffic = prof[dom["kernel"]]["total_bytes"]
            traffic_source = f"replayed from {HBM_TRAFFIC_FILE.relative_to(ROOT)} (rocprofv3 --pmc, not measured in this run)"
    except (OSError, KeyError, ValueError):
        pass
    measured_peak = None
    try:
        measured_peak = json.load(open(HBM_PEAK_FILE))
    except (OSError, ValueError):
        pass

    # ---- correctness stamp, outside every timed region: the rays just timed against the oracle ---------------
    check = None
    if rank == 0 and not args.no_check:
        check = correctness_stamp(ap, an, inc, dist_u, dist_e, tix, planar, ops, n_eval * n_eval)

    # ---- did sharding preserve the result?  (outside every timed region) --------------------------------------------------
    # The flux of the whole field as the ranks produced it - local per-target sums + ONE all-reduce, the exchange of
    # tutorials/02_heliostat_raytracing_distributed_tutorial.py:185-190 - against the same field traced by ONE rank: rank 0
    # gathers everybody's control points (they were trained for warmup + steps epochs), draws ALL rows of the seed-7 sample
    # and traces the 1000 heliostats alone.  An N = 1 run prints the same checksum, so lines of different N can be compared
    # with each other as well (bit-identical per-heliostat bitmaps; the per-target sums differ by fp32 summation order).
    with torch.no_grad():
        reduced = step(False)                                    # forward only: NURBS -> trace -> per-target sum -> all-reduce
    sharding = {"reduced_flux": flux_checksum(reduced)} if rank == 0 else None
    if (world > 1 or rehearse) and not args.no_check:
        cp_field = gather_owned_rows(cp.detach(), H_total)       # collective: every rank takes part
        if rank == 0:
            with torch.no_grad():
                du_all, de_all = sun.get_distortions_rows(list(range(H_total)), number_of_points=P,
                                                          number_of_active_heliostats=H_total, random_seed=7)
                ap_all, an_all = NURBSSurfaces(degrees, cp_field, device=dev).calculate_surface_points_and_normals(
                    uv[:1].expand(H_total, -1, -1, -1), group.active_canting, group.active_facet_translations,
                    orientations=orientation_all)
                flux_all, _ = ops.trace_rays(ap_all.reshape(H_total, P, 4), an_all.reshape(H_total, P, 4), inc_all, du_all, de_all,
                                             tix_all, planar.centers, planar.normals, planar.dimensions, 1.0, 0.0, 0.935, (256, 256),
                                             points_per_facet=n_eval * n_eval)
                single = ops.per_target_sum(flux_all, tix_all, T)
                rel = float((reduced.double() - single.double()).norm() / single.double().norm())
                del du_all, de_all, ap_all, an_all, flux_all
            sharding.update({"single_rank_flux": flux_checksum(single), "reduced_vs_single_rank_rel_l2": rel,
                             "tolerance": 2e-6, "ok": bool(rel < 2e-6)})

    if rank == 0:
        total_rays = H_total * R * P
        out = {
            "metric": "traced rays/s fwd+bwd, 1000-heliostat NURBS field",
            "value": total_rays * args.steps / dt,
            "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            # region 2 (same K steps again, one HIP-event pair per step on the launch stream): median and spread of a step
            "step_events": {"median_ms": median_step_ms, "min_ms": step_ms[0], "max_ms": step_ms[-1],
                            "wall_ms_per_step": dt_events / args.steps * 1e3,
                            "what": "second timed region of K steps, HIP events per step + per trace call; the headline region holds no event record"},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"metric config: {H_total} heliostats x {R} rays/point x {P} points "
                                   f"({total_rays:.3g} rays/step), {args.n_cp}x{args.n_cp} degree-3 NURBS, 256x256 bitmap, "
                                   "surface-reconstruction epoch fwd+bwd",
                       "heliostats": H_total, "rays_per_point": R, "points_per_heliostat": P,
                       "parallelism": f"heliostat-sharded dp{world}", "heliostats_per_rank": H},
            "fwd_only": {"value": total_rays * args.steps / dt_fwd, "unit": "rays/s",
                         "ms_per_step": dt_fwd / args.steps * 1e3},
            "kernels": {"where": "HIP events around art_trace_fwd / art_trace_bwd on their launch stream, mean over the K steps of the "
                                 "second timed region (art_trace_fwd = trace kernel + accumulator conversion + factors)",
                        "trace_fwd_ms": ms_fwd_call_timed, "trace_bwd_ms": ms_bwd_timed,
                        "trace_fwd_rays_per_s": rays_local / (ms_fwd_call_timed * 1e-3),
                        "trace_fwd_GBps": bytes_fwd / (ms_fwd_call_timed * 1e-3) / 1e9,
                        "trace_bwd_rays_per_s": rays_local / (ms_bwd_timed * 1e-3),
                        "trace_bwd_GBps": bytes_bwd / (ms_bwd_timed * 1e-3) / 1e9,
                        "back_to_back": {"what": "five calls of each in a row after the timed region (the chip's clock settles "
                                                 "differently under one kernel than under the epoch's mix)",
                                         "trace_fwd_ms": ms_fwd, "trace_bwd_ms": ms_bwd,
                                         "trace_fwd_rays_per_s": rays_local / (ms_fwd * 1e-3),
                                         "trace_bwd_rays_per_s": rays_local / (ms_bwd * 1e-3),
                                         "trace_fwd_GBps": bytes_fwd / (ms_fwd * 1e-3) / 1e9,
                                         "trace_bwd_GBps": bytes_bwd / (ms_bwd * 1e-3) / 1e9}},
            "roofline": {"bound": "hbm", "kernel": dom["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "bytes_per_launch": dom["bytes"], "ms_per_launch": dom["ms"], "launch": dom["what"],
                         # what a hand-written float4 streaming kernel reaches on this pool's MI355X (tools/hbm_peak.hip):
                         # replayed from profiles/, next to the 8 TB/s spec peak that `frac` is quoted against
                         "peak_measured": None if measured_peak is None else {
                             "copy": measured_peak["copy"], "read": measured_peak["read"], "write": measured_peak["write"],
                             "unit": "GB/s", "source": f"replayed from {HBM_PEAK_FILE.relative_to(ROOT)}",
                             "frac_of_measured_read": achieved / measured_peak["read"]},
                         # what binds the kernel (neither of the contract's two roofs): replayed analysis, DESIGN.md 4.0 / 4.1
                         "limiter_note": "not HBM: about two thirds of the kernel are vector issue of the ray arithmetic (reference operation order, "
                                         "no FMA contraction), the rest LDS-atomic, stray-ray and stream stalls - ablation table in DESIGN.md "
                                         "section 4.4, issue rates and the LDS conflict split in 4.0 (profiles/r02_issue_bench.json, r02_ablation.txt, r03_pmc_lds_conflicts.txt); "
                                         "with all 256 CUs busy the shader clock settles at ~2.0 GHz (2.4 GHz with an eighth of them: an item costs the same cycles at "
                                         "every load, profiles/r03_clock_vs_load.txt, DESIGN.md 4.2.1); stated from profiles/, not measured by this run"},
            "check": check,
            "sharding_check": sharding,
        }
        if world == 1 and not args.no_cpu_baseline:
            import numpy as np
            k = min(H, 64)
            npf = lambda x: x.detach().cpu().numpy()  # noqa: E731
            out["cpu_baseline"] = cpu_baseline(
                args, npf(cp[:k]), npf(canting[:k]), npf(transl[:k]), npf(uv[:1].contiguous()),
                npf(orientation[:k]), npf(inc[:k]), (npf(planar.centers), npf(planar.normals), npf(planar.dimensions)),
                args.cpu_seconds)
            # the reference-shaped path (eager PyTorch-CPU op chain) next to the C port
            out["cpu_baseline"]["reference_shape_torch"] = torch_eager_baseline(
                args, ap[:min(H, 64)].cpu(), an[:min(H, 64)].cpu(), inc[:min(H, 64)].cpu(), planar, args.cpu_seconds)
        print(json.dumps(out), flush=True)
        if sharding is not None and sharding.get("ok") is False:
            raise SystemExit(f"sharded flux differs from the single-rank flux: rel L2 {sharding['reduced_vs_single_rank_rel_l2']:.3e}")
    if world > 1 or rehearse:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
