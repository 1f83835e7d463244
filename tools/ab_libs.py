#!/usr/bin/env python3
"""Same-box, same-process A/B of libartist_hip builds (and of environment knobs) on the metric field.

Boxes of this pool differ by 10-20 % on the same binary, so timings from different gpurun calls cannot be compared.
This tool loads every given library into ONE process, builds the field once and times the forward / backward trace
ops of each (library, environment) variant in interleaved rounds; it prints the median and the minimum per variant.

usage:  python tools/ab_libs.py [--heliostats 1000] [--rounds 7] [--reps 3] NAME=path/to/lib.so[,ENV=VAL,...] ...
        (a bare NAME= uses artist_amd/libartist_hip.so; every library must have the ABI of artist_amd/_lib.py)
"""
from __future__ import annotations

import argparse
import ctypes
import os
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--heliostats", type=int, default=1000)
    ap.add_argument("--rays", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-facets", action="store_true", help="do not tell the kernels the facet structure")
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()

    from artist_amd import _lib, ops, scene
    default = _lib.lib()
    dev = torch.device("cuda:0")
    H, R = args.heliostats, args.rays
    scenario, uv = scene.build_synthetic_scenario(H, n_rays=R, device=dev)
    group = scenario.heliostat_field.heliostat_groups[0]
    mask = torch.ones(H, dtype=torch.int32, device=dev)
    group.activate_heliostats(mask)
    tix = torch.zeros(H, dtype=torch.long, device=dev)
    inc = torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=dev).repeat(H, 1)
    group.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask)
    ap_, an_ = group.active_surface_points.contiguous(), group.active_surface_normals.contiguous()
    P = ap_.shape[1]
    ppf = 0 if args.no_facets else P // group.number_of_facets_per_heliostat
    gen = torch.Generator(device=dev).manual_seed(7)
    both = torch.randn((H, R, P, 2), generator=gen, device=dev).mul_(4.3681e-06 ** 0.5)
    du, de = both[..., 0], both[..., 1]
    planar = scenario.solar_tower.target_areas[0]
    gflux = torch.rand((H, 256, 256), device=dev)

    variants = []
    for spec in args.variants:
        name, _, rest = spec.partition("=")
        parts = [x for x in rest.split(",") if x]
        path = parts[0] if parts and "=" not in parts[0] else ""
        env = dict(x.split("=", 1) for x in parts if "=" in x)
        handle = default
        if path:
            handle = ctypes.CDLL(str(pathlib.Path(path).resolve()))
            for fname, argtypes in _lib.SIGNATURES.items():
                fn = getattr(handle, fname)
                fn.argtypes = argtypes
                fn.restype = (ctypes.c_char_p if fname == "art_strerror"
                              else ctypes.c_int64 if fname in ("art_blocking_workspace_bytes", "art_trace_bwd_scratch_floats") else ctypes.c_int)
            assert handle.art_abi_version() == _lib.ABI_VERSION, (path, handle.art_abi_version())
        variants.append((name, handle, env))

    def run(handle, env, what):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        saved = _lib._LIB
        _lib._LIB = handle
        try:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if what == "fwd":
                a.record()
                for _ in range(args.reps):
                    flux, fac = ops.trace_rays(ap_, an_, inc, du, de, tix, planar.centers, planar.normals, planar.dimensions, 1.0, 0.0,
                                               0.935, (256, 256), points_per_facet=ppf)
                b.record()
            else:
                apg, ang = ap_.clone().requires_grad_(True), an_.clone().requires_grad_(True)
                flux, fac = ops.trace_rays(apg, ang, inc, du, de, tix, planar.centers, planar.normals, planar.dimensions, 1.0, 0.0,
                                           0.935, (256, 256), points_per_facet=ppf)
                a.record()
                for _ in range(args.reps):
                    torch.autograd.grad(flux, (apg, ang), gflux, retain_graph=True)
                b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / args.reps, float(flux.double().sum())
        finally:
            _lib._LIB = saved
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    times = {(name, what): [] for name, _, _ in variants for what in ("fwd", "bwd")}
    sums = {}
    for rnd in range(args.rounds + 1):
        for name, handle, env in variants:
            for what in ("fwd", "bwd"):
                ms, total = run(handle, env, what)
                if rnd > 0:
                    times[(name, what)].append(ms)
                sums[name] = total
    for name, _, env in variants:
        f, b = times[(name, "fwd")], times[(name, "bwd")]
        print(f"{name:24s} fwd {statistics.median(f):.3f} ms (min {min(f):.3f})   bwd {statistics.median(b):.3f} ms (min {min(b):.3f})   "
              f"sum(flux) {sums[name]:.6e}   {env if env else ''}")


if __name__ == "__main__":
    main()
