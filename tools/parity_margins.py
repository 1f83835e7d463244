#!/usr/bin/env python3
"""How far is the HIP path from the fp32 restatement (oracle) and from the reference's fixtures on the ill-conditioned cases,
beside the reference's own fp32-vs-fp64 distance that the tests use as their yardstick?  Prints one line per case and quantity."""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import conftest
from conftest import BLOCKING_CASES, CYL_CASES, rel_l2
import oracle
import test_gpu_parity as tp

def golden(name):
    return dict(np.load(ROOT / "tests" / "golden" / f"{name}.npz"))

from artist_amd import trace_rays
n, t = tp.n, tp.t
for name in CYL_CASES:
    d, d64 = golden(name), golden(name + "_f64")
    inp = tp.trace_inputs(d); inp["origins"].requires_grad_(True); inp["normals"].requires_grad_(True)
    flux, fac = trace_rays(**inp, cyl=tp.cyl_inputs(d))
    (flux * t(d["loss_weights"])).sum().backward()
    o_flux, o_fac = tp.oracle_fwd(d, cyl=oracle.cyl_tables(d))
    go, gn = oracle.trace_bwd(d["aligned_points"], d["aligned_normals"], d["incident"], d["distortions_u"], d["distortions_e"], d["target_idx"],
                              d["target_centers"], d["target_normals"], d["target_dims"], d["resolution"], d["loss_weights"], float(d["ray_magnitude"]),
                              float(d["extinction"]), float(d["reflectivity"]), cyl=oracle.cyl_tables(d))
    print(f"{name:18s} flux: hip-oracle {rel_l2(n(flux), o_flux):.2e}  hip-ref32 {rel_l2(n(flux), d['flux']):.2e}  ref32-ref64 {rel_l2(d['flux'], d64['flux']):.2e}  oracle-ref32 {rel_l2(o_flux, d['flux']):.2e}")
    for got, orc, key in ((inp["origins"].grad, go, "grad_aligned_points"), (inp["normals"].grad, gn, "grad_aligned_normals")):
        print(f"{'':18s} {key}: hip-oracle {rel_l2(n(got), orc):.2e}  hip-ref32 {rel_l2(n(got), d[key]):.2e}  ref32-ref64 {rel_l2(d[key], d64[key]):.2e}")
for name in BLOCKING_CASES:
    d, d64 = golden(name), golden(name + "_f64")
    H = d["aligned_points"].shape[0]
    flux, fac, flags = trace_rays(**tp.trace_inputs(d), blocking=tp.blocking_inputs(d))
    o_flux, o_fac = tp.oracle_fwd(d, blocking=oracle.blocking_tables(d, H))
    print(f"{name:18s} flux: hip-oracle {rel_l2(n(flux), o_flux):.2e}  hip-ref32 {rel_l2(n(flux), d['flux']):.2e}  ref32-ref64 {rel_l2(d['flux'], d64['flux']):.2e}  oracle-ref32 {rel_l2(o_flux, d['flux']):.2e}")
