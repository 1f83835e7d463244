"""ctypes/numpy front-end of the CPU oracle (``oracle/liboracle.so``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` - as the checker, never as the product.  ``artist_amd``
must not import this package (tests/test_boundary.py enforces it).

All functions take/return numpy arrays; float32 inputs run the ``_f32`` restatement (the
reference's arithmetic), float64 inputs the ``_f64`` yardstick.
"""
from __future__ import annotations

import ctypes
import pathlib
import subprocess

import numpy as np

_DIR = pathlib.Path(__file__).resolve().parent
_LIB = None


def build(force: bool = False) -> pathlib.Path:
    so = _DIR / "liboracle.so"
    srcs = [_DIR / "artist_oracle.c", _DIR / "oracle_impl.inc"]
    if force or not so.exists() or any(s.exists() and s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.check_call(["make", "-C", str(_DIR), "-s", "-B", "liboracle.so"])
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(str(build()))
        _LIB.orc_sampler_indices.restype = ctypes.c_int64
    return _LIB


def max_threads() -> int:
    return int(lib().orc_max_threads())


def _sfx(dtype) -> str:
    return {np.dtype(np.float32): "_f32", np.dtype(np.float64): "_f64"}[np.dtype(dtype)]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


_i64 = ctypes.c_int64
_dbl = ctypes.c_double


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with code {rc}")


def reflect(incident, normals):
    dt = normals.dtype
    inc, nrm = _c(incident, dt), _c(normals, dt)
    H, P = nrm.shape[0], nrm.shape[1]
    out = np.empty_like(nrm)
    _check(getattr(lib(), "orc_reflect" + _sfx(dt))(_p(inc), _p(nrm), _i64(H), _i64(P), _p(out)), "reflect")
    return out


def _dist_args(dist_u, dist_e):
    """Element strides of the [H,R,P] distortion views (they may be stride-2 views of one
    interleaved buffer, artist/scene/sun.py:227-234)."""
    assert dist_u.shape == dist_e.shape and dist_u.strides == dist_e.strides
    it = dist_u.itemsize
    assert all(s % it == 0 for s in dist_u.strides)
    return [_i64(s // it) for s in dist_u.strides]


def _cyl_args(cyl, dt):
    """Cylindrical target tables -> ctypes arguments (6 pointers + count); ``cyl`` is None or a dict with
    centers/normals/axes [Tc,4] and radii/heights/opening [Tc]."""
    if cyl is None or len(cyl["radii"]) == 0:
        return [None] * 6 + [_i64(0)], []
    keep = [_c(cyl[k], dt) for k in ("centers", "normals", "axes", "radii", "heights", "opening")]
    return [_p(a) for a in keep] + [_i64(keep[3].shape[0])], keep


def _blocking_args(blocking, dt, H):
    """``blocking`` = None or dict(corners [N,4,4], spans [N,2,4], normals [N,4], owner [H][, lbvh_compat]): the
    tables of ALL blocking primitives and, per traced heliostat, the index of its own rectangle.  ``lbvh_compat``
    (default True) restricts the filter to the primitives that are reachable in the reference's tree."""
    if blocking is None:
        return [None] * 4 + [_i64(0), ctypes.c_int(1)], [], 0
    keep = [_c(blocking["corners"], dt), _c(blocking["spans"], dt), _c(blocking["normals"], dt),
            _c(blocking["owner"], np.int32)]
    N = keep[0].shape[0]
    assert keep[0].shape == (N, 4, 4) and keep[1].shape == (N, 2, 4) and keep[2].shape == (N, 4) and keep[3].shape == (H,)
    return [_p(a) for a in keep] + [_i64(N), ctypes.c_int(1 if blocking.get("lbvh_compat", True) else 0)], keep, N


def trace_fwd(origins, normals, incident, dist_u, dist_e, target_idx, centers, plane_normals, dims,
              resolution, ray_magnitude=1.0, extinction=0.0, reflectivity=0.935, nthreads=0, debug=False, cyl=None,
              blocking=None):
    """Returns (flux [H,Hh,W], factors [3,H]) and, with debug=True, a dict of per-stage arrays (with blocking:
    also ``blocked`` [H,R,P] and ``filter_flags`` [N])."""
    dt = origins.dtype
    o, n, inc = _c(origins, dt), _c(normals, dt), _c(incident, dt)
    du = np.asarray(dist_u, dtype=dt)
    de = np.asarray(dist_e, dtype=dt)
    H, P = o.shape[0], o.shape[1]
    R = du.shape[1]
    assert du.shape == (H, R, P), (du.shape, (H, R, P))
    tix = _c(target_idx, np.int32)
    c, m, d = _c(centers, dt), _c(plane_normals, dt), _c(dims, dt)
    T = c.shape[0]
    W, Hh = int(resolution[0]), int(resolution[1])
    flux = np.empty((H, Hh, W), dtype=dt)
    factors = np.empty((3, H), dtype=dt)
    dbg = {}
    if debug:
        dbg = dict(reflected=np.empty((H, P, 4), dt), scattered=np.empty((H, R, P, 4), dt),
                   e_px=np.empty((H, R, P), dt), u_px=np.empty((H, R, P), dt),
                   distances=np.empty((H, R, P), dt), intensities=np.empty((H, R, P), dt))
    cargs, _keep = _cyl_args(cyl, dt)
    bargs, _keepb, N = _blocking_args(blocking, dt, H)
    if debug and N:
        dbg.update(blocked=np.empty((H, R, P), dt), filter_flags=np.empty((N,), np.int32))
    rc = getattr(lib(), "orc_trace_fwd" + _sfx(dt))(
        _p(o), _p(n), _p(inc), _p(du), _p(de), *_dist_args(du, de), _p(tix), _p(c), _p(m), _p(d), *cargs,
        *bargs, _p(dbg.get("filter_flags")),
        _dbl(ray_magnitude), _dbl(extinction), _dbl(reflectivity),
        _i64(H), _i64(R), _i64(P), _i64(T), _i64(W), _i64(Hh), _p(flux), _p(factors), ctypes.c_int(nthreads),
        _p(dbg.get("reflected")), _p(dbg.get("scattered")), _p(dbg.get("e_px")), _p(dbg.get("u_px")),
        _p(dbg.get("distances")), _p(dbg.get("intensities")), _p(dbg.get("blocked")))
    _check(rc, "trace_fwd")
    return (flux, factors, dbg) if debug else (flux, factors)


def trace_bwd(origins, normals, incident, dist_u, dist_e, target_idx, centers, plane_normals, dims,
              resolution, grad_flux, ray_magnitude=1.0, extinction=0.0, reflectivity=0.935, nthreads=0, cyl=None,
              blocking=None):
    """(grad_origins, grad_normals); with ``blocking`` also the DIRECT gradients w.r.t. the primitive tables
    (grad_corners [N,4,4] - corner 0 only -, grad_spans [N,2,4], grad_normals [N,4])."""
    dt = origins.dtype
    o, n, inc = _c(origins, dt), _c(normals, dt), _c(incident, dt)
    du = np.asarray(dist_u, dtype=dt)
    de = np.asarray(dist_e, dtype=dt)
    H, P = o.shape[0], o.shape[1]
    R = du.shape[1]
    tix = _c(target_idx, np.int32)
    c, m, d = _c(centers, dt), _c(plane_normals, dt), _c(dims, dt)
    T = c.shape[0]
    W, Hh = int(resolution[0]), int(resolution[1])
    g = _c(grad_flux, dt)
    assert g.shape == (H, Hh, W)
    go, gn = np.empty_like(o), np.empty_like(n)
    cargs, _keep = _cyl_args(cyl, dt)
    bargs, _keepb, N = _blocking_args(blocking, dt, H)
    gpc = np.zeros((N, 4, 4), dt) if N else None
    gps = np.zeros((N, 2, 4), dt) if N else None
    gpn = np.zeros((N, 4), dt) if N else None
    rc = getattr(lib(), "orc_trace_bwd" + _sfx(dt))(
        _p(o), _p(n), _p(inc), _p(du), _p(de), *_dist_args(du, de), _p(tix), _p(c), _p(m), _p(d), *cargs, *bargs,
        _dbl(ray_magnitude), _dbl(extinction), _dbl(reflectivity),
        _i64(H), _i64(R), _i64(P), _i64(T), _i64(W), _i64(Hh), _p(g), _p(go), _p(gn), _p(gpc), _p(gps), _p(gpn),
        ctypes.c_int(nthreads))
    _check(rc, "trace_bwd")
    return (go, gn, gpc, gps, gpn) if N else (go, gn)


def blocking_primitives(surfaces):
    """create_blocking_primitives_rectangles_by_index (artist/raytracing/blocking.py:123-209)."""
    dt = surfaces.dtype
    sfc = _c(surfaces, dt)
    N, P = sfc.shape[0], sfc.shape[1]
    corners, spans, normals = np.zeros((N, 4, 4), dt), np.zeros((N, 2, 4), dt), np.zeros((N, 4), dt)
    _check(getattr(lib(), "orc_blocking_primitives" + _sfx(dt))(_p(sfc), _i64(N), _i64(P), _p(corners), _p(spans),
                                                                _p(normals)), "blocking_primitives")
    return corners, spans, normals


def lbvh_live(corners):
    """Which primitives the reference's LBVH (blocking.py:514-749) can reach from its root (1 = reachable)."""
    dt = corners.dtype
    cc = _c(corners, dt)
    live = np.zeros((cc.shape[0],), np.int32)
    _check(getattr(lib(), "orc_lbvh_live" + _sfx(dt))(_p(cc), _i64(cc.shape[0]), _p(live)), "lbvh_live")
    return live


def blocking_filter(origins3, dirs3, t_target, owner, corners, lbvh_compat=True):
    """lbvh_filter_blocking_planes (:832-995) for independent rays -> indices of the primitives that are hit."""
    dt = origins3.dtype
    o, dd, tt, cc = _c(origins3, dt), _c(dirs3, dt), _c(t_target, dt), _c(corners, dt)
    ow = _c(owner, np.int32)
    flags = np.zeros((cc.shape[0],), np.int32)
    _check(getattr(lib(), "orc_blocking_filter" + _sfx(dt))(_p(o), _p(dd), _p(tt), _p(ow), _i64(o.shape[0]), _p(cc),
                                                            _i64(cc.shape[0]), ctypes.c_int(int(lbvh_compat)),
                                                            _p(flags)), "blocking_filter")
    return np.nonzero(flags)[0]


def soft_blocking(origins3, dirs3, corners, spans, normals):
    """soft_ray_blocking_mask (:212-354) for independent rays against all given primitives."""
    dt = origins3.dtype
    o, dd = _c(origins3, dt), _c(dirs3, dt)
    cc, ss, nn = _c(corners, dt), _c(spans, dt), _c(normals, dt)
    out = np.empty((o.shape[0],), dt)
    _check(getattr(lib(), "orc_soft_blocking" + _sfx(dt))(_p(o), _p(dd), _i64(o.shape[0]), _p(cc), _p(ss), _p(nn),
                                                          _i64(cc.shape[0]), _p(out)), "soft_blocking")
    return out


def flux_crop(flux, dims, crop_width=6.0, crop_height=6.0, grad_out=None):
    """crop_flux_distributions_around_center (artist/flux/bitmap.py:121-246) for bitmaps ``[B,Hh,W]`` whose target
    areas measure ``dims [B,2]`` metres.  Returns (cropped, centres [B,3]) or, with ``grad_out``, the gradient
    w.r.t. ``flux``."""
    dt = flux.dtype
    f, dd = _c(flux, dt), _c(dims, dt)
    B, Hh, W = f.shape
    if grad_out is None:
        out, com = np.empty_like(f), np.empty((B, 3), dt)
        _check(getattr(lib(), "orc_flux_crop_fwd" + _sfx(dt))(_p(f), _p(dd), _i64(B), _i64(Hh), _i64(W), _dbl(crop_width),
                                                              _dbl(crop_height), _p(out), _p(com)), "flux_crop_fwd")
        return out, com
    g, gf = _c(grad_out, dt), np.empty_like(f)
    _check(getattr(lib(), "orc_flux_crop_bwd" + _sfx(dt))(_p(f), _p(dd), _i64(B), _i64(Hh), _i64(W), _dbl(crop_width),
                                                          _dbl(crop_height), _p(g), _p(gf)), "flux_crop_bwd")
    return gf


def _loss(name, pred, truth, grad_loss):
    dt = pred.dtype
    p, g = _c(pred, dt), _c(truth, dt)
    B, npix = p.shape[0], int(np.prod(p.shape[1:]))
    loss = np.empty((B,), dt)
    gl = None if grad_loss is None else _c(grad_loss, dt)
    gp = None if grad_loss is None else np.empty_like(p)
    _check(getattr(lib(), name + _sfx(dt))(_p(p), _p(g), _i64(B), _i64(npix), _p(loss), _p(gl), _p(gp)), name)
    return loss if grad_loss is None else (loss, gp)


def pixel_loss(pred, truth, grad_loss=None):
    """PixelLoss (artist/optim/loss.py:251-318) per sample, reduction over the bitmap; with ``grad_loss`` [B] also
    the gradient w.r.t. the prediction."""
    return _loss("orc_pixel_loss", pred, truth, grad_loss)


def kl_loss(pred, truth, grad_loss=None):
    """KLDivergenceLoss (artist/optim/loss.py:321-410) per sample."""
    return _loss("orc_kl_loss", pred, truth, grad_loss)


def center_of_mass(bitmaps, grad_com=None):
    """get_center_of_mass (artist/flux/bitmap.py:12-71) in numpy, the reference's operation order in the bitmaps' dtype:
    normalise by (sum + 1e-8), then the first moments with pixel indices as coordinates.  Returns ``[B,2]`` = (e pixel,
    u pixel); with ``grad_com [B,2]`` the gradient w.r.t. the bitmaps instead ((index - centre) / (sum + 1e-8))."""
    dt = bitmaps.dtype
    b, hh, w = bitmaps.shape
    s = bitmaps.sum(axis=(1, 2), keepdims=True, dtype=dt) + dt.type(1e-8)                  # :46-53
    normalized = bitmaps / s
    e = np.linspace(0, w - 1, w, dtype=dt)[None, None, :]                                   # :55-61
    u = np.linspace(0, hh - 1, hh, dtype=dt)[None, :, None]
    com = np.stack([(e * normalized).sum(axis=(1, 2), dtype=dt), (u * normalized).sum(axis=(1, 2), dtype=dt)], axis=1)   # :64-71
    if grad_com is None:
        return com
    g = np.asarray(grad_com, dt)
    return (g[:, 0, None, None] * (e - com[:, 0, None, None]) + g[:, 1, None, None] * (u - com[:, 1, None, None])) / s


def bitmap_to_target_coordinates(bitmap_coordinates, resolution, target_idx, centers, dims, cyl=None):
    """bitmap_coordinates_to_target_coordinates (artist/geometry/coordinates.py:119-249): pixel (e, u) -> homogeneous world
    coordinates on the target surface.  ``centers [T,4]``, ``dims [T,2]`` are the planar tables, ``cyl`` the dict of
    ``cyl_tables``; global index, planar first.  ``resolution`` = (width, height)."""
    bc = np.asarray(bitmap_coordinates)
    dt = bc.dtype
    out = np.zeros((bc.shape[0], 4), dt)
    out[:, 3] = 1
    e_norm = (bc[:, 0] + dt.type(0.5)) / dt.type(resolution[0])                             # :185-186
    u_norm = (bc[:, 1] + dt.type(0.5)) / dt.type(resolution[1])
    T = 0 if centers is None else centers.shape[0]
    tix = np.asarray(target_idx)
    for k in range(bc.shape[0]):
        t = int(tix[k])
        if t < T:                                                                             # :193-218
            e_local = (dt.type(0.5) - e_norm[k]) * dims[t, 0]
            u_local = (dt.type(0.5) - u_norm[k]) * dims[t, 1]
            out[k, :3] = centers[t, :3] + e_local * np.array([1, 0, 0], dt) + u_local * np.array([0, 0, 1], dt)
        else:                                                                                 # :220-247
            c = t - T
            axis, normal = cyl["axes"][c, :3].astype(dt), cyl["normals"][c, :3].astype(dt)
            v = np.cross(axis, normal)
            theta = (e_norm[k] - dt.type(0.5)) * dt.type(np.ravel(cyl["opening"])[c])
            z = (dt.type(0.5) - u_norm[k]) * dt.type(np.ravel(cyl["heights"])[c])
            r = dt.type(np.ravel(cyl["radii"])[c])
            out[k, :3] = cyl["centers"][c, :3].astype(dt) + r * np.cos(theta) * normal + r * np.sin(theta) * v + z * axis
    return out


def focal_spot_loss(prediction, ground_truth, target_idx, centers, dims, cyl=None):
    """FocalSpotLoss (artist/optim/loss.py:124-250): distance between the world positions of the two bitmaps' centres
    of mass on their target area, per sample."""
    res = (prediction.shape[2], prediction.shape[1])                                        # (width, height), :212-216
    a = bitmap_to_target_coordinates(center_of_mass(prediction), res, target_idx, centers, dims, cyl)
    b = bitmap_to_target_coordinates(center_of_mass(ground_truth), res, target_idx, centers, dims, cyl)
    return np.linalg.norm(a[:, :3] - b[:, :3], axis=1).astype(prediction.dtype)               # :245-249


def rigid_body_orientations(positions, rot_dev, trans_dev, act_nonopt, act_opt, offsets, incident=None, aim=None,
                            motor_positions=None, max_iter=4, min_eps=1e-4):
    """RigidBody.incident_ray_directions_to_orientations (artist/field/kinematics_rigid_body.py:540-634) when
    ``incident``/``aim`` are given, ``motor_positions_to_orientations`` (:510-538) when ``motor_positions`` are.
    Returns (orientations [H,4,4], motor positions [H,2], number of forward evaluations)."""
    dt = positions.dtype
    pos, rd, td, an = _c(positions, dt), _c(rot_dev, dt), _c(trans_dev, dt), _c(act_nonopt, dt)
    H, rows = pos.shape[0], an.shape[1]
    ao = _c(act_opt, dt) if act_opt is not None and np.size(act_opt) else None
    off = _c(offsets, dt)
    mode = 0 if motor_positions is not None else 1
    motor = _c(motor_positions, dt).copy() if mode == 0 else np.zeros((H, 2), dt)
    inc = _c(incident, dt) if mode == 1 else None
    aimc = _c(aim, dt) if mode == 1 else None
    ori = np.empty((H, 4, 4), dt)
    rc = getattr(lib(), "orc_rigid_body" + _sfx(dt))(
        ctypes.c_int(mode), _p(pos), _p(rd), _p(td), _p(an), _i64(rows), _p(ao), _p(off), _p(inc), _p(aimc), _i64(H),
        ctypes.c_int(max_iter), _dbl(min_eps), _p(ori), _p(motor))
    if rc < 0:
        _check(rc, "rigid_body")
    return ori, motor, rc


def blocking_tables(d, H=None):
    """The blocking tables of a golden fixture as the ``blocking=`` argument (one group, every heliostat active:
    heliostat h owns primitive h)."""
    H = d["aligned_points"].shape[0] if H is None else H
    return dict(corners=d["prim_corners"], spans=d["prim_spans"], normals=d["prim_normals"],
                owner=np.arange(H, dtype=np.int32))


def per_target(bitmaps, target_idx, n_targets):
    dt = bitmaps.dtype
    b = _c(bitmaps, dt)
    H = b.shape[0]
    npix = int(np.prod(b.shape[1:]))
    out = np.empty((n_targets,) + b.shape[1:], dtype=dt)
    _check(getattr(lib(), "orc_per_target" + _sfx(dt))(_p(b), _p(_c(target_idx, np.int32)), _i64(H), _i64(n_targets),
                                                      _i64(npix), _p(out)), "per_target")
    return out


def uniform_knots(n_ctrl, degree, dtype=np.float32):
    """artist/nurbs/surfaces.py:128-147 - clamped uniform knot vector (torch.linspace values)."""
    import torch

    tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[np.dtype(dtype)]
    k = torch.zeros(n_ctrl + degree + 1, dtype=tdt)
    k[degree:-degree] = torch.linspace(0, 1, n_ctrl - degree + 1, dtype=tdt)
    k[-degree:] = 1
    return k.numpy()


def _nurbs_common(cp, uv, degrees, knots_u, knots_v, dt):
    cp, uv = _c(cp, dt), _c(uv, dt)
    H, F, nu, nv, _ = cp.shape
    M = uv.shape[2]
    p, q = int(degrees[0]), int(degrees[1])
    if knots_u is None:
        knots_u = uniform_knots(nu, p, dt)
    if knots_v is None:
        knots_v = uniform_knots(nv, q, dt)
    ku = _c(np.broadcast_to(np.asarray(knots_u, dtype=dt), (H, F, nu + p + 1)), dt)
    kv = _c(np.broadcast_to(np.asarray(knots_v, dtype=dt), (H, F, nv + q + 1)), dt)
    nuq_u = np.unique(ku.reshape(-1, ku.shape[-1]), axis=1).shape[1]
    nuq_v = np.unique(kv.reshape(-1, kv.shape[-1]), axis=1).shape[1]
    return cp, uv, ku, kv, H, F, M, nu, nv, p, q, nuq_u, nuq_v


def nurbs_fwd(cp, uv, degrees, canting=None, translations=None, knots_u=None, knots_v=None, uniform=True):
    dt = cp.dtype
    cp, uv, ku, kv, H, F, M, nu, nv, p, q, nuq_u, nuq_v = _nurbs_common(cp, uv, degrees, knots_u, knots_v, dt)
    cant, tr = _c(canting, dt), _c(translations, dt)
    pts = np.empty((H, F, M, 4), dtype=dt)
    nrm = np.empty((H, F, M, 4), dtype=dt)
    rc = getattr(lib(), "orc_nurbs_fwd" + _sfx(dt))(
        _p(cp), _p(uv), _p(ku), _p(kv), _p(cant), _p(tr), ctypes.c_int(p), ctypes.c_int(q), ctypes.c_int(int(uniform)),
        _i64(nuq_u), _i64(nuq_v), _i64(H), _i64(F), _i64(M), _i64(nu), _i64(nv), _p(pts), _p(nrm))
    _check(rc, "nurbs_fwd")
    return pts, nrm


def nurbs_bwd(cp, uv, degrees, g_points, g_normals, canting=None, knots_u=None, knots_v=None, uniform=True):
    dt = cp.dtype
    cp, uv, ku, kv, H, F, M, nu, nv, p, q, nuq_u, nuq_v = _nurbs_common(cp, uv, degrees, knots_u, knots_v, dt)
    cant = _c(canting, dt)
    gp, gn = _c(g_points, dt), _c(g_normals, dt)
    g_cp = np.empty_like(cp)
    rc = getattr(lib(), "orc_nurbs_bwd" + _sfx(dt))(
        _p(cp), _p(uv), _p(ku), _p(kv), _p(cant), ctypes.c_int(p), ctypes.c_int(q), ctypes.c_int(int(uniform)),
        _i64(nuq_u), _i64(nuq_v), _i64(H), _i64(F), _i64(M), _i64(nu), _i64(nv), _p(gp), _p(gn), _p(g_cp))
    _check(rc, "nurbs_bwd")
    return g_cp


def find_spans(x, knots, n_ctrl, degree, uniform=True):
    """Span search alone (artist/nurbs/surfaces.py:157-245) via a degenerate 1-facet evaluation is
    overkill; restated here in numpy on top of the same formulae for the known-answer test."""
    x = np.asarray(x)
    knots = np.asarray(knots)
    if uniform:
        n_unique = len(np.unique(knots))
        return (np.floor(x * x.dtype.type(n_unique - 1)).astype(np.int64) + degree)
    out = np.full(x.shape, degree, dtype=np.int64)
    for i, xv in enumerate(x.flat):
        for k in range(degree, n_ctrl):
            if xv >= knots[k] and xv < knots[k + 1]:
                out.flat[i] = k
                break
        if abs(xv - knots[n_ctrl]) <= 1e-5 + 1e-5 * abs(knots[n_ctrl]):
            out.flat[i] = n_ctrl - 1
    return out


def sampler_indices(n_samples, n_active_heliostats, world_size, rank):
    buf = np.empty(max(int(n_samples), 1), dtype=np.int64)
    n = lib().orc_sampler_indices(_i64(n_samples), _i64(n_active_heliostats), _i64(world_size), _i64(rank),
                                  _p(buf), _i64(buf.size))
    if n < 0:
        raise RuntimeError("sampler buffer too small")
    return buf[:n].copy()


# ---- stage entry points (known-answer tests) ------------------------------------------------
def scatter(e, u, dirs):
    """rotate_distortions(e, u) @ d for N independent triples (broadcast by the caller)."""
    dt = dirs.dtype
    e, u, dirs = _c(e, dt).reshape(-1), _c(u, dt).reshape(-1), _c(dirs, dt).reshape(-1, 4)
    assert e.shape == u.shape and e.shape[0] == dirs.shape[0]
    out = np.empty_like(dirs)
    _check(getattr(lib(), "orc_scatter" + _sfx(dt))(_p(e), _p(u), _p(dirs), _i64(e.shape[0]), _p(out)), "scatter")
    return out


def line_plane(dirs, mags, origins, centers, plane_normals, dims, target, resolution=(256, 256)):
    dt = dirs.dtype
    dirs, mags, origins = _c(dirs, dt).reshape(-1, 4), _c(mags, dt).reshape(-1), _c(origins, dt).reshape(-1, 4)
    N = dirs.shape[0]
    outs = [np.empty(N, dtype=dt) for _ in range(4)]
    rc = getattr(lib(), "orc_line_plane" + _sfx(dt))(
        _p(dirs), _p(mags), _p(origins), _i64(N), _p(_c(centers, dt)), _p(_c(plane_normals, dt)), _p(_c(dims, dt)),
        _i64(target), _i64(int(resolution[0])), _i64(int(resolution[1])), *[_p(o) for o in outs])
    _check(rc, "line_plane")
    return outs


def splat(e_px, u_px, inten, resolution):
    dt = e_px.dtype
    e, u, i = _c(e_px, dt).reshape(-1), _c(u_px, dt).reshape(-1), _c(inten, dt).reshape(-1)
    W, Hh = int(resolution[0]), int(resolution[1])
    out = np.empty((Hh, W), dtype=dt)
    _check(getattr(lib(), "orc_splat" + _sfx(dt))(_p(e), _p(u), _p(i), _i64(e.shape[0]), _i64(W), _i64(Hh), _p(out)),
           "splat")
    return out


def line_cylinder(dirs, mags, origins, cyl, target, resolution=(256, 256)):
    dt = dirs.dtype
    dirs, mags, origins = _c(dirs, dt).reshape(-1, 4), _c(mags, dt).reshape(-1), _c(origins, dt).reshape(-1, 4)
    N = dirs.shape[0]
    outs = [np.empty(N, dtype=dt) for _ in range(4)]
    cargs, _keep = _cyl_args(cyl, dt)
    rc = getattr(lib(), "orc_line_cylinder" + _sfx(dt))(
        _p(dirs), _p(mags), _p(origins), _i64(N), *cargs[:6], _i64(target), _i64(int(resolution[0])),
        _i64(int(resolution[1])), *[_p(o) for o in outs])
    _check(rc, "line_cylinder")
    return outs


def cyl_tables(d):
    """The cylinder tables of a golden fixture as the ``cyl=`` argument."""
    return dict(centers=d["cyl_centers"], normals=d["cyl_normals"], axes=d["cyl_axes"], radii=d["cyl_radii"],
                heights=d["cyl_heights"], opening=d["cyl_opening"])
