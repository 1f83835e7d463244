"""GPU tests of the C-ABI boundary's process-wide state (``-m gpu``): streams, the sticky status word, the work
counters, and what a heliostat the device had to skip leaves in the outputs (include/artist_hip.h, "Conventions")."""
import numpy as np
import pytest
import torch

from test_gpu_parity import DEV, blocking_inputs, n, trace_inputs

pytestmark = pytest.mark.gpu


def test_two_streams_trace_concurrently(golden):
    """Two streams drive art_trace_fwd / art_trace_bwd at the same time (each with its own accumulator buffer and work
    counters): every result equals the single-stream one bit for bit."""
    from artist_amd import trace_rays
    d = golden("mid_256")
    base = trace_inputs(d)
    ref_flux, ref_fac = trace_rays(**base)
    w = torch.rand(ref_flux.shape, device=DEV)
    o0, n0 = base["origins"].clone().requires_grad_(True), base["normals"].clone().requires_grad_(True)
    f0, _ = trace_rays(**dict(base, origins=o0, normals=n0))
    ref_go, ref_gn = torch.autograd.grad(f0, (o0, n0), w)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
    results = [[], []]
    for rep in range(6):                                  # interleaved submissions: the launches overlap on the device
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                o, nn_ = base["origins"].clone().requires_grad_(True), base["normals"].clone().requires_grad_(True)
                flux, fac = trace_rays(**dict(base, origins=o, normals=nn_))
                go, gn = torch.autograd.grad(flux, (o, nn_), w)
                results[k].append((flux, fac, go, gn))
    torch.cuda.synchronize()
    for per_stream in results:
        for flux, fac, go, gn in per_stream:
            np.testing.assert_array_equal(n(flux), n(ref_flux))
            np.testing.assert_array_equal(n(fac), n(ref_fac))
            np.testing.assert_array_equal(n(go), n(ref_go))
            np.testing.assert_array_equal(n(gn), n(ref_gn))


def test_gradient_of_a_summed_loss_needs_no_copy():
    """``FluxCropPixelLoss(...).sum().backward()``: autograd hands the op an expanded scalar (stride 0), which the kernel reads
    as it is (``grad_loss_stride = 0`` of art_flux_crop_pixel_loss_bwd) - the same gradient, bit for bit, as with a materialised
    ``[B]`` vector of that value, also when the value is not 1."""
    from artist_amd.flux import FluxCropPixelLoss
    gen = torch.Generator(device=DEV).manual_seed(5)
    B, Hh, W = 12, 64, 64
    ys, xs = torch.meshgrid(torch.arange(Hh, device=DEV, dtype=torch.float32), torch.arange(W, device=DEV, dtype=torch.float32), indexing="ij")
    cx, cy = 24 + 16 * torch.rand(B, generator=gen, device=DEV), 24 + 16 * torch.rand(B, generator=gen, device=DEV)
    flux = torch.exp(-((xs[None] - cx[:, None, None]) ** 2 + (ys[None] - cy[:, None, None]) ** 2) / (2 * 5.0 ** 2)).contiguous()
    truth = torch.rand((B, Hh, W), generator=gen, device=DEV) + 0.1
    dims = torch.full((B, 2), 8.0, device=DEV)
    grads = []
    for weight in (None, 0.37):
        for materialised in (False, True):
            f = flux.clone().requires_grad_(True)
            loss = FluxCropPixelLoss.apply(f, dims, truth, 6.0, 6.0)
            if materialised:
                g, = torch.autograd.grad(loss, f, torch.full_like(loss, 1.0 if weight is None else weight))
            else:
                (loss.sum() if weight is None else loss.sum() * weight).backward()
                g = f.grad
            grads.append(n(g))
    np.testing.assert_array_equal(grads[0], grads[1])
    np.testing.assert_array_equal(grads[2], grads[3])
    assert np.abs(grads[0]).max() > 0 and not np.array_equal(grads[0], grads[2])


def test_two_streams_share_no_part_sums():
    """Small batches of the fused crop + pixel-loss pair give a bitmap several workgroups and pass their part sums through
    library-owned scratch - one per (GPU, stream).  Two streams with DIFFERENT bitmaps, submitted interleaved: every loss and
    gradient equals the one-stream result bit for bit."""
    from artist_amd.flux import FluxCropPixelLoss
    gen = torch.Generator(device=DEV).manual_seed(11)
    B, Hh, W = 24, 128, 128
    ys, xs = torch.meshgrid(torch.arange(Hh, device=DEV, dtype=torch.float32), torch.arange(W, device=DEV, dtype=torch.float32), indexing="ij")

    def case(shift):
        cx = 40 + 40 * torch.rand(B, generator=gen, device=DEV) + shift
        cy = 40 + 40 * torch.rand(B, generator=gen, device=DEV) - shift
        flux = torch.exp(-((xs[None] - cx[:, None, None]) ** 2 + (ys[None] - cy[:, None, None]) ** 2) / (2 * 9.0 ** 2)).contiguous()
        truth = torch.rand((B, Hh, W), generator=gen, device=DEV) + 0.1
        dims = torch.full((B, 2), 8.0, device=DEV)
        w = torch.rand(B, generator=gen, device=DEV)
        return flux, truth, dims, w

    def run(c):
        flux, truth, dims, w = c
        f = flux.clone().requires_grad_(True)
        loss = FluxCropPixelLoss.apply(f, dims, truth, 6.0, 6.0)
        (loss * w).sum().backward()
        return loss.detach(), f.grad

    cases = [case(0.0), case(7.0)]
    refs = [run(c) for c in cases]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
    got = [[], []]
    for rep in range(8):
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                got[k].append(run(cases[k]))
    torch.cuda.synchronize()
    for k in range(2):
        for loss, grad in got[k]:
            np.testing.assert_array_equal(n(loss), n(refs[k][0]))
            np.testing.assert_array_equal(n(grad), n(refs[k][1]))
    assert not np.array_equal(n(refs[0][0]), n(refs[1][0]))


def test_status_word_is_per_device_and_any_stream_may_clear_it(golden):
    """The status word is one per GPU: a bad target index met by a kernel of stream A makes a trace call on stream B
    refuse, and a caller on stream B can clear it (``check_async_errors`` synchronises the device)."""
    from artist_amd import _lib, ops
    d = golden("small_deg3")
    inp = trace_inputs(d)
    good, _ = ops.trace_rays(**inp)
    torch.cuda.synchronize()
    a, b = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
    bad = dict(inp, target_idx=inp["target_idx"].clone())
    bad["target_idx"][0] = 5
    with torch.cuda.stream(a):
        ops.trace_rays(**bad)                                     # asynchronous: no error yet
    a.synchronize()
    with torch.cuda.stream(b):
        with pytest.raises(IndexError, match="out of range"):
            ops.trace_rays(**inp)                                 # another stream, same GPU: refused
        with pytest.raises(IndexError, match="out of range"):
            ops.check_async_errors(DEV)                           # ... and cleared from here
        assert _lib.lib().art_async_status(b.cuda_stream, 0) == 0
        again, _ = ops.trace_rays(**inp)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(n(again), n(good))


def test_skipped_heliostat_has_zero_gradients(golden):
    """The functional API has no host-side validation: a heliostat with a stale target index is skipped on the device -
    and its rows of BOTH gradient tensors are zero (not the allocator's garbage), the others' are untouched."""
    from artist_amd import ops
    d = golden("small_deg3")
    inp = trace_inputs(d)
    H = inp["origins"].shape[0]

    def grads(target_idx, bad=False):
        o, nn_ = inp["origins"].clone().requires_grad_(True), inp["normals"].clone().requires_grad_(True)
        flux, _ = ops.trace_rays(**dict(inp, origins=o, normals=nn_, target_idx=target_idx))
        if bad:          # the forward kernels have reported the index: clear, so that the backward call starts at all
            with pytest.raises(IndexError):
                ops.check_async_errors(DEV)
        return torch.autograd.grad(flux, (o, nn_), torch.ones_like(flux))

    go_ref, gn_ref = grads(inp["target_idx"])
    # poison the allocator's free blocks of that size, so that "uninitialised" cannot look like zero by luck
    for _ in range(4):
        junk = torch.full_like(inp["origins"], float("nan"))
        del junk
    bad = inp["target_idx"].clone()
    bad[1] = 9
    go, gn = grads(bad, bad=True)
    with pytest.raises(IndexError):                               # ... and the backward kernels report it again
        ops.check_async_errors(DEV)
    assert float(go[1].abs().sum()) == 0 and float(gn[1].abs().sum()) == 0
    assert not bool(torch.isnan(go).any()) and not bool(torch.isnan(gn).any())
    for h in range(H):
        if h != 1:
            np.testing.assert_array_equal(n(go[h]), n(go_ref[h]))
            np.testing.assert_array_equal(n(gn[h]), n(gn_ref[h]))


def test_candidate_overflow_poisons_the_overflowed_heliostat(golden, monkeypatch):
    """More rectangles inside one heliostat's ray cone than its candidate ROW holds - the caller's workspace, 32 entries here (the
    kernels themselves take lists of any length: test_more_candidates_than_the_tables_hold).  (1) Whatever the host learns and when, the
    RESULTS of the call that overflowed say so: NaN bitmap and factors for that heliostat, finite ones for the others -
    shown by clearing the status word between the filter and the trace call, which is the race a caller can lose.
    (2) Unpatched, the overflow is raised by the same call or the next one, with the hint how to clear the status."""
    from artist_amd import ArtistHipError, _lib, ops, trace_rays
    monkeypatch.setattr(ops, "BLOCKING_CANDIDATES", 32)
    d = golden("small_blocking")
    inp = trace_inputs(d)
    blk = blocking_inputs(d)
    good, good_fac, flags = trace_rays(**inp, blocking=blk)
    k = int(np.nonzero(n(flags))[0][0])                          # a rectangle that does block somebody
    shifts = torch.arange(1, 41, device=DEV, dtype=torch.float32)[:, None, None] * 1e-3
    extra = blk["corners"][k][None] + shifts * blk["normals"][k][None, None, :] * torch.tensor([1.0, 1.0, 1.0, 0.0], device=DEV)
    crowded = dict(blk, corners=torch.cat([blk["corners"], extra]), spans=torch.cat([blk["spans"], blk["spans"][k][None].expand(40, -1, -1)]),
                   normals=torch.cat([blk["normals"], blk["normals"][k][None].expand(40, -1)]))
    handle = _lib.lib()
    real = handle.art_trace_fwd
    stream = torch.cuda.current_stream(DEV).cuda_stream

    def trace_after_losing_the_race(*args):
        assert handle.art_async_status(stream, 1) == -5          # the filter did report it - and somebody cleared it
        return real(*args)

    monkeypatch.setattr(handle, "art_trace_fwd", trace_after_losing_the_race)
    flux, fac, _ = trace_rays(**inp, blocking=crowded)
    monkeypatch.undo()
    monkeypatch.setattr(ops, "BLOCKING_CANDIDATES", 32)
    torch.cuda.synchronize()
    over = np.isnan(n(fac)).any(axis=0)                          # [H]
    assert over.any() and not over.all()
    for h in range(flux.shape[0]):
        if over[h]:
            assert np.isnan(n(flux[h])).all() and np.isnan(n(fac[:, h])).all()
        else:
            assert np.isfinite(n(flux[h])).all()
    assert handle.art_async_status(stream, 0) == 0
    # (2) the product path
    try:
        trace_rays(**inp, blocking=crowded)                       # asynchronous: the status may or may not be visible yet
        with pytest.raises(ArtistHipError, match="check_async_errors"):
            trace_rays(**inp, blocking=blk)                       # refused, and the message says how to go on
    except ArtistHipError as exc:
        assert "check_async_errors" in str(exc)
    with pytest.raises(ArtistHipError, match="blocking rectangles"):
        ops.check_async_errors(DEV)
    again, again_fac, _ = trace_rays(**inp, blocking=blk)
    np.testing.assert_array_equal(n(again), n(good))
    np.testing.assert_array_equal(n(again_fac), n(good_fac))


def test_many_queued_launches_share_their_counters_safely(golden):
    """1500 trace calls queued without one synchronisation (round 2's ring of 1024 work-counter slots would have
    wrapped): the counters are per stream and reset by each launch's last fetch, so the last result equals the first."""
    from artist_amd import trace_rays
    d = golden("small_deg3")
    inp = trace_inputs(d)
    first, _ = trace_rays(**inp)
    o, nn_ = inp["origins"].clone().requires_grad_(True), inp["normals"].clone().requires_grad_(True)
    w = torch.ones_like(first)
    g_first = None
    last = None
    for k in range(750):
        last, _ = trace_rays(**dict(inp, origins=o, normals=nn_))
        g = torch.autograd.grad(last, (o, nn_), w)
        g_first = g if g_first is None else g_first
    torch.cuda.synchronize()
    np.testing.assert_array_equal(n(last), n(first))
    np.testing.assert_array_equal(n(g[0]), n(g_first[0]))
    np.testing.assert_array_equal(n(g[1]), n(g_first[1]))


def _poison_lds(pattern=0x7FC00000):
    """Fill every CU's LDS with a NaN bit pattern (tests/lds_poison.hip, built by ``__graft_entry__.build``)."""
    import ctypes
    import pathlib
    so = pathlib.Path(__file__).resolve().parent / "bin" / "liblds_poison.so"
    if not so.exists():
        pytest.skip("tests/bin/liblds_poison.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    lib = ctypes.CDLL(str(so))
    lib.lds_poison.argtypes = [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
    sink = torch.zeros(1, dtype=torch.int32, device=DEV)
    assert lib.lds_poison(pattern, torch.cuda.current_stream(DEV).cuda_stream, sink.data_ptr()) == 0
    torch.cuda.synchronize()
    assert int(sink) == 0


@pytest.mark.parametrize("blocking", [False, True])
def test_results_do_not_depend_on_what_the_lds_held_before(blocking):
    """LDS is not cleared between workgroups.  Heliostats that miss their target completely have an EMPTY window: their
    rays are all masked, and the lean backward kernel used to multiply whatever the unstaged cells held by zero weights -
    NaN whenever the CU's previous workgroup had left something that looks like one (round 3: a rare, box-dependent
    failure of test_random_scenes_split_calls).  With every CU's LDS filled with NaN patterns (and with all-ones) before
    each call, flux, factors and gradients must be the bits of the unpoisoned run - with blocking off (lean kernels) and
    on (split call: lean launch for the heliostats without candidate rectangles, among them the ones that miss)."""
    from artist_amd import trace_rays
    from test_gpu_parity import _random_feature_scene
    sc = _random_feature_scene(8, 12, 77, 17)                    # heliostats 4, 9, 10, 11 reflect past both planes
    tix = (sc["target_idx"] % 2).to(DEV)
    dv = lambda x: x.to(DEV).contiguous()  # noqa: E731
    both = dv(sc["both"])
    kw = dict(ray_magnitude=0.7, extinction=0.05, reflectivity=0.9, resolution=(96, 64))
    if blocking:
        kw["blocking"] = dict({k: dv(v) for k, v in sc["prims"].items()}, lbvh_compat=False)
    w = torch.rand((12, 64, 96), generator=torch.Generator().manual_seed(8)).to(DEV)

    def run(pattern, env_blocks):
        import os
        old = os.environ.get("ARTIST_HIP_FWD_BLOCKS")
        if env_blocks:
            os.environ["ARTIST_HIP_FWD_BLOCKS"] = env_blocks      # "1": samples stay in one chunk, the backward call splits too
        try:
            o, nn_ = dv(sc["origins"]).requires_grad_(True), dv(sc["normals"]).requires_grad_(True)
            if pattern is not None:
                _poison_lds(pattern)
            out = trace_rays(o, nn_, dv(sc["incident"]), both[..., 0], both[..., 1], tix, dv(sc["planes"]["centers"]),
                             dv(sc["planes"]["normals"]), dv(sc["planes"]["dims"]), **kw)
            if pattern is not None:
                _poison_lds(pattern)
            (out[0] * w).sum().backward()
            return n(out[0]), n(out[1]), n(o.grad), n(nn_.grad)
        finally:
            if old is None:
                os.environ.pop("ARTIST_HIP_FWD_BLOCKS", None)
            else:
                os.environ["ARTIST_HIP_FWD_BLOCKS"] = old

    for env_blocks in ("1", None):
        clean = run(None, env_blocks)
        assert float(clean[1][1, 9]) == 0 and np.isfinite(clean[2]).all() and np.abs(clean[2][9]).sum() == 0
        for pattern in (0x7FC00000, 0xFFFFFFFF, 0x7F800001):
            dirty = run(pattern, env_blocks)
            for a_, b_ in zip(clean, dirty):
                np.testing.assert_array_equal(a_, b_, err_msg=f"pattern {pattern:#x}, blocks {env_blocks}")


def test_every_entry_point_is_independent_of_lds_leftovers(monkeypatch):
    """The same check for the whole boundary: every asynchronous entry point of the library is wrapped so that each of its
    calls starts with all CUs' LDS full of NaN patterns, and an epoch that touches all of them - NURBS evaluation with fused
    alignment, a mixed planar / cylindrical tower with blocking on (filter, split launches, soft mask), per-target sums,
    centre of mass, crop + pixel loss, crop + KL, and every backward pass - must give the bits of the unwrapped epoch."""
    from artist_amd import HeliostatRayTracer, NURBSSurfaces, _lib, get_center_of_mass
    from artist_amd.flux import FluxCropKLLoss, FluxCropPixelLoss
    from artist_amd.scene import SolarTower, TowerTargetAreasCylindrical, build_synthetic_scenario
    H = 10
    handle = _lib.lib()

    def epoch():
        torch.manual_seed(0)
        scenario, uv = build_synthetic_scenario(H, 12, n_cp=(6, 6), n_eval=14, device=DEV)
        planar = scenario.solar_tower.target_areas[0]
        cyl = TowerTargetAreasCylindrical(
            names=["cyl"], centers=torch.tensor([[0.0, -12.0, 55.0, 1.0]], device=DEV),
            normals=torch.tensor([[0.0, 1.0, 0.0, 0.0]], device=DEV), axes=torch.tensor([[0.0, 0.0, 1.0, 0.0]], device=DEV),
            radii=torch.tensor([12.0], device=DEV), heights=torch.tensor([30.0], device=DEV), opening_angles=torch.tensor([2.0], device=DEV))
        scenario.solar_tower = SolarTower([planar, cyl], device=DEV)
        group = scenario.heliostat_field.heliostat_groups[0]
        mask = torch.ones(H, dtype=torch.int32, device=DEV)
        tix = torch.tensor([0, 1, 0, 0, 1, 0, 0, 0, 1, 0], device=DEV)
        inc = torch.tensor([0.0, 1.0, 0.0, 0.0], device=DEV).expand(H, 4).contiguous()       # a low sun in the south ...
        group.positions[:4] = torch.tensor([[0.0, 140.0, 0.0, 1.0], [0.0, 137.0, 0.0, 1.0], [1.2, 134.0, 0.0, 1.0],
                                            [0.4, 131.0, 0.0, 1.0]], device=DEV)             # ... and a column of mirrors in each other's beams
        group.activate_heliostats(mask, DEV)
        group.align_surfaces_with_incident_ray_directions(scenario.solar_tower.get_centers_of_target_areas(tix), inc, mask, DEV)
        from artist_amd import scene
        orientation = scene.ideal_orientations(group.active_positions, scenario.solar_tower.get_centers_of_target_areas(tix), inc)
        cp = group.active_nurbs_control_points.clone().requires_grad_(True)
        pts, nrm = NURBSSurfaces(group.nurbs_degrees, cp, device=DEV).calculate_surface_points_and_normals(
            uv, group.active_canting, group.active_facet_translations, orientations=orientation)
        group.active_surface_points, group.active_surface_normals = pts.reshape(H, -1, 4), nrm.reshape(H, -1, 4)
        rt = HeliostatRayTracer(scenario, group, blocking_active=True, bitmap_resolution=torch.tensor([64, 64]))
        rt.lbvh_compat = False
        flux, intercept, on_target, unblocked = rt.trace_rays(inc, mask, tix)
        per_target = rt.get_bitmaps_per_target(flux.detach(), tix)
        dims = torch.tensor([[8.0, 8.0]], device=DEV).expand(H, 2).contiguous()
        truth = torch.rand((H, 64, 64), generator=torch.Generator().manual_seed(1)).to(DEV) + 0.1
        loss = FluxCropPixelLoss.apply(flux, dims, truth, 6.0, 6.0).sum() + FluxCropKLLoss.apply(flux, dims, truth, 6.0, 6.0).sum() \
            + get_center_of_mass(flux).sum() * 1e-3
        loss.backward()
        torch.cuda.synchronize()
        return [n(x) for x in (pts, flux, intercept, on_target, unblocked, per_target, loss, cp.grad)]

    clean = epoch()
    assert np.isfinite(clean[7]).all() and np.abs(clean[7]).sum() > 0 and (clean[4] < 1).any()
    calls = []
    for name in _lib.SIGNATURES:
        if name in ("art_abi_version", "art_last_hip_error", "art_strerror", "art_async_status", "art_blocking_workspace_bytes",
                    "art_trace_bwd_scratch_floats", "art_trace_bwd_scratch_need"):
            continue
        real = getattr(handle, name)

        def wrapped(*args, _real=real, _name=name):
            _poison_lds(0x7FC00000)
            calls.append(_name)
            return _real(*args)

        monkeypatch.setattr(handle, name, wrapped)
    dirty = epoch()
    monkeypatch.undo()
    assert {"art_nurbs_fwd", "art_nurbs_bwd", "art_trace_fwd", "art_trace_bwd", "art_blocking_filter", "art_per_target_sum",
            "art_flux_crop_pixel_loss_fwd", "art_flux_crop_pixel_loss_bwd", "art_flux_crop_kl_loss_fwd", "art_flux_crop_kl_loss_bwd",
            "art_flux_center_of_mass", "art_flux_center_of_mass_bwd", "art_reflect"} <= set(calls), sorted(set(calls))
    for a_, b_ in zip(clean, dirty):
        np.testing.assert_array_equal(a_, b_)


def test_launch_event_recorder_brackets_the_trace_calls(golden):
    """``ops.record_launch_events`` (bench.py's roofline leg): one HIP-event pair per art_trace_fwd / art_trace_bwd call, on the
    stream the call is launched on - also a side stream -, results unchanged, nothing recorded once switched off."""
    from artist_amd import ops, trace_rays
    d = golden("mid_256")
    base = trace_inputs(d)
    ref_flux, _ = trace_rays(**base)
    assert ops._LAUNCH_EVENTS is None
    rec = ops.record_launch_events(True)
    side = torch.cuda.Stream(DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    for stream in (torch.cuda.current_stream(DEV), side):
        with torch.cuda.stream(stream):
            o = base["origins"].clone().requires_grad_(True)
            flux, _ = trace_rays(**dict(base, origins=o))
            torch.autograd.grad(flux, o, torch.ones_like(flux))
            np.testing.assert_array_equal(n(flux), n(ref_flux))
    assert ops.record_launch_events(False) is rec and ops._LAUNCH_EVENTS is None
    trace_rays(**base)                                    # off again: not recorded
    torch.cuda.synchronize()
    assert sorted(rec) == ["art_trace_bwd", "art_trace_fwd"] and all(len(v) == 2 for v in rec.values())
    for pairs in rec.values():
        for start, end in pairs:
            assert 0.0 < start.elapsed_time(end) < 1e3


def test_knobs_are_ignored_outside_debug_mode(golden, monkeypatch, capfd):
    """The launch-geometry / A-B knobs of the library (ARTIST_HIP_*) are read only when ARTIST_HIP_DEBUG=1: a caller's environment
    cannot change what the product launches (round-3 review: 25 variables were read unconditionally).  Shown with the knob that
    is visible from outside, ARTIST_HIP_PRINT_GEOMETRY (one line on stderr per trace call), and with ARTIST_HIP_FWD=global,
    which selects another kernel (results differ in the last bits from the windowed one)."""
    from test_gpu_parity import n, trace_inputs

    from artist_amd import trace_rays
    inp = trace_inputs(golden("mid_256"))
    monkeypatch.setenv("ARTIST_HIP_DEBUG", "1")
    monkeypatch.delenv("ARTIST_HIP_PRINT_GEOMETRY", raising=False)
    monkeypatch.delenv("ARTIST_HIP_FWD", raising=False)
    base = n(trace_rays(**inp)[0])
    torch.cuda.synchronize()
    capfd.readouterr()
    monkeypatch.setenv("ARTIST_HIP_PRINT_GEOMETRY", "1")
    monkeypatch.setenv("ARTIST_HIP_FWD", "global")
    debug = n(trace_rays(**inp)[0])
    torch.cuda.synchronize()
    capfd.readouterr()
    monkeypatch.setenv("ARTIST_HIP_FWD", "lds")
    trace_rays(**inp)
    torch.cuda.synchronize()
    assert "art_trace_fwd:" in capfd.readouterr().err                     # debug mode: the knob is honoured
    monkeypatch.setenv("ARTIST_HIP_FWD", "global")
    monkeypatch.setenv("ARTIST_HIP_DEBUG", "0")
    plain = n(trace_rays(**inp)[0])
    torch.cuda.synchronize()
    assert "art_trace_fwd:" not in capfd.readouterr().err                 # product mode: ignored
    np.testing.assert_array_equal(plain, base)                             # ... the windowed kernel ran, not the global-atomic one
    assert np.abs(debug - base).max() <= 1e-5 * np.abs(base).max()


def test_a_work_counter_left_behind_is_reset_and_reported(golden, monkeypatch):
    """The persistent kernels pull their work items from counters that the launch's last fetch puts back to zero (no memset per
    launch).  A launch that ends abnormally would leave a counter behind and every later launch on that stream would skip or repeat
    items - silently (advisor, round 3).  Now the call's first kernel checks the stream's counters, puts them back to zero and
    raises the sticky status (ART_EQUEUE): shown by corrupting a counter the way an aborted launch would
    (ARTIST_HIP_CORRUPT_COUNTER, debug mode): the call that meets it still computes the right bitmaps, the status is reported,
    later trace calls refuse to start until it is cleared, and then all is well."""
    from test_gpu_parity import n, trace_inputs

    from artist_amd import ArtistHipError, ops, trace_rays
    inp = trace_inputs(golden("mid_256"))
    base = n(trace_rays(**inp)[0])
    ops.check_async_errors()
    monkeypatch.setenv("ARTIST_HIP_DEBUG", "1")
    monkeypatch.setenv("ARTIST_HIP_CORRUPT_COUNTER", "1")
    hit = n(trace_rays(**inp)[0])
    monkeypatch.delenv("ARTIST_HIP_CORRUPT_COUNTER")
    np.testing.assert_array_equal(hit, base)                   # the counter was reset before the trace kernel fetched from it
    with pytest.raises(ArtistHipError, match="work counter"):
        trace_rays(**inp)                                      # sticky: nothing is launched on top of a suspect state
    with pytest.raises(ArtistHipError, match="work counter"):
        ops.check_async_errors()                               # ... reported (and cleared) here
    np.testing.assert_array_equal(n(trace_rays(**inp)[0]), base)
    ops.check_async_errors()
