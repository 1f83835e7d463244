"""ctypes binding of ``libartist_hip.so`` (C ABI declared in ``include/artist_hip.h``).

The HIP library is the product path and there is no fallback: if the shared object is missing
or fails to load, importing any op raises ``ArtistHipError`` - loudly - instead of silently
computing on the CPU.
"""
from __future__ import annotations

import ctypes
import os
import pathlib
import subprocess

_PKG = pathlib.Path(__file__).resolve().parent
LIB_PATH = pathlib.Path(os.environ.get("ARTIST_HIP_LIB", _PKG / "libartist_hip.so"))   # override: diagnostic builds only
CSRC = _PKG / "csrc"

ABI_VERSION = 13


class ArtistHipError(RuntimeError):
    """Raised when libartist_hip.so is unavailable or an entry point reports an error."""


_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_dbl = ctypes.c_double
_ptr = ctypes.c_void_p

# name -> argtypes (restype is int for every entry point except the two noted below);
# mirrors include/artist_hip.h one-to-one (tests/test_boundary.py checks every symbol).
SIGNATURES = {
    "art_trace_fwd": [_ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_i64, _c_i64, _ptr, _ptr, _ptr, _ptr,
                      _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                      _ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_dbl,
                      _c_dbl, _c_dbl, _c_dbl, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int,
                      _ptr, _ptr, _ptr, _ptr, _ptr],
    "art_async_status": [_ptr, _c_int],
    "art_trace_bwd": [_ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_i64, _c_i64, _ptr, _ptr, _ptr, _ptr,
                      _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                      _ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_i64, _c_dbl,
                      _c_dbl, _c_dbl, _c_dbl, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int,
                      _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _ptr],
    "art_trace_bwd_scratch_floats": [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64],
    "art_trace_bwd_scratch_need": [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64],
    "art_flux_crop_fwd": [_ptr, _ptr, _c_i64, _c_i64, _c_i64, _c_dbl, _c_dbl, _ptr, _ptr, _ptr],
    "art_flux_crop_bwd": [_ptr, _ptr, _ptr, _c_i64, _c_i64, _c_i64, _c_dbl, _c_dbl, _ptr, _ptr, _ptr, _ptr],
    "art_flux_loss": [_ptr, _ptr, _c_i64, _c_i64, _c_int, _ptr, _ptr, _ptr, _ptr],
    "art_flux_crop_pixel_loss_fwd": [_ptr, _ptr, _ptr, _c_i64, _c_i64, _c_i64, _c_dbl, _c_dbl, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "art_flux_crop_pixel_loss_bwd": [_ptr, _ptr, _ptr, _c_i64, _ptr, _ptr, _c_i64, _c_i64, _c_i64, _c_dbl, _c_dbl, _ptr, _ptr],
    "art_flux_crop_kl_loss_fwd": [_ptr, _ptr, _ptr, _c_i64, _c_i64, _c_i64, _c_dbl, _c_dbl, _ptr, _ptr, _ptr],
    "art_flux_crop_kl_loss_bwd": [_ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_i64, _c_i64, _c_dbl, _c_dbl, _ptr, _ptr, _ptr],
    "art_flux_center_of_mass": [_ptr, _c_i64, _c_i64, _c_i64, _ptr, _ptr],
    "art_flux_center_of_mass_bwd": [_ptr, _ptr, _c_i64, _c_i64, _c_i64, _ptr, _ptr],
    "art_rigid_body_fwd": [_c_int, _ptr, _ptr, _ptr, _ptr, _c_i64, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_int, _c_dbl,
                           _ptr, _ptr, _ptr, _ptr, _ptr],
    "art_rigid_body_bwd": [_c_int, _ptr, _ptr, _ptr, _ptr, _c_i64, _ptr, _ptr, _ptr, _ptr, _c_i64, _ptr, _ptr, _ptr,
                           _ptr, _ptr, _ptr, _ptr, _ptr],
    "art_blocking_workspace_bytes": [_c_i64, _c_i64],
    "art_blocking_filter": [_ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_i64, _c_i64, _ptr, _ptr, _ptr, _ptr,
                            _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_dbl,
                            _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64,
                            _ptr, _ptr, _c_i64, _c_dbl, _c_int, _c_i64, _ptr, _ptr, _ptr, _ptr, _ptr],
    "art_per_target_sum": [_ptr, _ptr, _c_i64, _c_i64, _c_i64, _ptr, _ptr],
    "art_nurbs_fwd": [_ptr, _ptr, _c_i64, _c_i64, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_i64, _c_i64,
                      _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _ptr, _ptr, _ptr, _ptr],
    "art_nurbs_bwd": [_ptr, _ptr, _c_i64, _c_i64, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_i64, _c_i64,
                      _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _ptr, _ptr, _ptr, _ptr, _ptr],
    "art_reflect": [_ptr, _ptr, _c_i64, _c_i64, _ptr, _ptr],
    "art_adam_step": [_ptr, _ptr, _ptr, _ptr, _c_i64, _c_dbl, _c_dbl, _c_dbl, _c_dbl, _c_dbl, _c_i64, _c_int, _c_i64, _c_i64, _ptr],
    "art_align_fwd": [_ptr, _ptr, _ptr, _c_i64, _c_i64, _ptr, _ptr, _ptr],
    "art_align_bwd": [_ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_i64, _ptr, _ptr, _ptr, _ptr],
    "art_abi_version": [],
    "art_last_hip_error": [],
    "art_strerror": [_c_int],
}

_LIB = None


def build(verbose: bool = False) -> pathlib.Path:
    """Compile every HIP source for gfx950 into artist_amd/libartist_hip.so (in-tree)."""
    cmd = ["make", "-C", str(CSRC), "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise ArtistHipError(f"building libartist_hip.so failed:\n{res.stdout}\n{res.stderr}")
    if not LIB_PATH.exists():
        raise ArtistHipError(f"build finished but {LIB_PATH} is missing")
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """Load the library (once).  torch is imported first so that the HIP runtime already in
    the process (torch/lib/libamdhip64.so, soname libamdhip64.so.7) is the one our DT_NEEDED
    entry resolves to - two HIP runtimes in one process cannot share streams or allocations."""
    global _LIB
    if _LIB is not None:
        return _LIB
    import torch  # noqa: F401  (loads torch's HIP runtime first)

    if not LIB_PATH.exists():
        raise ArtistHipError(
            f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {CSRC}`. There is no CPU fallback.")
    try:
        handle = ctypes.CDLL(str(LIB_PATH))
    except OSError as exc:  # pragma: no cover - depends on the host
        raise ArtistHipError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as exc:
            raise ArtistHipError(f"{LIB_PATH} does not export {name}") from exc
        fn.argtypes = argtypes
        fn.restype = (ctypes.c_char_p if name == "art_strerror"
                      else ctypes.c_int64 if name in ("art_blocking_workspace_bytes", "art_trace_bwd_scratch_floats", "art_trace_bwd_scratch_need")
                      else ctypes.c_int)
    if handle.art_abi_version() != ABI_VERSION:
        raise ArtistHipError(f"ABI mismatch: library {handle.art_abi_version()} vs binding {ABI_VERSION}")
    _LIB = handle
    return handle


def check(code: int, what: str) -> None:
    if code != 0:
        handle = lib()
        msg = handle.art_strerror(code).decode()
        raise ArtistHipError(f"{what}: {msg} (code {code}, hipError {handle.art_last_hip_error()})")


def loaded_hip_runtimes() -> list[str]:
    """Paths of libamdhip64 images mapped into this process (must be exactly one)."""
    paths = set()
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                paths.add(line.split()[-1])
    return sorted(paths)
