"""ctypes binding of ``libcar_hip.so`` (C ABI declared in ``include/car_hip.h``).

The library is built in-tree by ``__graft_entry__.build()``.  There is no fallback: if it cannot be loaded,
``load()`` raises, and every wrapper turns a negative return code into ``RuntimeError(car_last_error())``.
ctypes releases the GIL around each call; all work is enqueued on the stream the caller passes.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_size_t, c_void_p
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcar_hip.so")

_P = c_void_p


class CarDims(ctypes.Structure):          # struct car_dims
    _fields_ = [("b", c_int), ("V", c_int), ("R", c_int), ("P", c_int), ("H", c_int), ("W", c_int), ("n_levels", c_int),
                ("level_h", c_int * 4), ("level_w", c_int * 4), ("level_c", c_int * 4), ("repeat_attention", c_int), ("no_sample", c_int)]


# struct car_weights: device pointers in declaration order (attribute path on the module, flattened to [out, in])
WEIGHT_FIELDS = (
    ["query_encode_latent", "query_encode_latent_2", "latent_value", "key_map", "key_map_2", "query_embed", "query_embed_2",
     "query_repeat_embed", "query_repeat_embed_2", "encode_latent", "phi.lin_in", "phi.lin_out"],
    ["phi.lin_z", "phi.blocks.fc_0", "phi.blocks.fc_1"])


class CarWeights(ctypes.Structure):
    _fields_ = ([(f"{n.replace('.', '_')}_{k}", _P) for n in WEIGHT_FIELDS[0] for k in ("w", "b")]
                + [("phi_lin_z_w", _P * 3), ("phi_lin_z_b", _P * 3), ("phi_fc_0_w", _P * 3), ("phi_fc_0_b", _P * 3),
                   ("phi_fc_1_w", _P * 3), ("phi_fc_1_b", _P * 3)])


class CarInputs(ctypes.Structure):
    _fields_ = [("poses", _P), ("uv", _P), ("lattice", _P), ("gmeta", _P), ("steps", _P)]


class CarOutputs(ctypes.Structure):
    _fields_ = [("rgb", _P), ("valid_mask", _P), ("depth_ray", _P), ("at_wt", _P), ("at_wt_max", _P), ("coords", _P),
                ("pixel_val", _P)]


# name -> (restype, argtypes); mirrors include/car_hip.h line by line
SIGNATURES = {
    "car_version": (c_int, []),
    "car_last_error": (c_char_p, []),
    "car_device_cu_count": (c_int, []),
    "car_pose_setup": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P]),
    "car_ray_setup": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "car_sample_setup": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P,
                                 c_int, c_int, _P, _P]),
    "car_gather_encode_rows": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, _P, c_int, c_long, _P, c_int, _P]),
    "car_merge_lattice": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "car_merge_lattice_max": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P]),
    "car_lattice_encode_rows": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_long, _P, c_int, _P]),
    "car_lattice_encode_linear": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_long, _P, _P, c_int, c_int, _P, c_int, c_int, _P]),
    "car_project_points": (c_int, [_P, _P, c_int, c_long, c_int, c_int, c_int, c_int, _P, _P]),
    "car_gather_bilinear": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, c_long, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "car_gather_encode": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_long, _P, c_int, _P]),
    "car_fused_blob_floats": (c_size_t, []),
    "car_fused_bias_floats": (c_size_t, []),
    "car_fused_samples": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                  _P, _P, _P, _P, _P, _P]),
    "car_exchange_rows": (c_int, [_P, _P, _P, _P, c_int, c_int, c_long, c_int, c_int, _P, _P, _P, _P]),
    "car_fused_pack_rows": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "car_fused_rows": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "car_fused_tile_steps": (c_int, []),
    "car_fused_samples_parts": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        _P, _P, _P, _P, _P, _P, _P]),
    "car_linear_packed_floats": (c_size_t, [c_int, c_int]),
    "car_linear_pack": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P]),
    "car_linear": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int, c_long, c_int, _P]),
    "car_kq_tail_floats": (c_size_t, []),
    "car_kq_bias_floats": (c_size_t, []),
    "car_kq_pack": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "car_key_query_logits": (c_int, [_P, c_int, _P, _P, c_int, _P, _P, _P, c_long, _P, _P, _P]),
    "car_attend": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_float, _P, _P, c_int, c_int,
                           _P, _P, _P, _P, _P]),
    "car_attend_parts": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "car_round2_packed_floats": (c_size_t, []),
    "car_round2_bias_floats": (c_size_t, []),
    "car_round2_logits": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "car_round2q_packed_floats": (c_size_t, []),
    "car_round2q_bias_floats": (c_size_t, []),
    "car_round2_logits_from_g": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "car_round2q_pack": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "car_fused_pack": (c_int, [ctypes.POINTER(CarWeights), _P, _P, _P, _P]),
    "car_round2_pack": (c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "car_add_ray_bias_relu": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "car_finalize": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "car_chain_packed_floats": (c_size_t, [c_int, c_int]),
    "car_linear_x3_packed_floats": (c_size_t, [c_int, c_int]),
    "car_linear_x3_pack": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "car_linear_x3": (c_int, [_P, c_int, _P, _P, c_int, c_int, _P, c_int, c_long, c_int, _P]),
    "car_linear_x3_masked": (c_int, [_P, c_int, _P, _P, c_int, c_int, _P, c_int, c_long, c_int, _P, c_int, _P]),
    "car_chain_pack": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "car_ray_mid": (c_int, [_P, _P, _P, c_int, _P, _P, _P, c_int, _P, c_int, _P, _P, c_long, _P]),
    "car_ray_tail": (c_int, [_P, _P, _P, c_int, _P, _P, _P, c_int, _P, c_int, _P, c_int, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "car_plan_bytes": (c_size_t, [ctypes.POINTER(CarDims)]),
    "car_plan_build": (c_int, [ctypes.POINTER(CarDims), ctypes.POINTER(CarWeights), _P, _P]),
    "car_gmaps_floats": (c_size_t, [ctypes.POINTER(CarDims)]),
    "car_lattice_shape": (c_int, [ctypes.POINTER(CarDims), ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "car_gmeta_offset": (c_size_t, [ctypes.POINTER(CarDims)]),
    "car_workspace_find": (c_int, [ctypes.POINTER(CarDims), c_char_p, ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t)]),
    "car_profile_enable": (None, [c_int]),
    "car_profile_count": (c_int, []),
    "car_profile_read": (c_int, [c_int, ctypes.POINTER(c_char_p), ctypes.POINTER(c_float)]),
    "car_profile_reset": (None, []),
    "car_project_maps": (c_int, [ctypes.POINTER(CarDims), _P, _P, _P, _P]),
    "car_workspace_bytes": (c_size_t, [ctypes.POINTER(CarDims)]),
    "car_render_forward": (c_int, [ctypes.POINTER(CarDims), _P, ctypes.POINTER(CarInputs), ctypes.POINTER(CarOutputs), _P,
                                   c_size_t, _P]),
    "car_render_forward_phase": (c_int, [ctypes.POINTER(CarDims), _P, ctypes.POINTER(CarInputs), ctypes.POINTER(CarOutputs), _P,
                                         c_size_t, c_int, _P]),
    "car_linspace": (None, [c_float, c_float, c_int, _P]),
    "car_linear_wgrad": (c_int, [_P, c_int, _P, c_int, c_long, c_int, c_int, c_int, _P, c_int, _P, _P]),
    "car_attend_backward": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P, c_int, _P, _P]),
    "car_gather_bilinear_backward": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, c_long, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "car_scatter_workspace_bytes": (c_size_t, [_P, _P, c_int, c_int, c_long, c_int]),
    "car_gather_bilinear_backward_binned": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, _P, c_int, c_long, c_int, _P, c_int, c_int, _P, c_size_t, _P]),
    "car_relu_mask": (c_int, [_P, c_int, _P, c_int, c_long, c_int, _P]),
    "car_scale_rows": (c_int, [_P, c_int, _P, c_int, _P, c_long, c_float, c_long, c_int, c_int, _P]),
    "car_add": (c_int, [_P, c_int, _P, c_int, c_float, _P, c_int, c_float, c_long, c_int, _P]),
    "car_reduce_samples": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
}

_lib: Optional[ctypes.CDLL] = None


def source_hash() -> str:
    """sha256 over the kernel sources and the public header: build() stores it next to the library, load() refuses a library built
    from anything else (a pushed .so can not silently lag behind the tree)."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    files.append(os.path.join(os.path.dirname(here), "include", "car_hip.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _rebuild_stale() -> None:
    """The library on disk was built from other sources than the tree's: rebuild it with the tree's own build() when hipcc is here
    (always the real HIP library, never a substitute), refuse otherwise."""
    import fcntl
    import importlib.util
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    entry = os.path.join(root, "__graft_entry__.py")
    if os.path.exists(entry) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        # one builder at a time: ranks spawned together (experiment_scripts --gpus N, bench.py under torchrun) all land here at once.
        # The others wait on the lock, find the stamp current and load the library the first one linked (build_library links to a
        # temporary file and renames it into place, so a concurrent dlopen never sees a half-written .so).
        with open(LIB_PATH + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if not _stamp_current():
                    spec = importlib.util.spec_from_file_location("_car_graft_entry", entry)
                    mod = importlib.util.module_from_spec(spec)
                    spec.loader.exec_module(mod)
                    mod.build_library()
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
        if _stamp_current():
            return
    raise RuntimeError(f"{LIB_PATH} was built from other sources than the ones in this tree (csrc/, include/car_hip.h): rebuild it "
                       "with `python __graft_entry__.py`")


def _stamp_current() -> bool:
    stamp = LIB_PATH + ".srchash"
    try:
        return os.path.exists(LIB_PATH) and open(stamp).read().strip() == source_hash()
    except OSError:
        return False


def load() -> ctypes.CDLL:
    """Loads the HIP library (once) and sets the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported first: its wheel bundles its own libamdhip64.so, and the process must end up with ONE
    # HIP runtime (the one that owns torch's device context and streams) — loading ours first binds /opt/rocm's copy.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "The render path has no CPU/PyTorch fallback.")
    if not _stamp_current():                      # a missing stamp counts as stale: the library's sources are unknown
        _rebuild_stale()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check_exports() -> None:
    """Every symbol the header declares must be exported (used by build() and the CPU test-suite)."""
    lib = load()
    missing = [n for n in SIGNATURES if not hasattr(lib, n)]
    if missing:
        raise RuntimeError(f"libcar_hip.so lacks symbols: {missing}")
    if lib.car_version() < 300:
        raise RuntimeError("libcar_hip.so is older than the Python package")


def check(code: int, what: str = "") -> None:
    if code < 0:
        msg = load().car_last_error()
        raise RuntimeError(f"{what or 'libcar_hip'} failed ({code}): {msg.decode() if msg else '?'}")
