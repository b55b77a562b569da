"""ctypes binding of libmgacbam.so (C ABI: include/mgacbam.h).  There is NO fallback: if the library is missing
or an entry point is absent, loading raises -- device tensors never take another path."""
from __future__ import annotations

import ctypes as C
import os
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGACBAM_LIB") or os.path.join(_PKG, "libmgacbam.so")   # MGACBAM_LIB: A/B builds in tuning sweeps
ABI_VERSION = 14
MAX_LEVELS = 8
F32, F16, BF16 = 0, 1, 2
E_NULL, E_SHAPE, E_DTYPE, E_ALIGN, E_LEVELS, E_SIZE = -1, -2, -3, -4, -5, -6
# stage bit masks (include/mgacbam.h)
FWD_STAGES = dict(pool=1, chan=2, apply=4)
BWD_STAGES = dict(reduce1=1, convT=2, reduce2=4, wsa=8, params=16, apply=32)
BWD_FUSE = 64
FWD_SAVE_PROJ, BWD_HAVE_PROJ, PROJ_MAX_HIDDEN = 1, 1, 4
FWD_ALL, BWD_PARAMS, BWD_INPUTS, BWD_ALL = 7, 31, 32, 127
BWD_FOLD = 128   # with BWD_ALL: transposed conv folded into the k_bwd_reduce1 launch (ctx.sync zero-filled once by the caller)
FWD_FUSE = 8   # with FWD_ALL: one launch, in-launch hand-off through ctx.sync (caller zero-fills it once)

_c_float_p = C.POINTER(C.c_float)


class Params(C.Structure):                       # mgacbam_params_t
    _fields_ = [("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("wsa", C.c_void_p), ("beta", C.c_void_p),
                ("hidden", C.c_int32), ("k", C.c_int32), ("use_sigmoid_mask", C.c_int32),
                ("tiny_thr", C.c_float), ("eps", C.c_float)]


class FwdLevel(C.Structure):                     # mgacbam_fwd_level_t
    _fields_ = [("x", C.c_void_p), ("mask", C.c_void_p), ("y", C.c_void_p), ("ctx", C.c_void_p), ("ctx_bytes", C.c_size_t),
                ("p", Params), ("B", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("dtype", C.c_int32), ("flags", C.c_int32)]


class BwdLevel(C.Structure):                     # mgacbam_bwd_level_t
    _fields_ = [("x", C.c_void_p), ("mask", C.c_void_p), ("gy", C.c_void_p), ("ctx", C.c_void_p),
                ("scratch", C.c_void_p), ("ctx_bytes", C.c_size_t), ("scratch_bytes", C.c_size_t), ("gx", C.c_void_p), ("gmask", C.c_void_p),
                ("gw1", C.c_void_p), ("gb1", C.c_void_p), ("gw2", C.c_void_p), ("gb2", C.c_void_p),
                ("gwsa", C.c_void_p), ("gbeta", C.c_void_p),
                ("p", Params), ("B", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("dtype", C.c_int32), ("flags", C.c_int32)]


class EcaParams(C.Structure):                    # mgacbam_eca_params_t
    _fields_ = [("w", C.c_void_p), ("beta", C.c_void_p), ("k", C.c_int32), ("use_sigmoid_mask", C.c_int32),
                ("tiny_thr", C.c_float), ("eps", C.c_float)]


class EcaFwdLevel(C.Structure):                  # mgacbam_eca_fwd_level_t
    _fields_ = [("x", C.c_void_p), ("mask", C.c_void_p), ("y", C.c_void_p), ("ctx", C.c_void_p), ("ctx_bytes", C.c_size_t), ("p", EcaParams),
                ("B", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("dtype", C.c_int32)]


class EcaBwdLevel(C.Structure):                  # mgacbam_eca_bwd_level_t
    _fields_ = [("x", C.c_void_p), ("mask", C.c_void_p), ("gy", C.c_void_p), ("ctx", C.c_void_p), ("scratch", C.c_void_p),
                ("ctx_bytes", C.c_size_t), ("scratch_bytes", C.c_size_t),
                ("gx", C.c_void_p), ("gmask", C.c_void_p), ("gw", C.c_void_p), ("gbeta", C.c_void_p), ("p", EcaParams),
                ("B", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("dtype", C.c_int32)]


CTX_FIELDS = ("S", "use", "den", "avg", "mx", "mavg", "valid", "amax", "h_avg", "h_mx", "ca", "planes", "cidx", "sa", "proj", "sync", "total", "status")


class CtxLayout(C.Structure):                    # mgacbam_ctx_layout_t
    _fields_ = [(n, C.c_int64) for n in CTX_FIELDS]


# every symbol include/mgacbam.h declares: (restype, argtypes)
class SegLevel(C.Structure):                     # mgaseg_level_t
    _fields_ = [("logits", C.c_void_p), ("target", C.c_void_p), ("glogits", C.c_void_p),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Ht", C.c_int32), ("Wt", C.c_int32),
                ("dtype", C.c_int32), ("scale_weight", C.c_float), ("resize", C.c_int32)]


class SegCfg(C.Structure):                       # mgaseg_cfg_t
    _fields_ = [("bce_weight", C.c_float), ("dice_weight", C.c_float), ("smooth", C.c_float), ("loss_lambda", C.c_float),
                ("use_unified_focal", C.c_int32), ("ufl_lambda", C.c_float), ("ufl_delta", C.c_float), ("ufl_gamma", C.c_float)]


SEG_MAX_LEVELS = 4
SEG_NEAREST, SEG_BILINEAR = 0, 1


class HeadParams(C.Structure):                   # mgahead_params_t
    _fields_ = [("w1", C.c_void_p), ("bn_weight", C.c_void_p), ("bn_bias", C.c_void_p), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p), ("wh", C.c_void_p), ("bh", C.c_void_p),
                ("hidden", C.c_int32), ("eps", C.c_float), ("momentum", C.c_float), ("training", C.c_int32)]


class HeadFwdLevel(C.Structure):                 # mgahead_fwd_level_t
    _fields_ = [("x", C.c_void_p), ("logits", C.c_void_p), ("ctx", C.c_void_p), ("ctx_bytes", C.c_size_t), ("p", HeadParams),
                ("B", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("dtype", C.c_int32), ("flags", C.c_int32)]


class HeadBwdLevel(C.Structure):                 # mgahead_bwd_level_t
    _fields_ = [("x", C.c_void_p), ("g_logits", C.c_void_p), ("g_logits2", C.c_void_p), ("ctx", C.c_void_p), ("scratch", C.c_void_p),
                ("ctx_bytes", C.c_size_t), ("scratch_bytes", C.c_size_t), ("gx", C.c_void_p),
                ("gw1", C.c_void_p), ("gbn_weight", C.c_void_p), ("gbn_bias", C.c_void_p), ("gwh", C.c_void_p), ("gbh", C.c_void_p),
                ("p", HeadParams), ("B", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("dtype", C.c_int32),
                ("flags", C.c_int32)]


HEAD_BWD_ACCUM_GX, HEAD_LOGITS_F32 = 1, 2


class PmgCfg(C.Structure):                       # mgapmg_cfg_t
    _fields_ = [("tau", C.c_float), ("p_min", C.c_float), ("threshold", C.c_float), ("hard", C.c_int32)]


SYMBOLS = {
    "mgacbam_abi_version": (C.c_int, []),
    "mgacbam_last_error": (C.c_char_p, []),
    "mgacbam_build_info": (C.c_char_p, []),
    "mgacbam_reload_env": (None, []),
    "mgacbam_ctx_bytes": (C.c_size_t, [C.c_int] * 5),
    "mgacbam_bwd_scratch_bytes": (C.c_size_t, [C.c_int] * 6),
    "mgacbam_ctx_layout": (C.c_int, [C.c_int] * 5 + [C.POINTER(CtxLayout)]),
    "mgacbam_forward": (C.c_int, [C.POINTER(FwdLevel), C.c_int, C.c_void_p]),
    "mgacbam_backward": (C.c_int, [C.POINTER(BwdLevel), C.c_int, C.c_void_p]),
    "mgacbam_forward_stages": (C.c_int, [C.POINTER(FwdLevel), C.c_int, C.c_int, C.c_void_p]),
    "mgacbam_backward_stages": (C.c_int, [C.POINTER(BwdLevel), C.c_int, C.c_int, C.c_void_p]),
    "mgacbam_resize_nearest": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "mgacbam_eca_ctx_bytes": (C.c_size_t, [C.c_int] * 4),
    "mgacbam_eca_scratch_bytes": (C.c_size_t, [C.c_int] * 4),
    "mgacbam_eca_forward": (C.c_int, [C.POINTER(EcaFwdLevel), C.c_int, C.c_void_p]),
    "mgacbam_eca_backward": (C.c_int, [C.POINTER(EcaBwdLevel), C.c_int, C.c_void_p]),
    "mgaseg_ws_bytes": (C.c_size_t, [C.POINTER(SegLevel), C.c_int]),
    "mgaseg_forward": (C.c_int, [C.POINTER(SegLevel), C.c_int, C.POINTER(SegCfg), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "mgaseg_backward": (C.c_int, [C.POINTER(SegLevel), C.c_int, C.POINTER(SegCfg), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "mgahead_ctx_bytes": (C.c_size_t, [C.c_int] * 5),
    "mgahead_bwd_scratch_bytes": (C.c_size_t, [C.c_int] * 5),
    "mgahead_forward": (C.c_int, [C.POINTER(HeadFwdLevel), C.c_int, C.c_void_p]),
    "mgahead_backward": (C.c_int, [C.POINTER(HeadBwdLevel), C.c_int, C.c_void_p]),
    "mgakendall_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgakendall_backward": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 7),
    "mgaseg_kendall_forward": (C.c_int, [C.POINTER(SegLevel), C.c_int, C.POINTER(SegCfg), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgaseg_kendall_backward": (C.c_int, [C.POINTER(SegLevel), C.c_int, C.POINTER(SegCfg), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6),
    "mgapmg_forward": (C.c_int, [C.c_void_p] * 5 + [C.c_size_t, C.POINTER(PmgCfg), C.c_void_p]),
    "mgapmg_backward": (C.c_int, [C.c_void_p] * 4 + [C.c_size_t, C.POINTER(PmgCfg), C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


class LibraryMissing(RuntimeError):
    pass


def load():
    """dlopen libmgacbam.so (once).  Raises LibraryMissing with the build command if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise LibraryMissing(
                f"{LIB_PATH} not found: the HIP library is required (there is no fallback path). "
                "Build it with `python -m mga_yolo_amd.build` or `python -c 'import __graft_entry__ as g; g.build()'`.")
        import torch  # noqa: F401  -- loads torch's libamdhip64.so.7 first so the library binds to the same HIP runtime
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise LibraryMissing(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.restype, fn.argtypes = res, args
        v = lib.mgacbam_abi_version()
        if v != ABI_VERSION:
            raise LibraryMissing(f"{LIB_PATH} has ABI version {v}, expected {ABI_VERSION}; rebuild it")
        _lib = lib
        return lib


ENV_EPOCH = 0      # bumped by reload_env(): launch geometry may have changed, so pooled hand-off state of older epochs is not reused


def reload_env():
    """Make the library re-read its MGACBAM_* knobs (they are read once; tests and tuning sweeps change them in-process)."""
    global ENV_EPOCH
    load().mgacbam_reload_env()
    ENV_EPOCH += 1
    _size_cache.clear()          # scratch sizes depend on the launch geometry (tile counts), which the knobs can change


def available() -> bool:
    return os.path.exists(LIB_PATH)


def check(rc: int, what: str):
    if rc != 0:
        msg = load().mgacbam_last_error().decode(errors="replace")
        kind = "argument error" if rc < 0 else "HIP error"
        if rc == E_SIZE:
            kind = "work buffer too small (MGACBAM_E_SIZE)"
        raise RuntimeError(f"{what}: {kind} {rc}: {msg}")


_size_cache = {}     # (fn, shape...) -> bytes: the eager path asks on every call; the answer only depends on the shape


def ctx_bytes(B, Cc, H, W, hidden) -> int:
    key = ("ctx", B, Cc, H, W, hidden)
    n = _size_cache.get(key)
    if n is None:
        n = load().mgacbam_ctx_bytes(B, Cc, H, W, hidden)
        if n == 0:
            check(-2, "mgacbam_ctx_bytes")
        _size_cache[key] = n
    return n


def scratch_bytes(B, Cc, H, W, hidden, k) -> int:
    key = ("scratch", B, Cc, H, W, hidden, k)
    n = _size_cache.get(key)
    if n is None:
        n = load().mgacbam_bwd_scratch_bytes(B, Cc, H, W, hidden, k)
        if n == 0:
            check(-2, "mgacbam_bwd_scratch_bytes")
        _size_cache[key] = n
    return n


def ctx_layout(B, Cc, H, W, hidden) -> dict:
    key = ("layout", B, Cc, H, W, hidden)
    lay = _size_cache.get(key)
    if lay is None:
        L = CtxLayout()
        check(load().mgacbam_ctx_layout(B, Cc, H, W, hidden, C.byref(L)), "mgacbam_ctx_layout")
        lay = _size_cache[key] = {n: getattr(L, n) for n in CTX_FIELDS}
    return dict(lay)
