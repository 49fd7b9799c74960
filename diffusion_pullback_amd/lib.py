"""ctypes binding of libdpb.so (the C ABI declared in include/dpb.h).

There is NO fallback: if the HIP library is missing or a symbol cannot be bound this
module raises, so nothing in the product path can silently run on the CPU.
"""
from __future__ import annotations

import ctypes as C
import glob
import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DPB_LIB") or os.path.join(_HERE, "libdpb.so")   # DPB_LIB: another build of the same ABI (same-session A/B of kernel variants)
CSRC = os.path.join(_HERE, "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
STAMP_PATH = LIB_PATH + ".srchash"          # content hash of the sources the .so was built from (git-ignored, travels with gpurun)

DPB_F32, DPB_BF16, DPB_F16 = 0, 1, 2


def dtype_code(dtype) -> int:
    """torch dtype of the engine (storage + MFMA input type) -> DPB_* code of include/dpb.h"""
    import torch
    try:
        return {torch.float32: DPB_F32, torch.bfloat16: DPB_BF16, torch.float16: DPB_F16}[dtype]
    except KeyError:
        raise DpbError(f"engine dtype must be float32, bfloat16 or float16, got {dtype}") from None
OP_CONV, OP_GROUPNORM, OP_LAYERNORM, OP_ATTENTION, OP_GEGLU, OP_SILU, OP_CONCAT = 1, 2, 3, 4, 5, 6, 7
GATHER_NONE, GATHER_CONV, GATHER_UPCONV = 0, 1, 3
BUF_ACT, BUF_SHARED = 0, 1


class BufferDesc(C.Structure):
    _fields_ = [("rows", C.c_int32), ("channels", C.c_int32), ("kind", C.c_int32), ("valid_channels", C.c_int32)]


class OpDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in0", C.c_int32), ("in1", C.c_int32), ("in2", C.c_int32), ("out", C.c_int32),
                ("res", C.c_int32), ("rowbias", C.c_int32), ("ip", C.c_int32 * 12), ("fp", C.c_float * 4),
                ("w", C.c_void_p * 4)]


class NetDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("max_batch", C.c_int32), ("max_tangents", C.c_int32), ("n_buffers", C.c_int32),
                ("n_ops", C.c_int32), ("buffers", C.POINTER(BufferDesc)), ("ops", C.POINTER(OpDesc)),
                ("x_buf", C.c_int32), ("x_channels", C.c_int32), ("temb_buf", C.c_int32), ("temb_dim", C.c_int32),
                ("temb_flip_sin_to_cos", C.c_int32), ("temb_half_minus_one", C.c_int32), ("ctx_buf", C.c_int32)]


# every symbol include/dpb.h declares: name -> (restype, argtypes)
_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64
SYMBOLS = {
    "dpb_last_error": (C.c_char_p, []),
    "dpb_abi_version": (_I, []),
    "dpb_engine_create": (_I, [C.POINTER(NetDesc), C.POINTER(_P)]),
    "dpb_engine_destroy": (None, [_P]),
    "dpb_engine_set_stream": (_I, [_P, _P]),
    "dpb_engine_workspace_bytes": (C.c_size_t, [_P]),
    "dpb_engine_set_workspace": (_I, [_P, _P, C.c_size_t]),
    "dpb_primal": (_I, [_P, _P, _I, _F, _P, _I]),
    "dpb_forward": (_I, [_P, _P, _I, _F, _P, _I, _I, _P]),
    "dpb_read_buffer": (_I, [_P, _I, _I, _P]),
    "dpb_jvp": (_I, [_P, _I, _P, _I, _P]),
    "dpb_vjp": (_I, [_P, _I, _P, _I, _P]),
    "dpb_orth": (_I, [_P, _P, _P, _P, _P, _P, _I, _L, _P]),
    "dpb_orth_scratch_bytes": (C.c_size_t, [_I, _L]),
    "dpb_orth_checked": (_I, [_P, _P, _P, _P, _P, _P, C.c_size_t, _I, _L, _P]),
    "dpb_pullback_iterate": (_I, [_P, _I, _P, _P, _P, _P, _I, _I]),
    "dpb_ddim_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _P]),
    "dpb_lincomb": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _P]),
    "dpb_embed_tokens": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "dpb_engine_stats": (_I, [_P, C.POINTER(_L), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dpb_engine_profile": (_I, [_P, _I]),
    "dpb_debug_set": (_I, [C.c_char_p, _I]),
    "dpb_debug_gemm_plan": (_I, [_I, _I, _I, _I, _I, _I, _I, _L, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "dpb_engine_profile_dump": (_I, [_P, C.c_char_p]),
    "dpb_engine_profile_read": (_I, [_P, _I, C.POINTER(_L), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dpb_engine_profile_overhead": (_I, [_P, C.POINTER(C.c_double)]),
}

_lib = None


class DpbError(RuntimeError):
    pass


def _source_hash() -> str:
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.cpp"))
                   + [os.path.join(CSRC, "Makefile"), os.path.join(os.path.dirname(_HERE), "include", "dpb.h")])
    for f in files:
        if os.path.exists(f):
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def _built_hash() -> str:
    try:
        with open(STAMP_PATH) as fh:
            return fh.read().strip() if os.path.exists(LIB_PATH) else ""
    except OSError:
        return ""


def _needs_build() -> bool:
    """Stale or missing binary?  Only the content-hash stamp written by build() vouches for a libdpb.so: a binary without a stamp, or with another
    hash, is rebuilt -- file times prove nothing after a checkout or copy that preserves them (a stale binary would otherwise be stamped as current
    and edited kernels would run against it)."""
    return _source_hash() != _built_hash()


def build(force: bool = False) -> str:
    """Compile libdpb.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force or (os.path.exists(LIB_PATH) and not os.path.exists(STAMP_PATH)):
        # an unstamped binary says nothing about the objects next to it either: start from the sources
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise DpbError("hipcc build of libdpb.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    with open(os.path.join(_HERE, "libdpb.so.srchash"), "w") as fh:     # make always links the in-tree libdpb.so: stamp that one, wherever DPB_LIB points
        fh.write(_source_hash())
    return os.path.join(_HERE, "libdpb.so")


def load():
    """Load libdpb.so and bind every declared symbol; raises loudly if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.environ.get("DPB_LIB") and os.path.exists(HIPCC) and os.path.isdir(CSRC) and os.access(_HERE, os.W_OK) and _needs_build():
        # the sources changed since libdpb.so was linked (or it was never built): rebuild in-tree, incrementally -- an edited
        # .hip / dpb.h never runs against a stale binary, and never against a CPU substitute.  The stamp is a content hash, not
        # mtimes, so a copied tree (gpurun snapshot) with a matching .so does not rebuild.  One builder when several ranks (or several
        # hosts sharing the checkout) start together: the lock file lives next to libdpb.so, in tmp only if that directory refuses.
        import fcntl
        import tempfile
        lock = os.path.join(_HERE, ".build.lock")
        try:
            lk = open(lock, "w")
        except OSError:
            lk = open(os.path.join(tempfile.gettempdir(), "dpb-build-%s.lock" % hashlib.sha1(_HERE.encode()).hexdigest()[:12]), "w")
        with lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                if _needs_build():
                    build()
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    if not os.path.exists(LIB_PATH):
        raise DpbError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback for the product path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    if lib.dpb_abi_version() != 1:
        raise DpbError("libdpb.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise DpbError(load().dpb_last_error().decode())
