"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/dpb.h declares
(no compute calls: there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dpb.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dpb_[a-z_0-9]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from diffusion_pullback_amd import lib
    lib.build()
    l = lib.load()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(l, n), f"libdpb.so does not export {n}"
        assert n in lib.SYMBOLS, f"ctypes binding missing for {n}"
    assert set(lib.SYMBOLS) == set(names)
    assert l.dpb_abi_version() == 1


def test_engine_rejects_bad_descriptions_without_gpu():
    """engine_create is pure host code: argument validation is testable on CPU."""
    import ctypes as C
    from diffusion_pullback_amd import lib
    l = lib.load()
    net = lib.NetDesc()
    h = C.c_void_p()
    net.dtype = 7
    assert l.dpb_engine_create(C.byref(net), C.byref(h)) != 0
    assert b"dtype" in l.dpb_last_error()
    net.dtype = lib.DPB_F32; net.max_batch = 1; net.max_tangents = 1
    bufs = (lib.BufferDesc * 1)(lib.BufferDesc(16, 12, 0, 0))      # channels not a multiple of 8
    net.n_buffers = 1; net.buffers = bufs; net.n_ops = 0; net.x_buf = 0; net.x_channels = 3
    assert l.dpb_engine_create(C.byref(net), C.byref(h)) != 0
    assert b"multiple of 8" in l.dpb_last_error()


def test_product_has_no_cpu_fallback():
    import torch
    from diffusion_pullback_amd import PullbackUNet, DpbError
    from oracle import unet_ddpm
    cfg = unet_ddpm.DDPMConfig(ch=32, ch_mult=(1,), num_res_blocks=1, attn_resolutions=(), resolution=8)
    with pytest.raises(DpbError):
        PullbackUNet("ddpm", cfg, unet_ddpm.init_params(cfg), device="cpu")


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "diffusion_pullback_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            assert "oracle" not in re.sub(r'""".*?"""', "", open(os.path.join(pkg, fn)).read(), flags=re.S), fn
