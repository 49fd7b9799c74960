"""Runtime wrapper around one libdpb engine (one per GPU per process).

PyTorch is used for device memory (workspace / IO tensors) and the current HIP stream only;
all arithmetic happens in the HIP kernels behind the C ABI (include/dpb.h).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import lib as L
from .tape import Tape


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class Engine:
    def __init__(self, tape: Tape, temb_dim: int, flip_sin_to_cos: bool, half_minus_one: bool, x_channels: int,
                 max_batch: int = 1, max_tangents: int = 16):
        self.lib = L.load()
        if tape.device.type != "cuda":
            raise L.DpbError("the pullback engine needs a HIP device (there is no CPU fallback)")
        self.tape = tape
        self.device = tape.device
        self.max_batch, self.max_tangents = max_batch, max_tangents
        self.x_channels = x_channels
        nb, no = len(tape.buffers), len(tape.ops)
        self._bufs = (L.BufferDesc * nb)(*[L.BufferDesc(r, c, k, v) for (r, c, k), v in zip(tape.buffers, tape.valid)])
        ops = (L.OpDesc * no)()
        for i, d in enumerate(tape.ops):
            o = ops[i]
            o.kind, o.in0, o.in1, o.in2, o.out, o.res, o.rowbias = d["kind"], d["in0"], d["in1"], d["in2"], d["out"], d["res"], d["rowbias"]
            for j in range(12):
                o.ip[j] = int(d["ip"][j])
            for j in range(4):
                o.fp[j] = float(d["fp"][j])
                o.w[j] = d["w"][j] or None
        self._ops = ops
        net = L.NetDesc()
        net.dtype = L.dtype_code(tape.dtype)
        net.max_batch, net.max_tangents = max_batch, max_tangents
        net.n_buffers, net.n_ops = nb, no
        net.buffers, net.ops = self._bufs, self._ops
        net.x_buf, net.x_channels = tape.x, x_channels
        net.temb_buf, net.temb_dim = tape.temb_in, temb_dim
        net.temb_flip_sin_to_cos, net.temb_half_minus_one = int(flip_sin_to_cos), int(half_minus_one)
        net.ctx_buf = getattr(tape, "ctx", -1)
        self._net = net
        h = C.c_void_p()
        L.check(self.lib.dpb_engine_create(C.byref(net), C.byref(h)))
        self.h = h
        self.ws_bytes = int(self.lib.dpb_engine_workspace_bytes(h))
        with torch.cuda.device(self.device):
            self._ws = torch.empty(self.ws_bytes + 256, dtype=torch.uint8, device=self.device)
            off = (-self._ws.data_ptr()) % 256
            self._set_stream()
            L.check(self.lib.dpb_engine_set_workspace(h, C.c_void_p(self._ws.data_ptr() + off), self.ws_bytes))
        self.x_rows = tape.buffers[tape.x][0]
        self.n_in = self.x_rows * x_channels
        self.batch = 0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                torch.cuda.synchronize(self.device)
                self.lib.dpb_engine_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _set_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.dpb_engine_set_stream(self.h, C.c_void_p(s)))
        return s

    # ------------------------------------------------------------------ passes
    def tap_numel(self, tap) -> int:
        c, h, w = self.tape.tap_shape[self.tape.taps[tap]]
        return c * h * w

    def _inputs(self, x: torch.Tensor, ctx: Optional[torch.Tensor]):
        x = _f32(x, self.device)
        b = x.shape[0]
        if x[0].numel() != self.n_in:
            raise L.DpbError(f"input has {x[0].numel()} elements per sample, network expects {self.n_in}")
        cbuf = getattr(self.tape, "ctx", -1)
        if cbuf >= 0:
            if ctx is None:
                raise L.DpbError("this network needs encoder_hidden_states (ctx)")
            rows, cpad, _ = self.tape.buffers[cbuf]
            width = self.tape.valid[cbuf] or cpad
            if ctx.dim() != 3 or ctx.shape[1] != rows or ctx.shape[2] != width or ctx.shape[0] not in (1, b):
                raise L.DpbError(f"encoder_hidden_states has shape {tuple(ctx.shape)}, the engine was built for [{b} or 1, {rows}, {width}] "
                                 "(token count and width are fixed at engine build time: SDConfig.ctx_len / cross_dim)")
            ctx = _f32(ctx, self.device)
            if ctx.shape[0] != b:
                ctx = ctx.expand(b, -1, -1).contiguous()
        else:
            ctx = None
        return x, b, ctx

    def primal(self, x: torch.Tensor, t: float, ctx: Optional[torch.Tensor], tap) -> None:
        """x [B,C,H,W]; ctx [B,L,D] or None.  Keeps the activations resident for jvp/vjp."""
        buf = self.tape.taps[tap]
        with torch.cuda.device(self.device):
            self._set_stream()
            x, b, ctx = self._inputs(x, ctx)
            L.check(self.lib.dpb_primal(self.h, _ptr(x), b, float(t), _ptr(ctx), buf))
            self.batch = b

    def read(self, tap) -> torch.Tensor:
        buf = self.tape.taps[tap]
        c, h, w = self.tape.tap_shape[buf]
        if self.batch < 1:
            raise L.DpbError("no primal state to read: forward() keeps none (dpb_forward); call primal() first")
        with torch.cuda.device(self.device):
            self._set_stream()
            out = torch.empty(self.batch, c, h, w, dtype=torch.float32, device=self.device)
            L.check(self.lib.dpb_read_buffer(self.h, buf, c, _ptr(out)))
        return out

    def forward(self, x, t, ctx=None, tap="eps") -> torch.Tensor:
        """Forward only (dpb_forward): the U-Net calls of the DDIM / guidance loop and get_h.  Keeps no tangent / adjoint stash,
        so jvp / vjp / iterate need a primal() first (the engine refuses otherwise)."""
        buf = self.tape.taps[tap]
        c, h, w = self.tape.tap_shape[buf]
        with torch.cuda.device(self.device):
            self._set_stream()
            x, b, ctx = self._inputs(x, ctx)
            out = torch.empty(b, c, h, w, dtype=torch.float32, device=self.device)
            L.check(self.lib.dpb_forward(self.h, _ptr(x), b, float(t), _ptr(ctx), buf, c, _ptr(out)))
            self.batch = 0
        return out

    def jvp(self, tap, V: torch.Tensor) -> torch.Tensor:
        """V [nt, N_in] (NCHW-flattened) -> U [nt, N_h]"""
        buf = self.tape.taps[tap]
        with torch.cuda.device(self.device):
            self._set_stream()
            V = _f32(V, self.device).reshape(-1, self.n_in)
            U = torch.empty(V.shape[0], self.tap_numel(tap), dtype=torch.float32, device=self.device)
            L.check(self.lib.dpb_jvp(self.h, buf, _ptr(V), V.shape[0], _ptr(U)))
        return U

    def vjp(self, tap, U: torch.Tensor) -> torch.Tensor:
        buf = self.tape.taps[tap]
        with torch.cuda.device(self.device):
            self._set_stream()
            U = _f32(U, self.device).reshape(-1, self.tap_numel(tap))
            W = torch.empty(U.shape[0], self.n_in, dtype=torch.float32, device=self.device)
            L.check(self.lib.dpb_vjp(self.h, buf, _ptr(U), U.shape[0], _ptr(W)))
        return W

    def orth(self, W: torch.Tensor, Vprev: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        with torch.cuda.device(self.device):
            s_ = self._set_stream()
            W = _f32(W, self.device); Vprev = _f32(Vprev, self.device)
            k, n = W.shape
            V = torch.empty_like(W)
            s = torch.empty(k, dtype=torch.float32, device=self.device)
            conv = torch.empty(2, dtype=torch.float32, device=self.device)
            scratch = torch.empty(int(self.lib.dpb_orth_scratch_bytes(k, n)) // 8 + 1, dtype=torch.float64, device=self.device)
            L.check(self.lib.dpb_orth_checked(_ptr(W), _ptr(Vprev), _ptr(V), _ptr(s), _ptr(conv), _ptr(scratch), scratch.numel() * 8, k, n, C.c_void_p(s_)))
        return V, s, conv

    def iterate(self, tap, V: torch.Tensor, n_iters: int):
        """n_iters power iterations in place on V [B*k, N_in] (B = batch of the last primal, independent bases);
        returns (V, U [B*k, N_h], s [B*k], conv [B, 2]) device tensors, no host sync."""
        buf = self.tape.taps[tap]
        with torch.cuda.device(self.device):
            self._set_stream()
            assert V.is_cuda and V.dtype == torch.float32 and V.is_contiguous() and V.shape[0] % self.batch == 0
            nt = V.shape[0]
            k = nt // self.batch
            U = torch.empty(nt, self.tap_numel(tap), dtype=torch.float32, device=self.device)
            s = torch.empty(nt, dtype=torch.float32, device=self.device)
            conv = torch.empty(self.batch, 2, dtype=torch.float32, device=self.device)
            L.check(self.lib.dpb_pullback_iterate(self.h, buf, _ptr(V), _ptr(U), _ptr(s), _ptr(conv), k, n_iters))
        return V, U, s, conv

    def profile(self, enable: bool):
        L.check(self.lib.dpb_engine_profile(self.h, int(enable)))

    def profile_dump(self, path: str):
        L.check(self.lib.dpb_engine_profile_dump(self.h, path.encode()))

    def profile_read(self, big_tile: int):
        n = C.c_int64(); ms = C.c_double(); f = C.c_double()
        L.check(self.lib.dpb_engine_profile_read(self.h, int(big_tile), C.byref(n), C.byref(ms), C.byref(f)))
        return n.value, ms.value, f.value

    def profile_overhead_ms(self) -> float:
        """the calibrated empty-bracket time subtracted from every launch of profile_read / profile_dump"""
        v = C.c_double()
        L.check(self.lib.dpb_engine_profile_overhead(self.h, C.byref(v)))
        return v.value

    def stats(self):
        n = C.c_int64(); f = C.c_double(); b = C.c_double()
        L.check(self.lib.dpb_engine_stats(self.h, C.byref(n), C.byref(f), C.byref(b)))
        return n.value, f.value, b.value
