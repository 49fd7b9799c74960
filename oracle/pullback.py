"""ORACLE (test infrastructure, never imported by the product package).

CPU restatement of the reference's power-iteration low-rank SVD of the pullback
metric -- subspace iteration on J^T J with J = d get_h / d x.

Follows /root/reference/src/utils/utils.py
  * local_encoder_pullback_zt  :722-816   (SD variant:   chunks = k // chunk_size or 1)
  * local_encoder_pullback_xt  :165-249   (uncond variant: chunks = ceil(k / chunk_size))
and the identical copy in src/models/ddpm/diffusion.py:484-556.

Same autodiff calls in the same order: ``torch.func.jacfwd`` w.r.t. the scalar
``a`` of ``get_h(x + a*v_i)`` (:766-775), ``torch.autograd.functional.jacobian``
of ``<u_b, get_h(x)>`` (:790-797), ``torch.linalg.svd`` of the k x N matrix (:799),
stop iff ``allclose(V_prev, V, atol=thr)`` and ``i > min_iter`` (:806).

Differences from the reference, all deliberate and observable:
  * ``V0`` may be injected (the reference draws it on the device RNG, :750-753,
    which is not reproducible across CPU/HIP); ``V0=None`` reproduces the
    reference's draw ``QR(randn(N, k)).Q^T`` from the global CPU generator.
  * ``history=True`` also returns the per-iteration V and convergence distances.
  * no prints / no device bouncing (.cpu()/.to() are identities on CPU).
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
from einops import einsum


def _chunks(v: torch.Tensor, k: int, chunk_size: int, variant: str) -> List[torch.Tensor]:
    if variant == "zt":                      # utils.py:761-764
        n = v.size(0) // chunk_size
        return list(v.chunk(n)) if n != 0 else [v]
    num_chunk = k // chunk_size if k % chunk_size == 0 else k // chunk_size + 1   # utils.py:178
    return list(v.chunk(num_chunk))


def jvp_step(get_h: Callable, x: torch.Tensor, v: torch.Tensor, k: int, chunk_size: int, variant: str,
             batched_ctx: bool = False) -> torch.Tensor:
    """U = J V  (utils.py:766-775).  ``get_h(x_batch)`` must accept a batch."""
    a = torch.tensor(0.0, dtype=x.dtype)
    outs = []
    for vi in _chunks(v, k, chunk_size, variant):
        g = lambda a_: get_h(x + a_ * vi)
        outs.append(torch.func.jacfwd(g, argnums=0, has_aux=False, randomness="error")(a).detach().clone())
    return torch.cat(outs, dim=0)


def vjp_step(get_h: Callable, x: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """W = J^T U, one row per direction (utils.py:790-797)."""
    g = lambda x_: einsum(u, get_h(x_), "b c w h, i c w h -> b")
    w = torch.autograd.functional.jacobian(g, x)
    return w.reshape(u.shape[0], -1)


def pullback(get_h: Callable, x: torch.Tensor, pca_rank: int = 50, chunk_size: int = 25, min_iter: int = 10,
             max_iter: int = 100, convergence_threshold: float = 1e-3, variant: str = "zt",
             V0: Optional[torch.Tensor] = None, history: bool = False):
    """Returns ``(u, s, vT)`` exactly as the reference does (utils.py:810):
    ``u``  [N_h, k]  transposed view of J V_prev (un-normalised, one iteration stale),
    ``s``  [k]       sqrt of the singular values of J^T J V_prev,
    ``vT`` [k, N_in] orthonormal rows.
    ``get_h`` maps a batch ``[b, c, w, h]`` to features ``[b, c_o, w_o, h_o]``."""
    assert variant in ("zt", "xt")
    h_shape = get_h(x).shape
    c_i, w_i, h_i = x.size(1), x.size(2), x.size(3)
    c_o, w_o, h_o = h_shape[1], h_shape[2], h_shape[3]
    n_in = c_i * w_i * h_i

    if V0 is None:
        q, _ = torch.linalg.qr(torch.randn(n_in, pca_rank, dtype=torch.float))
        v = q.T
    else:
        v = V0.reshape(pca_rank, n_in).to(torch.float)
    v = v.reshape(-1, c_i, w_i, h_i)

    hist = {"V": [], "dist": [], "s": []}
    u = s = None
    for i in range(max_iter):
        v = v.to(dtype=x.dtype)
        v_prev = v.detach().clone()
        u = jvp_step(get_h, x, v, pca_rank, chunk_size, variant).to(x.dtype)
        v_ = vjp_step(get_h, x, u)
        _, s, v = torch.linalg.svd(v_, full_matrices=False)
        v = v.reshape(-1, c_i, w_i, h_i)
        u = u.reshape(-1, c_o, w_o, h_o)
        dist = torch.dist(v_prev, v).item()
        if history:
            hist["V"].append(v.reshape(pca_rank, n_in).clone()); hist["dist"].append(dist); hist["s"].append(s.sqrt().clone())
        if torch.allclose(v_prev, v, atol=convergence_threshold) and (i > min_iter):
            break
    out = (u.reshape(-1, c_o * w_o * h_o).T.detach(), s.sqrt().detach(), v.reshape(-1, n_in).detach())
    return out + (hist,) if history else out
