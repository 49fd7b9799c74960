import os
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def load_golden(name):
    return torch.load(os.path.join(G, name), weights_only=False)


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def abs_cos(a, b):
    """|cos| between matching rows"""
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a * b).sum(-1).abs() / (a.norm(dim=-1) * b.norm(dim=-1) + 1e-30))


def oracle_jvp(f, x, V):
    """rows of V pushed through f at x (fp32 CPU)"""
    outs = []
    for v in V:
        _, t = torch.func.jvp(f, (x,), (v.reshape(x.shape),))
        outs.append(t.reshape(-1))
    return torch.stack(outs)


def oracle_vjp(f, x, U):
    x = x.clone().requires_grad_(True)
    h = f(x)
    outs = []
    for u in U:
        (g,) = torch.autograd.grad(h, x, u.reshape(h.shape), retain_graph=True)
        outs.append(g.reshape(-1))
    return torch.stack(outs)
