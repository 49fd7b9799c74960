"""Wall-clock breakdown of a CLI run by phase (weight generation / engine build, prompt encoding, inversion, DDIM forward, pullback,
x-space guidance, decode loop, VAE, image files).  Off by default (no synchronisation is added to the loops); ``--timing`` or DPB_TIMING=1
turns it on: every phase boundary then synchronises the device, so the phases add up to the wall time of the run."""
from __future__ import annotations

import contextlib
import os
import time
from collections import OrderedDict

import torch

ENABLED = bool(int(os.environ.get("DPB_TIMING", "0")))
_acc: "OrderedDict[str, list]" = OrderedDict()
_stack = []


def enable(on: bool = True) -> None:
    global ENABLED
    ENABLED = on


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


@contextlib.contextmanager
def phase(name: str):
    """Time a phase; nested phases are charged to the innermost one only (the outer phase's own time excludes them)."""
    if not ENABLED:
        yield
        return
    _sync()
    t0 = time.perf_counter()
    _stack.append(0.0)
    try:
        yield
    finally:
        _sync()
        dt = time.perf_counter() - t0
        inner = _stack.pop()
        e = _acc.setdefault(name, [0.0, 0])
        e[0] += dt - inner
        e[1] += 1
        if _stack:
            _stack[-1] += dt


def report(total_s: float | None = None) -> str:
    rows = [(k, v[0], v[1]) for k, v in _acc.items()]
    tot = sum(r[1] for r in rows)
    lines = ["wall-clock breakdown (device synchronised at phase boundaries):"]
    for k, s, n in sorted(rows, key=lambda r: -r[1]):
        lines.append(f"  {k:34s} {s:8.3f} s  {100 * s / max(tot, 1e-9):5.1f} %   x{n}" + (f"   ({1e3 * s / n:.2f} ms each)" if n > 1 else ""))
    lines.append(f"  {'sum of phases':34s} {tot:8.3f} s" + (f"   (process wall {total_s:.1f} s: the rest is Python start-up / imports)" if total_s else ""))
    return "\n".join(lines)


def reset() -> None:
    _acc.clear()
