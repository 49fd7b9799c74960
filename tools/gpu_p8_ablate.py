"""Not a test: the 8-phase tile (code 530) with parts of its K loop compiled out (`make -C diffusion_pullback_amd/csrc ablate_p8`; wrong results,
timings only).  Run once per build:  DPB_LIB=diffusion_pullback_amd/csrc/build/p8abl<bits>/libdpb.so python tools/gpu_p8_ablate.py
bits: 1 no DMA in the loop, 2 no fragment reads, 4 no MFMA."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_gemm_bench import run
print("library:", os.environ.get("DPB_LIB", "(in-tree)"))
V = ((530, 1, 4), (530, 1, 4))
run("lin 64^2 2560->2560 b16", 64, 2560, 2560, 1, 16, variants=V)
run("lin 64^2 2560->2560 b4 (640 tiles)", 64, 2560, 2560, 1, 4, variants=V)
