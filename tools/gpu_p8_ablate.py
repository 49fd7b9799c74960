"""Not a test: the 8-phase tile with parts of its K loop compiled out (wrong results, timings only): 532 full, 534 no DMA, 535 no fragment reads,
536 neither, 537 no MFMA, 538 no MFMA and no DMA."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_gemm_bench import run
V = ((532, 1, 4), (534, 1, 4), (535, 1, 4), (536, 1, 4), (537, 1, 4), (538, 1, 4), (532, 1, 4))
for rep in range(2):
    run("lin 64^2 2560->2560 b16", 64, 2560, 2560, 1, 16, variants=V)
    run("lin 64^2 2560->2560 b4 (640 tiles)", 64, 2560, 2560, 1, 4, variants=V)
