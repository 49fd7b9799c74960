"""Not a test: the weight-heavy M = 64 k layers of the 8x8 level (k = 5 -> M = 320, k = 10 -> M = 640) with the 128x128 / 320x128 / 320x64
ring tiles at several split-K factors.   python tools/gpu_skinny_bench.py > gpurun_out/skinny.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_gemm_bench import run

if __name__ == "__main__":
    V = ((0, 0, 4), (515, 8, 4), (515, 16, 4), (520, 6, 4), (520, 12, 4), (520, 24, 4), (521, 4, 4), (521, 6, 4), (521, 12, 4))
    run("conv3x3 8^2 1280->1280 b5", 8, 1280, 1280, 3, 5, variants=V)
    run("conv3x3 8^2 1280->1280 b10", 8, 1280, 1280, 3, 10, variants=V)
    V = ((0, 0, 4), (515, 4, 4), (515, 8, 4), (520, 3, 4), (520, 6, 4), (520, 12, 4), (521, 2, 4), (521, 4, 4), (521, 6, 4))
    run("lin 8^2 5120->1280 b5", 8, 5120, 1280, 1, 5, variants=V)
    run("lin 8^2 1280->1280 b5", 8, 1280, 1280, 1, 5, variants=V)
