"""Not a test: BK=32 ring (tile 130, forced by 131) vs BK=64 ring (515 / 512) on the path's short-K plain-row products.
python tools/gpu_gemm_shortk.py > gpurun_out/gemm_shortk.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_gemm_bench import run

if __name__ == "__main__":
    V = ((0, 0, 4), (131, 1, 4), (133, 1, 4), (515, 1, 4), (512, 1, 4), (517, 1, 4), (65, 1, 4), (64, 1, 4))
    run("lin 64^2 320->320 b5", 64, 320, 320, 1, 5, variants=V)
    run("lin 64^2 320->960 b5", 64, 320, 960, 1, 5, variants=V)
    run("lin 64^2 320->1280 b5", 64, 320, 1280, 1, 5, variants=V)
    run("lin 64^2 320->2560 b5", 64, 320, 2560, 1, 5, variants=V)
    run("lin 32^2 640->640 b5", 32, 640, 640, 1, 5, variants=V)
    run("lin 16^2 1280->1280 b5", 16, 1280, 1280, 1, 5, variants=V)
    run("lin 8^2 1280->1280 b5", 8, 1280, 1280, 1, 5, variants=V)
    run("lin 64^2 320->320 b10", 64, 320, 320, 1, 10, variants=V)
    run("lin 32^2 640->640 b10", 32, 640, 640, 1, 10, variants=V)
    run("lin 16^2 1280->1280 b10", 16, 1280, 1280, 1, 10, variants=V)
