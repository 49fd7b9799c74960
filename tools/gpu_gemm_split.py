"""Not a test: split-K choice of the BK=64 ring on the path's under-filled plain-row products (heuristic = variant (0,0)).
python tools/gpu_gemm_split.py > gpurun_out/gemm_split.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_gemm_bench import run

if __name__ == "__main__":
    V = ((0, 0, 4), (515, 1, 4), (515, 2, 4), (515, 3, 4), (515, 4, 4), (517, 1, 4), (517, 2, 4), (518, 1, 4), (518, 2, 4))
    run("lin 16^2 1280->3840 b5", 16, 1280, 3840, 1, 5, variants=V)
    run("lin 16^2 1280->5120 b5", 16, 1280, 5120, 1, 5, variants=V)
    run("lin 16^2 5120->1280 b5", 16, 5120, 1280, 1, 5, variants=V)
    run("lin 16^2 3840->1280 b5", 16, 3840, 1280, 1, 5, variants=V)
    run("lin 32^2 5120->640 b5", 32, 5120, 640, 1, 5, variants=V)
    run("lin 32^2 2560->640 b5", 32, 2560, 640, 1, 5, variants=V)
    run("lin 32^2 1920->640 b5", 32, 1920, 640, 1, 5, variants=V)
    run("lin 32^2 640->640 b5", 32, 640, 640, 1, 5, variants=V)
    run("lin 32^2 640->1920 b5", 32, 640, 1920, 1, 5, variants=V)
    run("lin 32^2 640->2560 b5", 32, 640, 2560, 1, 5, variants=V)
    run("lin 64^2 2560->320 b5", 64, 2560, 320, 1, 5, variants=V)
    run("lin 64^2 1280->320 b5", 64, 1280, 320, 1, 5, variants=V)
    run("lin 64^2 320->960 b5", 64, 320, 960, 1, 5, variants=V)
    run("lin 8^2 1280->10240 b5", 8, 1280, 10240, 1, 5, variants=V)
    run("lin 8^2 5120->1280 b5", 8, 5120, 1280, 1, 5, variants=V)
