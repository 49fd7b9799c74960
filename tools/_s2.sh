timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k 'orth' 2>&1 | tail -5
timeout 600 python tools/gpu_eig_ab.py > gpurun_out/s2_eig_ab.txt 2> gpurun_out/s2_eig_ab.err; cat gpurun_out/s2_eig_ab.txt; tail -5 gpurun_out/s2_eig_ab.err
timeout 900 python -m pytest tests/test_gpu_edge.py -q -x -k 'rank or error_behaviour or chunking' 2>&1 | tail -5
python bench.py --k 50 --steps 6 --warmup 2 --no-cpu-baseline --no-unet-forward --no-sd21-leg --no-strong-leg --no-roofline --repeats 3 > gpurun_out/s2_k50.json 2> gpurun_out/s2_k50.err; cut -c1-260 gpurun_out/s2_k50.json
DPB_EIG_PAR=0 python bench.py --k 50 --steps 6 --warmup 2 --no-cpu-baseline --no-unet-forward --no-sd21-leg --no-strong-leg --no-roofline --repeats 3 > gpurun_out/s2_k50_old.json 2> gpurun_out/s2_k50_old.err; cut -c1-260 gpurun_out/s2_k50_old.json
