timeout 600 python tools/gpu_eig_ab.py > gpurun_out/s3_eig_ab.txt 2> gpurun_out/s3_eig_ab.err; head -3 gpurun_out/s3_eig_ab.txt; tail -5 gpurun_out/s3_eig_ab.err
