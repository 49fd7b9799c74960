R=$PWD
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-unet-forward --no-sd21-leg --no-strong-leg > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err; cut -c1-300 gpurun_out/s1_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s1_k50 -o k50 -- python $R/bench.py --k 50 --steps 6 --warmup 2 --profile-run > $R/gpurun_out/s1_k50.out 2>&1
cd $R; python tools/kernel_avgs.py $(ls gpurun_out/s1_k50/*/*kernel_stats.csv gpurun_out/s1_k50/*kernel_stats.csv 2>/dev/null | head -1) 8 40 > gpurun_out/s1_k50_avgs.txt 2>&1; head -50 gpurun_out/s1_k50_avgs.txt
