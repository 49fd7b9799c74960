R=$PWD
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/refresh_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/refresh_smoke.txt 2>&1
DPB_PROFILE_CSV=gpurun_out/refresh_gemm_launches.csv python bench.py 2>/dev/null > gpurun_out/refresh_bench_sd15.json
python bench.py --workload ddpm256 --dtype fp32 2>/dev/null > gpurun_out/refresh_bench_ddpm256.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/refresh_stats -o sd15 -- python $R/bench.py --profile-run > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/refresh_pmc_fetch -o f -- python $R/bench.py --steps 12 --warmup 0 --profile-run > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/refresh_pmc_write -o w -- python $R/bench.py --steps 12 --warmup 0 --profile-run > /dev/null 2>&1
cd $R; ls gpurun_out/refresh_*; cat gpurun_out/refresh_pytest.txt gpurun_out/refresh_smoke.txt | tail -5
