# One GPU-box session: the -m gpu tests, the headline bench line and the rocprofv3 kernel-trace summary of the same command.
#   gpurun -- 'bash tools/gpu_session.sh TAG [pytest-args...]'      (outputs under gpurun_out/TAG_*)
TAG=${1:-sess}; shift
R=$PWD
if [ -z "$SKIP_TESTS" ]; then   # (SKIP_TESTS=1: the suite ran in its own gpurun call -- the whole evidence run does not fit one 40-minute call)
  timeout 1500 python -m pytest tests -m gpu -q -s "$@" > gpurun_out/${TAG}_pytest.log 2>&1; tail -4 gpurun_out/${TAG}_pytest.log
fi
DPB_PROFILE_CSV=gpurun_out/${TAG}_gemm_launches.csv python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cut -c1-600 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o sd15 -- python $R/bench.py --steps 36 --warmup 12 --profile-run > /dev/null 2>&1
cd $R; python tools/kernel_avgs.py $(ls gpurun_out/${TAG}_stats/*/*kernel_stats.csv gpurun_out/${TAG}_stats/*kernel_stats.csv 2>/dev/null | head -1) 48 30 > gpurun_out/${TAG}_kernel_avgs.txt 2>&1; head -40 gpurun_out/${TAG}_kernel_avgs.txt
