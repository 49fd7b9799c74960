# Full evidence run of a round: tests, headline bench, kernel-trace stats, PMC passes, config sweep.  gpurun -- 'bash tools/gpu_final.sh TAG'
TAG=${1:-rXX}
bash tools/gpu_session.sh $TAG
bash tools/pmc_mfma.sh $TAG
bash tools/sweep_configs.sh > gpurun_out/${TAG}_config_sweep.jsonl 2> gpurun_out/${TAG}_config_sweep.err; cut -c1-200 gpurun_out/${TAG}_config_sweep.jsonl
