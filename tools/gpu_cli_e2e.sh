# The reference's SD editing script through this repo's CLI, end to end on the device (prompt encoder -> inversion -> pullback -> x-space guidance ->
# decode -> PNG), synthetic weights at the exact architecture shapes: SD-v1.5 in bf16 and SD-2.1-base (the id the reference's scripts pass) in fp16.
#   gpurun -- 'bash tools/gpu_cli_e2e.sh TAG'  ->  gpurun_out/TAG_main_cli_*.log
TAG=${1:-cli}
R=$PWD
for cfg in "runwayml/stable-diffusion-v1-5 bf16 sd15" "stabilityai/stable-diffusion-2-1-base fp16 sd21"; do
  set -- $cfg
  rm -rf /tmp/dpb_runs /tmp/inputs; mkdir -p /tmp/dpb_runs       # (the basis cache ./inputs/... is keyed by dataset / steps / rank, not by model -- as in the reference)
  ( cd /tmp && T0=$(date +%s.%N) && PYTHONPATH=$R python -m diffusion_pullback_amd.main --note demo --model_name $1 --dataset_name Examples --dtype $2 \
      --result_folder /tmp/dpb_runs --edit_prompt "sitting dog" --x_space_guidance_scale 1 --x_space_guidance_num_step 64 --edit_t 0.7 --pca_rank 2 \
      --run_edit_local_encoder_pullback_zt True --vae synthetic --text_encoder synthetic --timing True && python -c "import time,sys; print(\"Elapsed wall seconds:\", round(time.time() - float(sys.argv[1]), 1))" $T0 ) > $R/gpurun_out/${TAG}_main_cli_$3_end_to_end.log 2>&1
  find /tmp/dpb_runs -name "*.png" | sort >> gpurun_out/${TAG}_main_cli_$3_end_to_end.log
  grep -E "Elapsed|Error|Traceback|breakdown| s  " gpurun_out/${TAG}_main_cli_$3_end_to_end.log | tail -20
done
