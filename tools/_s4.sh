timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_bench.py -q -x 2>&1 | tail -5
for rep in 1 2; do for ob in 1 0; do
DPB_ORTH_BATCH=$ob python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-unet-forward --no-sd21-leg --no-roofline --repeats 5 > gpurun_out/s4_b$ob.$rep.json 2> gpurun_out/s4_b$ob.$rep.err
python - <<EOF
import json; d=json.loads(open('gpurun_out/s4_b$ob.$rep.json').read().strip().splitlines()[-1])
print('ORTH_BATCH=$ob rep $rep: headline', round(d['value'],2), 'batched4', round(d['batched_throughput']['value'],2), 'strong', round(d['strong_scaling']['value'],2) if 'strong_scaling' in d else None)
EOF
done; done
