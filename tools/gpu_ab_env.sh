# Same-session A/B of one environment switch on the headline bench (+ the U-Net forward leg):  gpurun -- 'bash tools/gpu_ab_env.sh VAR A B [extra bench flags]'
VAR=$1; A=$2; B=$3; shift 3
for rep in 1 2; do
  for v in $A $B; do
    env $VAR=$v python bench.py --no-cpu-baseline --no-strong-leg --no-sd21-leg "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
u=d.get('unet_forward',{}).get('batches',{})
print('$VAR=$v', 'ms/step %.3f' % d['ms_per_step'], 'launches', d.get('roofline',{}).get('launches_per_step'), 'batched', round(d.get('batched_throughput',{}).get('value',0),1), 'fwd ms B1/2/5', [round(u[b]['ms_per_forward'],3) for b in u], 'fwd launches', [u[b]['launches'] for b in u])"
  done
done
