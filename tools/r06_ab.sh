#!/bin/bash
# round 6 (ON THE GPU BOX): headline with the hidden layers of the forward / backward / both / no pass as ONE launch
mkdir -p gpurun_out
for mode in "" "--fused forward" "--fused backward" "--fused-stack"; do
  for ser in "" "--serial"; do
    for rep in 1 2 3; do
      timeout 200 python bench.py --no-cpu --no-extra $mode $ser 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('${mode:-per layer}', '${ser:-overlap}', 'ms/step %.4f' % d['ms_per_step'], {k: round(v, 4) for k, v in d['roofline']['kernel_ms_per_step'].items()})"
    done
  done
done
