#!/bin/bash
# round 6 (ON THE GPU BOX): headline with / without the fused stack launches, overlapped and serial
mkdir -p gpurun_out
for mode in fused perlayer; do
  for ser in "" "--serial"; do
    f=""; [ $mode = fused ] && f="--fused-stack"
    for rep in 1 2; do
      timeout 200 python bench.py --no-cpu --no-extra $f $ser 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$mode', '$ser' or 'overlap', 'ms/step %.4f' % d['ms_per_step'], {k: round(v, 4) for k, v in d['roofline']['kernel_ms_per_step'].items()})"
    done
  done
done
