#!/bin/bash
# round 6 (ON THE GPU BOX): polling rate of the per-cloud barrier (s_sleep argument between two polls) in the fused launches
for lib in "" devlibs/lib_sleep4.so devlibs/lib_sleep16.so; do
  for rep in 1 2; do
    CONV3P_HIP_LIB=$lib timeout 200 python bench.py --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('${lib:-shipped (sleep 1)}', 'cfg2 headline (fused forward) ms/step %.4f' % d['ms_per_step'])"
    CONV3P_HIP_LIB=$lib timeout 200 python bench.py --no-cpu --no-extra --serial --fused-stack 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('${lib:-shipped (sleep 1)}', 'cfg2 serial, both passes fused', {k: round(v, 4) for k, v in d['roofline']['kernel_ms_per_step'].items() if 'ward' in k})"
    echo -n "${lib:-shipped (sleep 1)} cfg4 fused forward, no prefetch: "; CONV3P_HIP_LIB=$lib timeout 200 python tools/cfg4_step.py --fused --no-prefetch 2>/dev/null | tail -1
  done
done
CONV3P_HIP_LIB=devlibs/lib_stamps.so timeout 300 python tools/fused_trace.py
