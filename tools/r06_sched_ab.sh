#!/bin/bash
# round 6 (ON THE GPU BOX): what the tiles' longest-first launch order (tile_sched_kernel) is worth: kernel times alone
for lib in "" devlibs/lib_nosched.so; do
  for rep in 1 2; do
    CONV3P_HIP_LIB=$lib timeout 200 python bench.py --no-cpu --no-extra --serial 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('${lib:-shipped}', 'serial ms/step %.4f' % d['ms_per_step'], {k: round(v, 4) for k, v in d['roofline']['kernel_ms_per_step'].items()})"
    CONV3P_HIP_LIB=$lib timeout 200 python bench.py --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('${lib:-shipped}', 'overlap ms/step %.4f' % d['ms_per_step'])"
  done
done
