#!/bin/bash
# round 6 (ON THE GPU BOX): what the tiles' longest-first launch order (tile_sched_kernel) is worth, launch included:
# shipped against a build that neither launches the kernel nor uses its order (-DCONV3P_DEV_NO_SCHED)
for lib in "" devlibs/lib_nosched.so; do
  for rep in 1 2 3; do
    CONV3P_HIP_LIB=$lib timeout 200 python bench.py --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('${lib:-shipped}', 'cfg2 headline ms/step %.4f' % d['ms_per_step'])"
    echo -n "${lib:-shipped} cfg4: "; CONV3P_HIP_LIB=$lib timeout 200 python tools/cfg4_step.py 2>/dev/null | tail -1
  done
done
