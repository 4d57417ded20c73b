#!/bin/bash
# round 6 (ON THE GPU BOX): the next batch's hash + sort + window tables started BEFORE the current forward (prefetch_sort)
for a in "" "--sort-first"; do
  for rep in 1 2 3; do echo -n "cfg4 $a: "; timeout 200 python tools/cfg4_step.py $a 2>/dev/null | tail -1; done
done
for a in "--no-sort-first" ""; do
  for rep in 1 2 3; do
    timeout 200 python bench.py --no-cpu --no-extra $a 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg2 headline ${a:-sort-first}', 'ms/step %.4f' % d['ms_per_step'])"
  done
done
