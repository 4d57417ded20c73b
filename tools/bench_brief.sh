#!/bin/bash
# quick look: headline (overlapped) and --serial, kernel ms per step
for extra in "" "--serial"; do
python bench.py --steps 50 --warmup 10 --no-cpu --no-extra $extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$extra', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
done
