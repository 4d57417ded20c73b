for a in "" "--fused-stack" "" "--fused-stack" "" "--fused-stack"; do timeout 200 python bench.py --no-cpu --no-extra $a 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg2 [$a]', d['ms_per_step'], d['config'].get('fused_status'), {k: round(v, 4) for k, v in d['roofline']['kernel_ms_per_step'].items()})"; done
