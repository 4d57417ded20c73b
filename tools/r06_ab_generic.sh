# developer (ON THE GPU BOX): headline / serial / cfg4 A/B of the shipped library against devlibs/lib_$1.so
for lib in "" devlibs/lib_$1.so; do
  for r in 1 2 3; do CONV3P_HIP_LIB=$lib timeout 200 python bench.py --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg2 lib=$lib', d['ms_per_step'], {k: round(v, 4) for k, v in d['roofline']['kernel_ms_per_step'].items()})"; done
  for r in 1 2; do CONV3P_HIP_LIB=$lib timeout 200 python bench.py --no-cpu --no-extra --serial 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('serial lib=$lib', d['ms_per_step'], {k: round(v, 4) for k, v in d['roofline']['kernel_ms_per_step'].items()})"; done
  for r in 1 2; do echo -n "cfg4 lib=$lib: "; CONV3P_HIP_LIB=$lib timeout 200 python tools/cfg4_step.py 2>/dev/null | tail -1; done
done
