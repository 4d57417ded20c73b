#!/bin/bash
# developer (ON THE GPU BOX): the geometry kernels alone -- durations (bench --serial under rocprofv3) and instruction
# counters -- for the shipped library and every devlibs/lib_*.so
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in $ROOT/pointwise_amd/csrc/libconv3p_hip.so $ROOT/devlibs/lib_*.so; do
  [ -f $lib ] || continue
  echo "=== $lib"
  export CONV3P_HIP_LIB=$lib
  python $ROOT/bench.py --steps 50 --warmup 10 --no-cpu --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['ms_per_step'], d['value'])"
  python $ROOT/tools/search_time.py cfg2; python $ROOT/tools/search_time.py cfg4
  rm -rf $OUT/sp_trace
  rocprofv3 --kernel-trace --stats -d $OUT/sp_trace -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-extra --serial > $OUT/sp_trace.log 2>&1
  python $ROOT/tools/pmc_query.py $OUT/sp_trace/t_results.db | grep -E "kernel|search|tile_|prep_sort" | head -8
  if [ "${PMC:-1}" = 1 ]; then
    rm -rf $OUT/sp_pmc
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/sp_pmc -o p -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-extra --serial > $OUT/sp_pmc.log 2>&1
    python $ROOT/tools/pmc_query.py $OUT/sp_pmc/p_results.db search | grep -E "INSTS|WAVE_CYCLES|SQ_WAVES"
    python $ROOT/tools/pmc_query.py $OUT/sp_pmc/p_results.db tile_tables | grep -E "INSTS_VALU"
  fi
done
rm -rf $OUT/sp_trace $OUT/sp_pmc
