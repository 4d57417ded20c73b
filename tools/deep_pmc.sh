#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  rm -rf $OUT/dp; rocprofv3 --kernel-trace --pmc $C -d $OUT/dp -o p -- python $ROOT/tools/deep_time.py > $OUT/dp.log 2>&1
  python $ROOT/tools/pmc_query.py $OUT/dp/p_results.db deep_gemm | grep "n="
done
rm -rf $OUT/dp
