#!/bin/bash
# developer (ON THE GPU BOX): what the kernels of the cfg4 step load -- instruction counts and unit-busy counters of
# the step WITHOUT prefetch (every kernel alone on the GPU), one rocprofv3 --pmc pass per counter group
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_HIT_sum TCP_TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  rm -rf $OUT/c4p
  rocprofv3 --kernel-trace --pmc $C -d $OUT/c4p -o p -- python $ROOT/tools/cfg4_step.py --no-prefetch > $OUT/c4p.log 2>&1
  tail -1 $OUT/c4p.log
  python $ROOT/tools/pmc_query.py $OUT/c4p/p_results.db conv3p | grep "n=" | grep -E "backward|forward_kernel|tap_"
done
rm -rf $OUT/c4p
