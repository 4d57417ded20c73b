#!/bin/bash
# developer (ON THE GPU BOX): what the kernels of the cfg2 step load -- instruction counts, LDS and address-unit counters of
# the step in --serial mode (every kernel alone on the GPU), one rocprofv3 --pmc pass per counter group
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_HIT_sum TCP_TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  rm -rf $OUT/c2p
  rocprofv3 --kernel-trace --pmc $C -d $OUT/c2p -o p -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu --no-extra --serial > $OUT/c2p.log 2>&1
  python $ROOT/tools/pmc_query.py $OUT/c2p/p_results.db conv3p | grep -E "n=|avg_us|kernel" | grep -E "backward|forward_kernel|search_fused|kernel "
done
rm -rf $OUT/c2p
