#!/bin/bash
# developer (ON THE GPU BOX): LDS / issue counters of the matrix-core path's kernels on the cfg5 shard step
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
for C in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  rm -rf $OUT/dp; rocprofv3 --kernel-trace --pmc $C -d $OUT/dp -o p -- python $ROOT/tools/deep_time.py > $OUT/dp.log 2>&1
  python $ROOT/tools/pmc_query.py $OUT/dp/p_results.db deep_ | grep -E "n=|avg_us" | grep -E "gemm|dw_kernel|kernel "
done
rm -rf $OUT/dp
