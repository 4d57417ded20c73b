#!/bin/bash
# Run ON THE GPU BOX (through gpurun): produces the evidence files that get committed under profiles/.
#   tools/collect_profiles.sh <tag>      e.g. r01
# 1. bench.py (full, with CPU baseline)                      -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command    -> gpurun_out/<tag>_kernel_stats.txt
# 3. PMC passes, each in its own run (FETCH_SIZE / WRITE_SIZE cannot share a pass; no trace domains
#    besides --kernel-trace)                                  -> gpurun_out/<tag>_pmc_*.txt, traffic json
set -u
TAG=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=20; WARM=5

python $ROOT/bench.py --steps 50 --warmup 10 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err

rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o t -- python $ROOT/bench.py --steps $STEPS --warmup $WARM --no-cpu --no-extra > $OUT/${TAG}_trace.log 2>&1
python $ROOT/tools/pmc_query.py $OUT/${TAG}_trace/t_results.db > $OUT/${TAG}_kernel_stats.txt 2>&1

for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  N=$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C -d $OUT/${TAG}_pmc_$N -o p -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-extra > $OUT/${TAG}_pmc_$N.log 2>&1
  python $ROOT/tools/pmc_query.py $OUT/${TAG}_pmc_$N/p_results.db conv3p > $OUT/${TAG}_pmc_$N.txt 2>&1
done
python $ROOT/tools/traffic_json.py $OUT/${TAG}_pmc_FETCH_SIZE/p_results.db $OUT/${TAG}_pmc_WRITE_SIZE/p_results.db > $OUT/${TAG}_traffic.json
python $ROOT/tools/valu_json.py $OUT/${TAG}_pmc_SQ_WAVES/p_results.db > $OUT/${TAG}_valu.json
# address-unit / cache counters of the accumulators (dependent latency vs address-unit throughput)
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_HIT_sum TCP_TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE -d $OUT/${TAG}_pmc_TA -o p -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-extra --serial > $OUT/${TAG}_pmc_TA.log 2>&1
python $ROOT/tools/pmc_query.py $OUT/${TAG}_pmc_TA/p_results.db conv3p > $OUT/${TAG}_pmc_TA.txt 2>&1
# 4. deep-channel path (cfg5 per-GPU shard: B=16, N=8192, 128->256): parity + timing, kernel summary, MFMA counters
python $ROOT/tools/deep_check.py > $OUT/${TAG}_deep_check.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_deep_trace -o t -- python $ROOT/tools/deep_time.py > $OUT/${TAG}_deep_trace.log 2>&1
python $ROOT/tools/pmc_query.py $OUT/${TAG}_deep_trace/t_results.db deep > $OUT/${TAG}_deep_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/${TAG}_deep_pmc -o p -- python $ROOT/tools/deep_time.py > $OUT/${TAG}_deep_pmc.log 2>&1
python $ROOT/tools/pmc_query.py $OUT/${TAG}_deep_pmc/p_results.db deep > $OUT/${TAG}_deep_pmc_MFMA.txt 2>&1
python $ROOT/tools/mfma_json.py $OUT/${TAG}_deep_pmc_MFMA.txt > $OUT/${TAG}_deep_mfma.json
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/${TAG}_deep_pmc_$C -o p -- python $ROOT/tools/deep_time.py > $OUT/${TAG}_deep_pmc_$C.log 2>&1
done
python $ROOT/tools/traffic_json.py $OUT/${TAG}_deep_pmc_FETCH_SIZE/p_results.db $OUT/${TAG}_deep_pmc_WRITE_SIZE/p_results.db > $OUT/${TAG}_deep_traffic.json
# 4b. the classification head's FC kernels (fc1 73 728 x 512) and the serial (no side stream) kernel durations
python $ROOT/tools/head_time.py > $OUT/${TAG}_head_time.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_serial_trace -o t -- python $ROOT/bench.py --steps $STEPS --warmup $WARM --no-cpu --no-extra --serial > $OUT/${TAG}_serial_trace.log 2>&1
python $ROOT/tools/pmc_query.py $OUT/${TAG}_serial_trace/t_results.db > $OUT/${TAG}_kernel_stats_serial.txt 2>&1
# 5. other shapes / dtypes: the models' stacks at cfg2 and cfg4, fp64 on the register path, the generic kernels
python $ROOT/tools/stack_time.py > $OUT/${TAG}_stack_time.txt 2>&1
python $ROOT/tools/f64_time.py > $OUT/${TAG}_f64_time.txt 2>&1
python $ROOT/tools/generic_time.py > $OUT/${TAG}_generic_time.txt 2>&1
# 6. single layers at the cfg4 cloud size (B=16, N=4096 rooms): the models' shapes on the register path, SceneNN's and
#    other mid-size shapes on the matrix-core path
for s in "9 9" "36 13" "36 41" "12 9" "16 16" "32 64" "64 64" "64 128"; do python $ROOT/tools/shape_time.py $s 16 4096 room 2>&1 | tail -1; done > $OUT/${TAG}_shape_time.txt
# 7. (round 4) the geometry alone and the two-stream timeline of the headline: which kernel runs beside which
( python $ROOT/tools/search_time.py cfg2; python $ROOT/tools/search_time.py cfg4; python $ROOT/tools/search_time.py cfg5 ) 2>/dev/null | grep geometry > $OUT/${TAG}_geometry_time.txt
$ROOT/tools/headline_timeline.sh 64 > $OUT/${TAG}_headline_timeline.txt 2>&1
$ROOT/tools/cfg4_timeline.sh > $OUT/${TAG}_cfg4_timeline.txt 2>&1
python $ROOT/tools/op_boundary.py 2>/dev/null | tail -1 > $OUT/${TAG}_op_boundary.txt
./tools/ubench/gather_rate > $OUT/${TAG}_gather_rate.txt 2>&1 || $ROOT/tools/ubench/gather_rate > $OUT/${TAG}_gather_rate.txt 2>&1
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_serial_trace $OUT/${TAG}_pmc_*/ $OUT/${TAG}_deep_trace $OUT/${TAG}_deep_pmc $OUT/${TAG}_deep_pmc_*/   # keep the text summaries, drop the databases
tail -c 1500 $OUT/${TAG}_bench.json; echo; head -14 $OUT/${TAG}_kernel_stats.txt; cat $OUT/${TAG}_traffic.json
