#!/bin/bash
# round 6, first measurement (ON THE GPU BOX): per-cloud barrier micro-probe + per-phase stamps of the narrow kernels
mkdir -p gpurun_out
./tools/ubench/cloud_barrier > gpurun_out/r06_cloud_barrier.txt 2>&1
: > gpurun_out/r06_phase_trace.txt
for s in 1 2 3 4; do
  ci=9; [ $s = 1 ] && ci=3
  CONV3P_HIP_LIB=devlibs/lib_stamps.so timeout 300 python tools/phase_trace.py $s $ci >> gpurun_out/r06_phase_trace.txt 2>&1
done
cat gpurun_out/r06_cloud_barrier.txt gpurun_out/r06_phase_trace.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "voxel_sizes or cache_ or golden" 2>&1 | tail -5
