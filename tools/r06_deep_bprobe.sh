#!/bin/bash
# round 6 (ON THE GPU BOX): is the B operand's L2 traffic what bounds stage 2 of deep_gemm?  shipped vs a timing build whose
# every B-operand group is the same 16 rows (L1 hits; wrong results)
echo "== shipped"; timeout 300 python tools/deep_time.py
echo "== every B-operand group from L1 (-DDEEP_DEV_B_L1, wrong results, timing only)"
CONV3P_HIP_LIB=devlibs/lib_bl1.so timeout 300 python tools/deep_time.py
