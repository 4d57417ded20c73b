#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for lib in /root/repo/variants/lib_*.so; do
  echo "== $lib"; rm -rf /tmp/tr
  CONV3P_HIP_LIB=$lib rocprofv3 --kernel-trace --stats -d /tmp/tr -o t -- python /root/repo/tools/stack_time.py cfg4 > /tmp/tr.log 2>&1
  python /root/repo/tools/pmc_query.py /tmp/tr/t_results.db | grep "backward"
done
