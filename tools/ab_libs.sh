#!/bin/bash
# developer (ON THE GPU BOX): tools/ab_libs.sh "<script args...>" lib1 lib2 ... -- runs `python <script args>` under each devlibs/lib_<name>.so
cd "$(dirname "$0")/.."
cmd=$1; shift
for lib in "$@"; do
  export CONV3P_HIP_LIB=$PWD/devlibs/lib_$lib.so
  echo "== $lib: $(python $cmd 2>&1 | tail -1)"
done
