#!/bin/bash
# developer: build A/B variants of the HIP library into devlibs/ (git-ignored, travels to the GPU box), in parallel
#   tools/build_variants.sh name1 "-DFLAG1 -DFLAG2" name2 "-DFLAG3" ...
cd "$(dirname "$0")/.."
mkdir -p devlibs
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fPIC -shared $f \
     -o devlibs/lib_$n.so pointwise_amd/csrc/conv3p_abi.hip 2> devlibs/build_$n.log &
done
wait
ls -la devlibs/*.so
