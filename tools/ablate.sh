#!/bin/bash
# developer tool: build ablated variants of the HIP library into build/ and time them on the GPU box
set -e
cd "$(dirname "$0")/.."
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fPIC -shared \
     -DCONV3P_ABLATE=$a -o build/libconv3p_ablate_$a.so pointwise_amd/csrc/conv3p_abi.hip
done
