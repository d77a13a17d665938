#!/bin/bash
# As tools/build_kernel_variant.sh, for flags that only touch rtx_api.hip.   tools/build_api_variant.sh name "-DRTX_CHUNK_MAX=1u"
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build_variants/$name
mkdir -p $out
cd $root/raytracing_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wno-unused-function $flags -x hip -c -o $out/rtx_api.o rtx_api.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/librtx.so $out/rtx_api.o build/rtx_kernels.o build/rtx_wavefront.o build/rtx_bvh.o -ldl -pthread
rm -f $out/*.o
echo built $out/librtx.so
