#!/bin/bash
# As tools/build_variant.sh, for flags that only touch rtx_kernels.hip: compiles that file alone and links it
# with the in-tree build's other objects.   tools/build_kernel_variant.sh name "-DRTX_BVH2_SCATTER_MIN=52u"
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build_variants/$name
mkdir -p $out
cd $root/raytracing_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wno-unused-function $flags -x hip -c -o $out/rtx_kernels.o rtx_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/librtx.so $out/rtx_kernels.o build/rtx_api.o build/rtx_wavefront.o build/rtx_bvh.o -ldl -pthread
rm -f $out/*.o
echo built $out/librtx.so
