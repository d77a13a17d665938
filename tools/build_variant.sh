#!/bin/bash
# Builds another librtx.so with extra compiler flags into build_variants/<name>/ for A/B timing:
#   tools/build_variant.sh name "-DRTX_WF_REFILL=8"        then      RTX_LIB=build_variants/name/librtx.so python bench.py ...
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build_variants/$name
mkdir -p $out
cd $root/raytracing_amd/csrc
for f in rtx_api.hip rtx_kernels.hip rtx_wavefront.hip rtx_bvh.cpp; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wno-unused-function $flags -x hip -c -o $out/${f%.*}.o $f &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/librtx.so $out/*.o -ldl -pthread
rm -f $out/*.o
echo built $out/librtx.so
