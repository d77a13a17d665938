#!/bin/bash
# Registers, spills, scratch, LDS and occupancy of every kernel of librtx.so, from the compiler's
# own resource report (-Rpass-analysis=kernel-resource-usage).  Usage: tools/resource_usage.sh > out.txt
set -e
cd "$(dirname "$0")/../raytracing_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -x hip -c -Rpass-analysis=kernel-resource-usage ${EXTRA_FLAGS}"
printf "%-72s %5s %5s %5s %7s %7s %8s %4s %7s\n" kernel SGPR VGPR AGPR spillS spillV scratchB occ ldsB
for f in rtx_kernels.hip rtx_wavefront.hip rtx_api.hip; do
  /opt/rocm/bin/hipcc $FLAGS -o /dev/null $f 2>&1 | awk '
    /Function Name:/ {name=$(NF-1)}
    / TotalSGPRs:/ {sg=$(NF-1)} / VGPRs:/ && !/Spill/ {vg=$(NF-1)} / AGPRs:/ {ag=$(NF-1)}
    /SGPRs Spill:/ {ss=$(NF-1)} /VGPRs Spill:/ {vs=$(NF-1)} /ScratchSize/ {sc=$(NF-1)}
    /Occupancy/ {oc=$(NF-1)}
    /LDS Size/ {lds=$(NF-1); cmd="c++filt " name; cmd | getline dn; close(cmd);
                gsub(/rtx::/,"",dn); gsub(/\(anonymous namespace\)::/,"",dn); gsub(/RenderParams.*/,"",dn); gsub(/void /,"",dn);
                printf "%-72s %5s %5s %5s %7s %7s %8s %4s %7s\n", substr(dn,1,72), sg, vg, ag, ss, vs, sc, oc, lds}'
done
