#!/bin/bash
# Static instruction histogram of one kernel of rtx_kernels.hip (default: the megakernel of the
# reference scene, k_megakernel_scan2<kSrcLds, no stats, few lights, no recording>), from the compiler's
# own assembly: which opcodes the VALU instructions of the kernel are made of.
#   tools/isa_histogram.sh [mangled kernel name] > profiles/<round>/isa_histogram_<kernel>.txt
set -e
K=${1:-_ZN3rtx18k_megakernel_scan2ILi0ELb0ELb0ELb0EEEvNS_12RenderParamsE}
cd "$(dirname "$0")/../raytracing_amd/csrc"
S=$(mktemp /tmp/rtxXXXX.s)
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -x hip -S --cuda-device-only -o $S rtx_kernels.hip 2>/dev/null
awk -v k="^$K:" '$0 ~ k {on=1} on {print} on && /^\.Lfunc_end/ {exit}' $S > $S.k
total=$(grep -cE "^\s+(v_|s_|ds_|global_|scratch_|buffer_|flat_)" $S.k)
valu=$(grep -cE "^\s+v_" $S.k)
echo "kernel: $(echo $K | c++filt)"
echo "instructions: $total  of which VALU (v_*): $valu"
echo
echo "classes (static count, share of VALU):"
cls() { n=$(grep -cE "^\s+($2)" $S.k || true); awk -v d="$1" -v n=$n -v v=$valu 'BEGIN { printf "  %-100s %5d  %5.1f %%\n", d, n, 100 * n / v }'; }
cls "division expansion: v_div_scale / v_div_fmas / v_div_fixup / v_rcp" "v_div_scale|v_div_fmas|v_div_fixup|v_rcp_f32"
cls "packed f32 arithmetic: v_pk_fma / v_pk_mul / v_pk_add" "v_pk_fma_f32|v_pk_mul_f32|v_pk_add_f32"
cls "scalar-lane f32 arithmetic: v_fma / v_fmac / v_mul / v_add / v_sub" "v_fma_f32|v_fmac_f32|v_mul_f32|v_add_f32|v_sub_f32"
cls "register moves: v_mov / v_pk_mov / v_accvgpr" "v_mov_b|v_pk_mov|v_accvgpr"
cls "selects: v_cndmask" "v_cndmask"
cls "compares: v_cmp*" "v_cmp"
cls "min / max / med3" "v_min|v_max|v_med3"
cls "square root: v_sqrt / v_rsq" "v_sqrt|v_rsq"
cls "integer and bit ops (Philox, indices, sign flips: v_xor, v_and, v_add_u32, v_mad_u64 ...)" "v_xor|v_and_|v_or_|v_add_u32|v_sub_u32|v_mad_u64|v_mul_lo|v_mul_hi|v_lsh|v_bfe|v_add_co|v_addc|v_not|v_bfi|v_alignbit|v_add3|v_lshl_add|v_mad_u32"
cls "conversions: v_cvt*" "v_cvt"
cls "cross-lane / spill traffic: v_readlane / v_writelane / v_readfirstlane / ds_bpermute" "v_readlane|v_writelane|v_readfirstlane|ds_bpermute"
echo
echo "other counts: s_nop $(grep -cE '^\s+s_nop' $S.k)  LDS (ds_*) $(grep -cE '^\s+ds_' $S.k)  global/scratch memory $(grep -cE '^\s+(global_|scratch_|buffer_|flat_)' $S.k)  branches $(grep -cE '^\s+s_cbranch|^\s+s_branch' $S.k)"
echo
echo "all opcodes:"
grep -E "^\s+(v_|s_|ds_|global_|scratch_|buffer_|flat_)" $S.k | awk '{print $1}' | sort | uniq -c | sort -rn
rm -f $S $S.k
