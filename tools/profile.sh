#!/bin/bash
# Runs on the GPU box:  tools/profile.sh <workload> <mode> [extra bench args]
# rocprofv3 kernel trace (with stats) of a 3-step bench run, then four counter passes of one
# launch each (separate runs: counters are not multiplexed), into gpurun_out/prof_<workload>_<mode>/.
set -e -o pipefail
W=${1:-c2}; M=${2:-megakernel}; shift 2 || true
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_${W}_${M}
mkdir -p "$OUT"
B="python3 $R/bench.py --workload $W --mode $M --steps 1 --warmup 0 --no-cpu-baseline --no-stats --no-host $*"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --workload "$W" --mode "$M" --steps 3 --warmup 1 --no-cpu-baseline --no-host "$@" > "$OUT/trace.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/a" -- $B > "$OUT/a.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_CVT --output-format csv -d "$OUT/b" -- $B > "$OUT/b.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/c" -- $B > "$OUT/c.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_VMEM --output-format csv -d "$OUT/d" -- $B > "$OUT/d.log" 2>&1
grep "^{" "$OUT/trace.log" | tail -1 | cut -c1-160
