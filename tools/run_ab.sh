# A/B timing of library variants on the GPU box:  tools/run_ab.sh "c3 c5" "name1 name2 ..."   (name "main" = the in-tree library)
set -o pipefail
mkdir -p gpurun_out
for w in $1; do
  for v in $2; do
    lib=build_variants/$v/librtx.so; [ "$v" = main ] && lib=raytracing_amd/librtx.so
    RTX_LIB=$lib timeout -k 10 120 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-stats 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['config']['workload'][:3], round(d['ms_per_step'],2), d['checksum'])" || exit 1
  done
done
for w in $1; do
  for v in $3; do
    RTX_LIB=build_variants/$v/librtx.so timeout -k 10 120 python tools/mk_debug.py $w 16 2>&1 | tail -1 | sed "s/^/$v $w: /" || exit 1
  done
done
