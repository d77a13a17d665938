# tools/run_ab2.sh "<bench args>" "variants..." : times one bench configuration per library variant
set -o pipefail
for v in $2; do
  lib=build_variants/$v/librtx.so; [ "$v" = main ] && lib=raytracing_amd/librtx.so
  RTX_LIB=$lib timeout -k 10 120 python bench.py $1 --steps 3 --warmup 1 --no-cpu-baseline --no-stats 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$1', round(d['ms_per_step'],2), d['checksum'])" || exit 1
done
