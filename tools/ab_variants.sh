# usage: bash tools/ab_variants.sh "variant1 variant2 ..." "c3:wavefront c5:wavefront" [bench args]
# "main" = raytracing_amd/librtx.so; other names = build_variants/<name>/librtx.so
set -e
mkdir -p gpurun_out/r2
for v in $1; do
  lib=build_variants/$v/librtx.so; [ "$v" = main ] && lib=raytracing_amd/librtx.so
  for item in $2; do
    wl=${item%%:*}; mode=${item##*:}
    RTX_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $wl --mode $mode --steps 3 --warmup 1 --no-cpu-baseline --no-stats ${@:3} > gpurun_out/r2/ab_${v}_${wl}_${mode}.json 2> gpurun_out/r2/ab_${v}_${wl}_${mode}.err || { echo "$v $wl $mode FAILED"; tail -3 gpurun_out/r2/ab_${v}_${wl}_${mode}.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/r2/ab_${v}_${wl}_${mode}.json"))
print("%-14s %s %-10s %8.2f ms %8.1f Msamples/s checksum %.6f" % ("$v", "$wl", "$mode", d["ms_per_step"], d["value"], d["checksum"]))
PY
  done
done
