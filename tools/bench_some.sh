# usage: bash tools/bench_some.sh "c3:wavefront c5:wavefront ..." [extra bench args]
set -e
mkdir -p gpurun_out/r2
for item in $1; do
  wl=${item%%:*}; mode=${item##*:}
  timeout -k 10 200 python bench.py --workload $wl --mode $mode --steps 3 --warmup 1 --no-cpu-baseline ${@:2} > gpurun_out/r2/b_${wl}_${mode}.json 2> gpurun_out/r2/b_${wl}_${mode}.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2/b_${wl}_${mode}.json"))
print("${wl} ${mode}", round(d["ms_per_step"],2),"ms", round(d["value"],1),"Msamples/s", "frac", round(d["roofline"]["frac"],4), "checksum", d["checksum"])
PY
done
