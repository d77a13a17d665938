set -e
mkdir -p gpurun_out/r2
for wl in c2 c3 c5; do for mode in megakernel wavefront; do
  timeout -k 10 200 python bench.py --workload $wl --mode $mode --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2/base_${wl}_${mode}.json 2> gpurun_out/r2/base_${wl}_${mode}.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2/base_${wl}_${mode}.json"))
print("${wl} ${mode}", round(d["ms_per_step"],2),"ms", round(d["value"],1),"Msamples/s", "frac", round(d["roofline"]["frac"],4), "rays/sample", round(d["roofline"]["rays_per_sample"],2))
PY
done; done
