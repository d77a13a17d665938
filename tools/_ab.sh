set -e
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/exp_test.log 2>&1
for rep in 1 2; do
for lib in build_variants/prev/librtx.so raytracing_amd/librtx.so; do
for wl in c5 c3 "c5 --mode wavefront" s200; do
  echo -n "$lib $wl " >> gpurun_out/ab.log
  RTX_LIB=$lib timeout -k 10 120 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-stats | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/ab.log
done
done
done
