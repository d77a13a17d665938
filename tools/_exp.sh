set -e
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/exp_test.log 2>&1
for cfg in "c2" "c5" "c3" "c5 --mode wavefront" "c3 --mode wavefront" "c2 --mode wavefront"; do
  echo "$cfg" >> gpurun_out/exp.log
  timeout -k 10 120 python bench.py --workload $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-stats | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/exp.log
done
