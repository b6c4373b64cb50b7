for b in 64 96 128; do
  timeout -k 10 500 python bench.py --steps 1 --warmup 1 --batch $b --no-secondary --no-cpu-baseline > gpurun_out/bench_b$b.json 2> gpurun_out/bench_b$b.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_b$b.json").read().strip().splitlines()[-1]); print($b, d["value"], d["ms_per_step"])
PY
done
