# Camera-ray hits kept between dispatches, off and on: frame loops (1 sample per dispatch) and the 256-spp step
set -e
for on in 0 1; do
  export PT_PRIMARY_CACHE=$on
  echo "{\"primary_cache\": $on}"
  for w in world cornell monkeyhead2 suzanne instanced; do
    timeout -k 10 120 python tools/one_dispatch.py --workload $w --spp 1 --reps 64
  done
  timeout -k 10 200 python bench.py --lean --steps 10 --warmup 3 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'bench_ms_per_step': d['ms_per_step']}))"
done
