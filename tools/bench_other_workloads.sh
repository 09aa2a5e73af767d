# bench.py lines of the other configurations (single GPU), appended to gpurun_out/bench_workloads.jsonl
set -e
out=gpurun_out/bench_workloads.jsonl
: > $out
timeout -k 10 200 python bench.py --lean --steps 10 --warmup 2 --workload monkeyhead2 >> $out
timeout -k 10 200 python bench.py --lean --steps 10 --warmup 2 --workload suzanne >> $out
timeout -k 10 200 python bench.py --lean --steps 8 --warmup 3 --workload instanced --spp 256 >> $out
timeout -k 10 200 python bench.py --lean --steps 10 --warmup 2 --workload world >> $out
timeout -k 10 200 python bench.py --lean --steps 10 --warmup 2 --workload torus --spp 64 >> $out
timeout -k 10 300 python bench.py --lean --steps 2 --warmup 4 --workload instanced --spp 4096 >> $out   # (warm-up: the cold dispatch and the three that time the two forms of the grid kernel)
