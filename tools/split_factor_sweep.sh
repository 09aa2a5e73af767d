# one line per (workload, split factor): kernel ms of 3 dispatches (tools/one_dispatch.py)
set -e
out=gpurun_out/split_sweep.log
: > $out
for w in cornell monkeyhead2 suzanne; do for f in 0 1 2 4 8; do
  echo "== $w F=$f" >> $out
  PT_SPLIT_FACTOR=$f timeout -k 10 100 python tools/one_dispatch.py --workload $w --spp 256 --reps 4 >> $out 2>&1
done; done
for f in 0 1 2 4 8; do
  echo "== instanced F=$f" >> $out
  PT_SPLIT_FACTOR=$f timeout -k 10 100 python tools/one_dispatch.py --workload instanced --spp 16 --reps 4 >> $out 2>&1
done
