set -e
out=gpurun_out/res_sweep2.log
: > $out
for w in cornell monkeyhead2 suzanne; do for r in 128 256 512 1024 2048; do
  echo "== $w R=$r" >> $out
  PT_GRID_RES=$r timeout -k 10 100 python tools/one_dispatch.py --workload $w --spp 256 --reps 3 >> $out 2>&1
done; done
for r in 512 1024 2048 4096; do
  echo "== instanced R=$r" >> $out
  PT_GRID_RES=$r timeout -k 10 100 python tools/one_dispatch.py --workload instanced --spp 16 --reps 3 >> $out 2>&1
done
