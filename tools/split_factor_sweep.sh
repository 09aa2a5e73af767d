# one line per (workload, split factor): per-dispatch ms of 5 dispatches and what the scheduler split (tools/one_dispatch.py)
out=gpurun_out/split_sweep.log
: > $out
for w in cornell monkeyhead2 suzanne world; do for f in 1 2 3 4 6; do
  echo "== $w F=$f" >> $out
  PT_DEBUG_SCHED=1 PT_SPLIT_FACTOR=$f timeout -k 10 100 python tools/one_dispatch.py --workload $w --spp 256 --reps 5 2>&1 | grep -o "per_rep_ms[^]]*\]\|[0-9]* items, [0-9]* split tiles" >> $out || exit 1
done; done
for w in instanced torus; do for f in 1 2 3 4 6; do
  echo "== $w F=$f" >> $out
  PT_DEBUG_SCHED=1 PT_SPLIT_FACTOR=$f timeout -k 10 100 python tools/one_dispatch.py --workload $w --spp 64 --reps 5 2>&1 | grep -o "per_rep_ms[^]]*\]\|[0-9]* items, [0-9]* split tiles" >> $out || exit 1
done; done
