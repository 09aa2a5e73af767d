# split factor of the work list (PT_SPLIT_FACTOR) for the two materials-mode kernels
cd $GRAFT_REPO_ROOT
for w in "suzanne --spp 64" "monkeyhead2 --spp 64" "instanced --spp 16" "cornell --spp 16 --accel bvh"; do
  for lw in 3 1; do
    for f in 0.5 1 2 4 16 64; do
      r=$(PT_SPLIT_FACTOR=$f PT_LANE_WALK=$lw timeout -k 10 120 python3 tools/one_dispatch.py --workload $w --shading materials --reps 5 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(min(d['per_rep_ms'][2:]), d['per_rep_ms'])") || exit 1
      echo "$w [walk $lw split $f] $r"
    done
  done
done
