# split factor of the work list (PT_SPLIT_FACTOR) for the materials-mode walks: PT_LANE_WALK=1 lanes decoupled, 2 lanes in step
# (profiles/r02_materials_split_factor.log was taken when 3 / 1 selected the decoupled kernel / the dropped pool kernel)
cd $GRAFT_REPO_ROOT
for w in "suzanne --spp 64" "monkeyhead2 --spp 64" "instanced --spp 16" "cornell --spp 16 --accel bvh"; do
  for lw in 1 2; do
    for f in 0.5 1 2 4 16 64; do
      r=$(PT_SPLIT_FACTOR=$f PT_LANE_WALK=$lw timeout -k 10 120 python3 tools/one_dispatch.py --workload $w --shading materials --reps 5 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(min(d['per_rep_ms'][2:]), d['per_rep_ms'])") || exit 1
      echo "$w [walk $lw split $f] $r"
    done
  done
done
