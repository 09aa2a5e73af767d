# Direction grid with the 4x4-block occupancy bits consulted first (PT_GRID_OCC=1), never (0), or as the build decides (unset)
cd $GRAFT_REPO_ROOT
for w in "instanced --spp 64" "instanced --spp 256" "cornell --spp 256" "suzanne --spp 256" "monkeyhead2 --spp 256" "torus_distorted --spp 64" "world --spp 256"; do
  for v in "PT_GRID_OCC=0" "PT_GRID_OCC=1" "PT_DUMMY=1"; do
    r=$(env $v timeout -k 10 120 python3 tools/one_dispatch.py --workload $w --reps 6 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(min(d['per_rep_ms'][2:]), d['per_rep_ms'])") || exit 1
    echo "$w [$v] $r"
  done
done
