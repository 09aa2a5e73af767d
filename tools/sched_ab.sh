cd $GRAFT_REPO_ROOT
for w in "instanced --spp 16 --shading materials" "instanced --spp 64" "suzanne --spp 64 --shading materials"; do
  for v in "PT_TILE_SCHEDULE=list" "PT_TILE_SCHEDULE=tile" "PT_TILE_SCHEDULE=0" "PT_SCHED_SORT=0"; do
    r=$(env $v timeout -k 10 120 python3 tools/one_dispatch.py --workload $w --reps 6 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(min(d['per_rep_ms'][2:]), d['per_rep_ms'])") || exit 1
    echo "$w [$v] $r"
  done
done
