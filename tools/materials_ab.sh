# Materials mode (diffuse/emissive), LBVH walk variants side by side: PT_LANE_WALK=1 lanes decoupled, 2 lanes in step, 0 per wave.
cd $GRAFT_REPO_ROOT
for w in "instanced --spp 16" "suzanne --spp 64" "monkeyhead2 --spp 64" "torus_distorted --spp 16"; do
  for m in 1 2; do
    echo "== $w PT_LANE_WALK=$m"
    PT_LANE_WALK=$m timeout -k 10 120 python3 tools/one_dispatch.py --workload $w --shading materials --reps 4 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('kernel_ms','per_rep_ms','Grays_per_s','traced_Grays_per_s','lane_utilisation','grid_tests_per_traced_ray')})" || exit 1
  done
done
