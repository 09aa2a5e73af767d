# Reference shading (direction grid): experimental builds against the product build. usage: bash tools/grid_ab.sh tag...
cd $GRAFT_REPO_ROOT
for w in "cornell --spp 256" "instanced --spp 64" "suzanne --spp 256" "monkeyhead2 --spp 256"; do
  for tag in "" "$@"; do
    lib=""; [ -n "$tag" ] && lib="--lib pathtracer_amd/lib/libpt_hip__$tag.so"
    r=$(timeout -k 10 120 python3 tools/one_dispatch.py --workload $w --reps 6 $lib | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(min(d['per_rep_ms'][1:]), d['per_rep_ms'])") || exit 1
    echo "$w [${tag:-product}] $r"
  done
done
