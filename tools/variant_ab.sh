# Times experimental builds of the library (pathtracer_amd/lib/libpt_hip__<tag>.so) against the product build on the
# materials-mode workloads. usage: bash tools/variant_ab.sh tag1 tag2 ...
cd $GRAFT_REPO_ROOT
for w in "instanced --spp 16" "suzanne --spp 64" "monkeyhead2 --spp 64"; do
  for tag in "" "$@"; do
    lib=""; [ -n "$tag" ] && lib="--lib pathtracer_amd/lib/libpt_hip__$tag.so"
    r=$(timeout -k 10 120 python3 tools/one_dispatch.py --workload $w --shading materials --reps 5 $lib | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(min(d['per_rep_ms'][1:]), d['per_rep_ms'])") || exit 1
    echo "$w [${tag:-product}] $r"
  done
done
