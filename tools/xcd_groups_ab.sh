# Work list dealt to the XCDs in image bands of equal cost (PT_XCD_GROUPS=1) against one cost-ordered list (=0): time and L2 hit rate
set -e
cd $GRAFT_REPO_ROOT
for cfg in "--workload instanced --shading materials --spp 16" "--workload instanced --spp 256" "--workload instanced --spp 64" "--workload suzanne --shading materials --spp 64" "--workload cornell --spp 256" "--workload monkeyhead2 --shading materials --spp 64"; do
  for g in 0 1; do
    echo "== $cfg PT_XCD_GROUPS=$g"
    PT_XCD_GROUPS=$g timeout -k 10 120 python3 tools/one_dispatch.py $cfg --reps 6 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['per_rep_ms'])"
  done
done
A="--workload instanced --shading materials --spp 16 --reps 3"
PT_XCD_GROUPS=0 bash tools/l2_quick.sh groups0 "" $A
PT_XCD_GROUPS=1 bash tools/l2_quick.sh groups1 "" $A
