# How the decoupled walk's time follows its vector instruction count: builds with 40 and 80 do-nothing vector instructions added to
# every node step (-DPT_V_EXTRA_VALU=n), against the product build. On the GPU box, after building the variants here:
#   python3 -c "from pathtracer_amd import build as b; [b.build_hip(force=True, extra_flags=['-DPT_V_EXTRA_VALU=%d' % n], tag='valu%d' % n) for n in (40, 80)]"
set -e
cd $GRAFT_REPO_ROOT
for lib in "" __valu40 __valu80; do
  echo "variant ${lib:-product}"
  PT_BVH_BUILDER=sah timeout -k 10 200 python3 tools/bvh_builder_ab.py --reps 3 --builders sah --workloads instanced:16,suzanne:64 --lib pathtracer_amd/lib/libpt_hip$lib.so | grep builder
done
