# The tracked counter profiles of the round: one tools/pmc_collect.py run per kernel family (on the GPU box).
# Before sending the tree: python3 tools/ab_bench.py --build --only countfull   (the counting build the loop kernels' entries need)
set -e
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; timeout -k 10 500 python3 tools/pmc_collect.py $tag "$@" > gpurun_out/pmc_$tag.log 2>&1 || { tail -20 gpurun_out/pmc_$tag.log; exit 1; }; tail -3 gpurun_out/pmc_$tag.log; }
run cornell_grid                                                   # pt_render_kernel<4,false>, the headline configuration
run instanced_grid --workload instanced --spp 256 --steps 8 --warmup 3   # <4,false> on the ~1M-triangle scene (fine grid: in use from the second warm-up step on)
run suzanne_bvh --workload suzanne --accel bvh --steps 10          # <2,false>: per-wave LBVH walk
run cornell_loop --accel brute --steps 10                          # <1,false>: the shader's own loop
run cornell_materials --shading materials --steps 6                # <1,true>: diffuse/emissive, all-triangles loop
run instanced_materials --workload instanced --shading materials --spp 16 --steps 6   # pt_render_decoupled_kernel: per-lane LBVH walk, lanes decoupled
run suzanne_materials --workload suzanne --shading materials --spp 64 --steps 6       # pt_render_decoupled_kernel on a mesh that fits the L2
