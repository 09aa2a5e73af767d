#!/bin/bash
# Materials mode, camera rays through the direction grid (default) against the hierarchy for every ray (PT_CAMERA_RAY_GRID=0).
set -e
mkdir -p gpurun_out
out=gpurun_out/camera_grid_ab.log
: > $out
for g in 0 1; do
  echo "PT_CAMERA_RAY_GRID=$g" >> $out
  for wl in instanced suzanne monkeyhead2; do
    for spp in 1 16 64; do
      PT_CAMERA_RAY_GRID=$g timeout -k 10 120 python3 tools/one_dispatch.py --workload $wl --shading materials --spp $spp --reps 4 | cut -c1-260 >> $out
    done
  done
done
cat $out
