#!/bin/bash
# Materials mode: dispatch time against the number of samples, to separate what a dispatch pays once per pixel (the camera
# ray) from what it pays per sample.
set -e
mkdir -p gpurun_out
out=gpurun_out/primary_share.log
: > $out
for wl in instanced suzanne; do
  for spp in 1 4 16 64; do
    timeout -k 10 120 python3 tools/one_dispatch.py --workload $wl --shading materials --spp $spp --reps 4 >> $out
  done
done
cat $out | cut -c1-330
