#!/bin/bash
# Work list: 4 / 8 / 16 cost buckets to an octave, interleaved in one process per configuration (tools/ab_bench.py variants
# base, sub4, sub16; build them first with: python tools/ab_bench.py --build --only base,sub4,sub16).
set -e
for cfg in "cornell reference 256" "instanced reference 256" "suzanne reference 256" "monkeyhead2 reference 256" "cornell materials 64" "instanced materials 32"; do
    set -- $cfg
    echo "== $1 $2 spp=$3"
    timeout -k 10 150 python tools/ab_bench.py --workload $1 --shading $2 --spp $3 --rounds 7 --only base,sub4,sub16
done
