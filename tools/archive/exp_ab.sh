#!/bin/bash
# Ad-hoc A/B on one box: bench line of the given library variants (names after libvrt_hip), tile tags on and off
for name in "$@"; do
  lib=voxel-raytracing_amd/csrc/libvrt_hip${name}.so
  for tags in 1 0; do
    VRT_TILE_TAGS=$tags VRT_LIB=$PWD/$lib timeout -k 10 300 python bench.py 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d['extra_configs']
print('lib${name}', 'tags=$tags', 'batched ms', d['ms_per_step'], 'single', d['roofline']['single_frame_launch']['kernel_ms'], 'cfg3', e[0]['geometry_ms'], 'defaults', e[2]['geometry_ms'], 'cfg5', e[3]['geometry_ms'], flush=True)"
  done
done
