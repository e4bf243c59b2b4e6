#!/bin/bash
# A/B of two builds of libstitch_gfx950.so in ONE gpurun call (same box): bash tools/ab_lib.sh [ROUNDS]
#   _ab/libstitch_base.so = the baseline build (git-ignored, travels with the snapshot), the in-tree library = the candidate.
# Alternates base / new ROUNDS times; prints pairs/s (3 in flight) and the 1-in-flight figure of every run.
export TMPDIR=/tmp
PKG=seamless-through-breaking-rethinking-image-stitching-for-optimal-alignment_amd
R=${1:-2}
cp $PKG/libstitch_gfx950.so /tmp/new.so
run() {
  python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-corr-roofline --harness none 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1 pairs/s', round(d['value'], 2), '1-in-flight', round(d.get('value_1_in_flight', 0), 2))"
}
for i in $(seq $R); do
  cp _ab/libstitch_base.so $PKG/libstitch_gfx950.so; run base
  cp /tmp/new.so $PKG/libstitch_gfx950.so; run new
done
cp /tmp/new.so $PKG/libstitch_gfx950.so
