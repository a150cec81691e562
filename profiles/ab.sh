#!/bin/bash
# A/B of two builds of the library on ONE box: profiles/ab.sh <config> <a.so> <b.so> [rounds]  (alternates a, b, a, b ...)
cfg=$1; a=$2; b=$3; rounds=${4:-3}
lib=parasail-rs_amd/lib/libparasail_amd.so
cp $lib /tmp/keep.so
for r in $(seq $rounds); do
  for which in $a $b; do
    cp $which $lib
    timeout -k 10 200 python bench.py --config $cfg --no-cpu-baseline --steps 40 --warmup 5 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$which', d['value'], d['ms_per_step'])" || exit 1
  done
done
cp /tmp/keep.so $lib
