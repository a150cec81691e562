#!/bin/bash
# A/B of two library builds under any script: profiles/ab_script.sh <a.so> <b.so> <python script> [grep pattern]
a=$1; b=$2; script=$3; pat=${4:-.}
lib=parasail-rs_amd/lib/libparasail_amd.so
cp $lib /tmp/keep.so
for which in $a $b $a $b; do
  cp $which $lib
  echo "== $which"
  timeout -k 10 300 python $script 2>/dev/null | grep -E "$pat" || exit 1
done
cp /tmp/keep.so $lib
