#!/bin/bash
# usage (inside a gpurun command): scripts/ab_variants.sh "bench command" base v1 v2 ...   -- alternates the library variants
# (gokalman_amd/_variants/lib<name>.so; "base" = the regular build) three times each and prints the command's output lines.
cmd=$1; shift
cp gokalman_amd/libgokalman_amd.so /tmp/libbase.so
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = base ]; then cp /tmp/libbase.so gokalman_amd/libgokalman_amd.so; else cp gokalman_amd/_variants/lib$v.so gokalman_amd/libgokalman_amd.so; fi
    echo "== $v (round $round)"; eval "$cmd"
  done
done
cp /tmp/libbase.so gokalman_amd/libgokalman_amd.so
