#!/bin/bash
# A/B inside ONE gpurun call: scripts/ab_libs.sh "<command>" <rounds> <variant> [<variant> ...]   ("base" = the regular library)
# Every round runs the command once per variant, in the given order; the variant library is copied over gokalman_amd/libgokalman_amd.so.
cmd=$1; rounds=$2; shift 2
cp gokalman_amd/libgokalman_amd.so /tmp/base_lib.so
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = base ]; then cp /tmp/base_lib.so gokalman_amd/libgokalman_amd.so; else cp gokalman_amd/_variants/lib$v.so gokalman_amd/libgokalman_amd.so; fi
    echo "== round $r variant $v"
    bash -c "$cmd" 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/base_lib.so gokalman_amd/libgokalman_amd.so
