#!/bin/bash
# usage: scripts/build_variant_multi.sh NAME "-DFOO=1 ..." file1.hip file2.hip ...   -> gokalman_amd/_variants/libNAME.so
# Several translation units rebuilt with extra defines (in parallel), linked with the cached objects of the regular build (see build_variant.sh).
set -e
cd "$(dirname "$0")/.."
name=$1; defs=$2; shift 2
obj=gokalman_amd/csrc/_obj
mkdir -p gokalman_amd/_variants /tmp/variant_$name
skip=""
for src in "$@"; do
  extra=$(python3 -c "import sys; sys.path.insert(0, '.'); from gokalman_amd import build as b; print(' '.join(b.EXTRA.get('$src', [])))")
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast-honor-pragmas -fno-fast-math $extra $defs -c gokalman_amd/csrc/$src -o /tmp/variant_$name/$src.o 2> /dev/null &
  skip="$skip -e /$src.o"
done
wait
objs=$(ls $obj/*.o | grep -v $skip)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o gokalman_amd/_variants/lib$name.so $objs /tmp/variant_$name/*.o
echo built gokalman_amd/_variants/lib$name.so
