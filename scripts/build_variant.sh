#!/bin/bash
# usage: scripts/build_variant.sh NAME file.hip "-DFOO=1 ..."   -> gokalman_amd/_variants/libNAME.so
# One translation unit rebuilt with extra defines, linked with the cached objects of the regular build: A/B kernel
# experiments on the GPU box (copy the variant over gokalman_amd/libgokalman_amd.so inside the gpurun command).
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; defs=$3
obj=gokalman_amd/csrc/_obj
# the per-file flags come from gokalman_amd/build.py (EXTRA): the variant is the library's kernel plus the defines, nothing else
extra=$(python3 -c "import sys; sys.path.insert(0, '.'); from gokalman_amd import build as b; print(' '.join(b.EXTRA.get('$src', [])))")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast-honor-pragmas -fno-fast-math $extra $defs -c gokalman_amd/csrc/$src -o /tmp/variant_$name.o
objs=$(ls $obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o gokalman_amd/_variants/lib$name.so $objs /tmp/variant_$name.o
echo built gokalman_amd/_variants/lib$name.so
