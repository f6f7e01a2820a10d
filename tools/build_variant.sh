#!/bin/bash
# A/B builds of ONE translation unit: tools/build_variant.sh <name> <file.hip> [-DFLAG ...] -> hiast_amd/csrc/_ab/libhiast_<name>.so
# (the other objects are the in-tree build's; select the library with HIAST_LIB=<path>)
set -e
cd "$(dirname "$0")/../hiast_amd/csrc"
name=$1; src=$2; shift 2
make -s libhiast_hip.so
mkdir -p _ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
    -Rpass-analysis=kernel-resource-usage "$@" -c "$src" -o "_ab/${name}.o" 2> "_ab/${name}.resources" || { grep -v "remark:" "_ab/${name}.resources"; exit 1; }
grep -E "ScratchSize \[bytes/lane\]: [1-9]" "_ab/${name}.resources" && echo "WARNING: ${name} spills"
objs=$(ls _obj/*.o | grep -v "_obj/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "_ab/libhiast_${name}.so" $objs "_ab/${name}.o"
echo "built _ab/libhiast_${name}.so"
