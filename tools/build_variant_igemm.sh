#!/bin/bash
# two translation units with one flag: igemm.hip + igemm_f16.hip -> hiast_amd/csrc/_ab/libhiast_<name>.so
set -e
cd "$(dirname "$0")/../hiast_amd/csrc"
name=$1; shift
make -s libhiast_hip.so
mkdir -p _ab
for src in igemm igemm_f16; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
    -Rpass-analysis=kernel-resource-usage "$@" -c "$src.hip" -o "_ab/${name}_$src.o" 2> "_ab/${name}_$src.resources" || { grep -v "remark:" "_ab/${name}_$src.resources"; exit 1; }
grep -E "ScratchSize \[bytes/lane\]: [1-9]" "_ab/${name}_$src.resources" && echo "WARNING: spills"
done
objs=$(ls _obj/*.o | grep -v "_obj/igemm.o" | grep -v "_obj/igemm_f16.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "_ab/libhiast_${name}.so" $objs "_ab/${name}_igemm.o" "_ab/${name}_igemm_f16.o"
echo "built _ab/libhiast_${name}.so"
