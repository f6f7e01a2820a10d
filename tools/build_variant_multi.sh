#!/bin/bash
# A/B build of SEVERAL translation units with the same flags:
#   tools/build_variant_multi.sh <name> "<a.hip b.hip ...>" [-DFLAG ...] -> hiast_amd/csrc/_ab/libhiast_<name>.so
# (the other objects are the in-tree build's; select the library with HIAST_LIB=<path>)
set -e
cd "$(dirname "$0")/../hiast_amd/csrc"
name=$1; srcs=$2; shift 2
make -s libhiast_hip.so
mkdir -p _ab
objs=$(ls _obj/*.o)
new=""
for src in $srcs; do
  base=${src%.hip}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
      -Rpass-analysis=kernel-resource-usage "$@" -c "$src" -o "_ab/${name}_$base.o" 2> "_ab/${name}_$base.resources" || { grep -v "remark:" "_ab/${name}_$base.resources"; exit 1; }
  grep -E "ScratchSize \[bytes/lane\]: [1-9]" "_ab/${name}_$base.resources" && echo "WARNING: ${name}_$base spills"
  objs=$(echo "$objs" | grep -v "_obj/$base.o")
  new="$new _ab/${name}_$base.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "_ab/libhiast_${name}.so" $objs $new
echo "built _ab/libhiast_${name}.so"
