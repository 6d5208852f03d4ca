#!/bin/bash
# A/B of a compile-time constant without touching the product library: build ONE source with extra -D flags and link it with the product
# objects of the others into gigapose_amd/lib<name>.so; run anything against it with GIGAPOSE_LIB=gigapose_amd/lib<name>.so.
#   tools/build_variant.sh <name> <source.hip> -DFOO=1 ...
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../gigapose_amd/csrc"
make -s -j8 ../libgigapose_hip.so
mkdir -p _build_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" -c $src -o _build_$name/$src.o
objs=$(ls _build/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib$name.so $objs _build_$name/$src.o
echo "built gigapose_amd/lib$name.so"
