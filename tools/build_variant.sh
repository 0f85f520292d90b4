#!/bin/bash
# Build an experimental variant of the library for tools/ab.sh:  tools/build_variant.sh NAME FILE.hip "-DSOMETHING ..."
# recompiles FILE with the extra flags, links it with the product build's other objects -> ab/libldpc_NAME.so
set -e -o pipefail
name=${1:?name}; src=${2:?source under libldpc_amd/csrc}; extra=${3:-}
cd "$(dirname "$0")/.."
[ -n "$SKIP_PRODUCT_BUILD" ] || python3 -m libldpc_amd.build > /dev/null
obj=$(python3 -c "from libldpc_amd import build; print(build.OBJ)")
mkdir -p ab /tmp/ldpc_variant
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden $extra \
  -c libldpc_amd/csrc/$src -o /tmp/ldpc_variant/$name.o
objs=$(ls $obj/*.o | grep -v "/$src.o" | grep -v ldpcsim_main)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ab/libldpc_$name.so $objs /tmp/ldpc_variant/$name.o
echo "ab/libldpc_$name.so"
