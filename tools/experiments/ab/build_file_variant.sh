#!/bin/bash
# usage: build_file_variant.sh <name> <csrc file without .hip> [extra hipcc flags...]  -> tools/experiments/ab/libhpx_<name>.so
# ONE translation unit rebuilt with the flags, the product's other objects reused (run `make -C hydra_pspec_amd/csrc` first).
set -e
ROOT=$(cd "$(dirname "$0")/../../.." && pwd)
CS=$ROOT/hydra_pspec_amd/csrc
name=$1; unit=$2; shift 2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I$CS "$@" -c $CS/$unit.hip -o /tmp/_fv_${name}_$unit.o
objs=""
for o in $CS/*.o; do
  if [ "$(basename $o)" = "$unit.o" ]; then objs="$objs /tmp/_fv_${name}_$unit.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/experiments/ab/libhpx_$name.so $objs
echo built libhpx_$name.so
