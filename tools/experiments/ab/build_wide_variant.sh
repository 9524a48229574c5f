#!/bin/bash
# usage: build_wide_variant.sh <name> [extra hipcc flags...]   -> tools/experiments/ab/libhpx_<name>.so
# hpx_factor.hip, hpx_factor_wide.hip and hpx_factor_split.hip rebuilt with the flags (e.g. -DHPX_WIDE_MIN=16 -DHPX_WIDE_KC=8), the
# product's other objects reused (run `make -C hydra_pspec_amd/csrc` first).
set -e
ROOT=$(cd "$(dirname "$0")/../../.." && pwd)
CS=$ROOT/hydra_pspec_amd/csrc
name=$1; shift
for f in hpx_factor hpx_factor_wide hpx_factor_split; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I$CS "$@" -c $CS/$f.hip -o /tmp/_v_${name}_$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/experiments/ab/libhpx_$name.so /tmp/_v_${name}_hpx_factor.o /tmp/_v_${name}_hpx_factor_wide.o /tmp/_v_${name}_hpx_factor_split.o \
  $CS/hpx_backsolve.o $CS/hpx_backsolve_lds.o $CS/hpx_transform.o $CS/hpx_plan.o $CS/hpx_setup.o $CS/hpx_chain.o $CS/hpx_post.o $CS/hpx_woodbury.o $CS/hpx_extra.o $CS/hpx_flat.o $CS/hpx_lowrank.o $CS/hpx_modes.o $CS/hpx_sqrtm.o $CS/hpx_eigh.o
echo built libhpx_$name.so
