#!/bin/bash
# Builds a variant of libmvk.so with extra compiler defines into build_diag/<name>/libmvk.so (git-ignored; it
# travels to the GPU box with the snapshot).  Run a probe against it with MVK_LIB=build_diag/<name>/libmvk.so.
# usage: bash profiles/build_variant.sh <name> [-DMVK_...=...]...
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build_diag/$NAME
mkdir -p $OUT
cd $ROOT/motion_planning_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC -fno-math-errno -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage "$@" -c mvk_kernels.hip -o $OUT/mvk_kernels.o 2> $OUT/kernel_resource_usage.txt || { tail -20 $OUT/kernel_resource_usage.txt; exit 1; }
/opt/rocm/bin/hipcc -shared -fPIC -Wl,-soname,libmvk.so -o $OUT/libmvk.so $OUT/mvk_kernels.o ../lib/obj/mvk_model.o ../lib/obj/mvk_space.o ../lib/obj/mvk_capi.o ../lib/obj/mvk_planner.o ../lib/obj/mvk_grid.o ../lib/obj/mvk_path.o ../lib/obj/mvk_dist.o ../lib/obj/mvk_pipe.o ../lib/obj/mvk_spline.o ../lib/obj/mvk_probe.o -ldl
echo "built $OUT/libmvk.so"
