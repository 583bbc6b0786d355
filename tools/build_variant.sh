#!/bin/bash
# A tuning build of the library with extra macro definitions: tools/build_variant.sh <name> -DX=1 ...
# (NO_LICM= in the environment builds with machine LICM left on, as the compiler defaults)
# -> tools/_build/librtx_<name>.so (load it with RTX_AMD_LIB).  Host objects come from the regular build.
set -e
name=$1; shift
cd "$(dirname "$0")/../raytracing_rust_amd/csrc"
mkdir -p ../../tools/_build build
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function ${NO_LICM--mllvm -disable-machine-licm} "$@" -c kernels.hip -o build/kernels_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/_build/librtx_$name.so build/kernels_$name.o build/bvh_build.o build/scene_graph.o build/scenes.o -lz
rm -f build/kernels_$name.o
echo tools/_build/librtx_$name.so
