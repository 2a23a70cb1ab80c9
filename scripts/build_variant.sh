#!/bin/bash
# build an experiment variant of the library: build_variant.sh name -DFLAG=..   -> build/ab_<name>.so
N=$1; shift
mkdir -p build; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value "$@" tangram_amd/csrc/tg_capi.hip -o build/ab_$N.so
