#!/bin/bash
# Builds a variant of the engine library for A/B measurements: tools/build_variant.sh <name> [-D... flags]
# -> abv/libkg_<name>.so (not in git; travels to the GPU box).  Run tools/ab.sh <tag> <name> ... inside a gpurun call.
set -e
NAME=$1; shift
mkdir -p abv build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-command-line-argument "$@" -c -o build/engine_capi_$NAME.o kangaroo_amd/csrc/engine_capi.hip
[ -f build/host_secp.o ] || g++ -O2 -std=c++17 -fPIC -c -o build/host_secp.o kangaroo_amd/csrc/host_secp.cpp
hipcc --offload-arch=gfx950 -shared -fPIC -o abv/libkg_$NAME.so build/engine_capi_$NAME.o build/host_secp.o -lpthread
python3 tools/check_store_hazard.py abv/libkg_$NAME.so
