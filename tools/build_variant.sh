#!/bin/bash
# dev: build_variants/libqpnet_<name>.so with extra compiler flags (select it with QPN_LIB=...)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build_variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" qpnet_amd/csrc/*.hip -o build_variants/libqpnet_$name.so
echo build_variants/libqpnet_$name.so
