#!/bin/bash
# Lab build of the library: one source recompiled with extra flags, linked against the product objects.
#   scripts/build_lablib.sh <tag> <source.hip> [extra hipcc flags...]   ->  scripts/lablib/libdss_hip_<tag>.so
# (DSS_HIP_LIBRARY=<that file> selects it; the product objects must be current: python deep-spectral-segmentation_amd/build.py)
set -e
cd "$(dirname "$0")/.."
TAG=$1; SRC=$2; shift 2
PKG=deep-spectral-segmentation_amd
mkdir -p scripts/lablib
EXTRA=""; [ "$SRC" = attention.hip ] && EXTRA="-fno-honor-nans -mno-amdgpu-ieee"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EXTRA "$@" -c $PKG/csrc/$SRC -o scripts/lablib/${SRC%.hip}_$TAG.o
OBJS=$(ls $PKG/lib/obj/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/lablib/libdss_hip_$TAG.so $OBJS scripts/lablib/${SRC%.hip}_$TAG.o
rm -f scripts/lablib/${SRC%.hip}_$TAG.o
echo "scripts/lablib/libdss_hip_$TAG.so"
