#!/bin/bash
# Lab build of the library: one source recompiled with extra flags, linked against the product objects.
#   scripts/build_lablib.sh <tag> <source.hip> [extra hipcc flags...]   ->  scripts/lablib/libdss_hip_<tag>.so
# <source.hip> is a file of deep-spectral-segmentation_amd/csrc/, or `linear384_r4_lab.hip` / `attention_r4_lab.hip` / `attention_r6_lab.hip` = the round-4 / round-6 lab snapshots of
# the Linear / attention kernels in scripts/probes/ (it replaces linear384.o: -DDSS_LIN_LAB_STAGGER, -DDSS_LIN_PLAIN_PREFETCH, -DDSS_LIN_ABL=n, -DDSS_GELU_SCALAR ...).
# (DSS_HIP_LIBRARY=<that file> selects it; the product objects must be current: python deep-spectral-segmentation_amd/build.py)
set -e
cd "$(dirname "$0")/.."
TAG=$1; SRC=$2; shift 2
PKG=deep-spectral-segmentation_amd
mkdir -p scripts/lablib
EXTRA=""; [ "$SRC" = attention.hip ] && EXTRA="-fno-honor-nans -mno-amdgpu-ieee"
PATHSRC=$PKG/csrc/$SRC; REPL=${SRC%.hip}
if [ "$SRC" = linear384_r4_lab.hip ]; then PATHSRC=scripts/probes/$SRC; REPL=linear384; fi
if [ "$SRC" = attention_r4_lab.hip ]; then PATHSRC=scripts/probes/$SRC; REPL=attention; EXTRA="-fno-honor-nans -mno-amdgpu-ieee"; fi
# attention_r6_lab.hip = the round-6 lab snapshot: -DDSS_ATTN_F16SUM=1 (row sums on packed f16), -DDSS_ATTN_PRIO=1|2|3 (s_setprio over the
# score / P.V MFMAs), -DDSS_ATTN_PIPE=4|8 (attn_fwd_pipe_kernel: next block's score tile in front of this block's softmax)
if [ "$SRC" = attention_r6_lab.hip ]; then PATHSRC=scripts/probes/$SRC; REPL=attention; EXTRA="-fno-honor-nans -mno-amdgpu-ieee -I $PKG/csrc"; fi
# LAB_SED='s/ATTN_NW = 8/ATTN_NW = 4/': the source is compiled from a sed-edited temporary copy (a constant of the product file changed
# for ONE A/B build, without a lab macro in the product file)
if [ -n "${LAB_SED:-}" ]; then sed -e "$LAB_SED" $PATHSRC > scripts/lablib/src_$TAG.hip; PATHSRC=scripts/lablib/src_$TAG.hip; EXTRA="$EXTRA -I $PKG/csrc"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EXTRA "$@" -c $PATHSRC -o scripts/lablib/${REPL}_$TAG.o
rm -f scripts/lablib/src_$TAG.hip
OBJS=$(ls $PKG/lib/obj/*.o | grep -v "/${REPL}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/lablib/libdss_hip_$TAG.so $OBJS scripts/lablib/${REPL}_$TAG.o -L/opt/rocm/lib -lhipblaslt -Wl,-rpath,/opt/rocm/lib
rm -f scripts/lablib/${REPL}_$TAG.o
echo "scripts/lablib/libdss_hip_$TAG.so"
