#!/bin/bash
# developer tool: build_exp/libps_<tag>.so = the library with orb_kernels.hip compiled with extra flags ("-DPS_EXP=1" ...)
# usage: tools/build_variant.sh <tag> <file.hip> <extra flags...>;  run with PS_LIB_PATH=build_exp/libps_<tag>.so
set -e
TAG=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../pointslot_amd/csrc"
mkdir -p ../../build_exp
OBJ=../../build_exp/${SRC%.hip}_$TAG.o
# same floating-point contraction as the Makefile: off everywhere except the FP64 optimiser kernels
CONTRACT=off
case $SRC in ba_kernels.hip|opt_kernels.hip) CONTRACT=fast;; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=$CONTRACT "$@" -c $SRC -o $OBJ
OTHERS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../build_exp/libps_$TAG.so $OBJ $OTHERS
echo build_exp/libps_$TAG.so
