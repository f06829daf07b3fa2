#!/bin/bash
# developer tool: build_exp/libps_<tag>.so = the WHOLE library compiled with extra flags (a copy of csrc is built beside the tree)
# usage: tools/build_all_variant.sh <tag> <extra flags...>;  run with PS_LIB_PATH=build_exp/libps_<tag>.so
set -e
TAG=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
D=$ROOT/build_exp/csrc_$TAG
rm -rf $D; mkdir -p $D
cp $ROOT/pointslot_amd/csrc/*.hip $ROOT/pointslot_amd/csrc/*.h $ROOT/pointslot_amd/csrc/*.inc $ROOT/pointslot_amd/csrc/Makefile $D/
make -C $D -j8 OUT=../libps_$TAG.so EXTRA="$*" > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
echo build_exp/libps_$TAG.so
