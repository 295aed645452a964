#!/bin/bash
# Variant build of the library for A/B runs:  TUNE_H='#define ...' tools/tune_build.sh NAME [extra hipcc flags]
#   -> levelsetpy_amd/csrc/libhj_vNAME.so   (select it with HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_vNAME.so)
# $TUNE_H is written to hj_tune.h and force-included in every translation unit (macro definitions with
# parentheses do not survive make's shell quoting).  -DHJ_TUNE_BUILD compiles the tiled kernels only for
# fp64 Dubins with the two WENO5 arithmetics (fast; TUNE_MODE=2: only the fp32 pendulum, C5); FULL=1 builds every instantiation.  Sources are copied
# to a scratch directory: the product build (libhj_mi355x.so and its objects) is not touched.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/hjbuild_$name
rm -rf $tmp; mkdir -p $tmp/levelsetpy_amd/csrc $tmp/include
cp $root/levelsetpy_amd/csrc/*.h $root/levelsetpy_amd/csrc/*.hip $root/levelsetpy_amd/csrc/Makefile $tmp/levelsetpy_amd/csrc/
cp $root/include/*.h $tmp/include/
printf '%s\n' "$TUNE_H" > $tmp/levelsetpy_amd/csrc/hj_tune.h
tune="-DHJ_TUNE_BUILD=${TUNE_MODE:-1}"; [ -n "$FULL" ] && tune=""
make -s -C $tmp/levelsetpy_amd/csrc -j8 EXTRA="$tune -include hj_tune.h $*" > $tmp/build.log 2>&1 || { tail -30 $tmp/build.log; exit 1; }
cp $tmp/levelsetpy_amd/csrc/libhj_mi355x.so $root/levelsetpy_amd/csrc/libhj_v$name.so
echo "built levelsetpy_amd/csrc/libhj_v$name.so (log: $tmp/build.log)"
