#!/bin/bash
# A/B build of the library with extra -D flags (every object is rebuilt, in a scratch copy of the
# sources):  tools/build_variant.sh NAME "-DPSVAE_PANEL_OCC=2 ..."  ->  exp_libs/libpsvae_NAME.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; DEFS=$2
TMP=/tmp/psvae_variant_$NAME
rm -rf $TMP && mkdir -p $TMP/ps-vae_amd $TMP/include
cp $ROOT/include/*.h $TMP/include/
mkdir -p $TMP/ps-vae_amd/csrc
cp $ROOT/ps-vae_amd/csrc/*.hip $ROOT/ps-vae_amd/csrc/*.h $ROOT/ps-vae_amd/csrc/Makefile $TMP/ps-vae_amd/csrc/
make -C $TMP/ps-vae_amd/csrc -j8 EXTRA="$DEFS" libpsvae_hip.so >/dev/null
mkdir -p $ROOT/exp_libs
cp $TMP/ps-vae_amd/csrc/libpsvae_hip.so $ROOT/exp_libs/libpsvae_$NAME.so
echo built $ROOT/exp_libs/libpsvae_$NAME.so
