#!/bin/bash
# A/B build of the kernel library: tools/build_variant.sh <name> "<extra hipcc flags>" ["<panel list>"]
# compiles a copy of ps-vae_amd/csrc in /tmp and leaves exp_libs/libpsvae_<name>.so (run it with
# PSVAE_HIP_LIB=exp_libs/libpsvae_<name>.so).
set -e
NAME=$1; EXTRA=$2; PANELS=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/psvae_build_$NAME
rm -rf $W && mkdir -p $W/ps-vae_amd && cp -r $ROOT/include $W/ && mkdir -p $W/ps-vae_amd/csrc
cp $ROOT/ps-vae_amd/csrc/*.hip $ROOT/ps-vae_amd/csrc/*.h $ROOT/ps-vae_amd/csrc/*.cpp $ROOT/ps-vae_amd/csrc/Makefile $W/ps-vae_amd/csrc/
if [ -n "$PANELS" ]; then
  make -C $W/ps-vae_amd/csrc -j8 libpsvae_hip.so EXTRA="$EXTRA" PANELS="$PANELS" > $W/build.log 2>&1
else
  make -C $W/ps-vae_amd/csrc -j8 libpsvae_hip.so EXTRA="$EXTRA" > $W/build.log 2>&1
fi
mkdir -p $ROOT/exp_libs && cp $W/ps-vae_amd/csrc/libpsvae_hip.so $ROOT/exp_libs/libpsvae_$NAME.so
echo built exp_libs/libpsvae_$NAME.so
