#!/bin/bash
# A/B build of the library with extra -D flags on gemm.hip / gine.hip (the other objects are
# reused):  tools/build_variant.sh NAME "-DPSVAE_PANEL_BK=16 ..."  ->  exp_libs/libpsvae_NAME.so
set -e
cd "$(dirname "$0")/../ps-vae_amd/csrc"
make -j8 >/dev/null
NAME=$1; DEFS=$2
OUT=../../exp_libs; mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS $DEFS -I. -c ${GEMM_SRC:-gemm.hip} -o $OUT/gemm_$NAME.o &
/opt/rocm/bin/hipcc $FLAGS $DEFS -c gine.hip -o $OUT/gine_$NAME.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/gemm_$NAME.o $OUT/gine_$NAME.o collate.o misc.o assemble.o node_embed.o optim.o gru.o -o $OUT/libpsvae_$NAME.so
rm -f $OUT/gemm_$NAME.o $OUT/gine_$NAME.o
echo built $OUT/libpsvae_$NAME.so
