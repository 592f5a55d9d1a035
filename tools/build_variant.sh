#!/bin/bash
# builds myria3d_amd/variants/libm3d_NAME.so with extra -D flags for ONE source file (tuning sweeps)
# usage: tools/build_variant.sh NAME file.hip -DFOO=1 ...
set -e
NAME=$1; SRC=$2; shift; shift
cd $(dirname $0)/../myria3d_amd/csrc
mkdir -p ../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-variable "$@" -c $SRC -o /tmp/var_$NAME.o 2>&1 | grep -E "error" || true
BASE=${VARIANT_REPLACES:-${SRC%.hip}}   # object of the stock build that this variant source replaces
OBJS=$(ls *.o | grep -v "^${BASE}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/var_$NAME.o -o ../variants/libm3d_$NAME.so
echo built $NAME
