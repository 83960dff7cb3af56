#!/bin/bash
# throw-away: diagnostic builds of csrc/drunet_wsplit.hip (switches WS_DIAG_*), each linked into its own copy of the library
set -e
cd "$(dirname "$0")/../../deepinv_amd/csrc"
make -j8 >/dev/null
OUT=../../scripts/r03/variants
mkdir -p $OUT
OTHERS=$(ls build/*.o | grep -v drunet_wsplit.o)
for v in "base:" "notransform:-DWS_DIAG_NO_TRANSFORM" "noexchange:-DWS_DIAG_NO_EXCHANGE" "nofence:-DWS_DIAG_NO_FENCE" "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize $flags -c drunet_wsplit.hip -o $OUT/ws_$name.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libdeepinv_amd_$name.so $OTHERS $OUT/ws_$name.o
  rm $OUT/ws_$name.o
  echo built $name
done
