#!/bin/bash
# Variant builds of libdeepinv_amd.so for the A/B timing of the MRI passes (scripts/r05/mri_bench.cpp): mri.hip and fft.hip are
# recompiled with other -D switches, every other object is the product's.  Output: scripts/r05/variants/lib<name>.so (git-ignored).
#   scripts/r05/build_mri_variants.sh name "flags" [name "flags" ...]
set -e
cd "$(dirname "$0")/../.."
make -C deepinv_amd/csrc -j8 > /dev/null
V=scripts/r05/variants
mkdir -p $V
OTHERS=$(ls deepinv_amd/csrc/build/*.o | grep -v -E "/(mri|fft)\.o$")
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  mkdir -p $V/obj_$name
  for f in mri fft; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize $flags -c deepinv_amd/csrc/$f.hip -o $V/obj_$name/$f.o 2>&1 | grep -E "error" || true &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/lib$name.so $V/obj_$name/mri.o $V/obj_$name/fft.o $OTHERS
  echo "built $V/lib$name.so  [$flags]"
done
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 scripts/r05/mri_bench.cpp -Iinclude -ldl -o scripts/r05/mri_bench
echo "built scripts/r05/mri_bench"
