#!/bin/bash
# builds experimental variants of libdeepinv_amd.so (A/B of occupancy choices; the winners are baked into the sources)
set -e
cd /root/repo/deepinv_amd/csrc
build() {  # name, flags
  name=$1; shift
  mkdir -p /tmp/exp_$name
  for f in *.hip; do
    o=/tmp/exp_$name/${f%.hip}.o
    case $f in
      radon_tiled.hip) /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast "$@" -c $f -o $o ;;
      *) cp build/${f%.hip}.o $o ;;
    esac
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scripts/r03/exp/libdeepinv_amd_$name.so /tmp/exp_$name/*.o
}
rm -f /root/repo/scripts/r03/exp/*.so
build v4 -DDINV_EXP_RADON_WPE=2
ls -la /root/repo/scripts/r03/exp/
