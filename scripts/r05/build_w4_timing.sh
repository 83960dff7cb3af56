#!/bin/bash
# product library + the timing variant of drunet_wino4.hip (-DDINV_W4_TIMING: phase stamps of wave 0) + the two harness binaries
set -e
cd "$(dirname "$0")/../.."
make -C deepinv_amd/csrc -j8 2>&1 | grep -E "error|Error" || true
V=scripts/r05/variants
mkdir -p $V/obj_w4t $V/w4t
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -DDINV_W4_TIMING $W4FLAGS -c deepinv_amd/csrc/drunet_wino4.hip -o $V/obj_w4t/drunet_wino4.o 2>&1 | grep -E " error" || true
OTHERS=$(ls deepinv_amd/csrc/build/*.o | grep -v -E "/drunet_wino4\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/w4t/libdeepinv_amd.so $V/obj_w4t/drunet_wino4.o $OTHERS
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -DTIMING scripts/r05/wino4_bench.cpp -Iinclude -L$V/w4t -ldeepinv_amd -o scripts/r05/wino4_time
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 scripts/r05/wino4_bench.cpp -Iinclude -Ldeepinv_amd -ldeepinv_amd -o scripts/r05/wino4_bench
echo built
