#!/bin/bash
# Host-side sanitizer run (GPU AddressSanitizer is not available on the pool; this instruments the HOST code of the library):
# builds libfhe_ring.so with -fsanitize=address,undefined for the host pass and drives the entry points that work without a
# device (set-up, tables, status codes) from tools/asan/host_driver.c.  Run in the build container: bash tools/scripts/asan_host.sh
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=${TMPDIR:-/tmp}/fhe_asan
mkdir -p $O
for f in ring_api rns_api fhew_api torus_api torusk_api keygen_api; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Xarch_host -fsanitize=address,undefined \
      -Xarch_host -fno-omit-frame-pointer -c -o $O/$f.o $R/learn-fhe_amd/csrc/$f.hip
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -o $O/libfhe_ring.so $O/*.o
gcc -std=c99 -g -I $R/include $R/tools/asan/host_driver.c -L $O -lfhe_ring -Wl,--allow-shlib-undefined -Wl,-rpath,$O \
    -fsanitize=address,undefined -o $O/host_driver
ASAN_OPTIONS=detect_leaks=1:protect_shadow_gap=0 $O/host_driver
echo "asan/ubsan host run: clean"
