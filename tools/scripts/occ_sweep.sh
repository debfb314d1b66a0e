#!/bin/bash
# experiment: the fused kernels compiled for 2 / 3 / 4 waves per SIMD (FHE_TEAM_OCC), secondary bench figures of each.
# Build the variants first (in the build container):
#   for occ in 2 3 4; do touch learn-fhe_amd/csrc/fhew_kernels.hpp; \
#     make -C learn-fhe_amd/csrc -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DFHE_TEAM_OCC=$occ"; \
#     cp learn-fhe_amd/lib/libfhe_ring.so tools/libfhe_occ$occ.so; done   (then rebuild the default library)
# The variant is selected through FHE_RING_LIB (learn-fhe_amd/_lib.py): the product library in learn-fhe_amd/lib/ is never touched.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for occ in 2 3 4; do
  echo "== occ $occ"
  FHE_RING_LIB=$R/tools/libfhe_occ$occ.so python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['fhew']); print(d['tfhe'])"
done
