#!/bin/bash
# Regenerates profiles/ntt14w_isa.json (what bench.py prices roofline.issue with) and profiles/<tag>_ntt14w_isa_hist.txt from the
# SHIPPED sources: hipcc -S of learn-fhe_amd/csrc/ring_api.hip, tools/isa_hist.py on the headline kernels.  No GPU needed.
# usage: tools/scripts/isa_report.sh <tag>
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
tag=${1:-r03}
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only -o $tmp/ring_api.s $R/learn-fhe_amd/csrc/ring_api.hip
{
  echo "# Instruction-class histogram of the shipped wave-local kernels (hipcc -O3 --offload-arch=gfx950 -S of learn-fhe_amd/csrc/ring_api.hip,"
  echo "# tools/isa_hist.py; static counts of the straight-line code = per-thread dynamic counts, the kernels have no loops)."
  echo "# 224 butterflies per thread at R0 = 3 (N = 2^14).  Regenerate: tools/scripts/isa_report.sh <tag>"
  python3 $R/tools/isa_hist.py $tmp/ring_api.s --json $R/profiles/ntt14w_isa.json 'ntt14w_fwd_kernelINS_7ArithDSILi60EEELb0ELi3' 'ntt14w_inv_kernelINS_7ArithDSILi60EEELb0ELb0ELi3'
  python3 $R/tools/isa_hist.py $tmp/ring_api.s 'ntt14w_mul_kernelINS_7ArithDSILi60EEELi3' 'ntt14w_inv_kernelINS_7ArithDSILi60EEELb1ELb1ELi3' 'ntt14w_fwd_kernelINS_7ArithDSILi60EEELb1ELi3'
} > $R/profiles/${tag}_ntt14w_isa_hist.txt
rm -rf $tmp
tail -n +4 $R/profiles/${tag}_ntt14w_isa_hist.txt | grep -E "^==|priced|VALU"
