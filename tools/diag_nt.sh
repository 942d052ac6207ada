#!/bin/bash
# Diagnostic builds of the tiled NT GEMM with one part compiled out (results are WRONG; timing only):
#   tools/diag_nt.sh build   (here)  ->  star_amd/diag/libstar_hip_n<mask>.so;   tools/diag_nt.sh run   (GPU box): bench_ops gemm with each
set -o pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
masks="${MASKS:-1 2 4 3 5 6 7}"
if [ "$1" = build ]; then
  mkdir -p $root/star_amd/diag
  for m in $masks; do
    ( cd $root/star_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -DNT_DIAG=$m -c gemm.hip -o build/gemm_n$m.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../diag/libstar_hip_n$m.so $(ls build/*.o | grep -v "gemm\.o\|gemm_[dn]\|gemm_old\|gemm_ln_g") build/gemm_n$m.o ) &
  done
  wait
else
  for m in 0 $masks; do
    if [ $m = 0 ]; then unset STAR_HIP_LIB; else export STAR_HIP_LIB=$root/star_amd/diag/libstar_hip_n$m.so; fi
    echo "== nt diag mask $m"
    STAR_AMD_GEMM_WREG=0 timeout -k 10 120 python3 $root/tools/bench_ops.py gemm 2>&1 | grep -E "qkv|fc1|fc2|dfc" || exit 1
  done
fi
