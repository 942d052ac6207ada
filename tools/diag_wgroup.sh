#!/bin/bash
# Diagnostic builds of the grouped weight-gradient kernel with one part of its loop compiled out (results are WRONG; timing only):
#   tools/diag_wgroup.sh build    (here, on the CPU)   ->  star_amd/diag/libstar_hip_d<mask>.so
#   tools/diag_wgroup.sh run      (on the GPU box)     ->  bench_ops wgroup for each
set -o pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
masks="${MASKS:-1 2 4 3 6 7 8}"
if [ "$1" = build ]; then
  mkdir -p $root/star_amd/diag
  for m in $masks; do
    ( cd $root/star_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -DTN_DIAG=$m -c gemm.hip -o build/gemm_d$m.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../diag/libstar_hip_d$m.so $(ls build/*.o | grep -v "gemm\.o\|gemm_[dn]\|gemm_old\|gemm_ln_g") build/gemm_d$m.o ) &
  done
  wait
else
  for m in 0 $masks; do
    if [ $m = 0 ]; then unset STAR_HIP_LIB; else export STAR_HIP_LIB=$root/star_amd/diag/libstar_hip_d$m.so; fi
    echo "== diag mask $m"
    timeout -k 10 120 python3 $root/tools/bench_ops.py wgroup 2>&1 | grep grouped || exit 1
  done
fi
