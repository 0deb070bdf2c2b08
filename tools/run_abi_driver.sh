#!/bin/bash
# The native stress driver (tests/sanitize/abi_driver.cpp) against the PRODUCT build, every mode, a few runs each:
# lookups served, rows checked against the recipe, and — the point of round 5's use — whether a run comes back at all.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
D=$R/hugectr_backend_amd/lib
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -D__HIP_PLATFORM_AMD__ -Iinclude -Ihugectr_backend_amd/csrc -I/opt/rocm/include tests/sanitize/abi_driver.cpp -L$D -lhps_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$D -Wl,-rpath,/opt/rocm/lib -pthread -o /tmp/abi_driver || exit 1
for mode in cpu gpu gpu_direct gpu_sharded; do
  for i in $(seq 1 ${1:-4}); do
    timeout 60 /tmp/abi_driver $mode 4 > /tmp/abi_out.txt 2>&1; rc=$?
    echo "$mode run $i: rc=$rc $(grep '^abi_driver' /tmp/abi_out.txt | tail -1)"
  done
done
