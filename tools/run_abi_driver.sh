set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
D=$R/hugectr_backend_amd/lib
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -D__HIP_PLATFORM_AMD__ -Iinclude -Ihugectr_backend_amd/csrc -I/opt/rocm/include tests/sanitize/abi_driver.cpp -L$D -lhps_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$D -Wl,-rpath,/opt/rocm/lib -pthread -o /tmp/abi_driver || exit 1
for mode in gpu gpu_direct gpu_sharded; do HPS_TRACE_TAIL=200 timeout 100 /tmp/abi_driver $mode 5 2>&1 | tail -4; done
