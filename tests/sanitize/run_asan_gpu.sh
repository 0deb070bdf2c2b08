set -u
D=/tmp/asan_libs; mkdir -p $D
export HPS_AMD_LIB_DIR=$D HPS_AMD_EXTRA_FLAGS="-fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -g -shared-libsan" HPS_AMD_EXTRA_LDFLAGS="-fsanitize=address -shared-libsan"
python -m hugectr_backend_amd.build > $D/build.log 2>&1 || { tail -5 $D/build.log; exit 1; }
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
# The SDMA engine wake-up (csrc/cache/copy_engines.cpp) is switched off for this job: with all 16 engine queues created, the
# HSA runtime's shutdown at process exit frees their buffers after ASan's device allocator has declared the device runtime
# unloaded ("CHECK failed: sanitizer_allocator_device.h:125" at exit, after a clean run) — a teardown-order limit of the
# sanitizer runtime, not a finding in this repository.
export HPS_WAKE_COPY_ENGINES=0
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:verify_asan_link_order=0 timeout 400 python tests/tools/asan_gpu_run.py > $D/run.log 2>&1
echo "rc=$?"; grep -c "ERROR: AddressSanitizer" $D/run.log; tail -4 $D/run.log
