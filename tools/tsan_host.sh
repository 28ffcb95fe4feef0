#!/bin/bash
# ThreadSanitizer run of the HOST code that is multi-threaded without a device: the verifier's shard threads (csrc/verify.cpp),
# the Lair interpreter / flattening called from several Python threads, the decoder.  Same recipe as tools/asan_host.sh: the .cpp
# files recompiled with -fsanitize=thread for the host pass only, linked with the unchanged device objects into
# lurk_amd/liblurkhip_tsan.so, the CPU tests that exercise threads run under it.  (GPU sanitizers are not available on this pool;
# the GPU half of the concurrency contract is tests/test_concurrency_gpu.py.)
set -e
cd /root/repo/lurk_amd/csrc
make >/dev/null
mkdir -p obj_tsan/lair
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.tsan-x86_64.so
for f in *.cpp lair/*.cpp; do
  o=obj_tsan/${f%.cpp}.o
  if [ ! -f $o ] || [ $f -nt $o ] || [ lair/lair.h -nt $o ] || [ ctx.h -nt $o ]; then
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off \
      -Xarch_host -fsanitize=thread -Xarch_host -fno-omit-frame-pointer -x hip -c $f -o $o
  fi
done
HIP_OBJS=$(for f in *.hip; do echo ${f%.hip}.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -shared-libsan -fsanitize=thread -o ../liblurkhip_tsan.so $HIP_OBJS obj_tsan/*.o obj_tsan/lair/*.o -L/opt/rocm/lib -lhiprtc
cd /root/repo
rm -f /tmp/tsan.log*
TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:log_path=/tmp/tsan.log LD_PRELOAD=$RT LURKHIP_LIB_PATH=lurk_amd/liblurkhip_tsan.so LURKHIP_VERIFY_THREADS=4 \
  python -m pytest tests/test_verify.py tests/test_lair_host.py -x -q "$@"
# the oracle's C side runs OpenMP regions (libgomp is not instrumented: its fork / join is invisible to the tool, every access of a
# worker reads as a race with the main thread); what counts are reports with a frame inside the product library
n_all=$(cat /tmp/tsan.log* 2>/dev/null | grep -c "WARNING: ThreadSanitizer" || true)
n_product=$(cat /tmp/tsan.log* 2>/dev/null | awk '/WARNING: ThreadSanitizer/{blk=""} {blk=blk $0 "\n"} /^$/{if (blk ~ /liblurkhip/) n++; blk=""} END{print n+0}')
echo "ThreadSanitizer: $n_all report(s), $n_product with a frame in liblurkhip"
[ "$n_product" = "0" ]
