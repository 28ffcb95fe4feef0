#!/bin/bash
# CPU-side AddressSanitizer + UBSan run of the HOST code (Lair parser / compiler / interpreter, bytecode import, wire format,
# ZStore host side): the .cpp files are recompiled with -fsanitize=address,undefined for the host pass only
# (-Xarch_host), linked with the unchanged device objects into lurk_amd/liblurkhip_asan.so, and the host tests run under it.
# GPU AddressSanitizer is not available on this pool; this covers the code that parses untrusted input.
set -e
cd /root/repo/lurk_amd/csrc
make >/dev/null
mkdir -p obj_asan/lair
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
for f in *.cpp lair/*.cpp; do
  o=obj_asan/${f%.cpp}.o
  if [ ! -f $o ] || [ $f -nt $o ] || [ lair/lair.h -nt $o ]; then
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off \
      -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -x hip -c $f -o $o
  fi
done
HIP_OBJS=$(for f in *.hip; do echo ${f%.hip}.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -shared-libsan -fsanitize=address,undefined -o ../liblurkhip_asan.so $HIP_OBJS obj_asan/*.o obj_asan/lair/*.o -L/opt/rocm/lib -lhiprtc
cd /root/repo
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0:log_path=/tmp/asan.log UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=/tmp/ubsan.log LD_PRELOAD=$RT LURKHIP_LIB_PATH=lurk_amd/liblurkhip_asan.so \
  python -m pytest tests/test_lair_host.py tests/test_bytecode.py tests/test_mix_programs.py tests/test_abi.py tests/test_lair_random.py tests/test_verify.py -x -q "$@"
