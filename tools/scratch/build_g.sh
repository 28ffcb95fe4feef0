#!/bin/bash
# builds tools/scratch/liblurkhip_g.so = the current objects with lair/execute.cpp recompiled with -g (for tools/pcsample)
cd /root/repo/lurk_amd/csrc && make >/dev/null 2>&1
/opt/rocm/bin/hipcc -O3 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -x hip -c lair/execute.cpp -o /tmp/execute_g.o || exit 1
OBJS=$(ls *.o lair/*.o | grep -v "lair/execute.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /root/repo/tools/scratch/liblurkhip_g.so $OBJS /tmp/execute_g.o -L/opt/rocm/lib -lhiprtc
