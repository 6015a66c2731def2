#!/bin/bash
# Developer tool: build t-route_amd/libtrmc<suffix>.so with LLVM backend options applied to the DEVICE code of trmc.hip only
# (hipcc hands -mllvm to the host compilation too, which does not know AMDGPU options):
#     tools/build_variant.sh _ifcvt -amdgpu-early-ifcvt [-early-ifcvt-limit=60 ...]
# then time it with tools/ab.sh "" _ifcvt on the GPU box.  The steps are the ones `hipcc -###` prints for the normal build.
set -e
cd "$(dirname "$0")/../t-route_amd/csrc"
sfx=$1; shift
F="-O3 -std=c++17 -fPIC -ffp-contract=off $TRMC_VARIANT_DEFS"
T=$(mktemp -d)
ml=""; for o in "$@"; do ml="$ml -mllvm $o"; done
# (the device-only output is already the offload bundle the host compilation embeds)
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F --offload-device-only $ml trmc.hip -o $T/trmc.hipfb
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F --offload-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/trmc.hipfb -c trmc.hip -o $T/trmc.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c diffusive.hip -o $T/diffusive.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c topology.cpp -o $T/topology.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c comm.hip -o $T/comm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libtrmc$sfx.so $T/trmc.o $T/diffusive.o $T/comm.o $T/topology.o -ldl -lrt
rm -rf $T
ls -la ../libtrmc$sfx.so
