#!/bin/bash
# builds rs-sync_amd/_variants/lib_NAME.so with extra hipcc flags:  bash tools/k2_build_variant.sh NAME [-DFOO=1 ...]
set -e
NAME=$1; shift
cd "$(dirname "$0")/../rs-sync_amd/csrc"
mkdir -p ../_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast-honor-pragmas -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-unused-function "$@" -c rssync_kernels.hip -o /tmp/var_$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC -o ../_variants/lib_$NAME.so /tmp/var_$NAME.o _build/sync_problem.o -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined
echo built $NAME
