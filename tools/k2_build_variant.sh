#!/bin/bash
# builds rs-sync_amd/_variants/lib_NAME.so with extra hipcc flags:  bash tools/k2_build_variant.sh NAME [-DFOO=1 ...]
# (the Makefile's `variant` target: the product's compiler, architecture and flags -- HIPCC / ARCH / ROCM can be overridden
# in the environment exactly as for the product build)
set -e
NAME=$1; shift
make -C "$(dirname "$0")/../rs-sync_amd/csrc" variant NAME="$NAME" EXTRA="$*"
echo built $NAME
