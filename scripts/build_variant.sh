#!/bin/bash
# scripts/build_variant.sh <name> [make EXTRA flags...]: an A/B build of the library next to the product one
# (ascii-chat_amd/lib_<name>.so, objects in ascii-chat_amd/build_<name>/; both git-ignored, the .so travels with gpurun)
set -e
name=$1; shift
cd "$(dirname "$0")/../ascii-chat_amd"
make -j16 BUILD=build_$name OUT=lib_$name.so EXTRA="$*" > /tmp/build_$name.log 2>&1 && echo "built lib_$name.so" || { tail -20 /tmp/build_$name.log; exit 1; }
