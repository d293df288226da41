#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..."  ->  morphsym_hgnn_amd/csrc/variants/NAME.so (a kernel-experiment build of the library; run it with MSHGNN_LIB=<path>)
set -e
cd "$(dirname "$0")/../morphsym_hgnn_amd/csrc"
mkdir -p variants
make OUT=variants/$1.so BUILD=build_$1 EXTRA="$2" -j8 2>&1 | grep -E "error|warning: v|Error" || true
ls -la variants/$1.so
