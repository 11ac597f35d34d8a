#!/bin/sh
# Builds oracle/_ref/libcuhdenc.so: the reference's CUHD encoder + decoder-table builder
# (cuhd-icpp/encoder/src/llhuffman_encoder.cc, cuhd-icpp/src/cuhd_codetable.cc), compiled UNMODIFIED from where
# they lie, plus oracle/ref_cuhd_shim.cpp (this repo's extern "C" shim over their public interface).
# `-include cstdint`: the reference relies on <cstdint> arriving through other headers, which GCC 11 no longer
# guarantees; it adds a standard header, nothing else.
set -e
REF=${REF:-/root/reference}
C=$REF/cuhd-icpp
HERE=$(cd "$(dirname "$0")" && pwd)
[ -f "$C/encoder/src/llhuffman_encoder.cc" ] || { echo "reference not present: keeping prebuilt _ref (if any)"; exit 0; }
mkdir -p "$HERE/_ref"
${CXX:-g++} -O2 -w -std=c++17 -include cstdint -fPIC -shared -I"$C/include" -I"$C/encoder/include" \
    -o "$HERE/_ref/libcuhdenc.so" "$C/encoder/src/llhuffman_encoder.cc" "$C/src/cuhd_codetable.cc" "$HERE/ref_cuhd_shim.cpp"
echo "built _ref/libcuhdenc.so from $C/encoder/src/llhuffman_encoder.cc + $C/src/cuhd_codetable.cc"
