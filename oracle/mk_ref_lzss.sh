#!/bin/sh
# Builds oracle/_ref/lzss_serial: the conventional serial LZSS the reference ships (Dipperstein lzss-0.6.2,
# cuda-lzss-unknown/lzss-0.6.2, plain ANSI C), compiled UNMODIFIED from where it lies with the match finder its own
# Makefile selects (brute.c).  Used by bench.py as the CPU LZSS baseline of config 3 ("kind": "reference").  It is
# a different format from CULZSS (12-bit offset / 4-bit length, 4 KiB window): a baseline, not a parity oracle.
set -e
REF=${REF:-/root/reference}
D=$REF/cuda-lzss-unknown/lzss-0.6.2
HERE=$(cd "$(dirname "$0")" && pwd)
[ -f "$D/lzencode.c" ] || { echo "reference not present: keeping prebuilt _ref (if any)"; exit 0; }
mkdir -p "$HERE/_ref"
${CC:-gcc} -O3 -w -I"$D" -o "$HERE/_ref/lzss_serial" "$D/sample.c" "$D/lzencode.c" "$D/lzdecode.c" "$D/lzvars.c" \
    "$D/brute.c" "$D/bitfile.c" "$D/optlist.c"
echo "built _ref/lzss_serial from $D (sample.c lzencode.c lzdecode.c lzvars.c brute.c bitfile.c optlist.c)"
