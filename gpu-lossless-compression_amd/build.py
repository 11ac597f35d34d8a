"""Builds libglc_amd.so (HIP kernels + the C ABI of include/cudpp.h and
include/culzss.h) in-tree for gfx950 with hipcc.  No torch dependency: the
library's boundary is plain pointers and sizes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libglc_amd.so")
SOURCES = ["cudpp_api.cpp", "bwt_sa.hip", "bwt_bucket.hip", "mtf.hip", "huffman.hip", "decode.hip", "culzss.hip",
           "culzss_api.cpp", "hd_decode.hip", "probe.hip"]


def _newest_source_mtime():
    m = 0.0
    for f in os.listdir(CSRC):
        m = max(m, os.path.getmtime(os.path.join(CSRC, f)))
    inc = os.path.join(os.path.dirname(HERE), "include")
    for f in os.listdir(inc):
        m = max(m, os.path.getmtime(os.path.join(inc, f)))
    return m


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_source_mtime():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
           "-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libglc_amd.so")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
