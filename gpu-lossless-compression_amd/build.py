"""Builds libglc_amd.so (HIP kernels + the C ABI of include/cudpp.h, include/culzss.h and include/glc_hd.h) in-tree for
gfx950 with hipcc.  No torch dependency: the library's boundary is plain pointers and sizes.
Sources are compiled to objects in parallel (build/ is git-ignored), an object is reused while it is newer than its
source and every header; the link step pulls in librccl for the multi-GPU exchange (csrc/exchange.cpp)."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
# A/B builds: GLC_CXXFLAGS adds compiler flags (e.g. -DGLC_IMD_POS=512), GLC_LIB_OUT names the library; such a build
# keeps its objects apart, in a directory named after a hash of ITS flags: two variants never share (and so never
# silently reuse) each other's objects.  tests / bench pick a library with GLC_LIB (glc_binding.py).
_VARIANT = bool(os.environ.get("GLC_CXXFLAGS") or os.environ.get("GLC_LIB_OUT"))
_TAG = hashlib.sha1(" ".join(os.environ.get("GLC_CXXFLAGS", "").split()).encode()).hexdigest()[:10]
OBJ = os.path.join(HERE, "build_variant", _TAG) if _VARIANT else os.path.join(HERE, "build")
LIB = os.environ.get("GLC_LIB_OUT") or os.path.join(HERE, "libglc_amd.so")
SOURCES = ["cudpp_api.cpp", "bwt_sa.hip", "bwt_bucket.hip", "bwt_periodic.hip", "mtf.hip", "huffman.hip", "decode.hip", "culzss.hip",
           "culzss_api.cpp", "hd_decode.hip", "probe.hip", "exchange.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread"] + os.environ.get("GLC_CXXFLAGS", "").split()


LAST = {}                                                      # what the last build() did: {sources, compiled, linked, forced}


def _header_mtime():
    m = 0.0
    for d in (CSRC, INC):
        for f in os.listdir(d):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return max(m, os.path.getmtime(os.path.abspath(__file__)))


def have_rccl():
    """the multi-GPU exchange (csrc/exchange.cpp) needs RCCL: header + library.  A single-GPU install without it still gets
    the whole compression path; the exchange entry points are then simply not exported (GLC_NO_RCCL=1 forces that)."""
    if os.environ.get("GLC_NO_RCCL") == "1":
        return False
    root = rocm_root()
    return os.path.exists(os.path.join(root, "include", "rccl", "rccl.h")) and any(
        os.path.exists(os.path.join(root, d, "librccl.so")) for d in ("lib", "lib64"))


def rocm_root():
    """ROCM_PATH, else the prefix of $HIPCC (<root>/bin/hipcc), else /opt/rocm"""
    if os.environ.get("ROCM_PATH"):
        return os.environ["ROCM_PATH"]
    h = os.environ.get("HIPCC")
    if h and os.path.basename(os.path.dirname(os.path.abspath(h))) == "bin":
        return os.path.dirname(os.path.dirname(os.path.abspath(h)))
    return "/opt/rocm"


def build(force=False, verbose=False):
    if os.environ.get("GLC_CXXFLAGS", "").strip() and os.path.abspath(LIB) == os.path.join(HERE, "libglc_amd.so"):
        raise RuntimeError("a build with GLC_CXXFLAGS is a variant: name its library with GLC_LIB_OUT (libglc_amd.so is always the "
                           "plain build -- tests, bench.py and smoke() load that one)")
    hipcc = os.environ.get("HIPCC", os.path.join(rocm_root(), "bin", "hipcc"))
    rccl = have_rccl()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s)) and (rccl or s != "exchange.cpp")]
    os.makedirs(OBJ, exist_ok=True)
    hm = _header_mtime()
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(hm, os.path.getmtime(src)):
            jobs.append((src, obj))
    global LAST
    LAST = {"sources": len(srcs), "compiled": len(jobs), "linked": False, "forced": bool(force)}
    if not jobs and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(os.path.join(OBJ, s + ".o")) for s in srcs):
        return LIB

    def cc(job):
        cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", job[0], "-o", job[1]]
        if verbose:
            print(" ".join(cmd))
        return job[0], subprocess.run(cmd, capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for src, r in ex.map(cc, jobs):
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError("hipcc failed compiling " + src)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + \
          [os.path.join(OBJ, s + ".o") for s in srcs] + (["-L" + os.path.join(rocm_root(), "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(rocm_root(), "lib")] if rccl else [])
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed linking libglc_amd.so")
    LAST["linked"] = True
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
