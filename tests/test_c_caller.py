"""The boundary is a C ABI: a plain-C program (gcc, not hipcc) shaped like the reference's own compress
test calls include/cudpp.h and must print the known answers of the reference vector."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_caller", "cudpp_rig.c")
PKG = os.path.join(ROOT, "gpu-lossless-compression_amd")


ORACLE = os.path.join(ROOT, "oracle")


def _build(tmp_path, src=SRC, name="cudpp_rig", with_oracle=False):
    exe = str(tmp_path / name)
    cmd = ["gcc", "-O1", "-std=gnu99", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", src, "-o", exe,
           "-L", PKG, "-lglc_amd", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    if with_oracle:
        cmd += ["-L", ORACLE, "-lglc_oracle", "-Wl,-rpath," + ORACLE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_caller_compiles_and_links_with_gcc(glc, tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    glc.lib()                                                  # builds libglc_amd.so if needed
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_caller_reproduces_the_reference_known_answers(glc, tmp_path):
    glc.lib()
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict(kv.split("=") for kv in r.stdout.split())
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_gold_1m.json")))
    ct, st = want["compressTest"], want["compressTest_stream_restatement"]
    assert out["in_crc"] == ct["crc_in"]                       # the generator matches the reference's (glibc rand)
    assert int(out["bwt_index"]) == ct["bwt_index"] == 296638
    assert int(out["size_words"]) == st["size_words"] == 262491
    assert out["crc_words"] == st["crc_words"] == "e12686dc"
    assert out["crc_offsets"] == st["crc_offsets"] == "62c9b10a"
    assert out["crc_hist"] == "aaaaa264"


@pytest.mark.gpu
def test_c_caller_culzss_pipeline_sequence(glc, tmp_path):
    """the reference pipeline's call sequence (culzss.c / deculzss.c) from plain C, four ring slots:
    candidates and packed bytes equal the oracle's, in-place decompression restores the input"""
    import oracle_lib as O
    glc.lib()
    O.lib()                                                    # builds liboracle if needed
    exe = _build(tmp_path, os.path.join(ROOT, "tests", "c_caller", "culzss_rig.c"), "culzss_rig", with_oracle=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("packed_equal=1") == 4 and r.stdout.count("candidates_equal=1") == 4


@pytest.mark.gpu
def test_c_caller_culzss_ring_of_four_slots(glc, tmp_path):
    """the reference's caller shape (culzss.c:85-176): four ring slots in flight -- from one thread, and from a producer, a
    GPU thread and a CPU thread -- gives the bytes of one buffer at a time (tests/c_caller/culzss_ring_bench.c)"""
    glc.lib()
    exe = str(tmp_path / "culzss_ring_bench")
    cmd = ["gcc", "-O2", "-std=gnu99", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "c_caller", "culzss_ring_bench.c"), "-o", exe,
           "-L", PKG, "-lglc_amd", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "48", "12"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "bytes_equal=1" in r.stdout, r.stdout + r.stderr


def test_exchange_rig_compiles_with_gcc(glc, tmp_path):
    """include/glc_exchange.h needs neither hipcc nor the RCCL headers on the caller's side"""
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    glc.lib()
    assert os.path.exists(_build(tmp_path, os.path.join(ROOT, "tests", "c_caller", "exchange_rig.c"), "exchange_rig"))


@pytest.mark.gpu
def test_c_caller_exchange_sequence_world_of_one(glc, tmp_path):
    """a rank of the multi-GPU path written in C: encode, compact, records, RCCL count exchange, gather, scatter,
    unpack, expand, decode -- the input comes back (tests/c_caller/exchange_rig.c)"""
    glc.lib()
    exe = _build(tmp_path, os.path.join(ROOT, "tests", "c_caller", "exchange_rig.c"), "exchange_rig")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL OK" in r.stdout and "round_trip=1" in r.stdout, r.stdout + r.stderr


def _build_cpp(tmp_path):
    exe = str(tmp_path / "cuhd_adapter_rig")
    cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "c_caller", "cuhd_adapter_rig.cpp"), "-o", exe,
           "-L", PKG, "-lglc_amd", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_cuhd_adapter_compiles_with_gpp(glc, tmp_path):
    """include/glc_cuhd_adapter.hpp (the reference's cuhd::CUHDGPUDecoder::decode signature over the C ABI) is plain
    host C++: g++, not hipcc"""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    glc.lib()
    assert os.path.exists(_build_cpp(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("case", [0, 1])
def test_cuhd_adapter_decodes_reference_encoded_stream(glc, tmp_path, case):
    """a C++ caller with the reference's call shape decodes a stream written by the reference's own encoder
    (tests/golden/ref_cuhd_gold.npz) -- table in device memory, as cuhd::CUHDGPUCodetable holds it"""
    import numpy as np
    glc.lib()
    ref = np.load(os.path.join(ROOT, "tests", "golden", "ref_cuhd_gold.npz"))
    name = [str(c) for c in ref["cases"]][case]
    import test_hd
    sym = test_hd._ref_symbols(name)
    ref[name + "_units"].astype(np.uint32).tofile(str(tmp_path / "units.bin"))
    np.ascontiguousarray(ref[name + "_table"]).astype(np.uint8).tofile(str(tmp_path / "table.bin"))
    sym.astype(np.uint8).tofile(str(tmp_path / "symbols.bin"))
    exe = _build_cpp(tmp_path)
    r = subprocess.run([exe, str(tmp_path / "units.bin"), str(tmp_path / "table.bin"), str(tmp_path / "symbols.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "decoded_equals_original=1" in r.stdout, r.stdout + r.stderr
