"""The C++ host wrapper (basevar_amd/host/basetype_gpu.hpp): compiles against the C ABI on CPU,
refuses to run without a GPU, and on a GPU reproduces the oracle on the sites it packed."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_example(tmp_path):
    import __graft_entry__ as g
    g.build()
    exe = str(tmp_path / "cpp_host_example")
    lib = os.path.join(ROOT, "basevar_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "cpp_host_example.cpp"), "-L", lib, "-lbasevar_amd",
                           "-Wl,-rpath," + lib, "-o", exe])
    return exe


def test_cpp_wrapper_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = build_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode != 0 and "no HIP device" in p.stderr


@pytest.mark.gpu
def test_cpp_wrapper_matches_oracle(tmp_path, restatement):
    import oracle
    from parity import compare_sites, describe
    exe = build_example(tmp_path)
    dump = str(tmp_path / "dump.bin")
    out = subprocess.check_output([exe, dump], text=True)
    assert "ALT=" in out and "tagged rank layout: records identical" in out
    raw = open(dump, "rb").read()
    S, N, P = np.frombuffer(raw[:24], dtype=np.uint64).astype(int)
    o = 24
    planes = {}
    for name, dt, w in (("base_strand", np.uint8, 1), ("qual", np.uint8, 1), ("mapq", np.uint8, 1), ("rpr", np.uint16, 2)):
        planes[name] = np.frombuffer(raw[o:o + S * P * w], dtype=dt).reshape(S, P)
        o += S * P * w
    planes["ref_base"] = np.frombuffer(raw[o:o + S], dtype=np.uint8)
    o += S
    got = np.frombuffer(raw[o:o + S * oracle.SITE_DTYPE.itemsize], dtype=oracle.SITE_DTYPE)
    planes["n_samples"] = N
    exp, _ = restatement.run(planes, restatement.min_af(N, 0.01))
    bad = compare_sites(got, exp)
    assert not bad, describe(bad, got, exp)
    assert ((exp["status"] & 2) != 0).sum() >= 5


@pytest.mark.gpu
def test_python_example_runs():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "python_example.py")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.count("site ") == 3 and "alts ['C']" in p.stdout
