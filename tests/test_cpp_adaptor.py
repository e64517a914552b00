"""include/ndzip_hip.hh: compiles stand-alone and (where the reference tree exists) against the reference's own headers;
on the GPU box the reference-style round-trip program runs through it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "adaptor_roundtrip.cc")


def test_adaptor_compiles_standalone():
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), SRC], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/include/ndzip"), reason="reference tree only exists in the authoring container")
def test_adaptor_compiles_against_reference_headers():
    """Drop-in check: with the reference's ndzip.hh / offload.hh, hip_offloader<T> IS an ndzip::offloader<T>."""
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-DNDZIP_HIP_WITH_REFERENCE_HEADERS", "-I" + os.path.join(ROOT, "include"),
                        "-I/root/reference/include", SRC], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/include/ndzip"), reason="reference tree only exists in the authoring container")
def test_integration_factory_unit_compiles_in_the_reference_tree(tmp_path):
    """the proposed src/ndzip/hip_factory.cc builds to an object against the reference's headers and
    defines the two factory instantiations the reference's make_offloader switch would call."""
    obj = str(tmp_path / "hip_factory.o")
    r = subprocess.run(["g++", "-std=c++17", "-c", "-I" + os.path.join(ROOT, "include"), "-I/root/reference/include",
                        os.path.join(ROOT, "tests", "cpp", "hip_factory_example.cc"), "-o", obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    syms = subprocess.run(["nm", "-C", obj], capture_output=True, text=True, check=True).stdout
    assert "ndzip::make_hip_offloader<float>(int)" in syms or "make_hip_offloader<float>" in syms
    assert "make_hip_offloader<double>" in syms
    # hip_host_compressor<T> IS an ndzip::compressor<T> of the reference's own header
    assert "make_hip_host_compressor<float>" in syms and "make_hip_host_decompressor<double>" in syms


@pytest.mark.gpu
@pytest.mark.hardware_only
def test_adaptor_roundtrip_on_gpu(tmp_path):
    exe = str(tmp_path / "adaptor_rt")
    libdir = os.path.join(ROOT, "ndzip_amd")
    import torch  # the program must use the same HIP runtime the .so was built against when torch is absent: system ROCm

    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe, "-L" + libdir, "-lndzip_hip",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "adaptor round trips ok" in r.stdout


def test_adaptor_roundtrip_on_the_cpu_model(tmp_path):
    """The same reference-style program, linked against the wave64 functional model of the library (tests/wavesim, test
    infrastructure): the adaptor classes, the C ABI behind them and the kernels' logic, without a GPU."""
    from tests.wavesim import build as simbuild

    lib = simbuild.build()
    exe = str(tmp_path / "adaptor_rt_model")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe, lib, "-Wl,-rpath," + os.path.dirname(lib)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "adaptor round trips ok" in r.stdout
