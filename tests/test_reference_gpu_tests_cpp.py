"""The reference's own GPU tests for the path (tfhe/src/core_crypto/gpu/algorithms/test/*.rs), restated in C++ in
tests/cpp/reference_gpu_tests.cpp on top of the compiled host mirror tfhe_rs_amd/host/core_crypto_gpu.hpp and LINKED
against the backend library like the Rust crate would be (the Rust originals cannot be compiled
here).  [emu] small parameter sets against the host-emulation build of the kernels (CPU tier);
[hip] the reference's parameter sets (TEST_PARAMS_4_BITS_NATIVE_U64, MULTI_BIT_2_2_{2,3,4}_PARAMS) on the MI355X."""
import os
import subprocess

import pytest

from .harness import build_emu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PRODUCT_LIB = os.path.join(ROOT, "tfhe_rs_amd", "lib", "libtfhe_hip_backend.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "libtfhe_oracle.so")


def build_tests(lib, exe, source="reference_gpu_tests.cpp"):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-o", exe,
           os.path.join(HERE, "cpp", source), lib, ORACLE_LIB,
           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath," + os.path.dirname(ORACLE_LIB)]
    subprocess.check_call(cmd)
    return exe


def run(exe, *args, timeout, env=None):
    env = dict(env or {}, TFHE_FFT_GOLDEN=os.path.join(HERE, "golden", "fft16x4x16_golden_v1.json"))
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **env))
    for line in r.stderr.splitlines():  # TFHE_HIP_ARENA_REDZONE=1 (tools/redzone_pass.sh): what the binary's library checked
        if "[arena red zone]" in line:
            print(os.path.basename(exe), " ".join(args), line)
    assert r.returncode == 0, r.stdout + r.stderr
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("test result: ok."), r.stdout
    return r.stdout


@pytest.fixture(scope="module")
def hip_exe(tmp_path_factory):
    assert os.path.exists(PRODUCT_LIB)
    return build_tests(PRODUCT_LIB, str(tmp_path_factory.mktemp("cpp") / "reference_gpu_tests"))


def test_reference_gpu_tests_on_the_host_emulation(tmp_path):
    exe = build_tests(build_emu(), str(tmp_path / "reference_gpu_tests_emu"))
    out = run(exe, "toy", timeout=1500)
    assert out.count(" ... ok") == 23, out


@pytest.mark.gpu
def test_reference_gpu_tests_with_the_reference_parameter_sets(hip_exe):
    out = run(hip_exe, "reference", timeout=1500)
    # 4 classic + 3 multi-bit bootstraps + 2 multi-bit keyswitches + 2 noise-test flows (multi-bit switch -> blind rotation)
    # + KS32 keyswitch + 2 closest-representable + modulus switch + 2 cooperative modulus switches + 3 transform tests + panics
    assert out.count(" ... ok") == 21, out
    print(out)


@pytest.mark.gpu
def test_reference_gpu_tests_small_sets_on_the_gpu(hip_exe):
    out = run(hip_exe, "toy", timeout=600)
    assert out.count(" ... ok") == 23, out

