"""The reference's GPU tests of the radix operations the backend wires (integer/gpu/server_key/radix/tests_unsigned:
unchecked_add_test, default_add_test, default_overflowing_add_test, default_mul_test and, round 6, the default sub / bitop /
comparison / if_then_else / scalar shift tests), restated in C++ in
tests/cpp/reference_integer_gpu_tests.cpp on the compiled host mirror tfhe_rs_amd/host/integer_gpu.hpp and linked against
the library.  [emu] small sets on the host emulation; [hip] PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128 and the GPU
multi-bit g = 4 set on the MI355X.  (The file sorts last: it is the longest of the tier, and a surprise here must not hide the rest of it behind `pytest -x`;
it has passed on the MI355X in every full run of round 5, `profiles/r05_gputest.log`.)"""
import pytest

from .harness import build_emu
from .test_reference_gpu_tests_cpp import PRODUCT_LIB, build_tests, run


@pytest.fixture(scope="module")
def emu_exe(tmp_path_factory):
    return build_tests(build_emu(), str(tmp_path_factory.mktemp("cpp") / "reference_integer_gpu_tests_emu"),
                       "reference_integer_gpu_tests.cpp")


@pytest.fixture(scope="module")
def hip_exe(tmp_path_factory):
    return build_tests(PRODUCT_LIB, str(tmp_path_factory.mktemp("cpp") / "reference_integer_gpu_tests"),
                       "reference_integer_gpu_tests.cpp")


def test_reference_integer_gpu_tests_on_the_host_emulation(emu_exe):
    out = run(emu_exe, "toy", timeout=1500)
    assert out.count(" ... ok") == 20, out   # 10 tests on each of the two small sets


@pytest.mark.parametrize("no_peer", [0, 1], ids=["peer_access", "host_staged"])
def test_multi_device_integer_add_on_the_emulated_device_model(emu_exe, no_peer):
    """multi_device_integer_add (GpuMultiDeviceFunctionExecutor: the server key on a random subset of the GPUs in a random
    order) on the emulation's device model — four pretend devices, with and without peer access between them."""
    out = run(emu_exe, "toy", "multi_device", timeout=1500, env={"HIPEMU_DEVICES": "4", "HIPEMU_NO_PEER": str(no_peer)})
    assert out.count(" ... ok") == 2 and "Setting up server key on GPUs" in out, out


@pytest.mark.gpu
def test_reference_integer_gpu_tests_with_the_reference_parameter_sets(hip_exe):
    out = run(hip_exe, "reference", timeout=1500)
    assert out.count(" ... ok") >= 20 and "FAILED" not in out, out   # + 2 multi-device additions on a node with several GPUs
    print(out)


@pytest.mark.gpu
def test_reference_integer_gpu_tests_small_sets_on_the_gpu(hip_exe):
    out = run(hip_exe, "toy", timeout=600)
    assert out.count(" ... ok") >= 20 and "FAILED" not in out, out
