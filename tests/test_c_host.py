"""The boundary from compiled code: tests/cpp/abi_smoke.c is plain C that dlopen()s the library, binds the
reference's symbol names and runs scratch -> key conversion -> PBS -> cleanup with raw pointers, then compares
the output with the oracle's bits from a fixture written here.  [emu] the host-emulation build (CPU),
[hip] the product library on the MI355X."""
import os
import subprocess

import numpy as np
import pytest

from . import oracle as orc
from .common import TOY_2048, TOY_K2, encrypt_small, make_keys
from .harness import EMU_LIB, build_emu, oracle_pbs

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PRODUCT_LIB = os.path.join(ROOT, "tfhe_rs_amd", "lib", "libtfhe_hip_backend.so")


def build_smoke(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-o", exe, os.path.join(HERE, "cpp", "abi_smoke.c"), "-ldl"])
    return exe


def write_fixture(path, p):
    keys = make_keys(p, with_ksk=False)
    msgs = [m % p.plaintext_modulus for m in range(5)]
    cts = encrypt_small(p, keys, msgs, seed=77)
    lut = orc.generate_lut(p.k, p.N, p.plaintext_modulus, p.delta, lambda x: (x + 3) % p.plaintext_modulus)
    want = oracle_pbs(p, keys, "fft64", cts, lut)
    hdr = np.array([p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.ms_type, len(msgs)], dtype="<u8")
    with open(path, "wb") as f:
        for a in (hdr, keys.bsk, lut, cts, want):
            f.write(np.ascontiguousarray(a, dtype="<u8").tobytes())


@pytest.mark.parametrize("kind", [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("p", [TOY_2048, TOY_K2], ids=lambda p: p.name)
def test_pbs_from_plain_c(kind, p, tmp_path):
    lib = build_emu() if kind == "emu" else PRODUCT_LIB
    assert os.path.exists(lib)
    exe = build_smoke(tmp_path)
    fixture = str(tmp_path / "fixture.bin")
    write_fixture(fixture, p)
    r = subprocess.run([exe, lib, fixture], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("OK")
