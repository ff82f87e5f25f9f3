"""The NTT-bnf engines of the MI355X on the inputs of the reference's GPU golden-value test (tests/test_pbs_golden.py has the
rest).  Written after the round's GPU minutes had run out: its logic ran on the host emulation (both engines, one input,
bit-equal to the oracle and 2^49 from the golden bytes in phase), not yet on hardware — hence a file that sorts last."""
import numpy as np
import pytest

from .harness import Ctx, oracle_pbs
from .test_pbs_golden import check_against_golden, golden_setup


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["ntt64", "ntt64_split"])
def test_ntt_engines_on_the_golden_inputs(engine):
    """The NTT-bnf engines of the MI355X (integer Goldilocks kernel and its split-key f64 form) on the reference's golden
    inputs: bit-equal to the oracle's NTT path, within transform noise of the H100's f64 bytes in phase."""
    p, keys, lut, inputs, messages, golden, _ = golden_setup("classical")
    c = Ctx("hip", p, keys, engine)
    out = c.pbs(np.repeat(inputs, 3, axis=0), lut)
    ref = oracle_pbs(p, keys, "ntt64", inputs, lut)
    for i, m in enumerate(messages):
        for lane in range(3):
            assert np.array_equal(out[3 * i + lane], ref[i]), (engine, m, lane)
        check_against_golden(out[3 * i], golden[i], keys.glwe_sk, m, f"{engine} engine")
