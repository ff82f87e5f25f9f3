"""bench.py's multi-GPU logic on the CPU tier: `python bench.py --gpus N` drives N GPUs from one process — one host
thread, stream, key replica and shard per GPU, the shape of the reference's own throughput bench
(tfhe-benchmark/benches/core_crypto/pbs_bench.rs:1050-1160; shards by helper_multi_gpu.cu:71-101).  Here the same
functions run against the host emulation of the kernel sources with two pretend devices and a toy key; the GPU tier
runs the real command line with TFHE_BENCH_FAKE_MULTI_GPU=1 (two streams of the one GPU of the test box)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from .common import TOY_2048, make_keys
from .harness import oracle_pbs, use_backend

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))


def test_in_process_two_gpus_shards_verified_and_counted(monkeypatch):
    monkeypatch.setenv("HIPEMU_DEVICES", "2")
    import bench
    lib = use_backend("emu")
    p, keys = TOY_2048, make_keys(TOY_2048)
    devices, fake = bench.pick_devices(lib, 2)
    assert devices == [0, 1] and not fake
    B, steps = 5, 2
    elapsed, per_gpu, shard0, kernel_ms, kernel_id = bench.run_in_process(lib, p, keys, devices, B, steps, 1)
    assert len(per_gpu) == 2 and [g["device"] for g in per_gpu] == [0, 1]
    assert all(g["verified"] and g["lwes"] == B for g in per_gpu)
    assert len(kernel_ms) == steps and elapsed >= max(g["seconds"] for g in per_gpu) > 0
    # the shards are the two halves of ONE global batch (messages continue across the shard boundary) and shard 0's
    # device output is the oracle's, bit for bit
    assert shard0.msgs == [i % p.plaintext_modulus for i in range(B)]
    assert np.array_equal(shard0.outputs(), oracle_pbs(p, keys, "fft64", shard0.cts, shard0.lut))
    shard0.close()


def test_ragged_global_batch_over_three_gpus(monkeypatch):
    """A global batch that does not divide by the GPU count (11 LWEs over 3 devices: 4 + 4 + 3, the CPU-tier image of
    4099 over 3) is split by the reference's rule (helper_multi_gpu.cu:71-101: the first B mod G GPUs take one more),
    contiguously and without a gap or an overlap; every shard is verified and the shards together are the global batch."""
    monkeypatch.setenv("HIPEMU_DEVICES", "3")
    import bench
    from tfhe_rs_amd.multi_gpu import get_num_inputs_on_gpu
    lib = use_backend("emu")
    p, keys = TOY_2048, make_keys(TOY_2048)
    devices, fake = bench.pick_devices(lib, 3)
    assert devices == [0, 1, 2] and not fake
    G, steps = 11, 1
    elapsed, per_gpu, shard0, kernel_ms, _ = bench.run_in_process(lib, p, keys, devices, None, steps, 0, global_batch=G)
    assert [g["lwes"] for g in per_gpu] == [get_num_inputs_on_gpu(G, i, 3) for i in range(3)] == [4, 4, 3]
    assert [g["first_lwe"] for g in per_gpu] == [0, 4, 8] and sum(g["lwes"] for g in per_gpu) == G
    assert all(g["verified"] for g in per_gpu)
    assert shard0.msgs == [i % p.plaintext_modulus for i in range(4)]
    assert np.array_equal(shard0.outputs(), oracle_pbs(p, keys, "fft64", shard0.cts, shard0.lut))
    shard0.close()


def test_more_shards_than_gpus_is_refused_unless_faked(monkeypatch):
    monkeypatch.setenv("HIPEMU_DEVICES", "1")
    import bench
    lib = use_backend("emu")
    monkeypatch.delenv("TFHE_BENCH_FAKE_MULTI_GPU", raising=False)
    with pytest.raises(SystemExit):
        bench.pick_devices(lib, 4)
    monkeypatch.setenv("TFHE_BENCH_FAKE_MULTI_GPU", "1")
    assert bench.pick_devices(lib, 4) == ([0, 0, 0, 0], True)
    assert bench.pick_devices(lib, 1) == ([0], False)


@pytest.mark.gpu
def test_bench_command_line_with_two_fake_gpus():
    """The driver's command shape, `python bench.py --gpus 2`, with no torch.distributed launcher around it: prints
    n_gpus = 2, both shards verified; config 5 runs through both shardings (by the caller and inside the library)."""
    env = dict(os.environ, TFHE_BENCH_FAKE_MULTI_GPU="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                                   "--warmup", "1"], env=env, text=True, timeout=1500)
    r = json.loads(out.strip().splitlines()[-1])
    assert r["n_gpus"] == 2 and r["fake_multi_gpu"] is True
    assert len(r["per_gpu"]) == 2 and all(g["verified"] for g in r["per_gpu"])
    assert r["value"] == pytest.approx(2 * 4096 * 2 / (r["ms_per_step"] * 2e-3), rel=1e-6)
    for key in ("fheuint64", "fheuint64_in_library_sharding"):
        for op in ("add", "mul"):
            assert r["extra"][key][op]["results_decrypt_to_clear_arithmetic"], (key, op)
            assert sum(r["extra"][key][op]["per_shard_integers"]) == r["extra"][key][op]["batch"]
