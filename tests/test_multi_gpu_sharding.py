"""Multi-GPU path on CPU: the sharding rule of the reference (helper_multi_gpu.cu:71-101) and a
world-size-2 run over gloo where each rank bootstraps its own contiguous shard (host-emulation
backend, no GPU) — no data-path collective, results gathered only to check them."""
import os
import subprocess
import sys

import numpy as np

from tfhe_rs_amd.multi_gpu import get_gpu_offset, get_num_inputs_on_gpu, shard_range

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))


def test_get_num_inputs_on_gpu_rule():
    for total in (0, 1, 7, 8, 9, 4096, 4099):
        for g in (1, 2, 3, 4, 8):
            counts = [get_num_inputs_on_gpu(total, i, g) for i in range(g)]
            assert sum(counts) == total
            assert max(counts) - min(counts) <= 1
            assert counts == sorted(counts, reverse=True)      # ceil(B/G) first, floor after
            offs = [get_gpu_offset(total, i, g) for i in range(g)]
            assert offs == [sum(counts[:i]) for i in range(g)]
            assert [shard_range(total, i, g) for i in range(g)] == [(o, o + c) for o, c in zip(offs, counts)]


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["REPO_ROOT"])
import numpy as np
import torch
import torch.distributed as dist
from tests import oracle as orc
from tests.common import TOY_K1, make_keys, encrypt_small
from tests.harness import Ctx, oracle_pbs
from tfhe_rs_amd.multi_gpu import shard_range

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
p = TOY_K1
keys = make_keys(p)                      # every rank holds a replica of the key
B = 11                                   # ragged: 6 + 5
msgs = [m % p.plaintext_modulus for m in range(B)]
cts = encrypt_small(p, keys, msgs, seed=77)
lut = orc.generate_lut(p.k, p.N, p.plaintext_modulus, p.delta, lambda x: (x + 1) % p.plaintext_modulus)
lo, hi = shard_range(B, rank, world)
c = Ctx("emu", p, keys, "fft64")
out = c.pbs(cts[lo:hi], lut)             # this rank's shard only
ref = oracle_pbs(p, keys, "fft64", cts[lo:hi], lut)
ok = torch.tensor([int(np.array_equal(out, ref)), hi - lo])
dist.barrier()
dist.all_reduce(ok, op=dist.ReduceOp.SUM)   # check-only reduction, not part of the data path
if rank == 0:
    print("RESULT", int(ok[0]), int(ok[1]))
dist.destroy_process_group()
'''


def test_world_size_2_gloo_shards(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, REPO_ROOT=ROOT, OMP_NUM_THREADS="2")
    out = subprocess.check_output(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29531", str(script)], env=env, text=True, stderr=subprocess.STDOUT, timeout=600)
    line = [l for l in out.splitlines() if l.startswith("RESULT")][-1].split()
    assert line[1] == "2", out      # both ranks matched the oracle bit-for-bit
    assert line[2] == "11", out     # the shards cover the batch exactly once
