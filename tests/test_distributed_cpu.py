"""world_size-2 `gloo` test on CPU of the tensor-parallel host logic: shard a transformer
MLP + attention-projection pair the Megatron way (infinitensor_amd/tp.py, mirroring
examples/distributed/parallel_opt.py), run each rank's partial with the oracle, all-reduce(sum) the
row-parallel outputs, and compare with the unsharded result — the same check the reference launcher
does against its single-GPU output (cuda_launch.py:70-76). The RCCL collective itself needs GPUs; here
the collective is gloo's, what is tested is the sharding / reduction placement."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_tensor_parallel_sharding_world2():
    import json
    import subprocess

    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="2")
        procs.append(subprocess.Popen([sys.executable, str(REPO / "tests" / "_tp_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT ")][0]
    res = json.loads(line[7:])
    assert res["mlp"] < 1e-10 and res["attn"] < 1e-10


def test_shard_helpers():
    from infinitensor_amd import tp

    w = np.arange(24).reshape(4, 6)
    a, _ = tp.shard_column(w, 2, 1)
    assert np.array_equal(a, w[:, 3:])
    assert np.array_equal(tp.shard_row(w, 2, 0), w[:2])
    with pytest.raises(ValueError):
        tp.shard_row(w, 3, 0)
    assert tp.llama_block_flops(2048, 4096, 11008, 8) == pytest.approx(2 * 2048 * (4 * 4096 ** 2 + 3 * 4096 * 11008) / 8)
