"""N > 1 path on CPU: two gloo ranks shard a batch of independent images, each 'processes' its shard, and
one all-gather of uint8 images reassembles the batch in image order."""
import os
import socket
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_shard_and_gather():
    _ranks(8, 2)
    _ranks(5, 2)      # uneven: rank 0 holds 3 images, rank 1 holds 2 and pads


def test_eight_rank_shard_and_gather():
    """BASELINE configs[4]'s partition: 64 images over 8 ranks (8 each), and 61 (ranks 0-4 hold 8, ranks 5-7 hold 7 and pad) --
    the rank count the driver's scaling run uses, rehearsed on gloo."""
    _ranks(64, 8)
    _ranks(61, 8)


def _ranks(n, world):
    import json
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(ROOT, "tests", "_dist_worker.py")
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, worker, str(n)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    res = []
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out
        res.append(json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][7:]))
    res.sort(key=lambda d: d["rank"])
    want = [int(round(((i / n * 2 - 1) + 1) * 0.5 * 255)) for i in range(n)]
    for r in range(world):
        assert res[r]["idx"] == list(range(r, n, world))
        assert res[r]["vals"] == want                           # every rank holds all images, in image order
        assert res[r]["rows"] == [[q + 0.5, 10.0 * q] for q in range(world)]      # bench.dist_gather_rows: every rank's figures on every rank
        assert res[r]["max"] == [float(world - 1), 7.0]                            # bench.dist_max
    assert sorted(i for d in res for i in d["idx"]) == list(range(n))   # each image processed exactly once


def test_shard_and_unshard_are_inverse():
    sys.path.insert(0, ROOT)
    from rsvld_amd import parallel
    for n, world in ((8, 2), (64, 8), (16, 4), (5, 4), (7, 2), (9, 4)):   # incl. uneven shards: short ranks pad their tail
        per = -(-n // world)
        rows = []
        for r in range(world):
            idx = parallel.shard_indices(n, r, world)
            rows += idx + [idx[-1]] * (per - len(idx))
        assert parallel.unshard(torch.tensor(rows), n, world).tolist() == list(range(n))
    x = torch.tensor([-2.0, -1.0, 0.0, 0.999, 1.0, 3.0])
    assert parallel.to_uint8(x).tolist() == [0, 0, 128, 255, 255, 255]
