"""Data-parallel scaling parity (last sentence): the images a 2-rank job produces are bit-identical
to the 1-rank job's, image by image.  Both jobs run the real two-stage pipeline on the GPU in fresh child processes;
the two ranks share GPU 0 (RCCL refuses two ranks on one device, so this rehearsal gathers through gloo: the
shard / pad / gather / unshard control flow is the one `bench.py --gpus N` and infer_dir use with RCCL)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_dp_pipeline_worker.py")


def _start(world, n_images, out_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world),
                   LOCAL_RANK=str(r), RSVLD_DIST_BACKEND="gloo", RSVLD_DEVICE_OVERRIDE="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, WORKER, str(n_images), out_path], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    return procs


def _finish(procs, out_path):
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0 and "DP_WORKER_OK" in out, out[-3000:]
    return np.load(out_path)


def test_two_ranks_equal_one_rank_bit_for_bit(cuda, tmp_path):
    # the two jobs are independent: 3 child processes share GPU 0 side by side (3 images: uneven over 2 ranks, rank 1 pads)
    j1, j2 = _start(1, 3, str(tmp_path / "w1.npy")), _start(2, 3, str(tmp_path / "w2.npy"))
    one, two = _finish(j1, str(tmp_path / "w1.npy")), _finish(j2, str(tmp_path / "w2.npy"))
    assert one.shape == two.shape == (3, 3, 64, 64) and one.dtype == np.uint8
    assert one.std() > 1.0
    for i in range(3):
        assert np.array_equal(one[i], two[i]), f"image {i} differs between the 1-rank and the 2-rank job"
    assert not np.array_equal(one[0], one[1])            # per-image seeds: the images are different images
