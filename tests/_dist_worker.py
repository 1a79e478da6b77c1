"""One gloo rank of the shard-and-gather test (launched by test_parallel.py with RANK/WORLD_SIZE/MASTER_* set)."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsvld_amd import parallel

n_images = int(sys.argv[1])
rank, world, _ = parallel.init_from_env(backend="gloo")
idx = parallel.shard_indices(n_images, rank, world)
# stand-in for the sampler: image i is a deterministic function of i only (per-image seeds)
allimgs = parallel.run_sharded(lambda i: parallel.to_uint8(torch.full((3, 4, 4), float(i) / n_images * 2 - 1)),
                               n_images, rank, world)
# bench.py's own cross-rank helpers (the per-rank phase table and the max-over-ranks timing of the driver's line) on this backend
import bench
rows = bench.dist_gather_rows([rank + 0.5, 10.0 * rank], torch.device("cpu"), world)
mx = bench.dist_max([float(rank), 7.0 - rank], torch.device("cpu"), world)
dist.barrier()
print("RESULT " + json.dumps({"rank": rank, "idx": idx, "vals": allimgs[:, 0, 0, 0].tolist(), "rows": rows, "max": mx}), flush=True)
dist.destroy_process_group()
