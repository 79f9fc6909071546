"""N>1 plumbing on CPU (gloo, world_size 2): data-set broadcast, problem/chain partition, summary gather.
The sampler itself only runs on a GPU; what must be right by construction here is that shards are disjoint,
complete, balanced, and that Philox ids make the partition invisible in the results."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def test_shard_problems_is_a_balanced_partition(g4_pairs):
    from pyhillfit_amd import distributed as pd
    costs = [p["n_total"] for p in g4_pairs.values()]
    for world in (1, 2, 4, 8):
        parts = pd.shard_problems(costs, world)
        allq = np.sort(np.concatenate(parts))
        assert np.array_equal(allq, np.arange(len(costs)))
        loads = [sum(costs[q] for q in p) for p in parts]
        assert max(loads) - min(loads) <= max(costs)
    assert [pd.shard_chains(65536, r, 8) for r in (0, 7)] == [(0, 8192), (57344, 8192)]
    spans = [pd.shard_chains(10, r, 4) for r in range(4)]
    assert spans == [(0, 3), (3, 3), (6, 2), (8, 2)]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from pyhillfit_amd import distributed as pd
    from pyhillfit_amd import doseresponse as dr
    r, _, w = pd.init(backend="gloo")
    assert (r, w) == (rank, world) and dist.is_initialized()
    # only rank 0 may read the data file: hide it from the others
    path = os.path.join(REPO, "data", "crumb_dataset.json") if rank == 0 else "/nonexistent/crumb_dataset.json"
    pd.setup_data_file(path, src=0)
    assert len(dr.drugs) == 30 and len(dr.channels) == 7 and dr.dir_name == "crumb_dataset" and len(dr.table.dose) == 2585
    packed = None
    if rank == 0:
        packed = dr.pack_single_level([(d, c) for d in dr.drugs for c in dr.channels])
    got = pd.broadcast_packed_points(packed, "cpu", src=0)
    mine = pd.shard_problems(got.counts[:, 3], world)[rank]
    # stand-in for per-problem summaries computed on this rank's shard
    local = torch.tensor(np.column_stack([mine, got.pi_bit[mine]]), dtype=torch.float64)
    rows = pd.gather_rows(local, dst=0)
    q.put((rank, got.num_pairs, got.stride, float(got.ln_conc.sum()) + float(got.weight.sum()) + float(got.extra.sum()), int(got.counts.sum()),
           None if rows is None else [r_.tolist() for r_ in rows], float(got.weight.sum()), float(got.extra[:, 0].sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_gather_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60); assert p.exitcode == 0
    assert res[0][1:5] == res[1][1:5] and res[0][1] == 210 and res[0][2] == 8        # identical data set on both ranks (<= 8 entries per pair)
    assert res[0][6:] == res[1][6:] and res[1][6] == 2584 and res[1][7] == 1805      # weights = points inside [0,100]; uncensored points
    rows = res[0][5]
    assert res[1][5] is None and len(rows) == 2
    ids = np.sort(np.concatenate([np.array(r_)[:, 0] for r_ in rows]))
    assert np.array_equal(ids, np.arange(210))                                        # every problem summarised once
    allrows = np.vstack([np.array(r_) for r_ in rows])
    assert np.array_equal(np.unique(allrows[:, 1] / (0.5 * np.log(2 * np.pi))).round().astype(int), [6, 7, 12, 13, 14, 15, 16, 18, 19, 20])   # pi_bit payload intact
