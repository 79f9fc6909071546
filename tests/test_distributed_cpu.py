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


def test_shard_blocks_covers_every_pair_block_once_and_balances(g4_pairs):
    """strong scaling of one batch: (pair, 64-chain block) units by LPT — every unit on exactly one rank, block counts within one
    of each other, cost within one unit; the Philox ids a rank hands its kernels (problem id = pair, chain offset = 64 x block)
    are the one-rank run's"""
    from pyhillfit_amd import distributed as pd
    costs = np.array([525.0 + 30.0 * p["n_total"] for p in g4_pairs.values()])
    P, B = len(costs), 64                                            # the full Crumb set at 4 096 chains per pair: 13 440 blocks
    everything = {(q, b) for q in range(P) for b in range(B)}
    for world in (1, 2, 3, 4, 8):
        parts = pd.shard_blocks(costs, B, world)
        assert len(parts) == world and sum(len(p) for p in parts) == P * B
        seen = set()
        for p in parts:
            units = set(map(tuple, p.tolist()))
            assert len(units) == len(p) and not (units & seen)
            seen |= units
        assert seen == everything
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 1 + (world == 3), sizes     # 1 680 each at 8 ranks
        loads = [float(costs[p[:, 0]].sum()) for p in parts]
        assert max(loads) - min(loads) <= costs.max(), loads
        # global chain ids of a rank's units: chain_offset 64 b + lane — disjoint over ranks, complete over the batch
        ids = np.concatenate([(p[:, 0] * (B * 64) + p[:, 1] * 64)[:, None] + np.arange(64)[None, :] for p in parts]).ravel()
        assert np.array_equal(np.sort(ids), np.arange(P * B * 64))
    # ragged: different block counts per problem (a ladder of few-chain rungs next to a big pair), more ranks than some problems' blocks
    parts = pd.shard_blocks([3.0, 1.0, 2.0], [1, 5, 2], 4)
    assert sorted(u for p in parts for u in map(tuple, p.tolist())) == [(0, 0), (1, 0), (1, 1), (1, 2), (1, 3), (1, 4), (2, 0), (2, 1)]
    assert sorted(len(p) for p in parts) == [1, 2, 2, 3] and parts[0].tolist()[0] == [0, 0]


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


# ---- PyHillTemp: (pair, rung) work units over the ranks (python/PyHillTemp.py:151-161 maps the rungs of ONE pair over its pool) ----
def test_tempered_units_partition_covers_every_pair_and_rung():
    from pyhillfit_amd import PyHillTemp as T
    for points, R, world in (([12], 41, 8), ([12, 20, 6], 32, 2), ([12] * 210, 32, 8), ([7], 3, 8)):
        parts = T.partition_units(points, R, world)
        assert len(parts) == world
        allu = np.sort(np.concatenate(parts))
        assert np.array_equal(allu, np.arange(len(points) * R))                       # disjoint and complete
        sizes = [len(p) for p in parts]
        if len(set(points)) == 1:
            assert max(sizes) - min(sizes) <= 1                                        # ONE pair on 8 GPUs: 41 rungs -> 6,5,5,5,5,5,5,5
    assert sorted(len(p) for p in T.partition_units([12], 41, 8)) == [5] * 7 + [6]


def _tempered_worker(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from pyhillfit_amd import PyHillTemp as T
    from pyhillfit_amd import distributed as pd
    from pyhillfit_amd import doseresponse as dr
    pd.init(backend="gloo")
    try:
        pd.setup_data_file(os.path.join(REPO, "data", "crumb_dataset.json") if rank == 0 else "/nonexistent.json", src=0)
        dr.define_model(2)
        dr.output_root = tmp
        pairs = [("Amiodarone", "hERG"), ("Bepridil", "Kv4.3")]
        temperatures = dr.temperature_ladder(4)                                        # 5 rungs
        R, d = len(temperatures), 3
        mine = T.partition_units([12, 16], R, world)[rank]
        # stand-in for the sampler's per-unit expectations: a known function of (pair, rung), so the assembly can be checked
        rows = np.array([[u // R, u % R, -100.0 + 7 * (u // R) + 10 * temperatures[u % R], -99.0 + u, 6.0, 1.0, 8.0, -40.0 - u] for u in mine]).reshape(-1, 4 + d + 1)
        gathered = pd.gather_rows(torch.as_tensor(rows, device=pd.collective_device("cuda:0")), dst=0)
        out = None
        if rank == 0:
            rungs, tis = T.assemble_thermodynamic_integration(np.concatenate(gathered), pairs, temperatures, 2, {"chains": 64, "ranks": world})
            out = (len(rungs), [r_["temperature"] for r_ in rungs[:R]], [t["expectation_pooled"] for t in tis], tis[1]["log_py_chain0"],
                   rungs[R]["drug"], rungs[R]["chain_file"], tis[0]["ranks"])
        q.put((rank, len(mine), out))
        dist.barrier()
    finally:
        pd.finalize()
    assert not dist.is_initialized()


def test_tempered_pair_rung_units_gathered_and_assembled_world_size_2(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tempered_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60); assert p.exitcode == 0
    assert res[0][1] + res[1][1] == 10 and min(res[0][1], res[1][1]) >= 4             # both ranks sampled rungs of the 2 pairs
    n, temps, expect, chain0, drug, chain_file, ranks = res[0][2]
    lad = (np.arange(5) / 4.) ** 3
    assert res[1][2] is None and n == 10 and np.allclose(temps, lad) and ranks == 2
    trap = lambda v: float(np.sum(0.5 * (v[1:] + v[:-1]) * np.diff(lad)))             # doseresponse.py:192-193
    assert expect == pytest.approx([trap(-100.0 + 10 * lad), trap(-93.0 + 10 * lad)], rel=1e-12)
    assert chain0 == [-99.0 + u for u in range(5, 10)] and drug == "Bepridil"
    assert chain_file.endswith("single-level/Bepridil/Kv4.3/model_2/temperature_0.0/chain/Bepridil_Kv4.3_model_2_temp_0.0_chain_single-level.txt")
    with pytest.raises(RuntimeError):                                                  # a missing unit is an error, not a silent gap
        from pyhillfit_amd import PyHillTemp as T
        from pyhillfit_amd import doseresponse as dr
        dr.setup(os.path.join(REPO, "data", "crumb_dataset.json")); dr.define_model(2)
        T.assemble_thermodynamic_integration(np.zeros((3, 8)), [("Amiodarone", "hERG")], lad, 2, {})


def test_num_cores_flag_starts_ranks_through_the_cli_entry_point(tmp_path):
    """`PyHillFit.py ... -c 2` (the reference's pool size, python/PyHillFit.py:40,997-1003) starts two ranks as a child torchrun
    and passes the exit code on.  On the CPU: gloo, and the one step of the command line that needs no GPU (-bfo: best fits only,
    python/PyHillFit.py:739-746) — both ranks go through broadcast, partition and gather, together they write every pair's file."""
    import subprocess
    import sys
    from pyhillfit_amd import distributed as pd
    from pyhillfit_amd import doseresponse as dr
    assert pd.ranks_for_cores(1) == 0 and pd.ranks_for_cores(0) == 0
    dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
    csv = str(tmp_path / "crumb_data.csv")
    dr.table.to_csv(csv)
    out = str(tmp_path / "output")
    env = dict(os.environ, PHF_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(REPO, "python", "PyHillFit.py"), "--data-file", csv, "-m", "2", "-a", "-bfo", "-c", "2",
           "--output-root", out, "--write-workers", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "gathered summaries from 2 ranks" in r.stdout
    import glob
    assert len(glob.glob(os.path.join(out, "crumb_data", "single-level", "*", "*", "model_2", "temperature_1", "figures", "*_best_fit_params.txt"))) == 210
    # a failing rank's exit code comes back through the parent (a data file that does not exist)
    bad = subprocess.run(cmd[:3] + [str(tmp_path / "missing.csv")] + cmd[4:], env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0
    # inside a torchrun world the flag starts nothing
    os.environ["WORLD_SIZE"] = "2"
    try:
        assert pd.ranks_for_cores(8) == 0
    finally:
        del os.environ["WORLD_SIZE"]
