"""SURVEY 8(d)'s synthetic scaling set S3 (pyhillfit_amd/synthetic.py): the generator, its packing, the partition of its (pair,
64-chain block) units over ranks, and the broadcast of the packed set over gloo with two ranks.  CPU only."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402

from conftest import REPO  # noqa: E402


def test_generator_follows_the_survey_recipe_and_is_deterministic():
    from pyhillfit_amd import synthetic as S
    ex, truth = S.generate(210)
    ex2, _ = S.generate(210)
    assert len(ex) == 210 and all(len(e) == 3 and all(x.shape == (4, 2) for x in e) for e in ex)
    assert all(np.array_equal(a, b) for e1, e2 in zip(ex, ex2) for a, b in zip(e1, e2))
    assert 3 <= truth["pic50"].min() and truth["pic50"].max() <= 9 and 0.5 <= truth["hill"].min() and truth["hill"].max() <= 2
    assert 2 <= truth["sigma"].min() and truth["sigma"].max() <= 10
    # the recipe, re-stated: default_rng(12345); three vectors of P uniforms, then the [P][3][4] normals
    rng = np.random.default_rng(12345)
    pic50, hill, sigma = rng.uniform(3, 9, 210), rng.uniform(0.5, 2, 210), rng.uniform(2, 10, 210)
    z = rng.standard_normal((210, 3, 4))
    assert np.array_equal(pic50, truth["pic50"]) and np.array_equal(hill, truth["hill"]) and np.array_equal(sigma, truth["sigma"])
    p = 17
    ic50 = 10.0 ** (6.0 - pic50[p])
    doses = ic50 * 10.0 ** np.linspace(-2, 2, 4)
    assert np.allclose(ex[p][1][:, 0], doses, rtol=1e-15) and np.isclose(doses[-1] / doses[0], 1e4)
    pred = 100.0 * (1.0 - 1.0 / (1.0 + (doses / ic50) ** hill[p]))
    assert np.allclose(ex[p][2][:, 1], np.clip(pred + sigma[p] * z[p, 2], 0, 100), rtol=1e-13, atol=1e-13)
    # responses are clipped to the measurable range; the censored share is what the recipe gives (Crumb: 783 of 2 585 = 30 %)
    for P, lo, hi in ((210, 0.20, 0.27), (1680, 0.21, 0.26)):
        e, _ = S.generate(P)
        y = np.concatenate([x[:, 1] for pair in e for x in pair])
        assert y.min() == 0.0 and y.max() == 100.0 and lo < S.censoring_rate(e) < hi, S.censoring_rate(e)


def test_packed_set_and_block_partition_over_8_ranks():
    from pyhillfit_amd import distributed as pd
    from pyhillfit_amd import doseresponse as dr
    from pyhillfit_amd import hierarchical as H
    from pyhillfit_amd import synthetic as S
    ex, _ = S.generate(1680)
    packed = dr.PackedPoints(S.single_level_pairs(ex))
    assert packed.num_pairs == 1680 and packed.stride <= 12 and int(packed.counts[:, 3].min()) == 12 == int(packed.counts[:, 3].max())
    assert all(H.group_key(e) == (3, 4) for e in ex)                      # the hierarchical sampler would run its gfx950 assembly build on them
    cnt = packed.counts
    costs = 525.0 + 28.0 * cnt[:, 0] + 115.0 * (cnt[:, 1] + cnt[:, 2])    # bench.py: instructions per iteration
    for world in (2, 4, 8):
        parts = pd.shard_blocks(costs, 4096 // 64, world)
        units = np.concatenate(parts)
        assert len(units) == 1680 * 64 and len({(int(a), int(b)) for a, b in units}) == 1680 * 64          # every (pair, block) once
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 64 and sum(sizes) == 107520
        loads = [float(costs[p[:, 0]].sum()) for p in parts]
        assert max(loads) / min(loads) < 1.001                               # 13 440 blocks per GPU at 8: balanced to a block


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from pyhillfit_amd import distributed as pd
    from pyhillfit_amd import doseresponse as dr
    from pyhillfit_amd import synthetic as S
    pd.init(backend="gloo")
    packed = dr.PackedPoints(S.single_level_pairs(S.generate(210)[0])) if rank == 0 else None      # bench.py: rank 0 generates
    got = pd.broadcast_packed_points(packed, "cpu", src=0)
    mine = pd.shard_blocks(525.0 + 28.0 * got.counts[:, 0] + 115.0 * (got.counts[:, 1] + got.counts[:, 2]), 1024 // 64, world)[rank]
    q.put((rank, got.num_pairs, float(got.ln_conc.sum()) + float(got.response.sum()) + float(got.weight.sum()), int(got.counts.sum()), mine.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_of_the_generated_set_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60); assert p.exitcode == 0
    assert res[0][1:4] == res[1][1:4] and res[0][1] == 210
    units = sorted(tuple(u) for r in res for u in r[4])
    assert units == [(p, b) for p in range(210) for b in range(16)]          # the two ranks' shares: every (pair, block) exactly once
