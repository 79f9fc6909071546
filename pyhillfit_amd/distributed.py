"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The MH path shards embarrassingly — every (pair, temperature, chain) is an independent Markov chain
(python/PyHillFit.py:978-1003 maps pairs over a process pool; python/PyHillTemp.py:155-161 maps rungs) — so there
is NO collective inside the sampling loop.  Collectives are used only around it:
  before: rank 0 reads the data file and broadcasts the packed points (a few tens of KB: one latency-bound hop);
  after:  per-problem summaries are gathered to rank 0 (a few MB).
Philox streams are addressed by global (problem id, chain id), so results do not depend on the partition."""
import os

import numpy as np

# torch is imported inside the functions: the chain-file writer processes (chainio.WriterPool) re-import the CLI
# module that imports this one, and must stay GPU- and torch-free.


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for a single process)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def collective_device(device):
    """where a tensor must live to go through the current process group: the GPU for RCCL, host memory for gloo"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return device
    return "cpu"


def finalize():
    """leave the process group (every CLI does this in a `finally`, so that a rank that fails does not leave the others
    waiting in a collective for ever: their next collective then raises instead of hanging)"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def shard_problems(costs, world):
    """Partition problems over ranks, balancing cost (a problem costs ~ its number of points):
    longest-processing-time greedy, deterministic.  Returns a list of index arrays, one per rank."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world)
    parts = [[] for _ in range(world)]
    for q in order:
        r = int(np.argmin(load))
        parts[r].append(int(q)); load[r] += costs[q]
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


def shard_chains(num_chains, rank, world):
    """Contiguous block of chain ids for this rank: (first, count).  Used when there are fewer problems than GPUs
    (e.g. BASELINE config 2: one pair, 65 536 chains per GPU)."""
    base, rem = divmod(int(num_chains), world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def broadcast_packed_points(packed, device, src=0):
    """Rank `src` holds a doseresponse.PackedPoints; everybody returns an identical copy.
    (the 'scatter the dataset' step: 6 small tensors, one broadcast each)"""
    import torch
    import torch.distributed as dist
    from .doseresponse import PackedPoints
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return packed
    rank = dist.get_rank()
    dev = torch.device(device)
    shape = torch.zeros(2, dtype=torch.int64, device=dev)
    if rank == src:
        shape[0], shape[1] = packed.num_pairs, packed.stride
    dist.broadcast(shape, src)
    P, stride = int(shape[0]), int(shape[1])
    if rank == src:
        bufs = [torch.from_numpy(packed.ln_conc).to(dev), torch.from_numpy(packed.response).to(dev),
                torch.from_numpy(packed.weight).to(dev), torch.from_numpy(packed.counts).to(dev),
                torch.from_numpy(packed.pi_bit).to(dev), torch.from_numpy(packed.extra).to(dev)]
    else:
        f64 = dict(dtype=torch.float64, device=dev)
        bufs = [torch.empty((P, stride), **f64), torch.empty((P, stride), **f64), torch.empty((P, stride), **f64),
                torch.empty((P, 4), dtype=torch.int32, device=dev), torch.empty(P, **f64), torch.empty((P, 2), **f64)]
    for b in bufs:
        dist.broadcast(b, src)
    out = PackedPoints.__new__(PackedPoints)
    out.num_pairs, out.stride = P, stride
    out.ln_conc, out.response, out.weight, out.counts, out.pi_bit, out.extra = [b.cpu().numpy() for b in bufs]
    return out


def gather_rows(local, dst=0):
    """Gather per-problem rows [n_local, k] (n_local may differ per rank) to rank dst; returns list of arrays on dst.
    (the 'gather samples/summaries' step)"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local.cpu().numpy()]
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    nmax = int(max(int(s) for s in sizes))
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    outs = [torch.zeros_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, outs, dst=dst)
    if rank != dst:
        return None
    return [o[:int(s)].cpu().numpy() for o, s in zip(outs, sizes)]


def setup_data_file(path, src=0):
    """dr.setup(path) on every rank with ONE reader: rank `src` parses the file and broadcasts the table
    (a few tens of KB through RCCL/gloo) — the reference forks workers that inherit the parsed DataFrame
    (python/PyHillFit.py:61,997-1003)."""
    import torch.distributed as dist
    from . import doseresponse as dr
    if not dist.is_initialized() or dist.get_world_size() == 1:
        dr.setup(path)
        return
    box = [None]
    if dist.get_rank() == src:
        dr.setup(path)
        t = dr.table
        box[0] = (list(t.drug), list(t.channel), t.experiment.tolist(), t.dose.tolist(), t.response.tolist())
    dist.broadcast_object_list(box, src=src)
    if dist.get_rank() != src:
        dr.setup_from_table(path, dr.Table(*box[0]))
