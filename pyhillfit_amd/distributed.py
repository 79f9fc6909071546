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


def _gpu_count_from_driver_topology():
    """GPUs by the topology the kernel driver publishes (/sys/class/kfd: a node with SIMDs is a GPU), cut down by the visibility
    variables the runtime honours (ROCR_VISIBLE_DEVICES, then HIP_/CUDA_VISIBLE_DEVICES select from what ROCR left); None if the
    topology is not readable.  No HIP call — but also blind to visibility restricted by other means (device-node permissions)."""
    import glob
    try:
        gpus = 0
        for props in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            with open(props) as f:
                for line in f:
                    if line.startswith("simd_count"):
                        gpus += int(line.split()[1]) > 0
                        break
    except (OSError, ValueError):
        return None
    if gpus == 0:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            gpus = min(gpus, len([x for x in v.split(",") if x.strip() != ""]))
    return gpus


def visible_gpu_count():
    """GPUs this process (and the ranks it starts) can use, WITHOUT touching the HIP runtime in THIS process (ADVICE r03:
    torch.cuda.device_count() may initialise it when amdsmi is absent, and the process then holds it while its ranks run).
    First the driver's topology files (/sys/class/kfd: no HIP call, milliseconds — a `-c N` start costs 3 s in all); only where they
    are unreadable is the runtime's own answer asked of a short-lived CHILD interpreter (seconds: it imports torch); torch in this
    process is the last resort (tolerable only because ranks are started as a child process, never by an exec).  PHF_GPU_COUNT_FROM=
    child forces the child's answer (exact whatever restricts visibility, e.g. device-node permissions the topology does not show)."""
    import subprocess
    import sys
    if os.environ.get("PHF_GPU_COUNT_FROM") != "child":
        n = _gpu_count_from_driver_topology()
        if n is not None:
            return n
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=180)
        if out.returncode == 0:
            return int(out.stdout.strip().splitlines()[-1])
    except (OSError, ValueError, IndexError, subprocess.SubprocessError):
        pass
    import torch
    return torch.cuda.device_count()


def torchrun_command(n, tail):
    """`python -m torch.distributed.run` for n ranks on this node: --standalone lets the launcher pick a free rendezvous port itself
    (a port found by bind-and-close could be taken again before the ranks bind it); --local-addr 127.0.0.1 because the container's
    host name may not resolve."""
    import sys
    return [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
            "--nproc-per-node", str(int(n))] + list(tail)


def ranks_for_cores(num_cores):
    """How many ranks a command line's `-c/--num-cores N` (PyHillFit.py:40) / `-nc N` (PyHillTemp.py:25) starts: the reference
    sizes its process pool with that flag (python/PyHillFit.py:997-1003: min(N, cpu_count - 1) workers; PyHillTemp.py:155-159);
    here a worker is a GPU, so N ranks, at most one per visible GPU.  0 = do not start anything (N <= 1, or this process is
    already a rank of a torchrun world).  PHF_DIST_BACKEND=gloo lifts the GPU cap (rehearsal / CPU-only steps such as -bfo)."""
    if int(num_cores) <= 1 or "WORLD_SIZE" in os.environ:
        return 0
    if os.environ.get("PHF_DIST_BACKEND") == "gloo":
        return int(num_cores)
    n = min(int(num_cores), visible_gpu_count())
    return n if n > 1 else 0


def spawn_ranks(module, argv, n):
    """Start `python -m torch.distributed.run --nproc-per-node n -m <module> <argv>` as a CHILD process — this process has made no
    GPU call yet, and it is a child, never an exec — and return its exit code (as bench.py --gpus N does)."""
    import subprocess
    cmd = torchrun_command(n, ["-m", module] + list(argv))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    return subprocess.call(cmd, env=env)


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for a single process)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("PHF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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


def shard_blocks(costs, blocks_per_problem, world):
    """Partition the (problem, 64-chain block) UNITS of a batch over ranks, balancing cost: longest-processing-time greedy over
    the units (every block of a problem costs what the problem costs per chain), deterministic.  The reference's unit of
    distribution is a whole pair (python/PyHillFit.py:978-1003: pool.map_async(run, pairs)); with thousands of chains per pair a
    pair is many wavefronts, and splitting by blocks keeps every rank's share within one block of the others' — 13 440 blocks of
    the full Crumb set over 8 GPUs are 1 680 each — where whole pairs leave 26 or 27 pairs of unequal cost per rank.
    Returns, per rank, an int64 array [n_units][2] of (problem index, block index), sorted; a unit keeps its chains' Philox
    streams through phf_problems.chain_offset = 64 * block (sampler `chain_offsets`)."""
    import heapq
    costs = np.asarray(costs, dtype=np.float64)
    bpp = np.broadcast_to(np.asarray(blocks_per_problem, dtype=np.int64), costs.shape)
    order = np.argsort(-costs, kind="stable")
    heap = [(0.0, r) for r in range(world)]
    parts = [[] for _ in range(world)]
    for q in order:
        for b in range(int(bpp[q])):
            load, r = heapq.heappop(heap)
            parts[r].append((int(q), b))
            heapq.heappush(heap, (load + float(costs[q]), r))
    return [np.array(sorted(p), dtype=np.int64).reshape(-1, 2) for p in parts]


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
