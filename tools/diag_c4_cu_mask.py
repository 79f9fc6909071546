"""C4 with the Ne groups' streams confined to disjoint sets of CUs (hipExtStreamCreateWithCUMask): the assembly kernel's 256-thread
workgroups need whole CUs (half of all four SIMDs + 81 KB of LDS), a hipcc wavefront needs a whole SIMD — side by side on the same CUs
they fragment each other (profiles/r05/c4_timeline.txt: every kernel stretched 1.5..2.5x, 81 % of the SIMD time used while they mix).
    python tools/diag_c4_cu_mask.py [CUs for the assembly kernel, ...]        (0 = no masks: bench.py's own streams)
Produced profiles/r05/c4_cu_partition.txt."""
import ctypes
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from pyhillfit_amd import doseresponse as dr

hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))     # the runtime torch has loaded
TOTAL_CUS = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(lo, hi):
    """a stream whose kernels run on CUs lo .. hi - 1 of the mask's numbering"""
    words = (TOTAL_CUS + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in range(lo, hi):
        mask[cu // 32] |= 1 << (cu % 32)
    s = ctypes.c_void_p()
    e = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), mask)
    if e:
        raise RuntimeError("hipExtStreamCreateWithCUMask: hip error %d" % e)
    return torch.cuda.ExternalStream(s.value, device="cuda:0")


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
    names = [(d, c) for d in dr.drugs for c in dr.channels]
    ctx = {"torch": torch, "dist": None, "dev": dev, "world": 1, "backend": None}
    print("device reports %d CUs" % TOTAL_CUS, flush=True)
    for n_asm in [int(x) for x in (sys.argv[1:] or ["0", "136", "0", "136"])]:
        b = bench.HierarchicalBatch(dr, names, 1024, 5, 0, dev, torch)
        if n_asm:
            streams = []
            for h in b.samplers:
                is_asm = h.points.packed.n_expts == 3 and h.points.packed.points_per_expt == 4
                streams.append(masked_stream(0, n_asm) if is_asm else masked_stream(n_asm, TOTAL_CUS))
            b.streams = streams
        dt, kernel_ms, _ = bench.timed_region(b, 2000, 10, 3, ctx)
        print("assembly kernel on %3d CUs, the hipcc groups on the other %3d: %.2f ms per step of 2 000 iterations (wall %.2f)"
              % (n_asm, TOTAL_CUS - n_asm if n_asm else TOTAL_CUS, kernel_ms, dt / 10 * 1e3), flush=True)
        del b
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
