"""Diagnostic: which sampler of a hierarchical command-line run reports a drained queue, and what its workspace holds."""
import gc, os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from pyhillfit_amd import PyHillFit, doseresponse as dr, hierarchical as H
dr.setup(os.path.join(R, "data", "crumb_dataset.json"))
with tempfile.TemporaryDirectory() as tmp:
    csv = os.path.join(tmp, "crumb_data.csv"); dr.table.to_csv(csv)
    if os.environ.get("PHF_DIAG_ONLY444") == "1":          # only the pairs of 4 + 4 + 4 points: the queued assembly kernel alone on the chip
        load = dr.load_crumb_data

        def only444(d, c):
            r = load(d, c)
            also = os.environ.get("PHF_DIAG_ALSO", "")          # e.g. "2,2,2" or "ne4": one more group beside it
            sh = [len(e) for e in r[2]]
            if sh != [4, 4, 4] and ",".join(str(n) for n in sh) != also and "ne%d" % len(sh) != also:
                raise ValueError("skipped")
            return r
        dr.load_crumb_data = only444
    if os.environ.get("PHF_DIAG_EAGER") == "1":             # allocate every sampler's workspace when the sampler is built (default stream)
        init0 = H.HierarchicalSampler.init

        def init(o, *a, **kw):
            o.queue
            return init0(o, *a, **kw)
        H.HierarchicalSampler.init = init
    if os.environ.get("PHF_DIAG_222_HIPCC") == "1":         # the 2 + 2 + 2 group on the hipcc kernel
        H.GROUPED_SHAPES = {k_ for k_ in H.GROUPED_SHAPES if k_ != (3, 2)}
    if os.environ.get("PHF_DIAG_SIDE_BY_SIDE", "1") == "1":     # undo run_hierarchical's avoidance: assembly launches side by side, as before
        hint0 = H.HierarchicalSampler.set_kernel_hint

        def set_hint(o, lanes=0, wps=0, isa=None):
            return hint0(o, lanes=lanes, wps=wps, isa=None if isa is False and os.environ.get("PHF_DIAG_222_HIPCC") != "1" else isa)
        H.HierarchicalSampler.set_kernel_hint = set_hint
    orig = H.HierarchicalSampler.check_queue

    def check(o):
        q = o._queue
        if q is not None:
            print("   queue at", hex(q.data_ptr()), "bytes", q.numel() * 4, "end", hex(q.data_ptr() + q.numel() * 4), "state at", hex(o.state.data_ptr()),
                  "moments at", None if o.moments is None else hex(o.moments.data_ptr()), flush=True)
        print(o.n_expts, o.points.packed.points_per_expt, "Q", o.Q, "C", o.C, "blocks", o.nblocks, "t", o.t, "hint", o.prob.kernel_hint, "last kernel", H.last_kernel(),
              "queue", None if q is None else (q.numel(), int(q[0].item()), int(q[1 + o.nblocks].item()), q[1:5].tolist(), q[o.nblocks - 2:o.nblocks + 8].tolist()), flush=True)
        return orig(o)
    H.HierarchicalSampler.check_queue = check
    adv = H.HierarchicalSampler.advance

    def advance(o, n, out=None, save=True):
        if os.environ.get("PHF_DIAG_222_HIPCC") == "1" and o.points.packed.points_per_expt == 2:
            o.set_kernel_hint(lanes=1, isa=False)
        r = adv(o, n, out=out, save=save)
        if o.n_expts == 3 and o.points.packed.points_per_expt == 4 and os.environ.get("PHF_DIAG_EACH") == "1":
            torch.cuda.synchronize()
            q = o._queue
            print("after advance to t =", o.t, "kernel", H.last_kernel(), "counter", int(q[0].item()), "fault", int(q[1 + o.nblocks].item()), "progress min/max",
                  int(q[1:1 + o.nblocks].min().item()), int(q[1:1 + o.nblocks].max().item()), flush=True)
        return r
    H.HierarchicalSampler.advance = advance
    try:
        PyHillFit.main(["--data-file", csv, "-m", "2", "-a", "--hierarchical", "--num-chains", "1024", "-i", sys.argv[1] if len(sys.argv) > 1 else "40000",
                        "--output-root", os.path.join(tmp, "out"), "--fused-launch", os.environ.get("PHF_DIAG_FUSED", "off")] + sys.argv[2:])
        print("no error")
    except Exception as e:
        print("ERROR", str(e)[:80])
    for o in gc.get_objects():
        if isinstance(o, H.HierarchicalSampler):
            q = o._queue
            print(o.n_expts, o.points.packed.points_per_expt, "Q", o.Q, "C", o.C, "blocks", o.nblocks, "t", o.t, "hint", o.prob.kernel_hint,
                  "queue", None if q is None else (q.numel(), int(q[0].item()), int(q[1 + o.nblocks].item()), q[1:5].tolist()), flush=True)
