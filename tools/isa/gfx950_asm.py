"""A small assembler-level code builder for gfx950 (CDNA4): the author places every instruction and owns every register.

Used by tools/gen_hier_isa.py to emit the hand-allocated hierarchical kernel (pyhillfit_amd/csrc/generated/*.s).  What it does:
  * registers: explicit pools of VGPRs / SGPRs (64-bit values in even-aligned pairs, as gfx90a+ requires), allocate / free by the
    author, high-water marks reported; nothing is ever spilled behind the author's back — running out is an error;
  * instructions: one method call = one instruction, operands checked (inline constants, one constant-bus read per VOP3,
    even-aligned tuples);
  * s_waitcnt: LDS / scalar-memory / vector-memory results are tracked per destination register and the counted wait is
    inserted right before the first instruction that touches the register (lgkmcnt(N) with N = younger LDS operations still
    allowed in flight; scalar loads return out of order, so anything behind one waits for lgkmcnt(0));
  * hazards: the software wait states gfx940/gfx950 need and the assembler does not insert (LLVM's GCNHazardRecognizer runs in the
    compiler, not in llvm-mc): a transcendental's result read by the next VALU (1), an SGPR / VCC written by a VALU and read by a
    VALU (2) or a vector-memory instruction (5), v_readfirstlane of a VGPR written by the previous VALU (1); s_nop as needed;
  * control flow: labels and scalar branches; at every label and branch the pending-load state is drained (s_waitcnt) and the
    hazard history is made worst-case, so that every path into a block is safe.

TEST INFRASTRUCTURE?  No: this is build-time tooling of the product (the generated .s is what ships)."""
import struct

INLINE_F64 = {0.5: "0.5", -0.5: "-0.5", 1.0: "1.0", -1.0: "-1.0", 2.0: "2.0", -2.0: "-2.0", 4.0: "4.0", -4.0: "-4.0"}


def f64_bits(x):
    return struct.unpack("<Q", struct.pack("<d", float(x)))[0]


class Reg(object):
    __slots__ = ("file", "idx", "n")

    def __init__(self, file, idx, n=1):
        self.file, self.idx, self.n = file, idx, n

    def __repr__(self):
        if self.file == "vcc":
            return "vcc"
        if self.file == "exec":
            return "exec"
        return "%s%d" % (self.file, self.idx) if self.n == 1 else "%s[%d:%d]" % (self.file, self.idx, self.idx + self.n - 1)

    def regs(self):
        if self.file == "vcc":
            return {("s", 106), ("s", 107)}
        if self.file == "exec":
            return {("s", 126), ("s", 127)}
        return {(self.file, self.idx + i) for i in range(self.n)}

    def lo(self):
        return Reg(self.file, self.idx, 1)

    def hi(self):
        assert self.n == 2
        return Reg(self.file, self.idx + 1, 1)

    def sub(self, i, n=2):
        """registers i .. i+n-1 of a tuple (a double out of a b128 load)"""
        assert i + n <= self.n
        return Reg(self.file, self.idx + i, n)


VCC = Reg("vcc", 106, 2)
EXEC = Reg("exec", 126, 2)


class Neg(object):
    def __init__(self, x):
        self.x = x


class Abs(object):
    def __init__(self, x):
        self.x = x


class Lit(object):
    """a 32-bit literal (VOP1 / VOP2 / VOPC / SOP encodings only)"""

    def __init__(self, v):
        self.v = v & 0xffffffff


def base(o):
    while isinstance(o, (Neg, Abs)):
        o = o.x
    return o


def fmt(o):
    if isinstance(o, Neg):
        return "-" + fmt(o.x)
    if isinstance(o, Abs):
        return "|" + fmt(o.x) + "|"
    if isinstance(o, Reg):
        return repr(o)
    if isinstance(o, Lit):
        return "0x%x" % o.v
    if isinstance(o, float):
        if o == 0.0:
            return "0"
        if o in INLINE_F64:
            return INLINE_F64[o]
        raise ValueError("%r is not an inline fp64 constant" % o)
    if isinstance(o, int):
        if -16 <= o <= 64:
            return str(o)
        raise ValueError("%r is not an inline integer constant" % o)
    if isinstance(o, str):
        return o
    raise TypeError(o)


def regs_of(o):
    o = base(o)
    return o.regs() if isinstance(o, Reg) else set()


TRANS = {"v_rcp_f64_e32", "v_rsq_f64_e32", "v_sqrt_f64_e32", "v_rcp_f32_e32", "v_exp_f32_e32", "v_log_f32_e32"}


class Pool(object):
    def __init__(self, file, lo, hi):
        self.file, self.lo, self.hi = file, lo, hi
        self.free = set(range(lo, hi))
        self.high = lo

    def alloc(self, n, align=None):
        align = align or (2 if n >= 2 else 1)
        for i in range(self.lo + (-self.lo) % align, self.hi - n + 1, align):
            if all((i + j) in self.free for j in range(n)):
                for j in range(n):
                    self.free.discard(i + j)
                self.high = max(self.high, i + n)
                return Reg(self.file, i, n)
        raise RuntimeError("out of %sGPRs (wanted %d)" % (self.file.upper(), n))

    def release(self, r):
        for j in range(r.n):
            assert self.lo <= r.idx + j < self.hi and (r.idx + j) not in self.free, "double free of %r" % r
            self.free.add(r.idx + j)

    def in_use(self):
        return (self.hi - self.lo) - len(self.free)


class Kernel(object):
    def __init__(self, name, num_vgpr=256, num_sgpr=100, first_sgpr=0, first_vgpr=0):
        self.name = name
        self.lines = []
        self.v = Pool("v", first_vgpr, num_vgpr)
        self.s = Pool("s", first_sgpr, num_sgpr)
        self.max_v_in_use = 0
        self.hist = []                      # recent instructions: dicts kind, vw, sw (written regs)
        self.lgkm = []                      # outstanding LDS / SMEM operations, oldest first: (kind, set of dst regs)
        self.vm = []                        # outstanding vector-memory operations: set of dst regs
        self.counts = {}
        self.labels = 0
        self.count_on = True
        self.prog = []                      # instruction records in program order; text, waits and hazards come from finalize()
        self._streams = None                # inside parallel(): the instruction lists being filled
        self._deferred = None               # inside parallel(): registers freed by the streams, returned to the pools at the end
        self._stream_free = self._stream_owned = self._stream_regs = None

    # ---- registers ----
    def _alloc(self, pool, n, align=None):
        """inside a parallel region a stream first re-uses what IT has freed (another stream's freed registers are off limits until the
        region ends: the merge interleaves the streams, so their lifetimes overlap whatever the order they were written in)"""
        if self._deferred is not None and self._streams:
            own = self._stream_free[len(self._streams) - 1]
            for i, r in enumerate(own):
                if r.file == pool.file and r.n == n and (align is None or r.idx % align == 0):
                    return own.pop(i)
            r = pool.alloc(n, align)
            self._stream_owned[len(self._streams) - 1].add(id(r))
            self._stream_regs[len(self._streams) - 1].append(r)
            return r
        return pool.alloc(n, align)

    def vd(self, n=1):
        """n fp64 VGPR pairs"""
        r = [self._alloc(self.v, 2) for _ in range(n)]
        self._note()
        return r[0] if n == 1 else r

    def v1(self):
        r = self._alloc(self.v, 1)
        self._note()
        return r

    def vq(self):
        r = self._alloc(self.v, 4, 2)
        self._note()
        return r

    def sd(self):
        return self._alloc(self.s, 2)

    def s1(self):
        return self._alloc(self.s, 1)

    def sx(self, n):
        return self._alloc(self.s, n, 4 if n >= 4 else 2)

    def free(self, *rs):
        for r in rs:
            if isinstance(r, (list, tuple)):
                self.free(*r)
            elif self._deferred is not None:
                if self._streams and id(r) in self._stream_owned[len(self._streams) - 1]:
                    self._stream_free[len(self._streams) - 1].append(r)      # allocated by this stream: this stream may have it again
                else:
                    self._deferred.append(r)
            elif r.file == "v":
                self.v.release(r)
            else:
                self.s.release(r)

    def _note(self):
        self.max_v_in_use = max(self.max_v_in_use, self.v.in_use())

    # ---- text ----
    def raw(self, text):
        self.lines.append("\t" + text)

    def _rec(self, r):
        (self.prog if self._streams is None else self._streams[-1]).append(r)

    def comment(self, text):
        self._rec({"kind": "comment", "text": text})

    # ---- side-by-side instruction streams ----
    def parallel(self):
        """with k.parallel() as par: ... par.stream() ... par.stream() ...: the instructions emitted after each stream() call form
        one stream; at the end the streams are merged round-robin (in proportion to their lengths), so that independent dependency
        chains advance side by side.  The author guarantees the streams are independent (no register written by one and touched by
        another, except registers freed... which are only recycled after the region); registers freed inside the region go back to
        the pools when it ends."""
        return _Parallel(self)

    def new_label(self, stem="L"):
        self.labels += 1
        return ".%s_%s_%d" % (self.name, stem, self.labels)

    # ---- waits ----
    def _wait_for(self, touched):
        # LDS / scalar memory
        need = None
        for i, (kind, regs) in enumerate(self.lgkm):
            if regs & touched:
                need = i
        if need is not None:
            younger = self.lgkm[need + 1:]
            if any(k == "smem" for k, _ in self.lgkm[:need + 1]) or any(k == "smem" for k, _ in younger):
                n = 0                        # scalar loads return out of order: only lgkmcnt(0) says anything
            else:
                n = len(younger)
            if n > 15:
                n = 15
            self.raw("s_waitcnt lgkmcnt(%d)" % n)
            self.lgkm = [] if n == 0 else self.lgkm[len(self.lgkm) - n:]
            self._hist_push({"kind": "wait", "vw": set(), "sw": set()})
        need = None
        for i, regs in enumerate(self.vm):
            if regs & touched:
                need = i
        if need is not None:
            n = min(len(self.vm) - need - 1, 63)
            self.raw("s_waitcnt vmcnt(%d)" % n)
            self.vm = self.vm[len(self.vm) - n:] if n else []
            self._hist_push({"kind": "wait", "vw": set(), "sw": set()})

    def drain(self, lgkm=True, vm=True):
        """wait for every outstanding load (stores need no wait)"""
        if lgkm and any(regs for _, regs in self.lgkm):
            self.raw("s_waitcnt lgkmcnt(0)")
            self.lgkm = []
        if vm and any(regs for regs in self.vm):
            self.raw("s_waitcnt vmcnt(0)")
            self.vm = []

    # ---- hazards ----
    def _hist_push(self, e):
        self.hist.append(e)
        if len(self.hist) > 6:
            self.hist.pop(0)

    def _nops_needed(self, kind, vr, sr, opc):
        need = 0
        n = len(self.hist)
        for back, h in enumerate(reversed(self.hist)):      # back = instructions between h and the new one
            if h["kind"] == "unknown":
                # worst case at a join: a VALU wrote every SGPR, a transcendental every VGPR
                if kind in ("valu", "trans") and (sr or vr):
                    need = max(need, (2 if sr else 1) - back)
                if kind == "vmem" and sr:
                    need = max(need, 5 - back)
                if opc.startswith("v_readfirstlane") or opc.startswith("v_readlane"):
                    need = max(need, 1 - back)
                continue
            if h["kind"] == "trans" and kind in ("valu", "trans") and (h["vw"] & vr):
                need = max(need, 1 - back)
            if h["kind"] in ("valu", "trans") and h["sw"]:
                if kind in ("valu", "trans") and (h["sw"] & sr):
                    need = max(need, 2 - back)
                if kind == "vmem" and (h["sw"] & sr):
                    need = max(need, 5 - back)
            if h["kind"] in ("valu", "trans") and (opc.startswith("v_readfirstlane") or opc.startswith("v_readlane")) and (h["vw"] & vr):
                need = max(need, 1 - back)
        return max(need, 0)

    # ---- the one emitter ----
    def emit(self, opc, dsts, srcs, kind, suffix="", count=None, mem=None):
        dsts = [d for d in dsts if d is not None]
        self._rec({"kind": kind, "opc": opc, "dsts": list(dsts), "srcs": list(srcs), "suffix": suffix, "count": count or kind, "mem": mem})
        return dsts[0] if dsts else None

    def _emit_now(self, rec):
        opc, dsts, srcs, kind, suffix, count = rec["opc"], rec["dsts"], rec["srcs"], rec["kind"], rec["suffix"], rec["count"]
        vr, sr, vw, sw = set(), set(), set(), set()
        for o in srcs:
            for f, i in regs_of(o):
                (vr if f == "v" else sr).add((f, i))
        for o in dsts:
            for f, i in regs_of(o):
                (vw if f == "v" else sw).add((f, i))
        self._wait_for(vr | sr | vw | sw)
        nops = self._nops_needed(kind, vr, sr, opc)
        if nops:
            self.raw("s_nop %d" % (nops - 1))
            for _ in range(nops):
                self._hist_push({"kind": "nop", "vw": set(), "sw": set()})
        ops = ", ".join(fmt(o) for o in list(dsts) + list(srcs))
        self.raw("%s %s%s" % (opc, ops, (" " + suffix) if suffix else ""))
        self._hist_push({"kind": kind, "vw": vw, "sw": sw})
        self.counts[count] = self.counts.get(count, 0) + 1
        mem = rec["mem"]
        if mem == "lds":
            self.lgkm.append(("lds", set().union(*[d.regs() for d in dsts]) if dsts else set()))
        elif mem == "smem":
            self.lgkm.append(("smem", set().union(*[d.regs() for d in dsts])))
        elif mem == "vm":
            self.vm.append(set().union(*[d.regs() for d in dsts]) if dsts else set())

    def _bus_check(self, srcs):
        """VOP3 on gfx9: one constant-bus read (the same SGPR pair may be named more than once)"""
        seen = set()
        for o in srcs:
            b = base(o)
            if isinstance(b, Reg) and b.file in ("s", "vcc", "exec"):
                seen.add((b.file, b.idx))
            if isinstance(b, Lit):
                seen.add(("lit", b.v))
        assert len(seen) <= 1, "constant-bus limit: %s" % [fmt(o) for o in srcs]

    # ---- fp64 VALU ----
    def _valu3(self, opc, d, srcs, count="valu_f64"):
        self._bus_check(srcs)
        for o in [d] + list(srcs):
            b = base(o)
            if isinstance(b, Reg) and b.n >= 2 and b.file in ("v", "s"):
                assert b.idx % 2 == 0, "unaligned 64-bit operand %r" % b
        return self.emit(opc, [d], srcs, "valu", count=count)

    def fma(self, d, a, b, c):
        return self._valu3("v_fma_f64", d, [a, b, c])

    def mul(self, d, a, b):
        return self._valu3("v_mul_f64", d, [a, b])

    def add(self, d, a, b):
        return self._valu3("v_add_f64", d, [a, b])

    def sub(self, d, a, b):
        return self._valu3("v_add_f64", d, [a, Neg(b)])

    def fmin(self, d, a, b):
        return self._valu3("v_min_f64", d, [a, b])

    def fmax(self, d, a, b):
        return self._valu3("v_max_f64", d, [a, b])

    def ldexp(self, d, a, e):
        return self._valu3("v_ldexp_f64", d, [a, e])

    def rcp_est(self, d, a):
        return self.emit("v_rcp_f64_e32", [d], [a], "trans", count="valu_trans")

    def rsq_est(self, d, a):
        return self.emit("v_rsq_f64_e32", [d], [a], "trans", count="valu_trans")

    def cvt_f64_i32(self, d, a):
        return self.emit("v_cvt_f64_i32_e32", [d], [a], "valu", count="valu_f64")

    def cvt_f64_u32(self, d, a):
        return self.emit("v_cvt_f64_u32_e32", [d], [a], "valu", count="valu_f64")

    def cmp(self, rel, mask, a, b):
        """mask (an SGPR pair or VCC) <- a rel b per lane, fp64; rel in lt gt le ge eq lg neq nlt ..."""
        self._bus_check([a, b])
        return self.emit("v_cmp_%s_f64_e64" % rel, [mask], [a, b], "valu", count="valu_f64")

    def cmp_lit(self, rel, hi_word, v):
        """VCC <- C rel v per lane, C the fp64 constant whose HIGH word is hi_word and whose low word is zero (VOPC e32: a 32-bit
        literal stands for the high half of a 64-bit operand)"""
        return self.emit("v_cmp_%s_f64_e32" % rel, [VCC], [Lit(hi_word), v], "valu", count="valu_f64")

    def cmp_u32(self, rel, mask, a, b):
        self._bus_check([a, b])
        return self.emit("v_cmp_%s_u32_e64" % rel, [mask], [a, b], "valu", count="valu_int")

    def cmp_i32(self, rel, mask, a, b):
        self._bus_check([a, b])
        return self.emit("v_cmp_%s_i32_e64" % rel, [mask], [a, b], "valu", count="valu_int")

    def cnd32(self, d, f, t, mask):
        """d = mask ? t : f (32 bits)"""
        srcs = [f, t, mask]
        n = {(base(o).file, base(o).idx) for o in (f, t) if isinstance(base(o), Reg) and base(o).file == "s"}
        assert not n, "v_cndmask: the mask is the constant-bus read"
        return self.emit("v_cndmask_b32_e64", [d], srcs, "valu", count="valu_int")

    def cnd32_vcc(self, d, f, t):
        """d = VCC ? t : f; f: a VGPR or an inline constant, t: a VGPR.  VOP3 encoding: 64 v_cndmask_b32_e32 in a row measure ~20 cycles each
        against 4 in this form (profiles/r05/gfx950_instruction_costs.txt); in the kernels the two encodings time the same"""
        assert isinstance(t, Reg) and t.file == "v"
        assert not isinstance(f, Lit) and not (isinstance(f, Reg) and f.file == "s"), "VCC is the constant-bus read; no VOP3 literals"
        return self.emit("v_cndmask_b32_e64", [d], [f, t, VCC], "valu", count="valu_int")

    def cnd64(self, d, f, t, mask):
        """d = mask ? t : f (a double: two v_cndmask_b32); f / t: VGPR pairs, or a python float whose halves are inline constants"""
        def half(x, hi):
            if isinstance(x, Reg):
                return x.hi() if hi else x.lo()
            bits = f64_bits(x)
            w = (bits >> 32) if hi else (bits & 0xffffffff)
            if w == 0:
                return 0
            if hi and x in INLINE_F64:      # fp32 inline constants do not share the fp64 ones' high words: only zero works
                raise ValueError("cnd64 with constant %r: put its high word in a VGPR" % x)
            raise ValueError("cnd64 with constant %r" % x)
        self.cnd32(d.lo(), half(f, 0), half(t, 0), mask)
        self.cnd32(d.hi(), half(f, 1), half(t, 1), mask)
        return d

    # ---- 32-bit VALU ----
    def vop(self, opc, d, *srcs, **kw):
        if not opc.endswith("_e32"):
            self._bus_check(srcs)
        return self.emit(opc, [d], list(srcs), "valu", suffix=kw.get("suffix", ""), count="valu_int")

    def mov32(self, d, a):
        return self.vop("v_mov_b32_e32", d, a)

    def mov64(self, d, a):
        if isinstance(a, Reg):
            return self.emit("v_mov_b64_e32", [d], [a], "valu", count="valu_int")
        bits = f64_bits(a)
        self.mov32(d.lo(), Lit(bits & 0xffffffff) if (bits & 0xffffffff) else 0)
        self.mov32(d.hi(), Lit(bits >> 32) if (bits >> 32) else 0)
        return d

    def readfirstlane(self, sdst, v):
        return self.emit("v_readfirstlane_b32", [sdst], [v], "valu", count="valu_int")

    def mad_u64_u32(self, d, a, b):
        """d (64 bits) = a * b (32 x 32); the carry-out goes to VCC (not used)"""
        self._bus_check([a, b])
        return self.emit("v_mad_u64_u32", [d, VCC], [a, b, 0], "valu", count="valu_int")

    def xor3(self, d, a, b, c):
        self._bus_check([a, b, c])
        return self.emit("v_bitop3_b32", [d], [a, b, c], "valu", suffix="bitop3:0x96", count="valu_int")

    # ---- LDS ----
    def ds_read(self, d, addr, offset=0):
        opc = {2: "ds_read_b64", 4: "ds_read_b128", 1: "ds_read_b32"}[d.n]
        assert 0 <= offset < 65536
        if d.n == 4:
            assert offset % 16 == 0
        self.emit(opc, [d], [addr], "lds", suffix="offset:%d" % offset if offset else "", mem="lds")
        return d

    def ds_write(self, addr, data, offset=0):
        opc = {2: "ds_write_b64", 4: "ds_write_b128", 1: "ds_write_b32"}[data.n]
        assert 0 <= offset < 65536
        self.emit(opc, [], [addr, data], "lds", suffix="offset:%d" % offset if offset else "", mem="lds")

    # ---- global memory (saddr form: 64-bit SGPR base + 32-bit VGPR byte offset + immediate) ----
    def gload(self, d, voff, sbase, offset=0):
        opc = {2: "global_load_dwordx2", 4: "global_load_dwordx4", 1: "global_load_dword"}[d.n]
        assert -4096 <= offset < 4096
        self.emit(opc, [d], [voff, sbase], "vmem", suffix="offset:%d" % offset if offset else "", mem="vm")
        return d

    def gstore(self, voff, data, sbase, offset=0, scope=""):
        """scope: "" (wavefront: the line may stay dirty in L2) or "sc0 sc1" (system: written through to memory)"""
        opc = {2: "global_store_dwordx2", 4: "global_store_dwordx4", 1: "global_store_dword"}[data.n]
        assert -4096 <= offset < 4096
        self.emit(opc, [], [voff, data, sbase], "vmem", suffix=" ".join(x for x in ("offset:%d" % offset if offset else "", scope) if x), mem="vm")

    def gatomic_add_f64(self, voff, data, sbase, offset=0, scope=""):
        """scope: "" (agent: performed in this XCD's L2) or "sc1" (system: performed at the memory side)"""
        self.emit("global_atomic_add_f64", [], [voff, data, sbase], "vmem", suffix=" ".join(x for x in ("offset:%d" % offset if offset else "", scope) if x), mem="vm")

    # ---- scalar ----
    def sop(self, opc, d, *srcs):
        return self.emit(opc, [d] if d is not None else [], list(srcs), "salu")

    def s_load(self, d, sbase, offset=0):
        """offset: an immediate (bytes) or an SGPR holding bytes"""
        opc = {1: "s_load_dword", 2: "s_load_dwordx2", 4: "s_load_dwordx4", 8: "s_load_dwordx8", 16: "s_load_dwordx16"}[d.n]
        off = offset if isinstance(offset, Reg) else ("0x%x" % offset)
        self.emit(opc, [d], [sbase, off], "smem", mem="smem")
        return d

    def raw_rec(self, text):
        """an instruction the builder knows nothing about (no register operands: s_sleep, buffer_inv, buffer_wbl2, s_waitcnt ...)"""
        self._rec({"kind": "rawinst", "text": text})

    def s_waitcnt_all(self):
        self._rec({"kind": "waitall"})

    # ---- control flow ----
    def label(self, name, drain=True):
        """drain=False: loads in flight stay in flight across the label — only where every path into the label has issued the same
        loads or none that the code behind it touches without a wait of its own (the tracker keeps its queue: a load issued on one
        path only is waited for on both, which is safe: s_waitcnt on a counter that is already low does not block)"""
        assert self._streams is None
        self._rec({"kind": "label", "name": name, "drain": drain})

    def branch(self, opc, target):
        """s_branch / s_cbranch_scc0 / scc1 / vccz / vccnz / execz / execnz"""
        assert self._streams is None
        self._rec({"kind": "branch", "opc": opc, "target": target})

    def long_jump(self, target, tmp):
        """an unconditional jump beyond s_branch's +-128 KB: s_getpc_b64 + the label's distance + s_setpc_b64 through the SGPR pair `tmp`"""
        assert self._streams is None and tmp.file == "s" and tmp.n == 2
        self.labels += 1
        self._rec({"kind": "longjump", "target": target, "tmp": tmp, "here": ".%s_pc_%d" % (self.name, self.labels)})

    def barrier(self):
        self._rec({"kind": "barrier"})

    def endpgm(self):
        self._rec({"kind": "endpgm"})

    def count_marker(self, name):
        """instruction counts (self.counts) are snapshotted under `name` when the final pass reaches this point"""
        self._rec({"kind": "marker", "name": name})
        # registers in use at most since the previous marker, and now (generation time: that is when they are allocated)
        self.phase_vgprs = getattr(self, "phase_vgprs", [])
        self.phase_vgprs.append((name, self.max_v_in_use + self.v.lo, self.v.in_use() + self.v.lo))
        self.max_v_in_use = self.v.in_use()

    def finalize(self):
        """program order is final: insert s_waitcnt / s_nop and produce the text"""
        assert self._streams is None
        self.snapshots = {}
        for r in self.prog:
            kd = r["kind"]
            if kd == "comment":
                self.lines.append("\t; " + r["text"])
            elif kd == "label":
                if r.get("drain", True):
                    self.drain()
                self.lines.append(r["name"] + ":")
                self.hist = [{"kind": "unknown", "vw": set(), "sw": set()}]
            elif kd == "branch":
                self.drain()
                self.raw("%s %s" % (r["opc"], r["target"]))
                self._hist_push({"kind": "salu", "vw": set(), "sw": set()})
                self.counts["branch"] = self.counts.get("branch", 0) + 1
            elif kd == "longjump":
                self.drain()
                t = r["tmp"]
                self.raw("s_getpc_b64 %s" % fmt(t))
                self.lines.append(r["here"] + ":")
                self.raw("s_add_u32 %s, %s, (%s-%s)&0xffffffff" % (fmt(t.lo()), fmt(t.lo()), r["target"], r["here"]))
                self.raw("s_addc_u32 %s, %s, (%s-%s)>>32" % (fmt(t.hi()), fmt(t.hi()), r["target"], r["here"]))
                self.raw("s_setpc_b64 %s" % fmt(t))
                self._hist_push({"kind": "salu", "vw": set(), "sw": set()})
                self.counts["branch"] = self.counts.get("branch", 0) + 1
            elif kd == "barrier":
                self.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
                self.lgkm, self.vm = [], []
                self.raw("s_barrier")
            elif kd == "waitall":
                self.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
                self.lgkm, self.vm = [], []
            elif kd == "endpgm":
                self.raw("s_endpgm")
            elif kd == "rawinst":
                self.raw(r["text"])
                self._hist_push({"kind": "salu", "vw": set(), "sw": set()})
            elif kd == "marker":
                self.snapshots[r["name"]] = dict(self.counts)
            else:
                self._emit_now(r)

    # ---- the code object text ----
    def finish(self, lds_bytes, kernarg_bytes, sgprs=None):
        self.finalize()
        nv = self.v.high
        ns = sgprs or self.s.high
        head = [
            "\t.text",
            "\t.protected\t%s" % self.name,
            "\t.globl\t%s" % self.name,
            "\t.p2align\t8",
            "\t.type\t%s,@function" % self.name,
            "%s:" % self.name,
        ]
        tail = [
            "\t.section\t.rodata,\"a\",@progbits",
            "\t.p2align\t6, 0x0",
            "\t.amdhsa_kernel %s" % self.name,
            "\t\t.amdhsa_group_segment_fixed_size %d" % lds_bytes,
            "\t\t.amdhsa_private_segment_fixed_size 0",
            "\t\t.amdhsa_kernarg_size %d" % kernarg_bytes,
            "\t\t.amdhsa_user_sgpr_count 2",
            "\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1",
            "\t\t.amdhsa_system_sgpr_workgroup_id_x 1",
            "\t\t.amdhsa_system_sgpr_workgroup_id_y 0",
            "\t\t.amdhsa_system_sgpr_workgroup_id_z 0",
            "\t\t.amdhsa_system_vgpr_workitem_id 0",
            "\t\t.amdhsa_next_free_vgpr %d" % nv,
            "\t\t.amdhsa_next_free_sgpr %d" % ns,
            "\t\t.amdhsa_accum_offset %d" % (((nv + 3) // 4) * 4),
            "\t\t.amdhsa_reserve_vcc 1",
            "\t\t.amdhsa_float_round_mode_32 0",
            "\t\t.amdhsa_float_round_mode_16_64 0",
            "\t\t.amdhsa_float_denorm_mode_32 3",
            "\t\t.amdhsa_float_denorm_mode_16_64 3",
            "\t\t.amdhsa_dx10_clamp 1",
            "\t\t.amdhsa_ieee_mode 1",
            "\t.end_amdhsa_kernel",
            "\t.text",
            ".L%s_end:" % self.name,
            "\t.size\t%s, .L%s_end-%s" % (self.name, self.name, self.name),
        ]
        meta = {"name": self.name, "lds": lds_bytes, "kernarg": kernarg_bytes, "sgpr": ns + 6, "vgpr": nv}
        return head + self.lines + tail, meta


class _Parallel(object):
    def __init__(self, k):
        self.k = k

    def __enter__(self):
        assert self.k._streams is None, "parallel regions do not nest"
        self.k._streams = []
        self.k._deferred = []
        self.k._stream_free, self.k._stream_owned, self.k._stream_regs = [], [], []
        return self

    def stream(self):
        self.k._streams.append([])
        self.k._stream_free.append([])
        self.k._stream_owned.append(set())
        self.k._stream_regs.append([])

    def __exit__(self, et, ev, tb):
        k = self.k
        streams, deferred = k._streams, k._deferred
        for own in k._stream_free:                   # what the streams freed of their own and did not take again goes back now
            deferred.extend(own)
        k._streams, k._deferred = None, None
        k._stream_free = k._stream_owned = k._stream_regs = None
        if et is not None:
            return False
        real = [[r for r in s if r["kind"] != "comment"] for s in streams]
        total = sum(len(s) for s in real)
        pos = [0] * len(real)
        # proportional round-robin: always advance the stream that is furthest behind its share
        for _ in range(total):
            best, bi = None, -1
            for i, s in enumerate(real):
                if pos[i] < len(s):
                    frac = pos[i] / float(len(s))
                    if best is None or frac < best:
                        best, bi = frac, i
            k.prog.append(real[bi][pos[bi]])
            pos[bi] += 1
        k.free(*deferred)
        return False


def module_text(kernels):
    """kernels: list of (lines, meta) from Kernel.finish"""
    out = ["\t.amdgcn_target \"amdgcn-amd-amdhsa--gfx950\"", "\t.amdhsa_code_object_version 6"]
    for lines, _ in kernels:
        out += lines
    out += ["\t.text", "\t.p2alignl 6, 3212836864", "\t.fill 256, 4, 3212836864"]       # s_code_end padding behind the last kernel (prefetch)
    out += ["\t.amdgpu_metadata", "---", "amdhsa.kernels:"]
    for _, m in kernels:
        out += [
            "  - .agpr_count:     0",
            "    .args:",
            "      - .offset:         0",
            "        .size:           %d" % m["kernarg"],
            "        .value_kind:     by_value",
            "    .group_segment_fixed_size: %d" % m["lds"],
            "    .kernarg_segment_align: 8",
            "    .kernarg_segment_size: %d" % m["kernarg"],
            "    .max_flat_workgroup_size: 256",
            "    .name:           %s" % m["name"],
            "    .private_segment_fixed_size: 0",
            "    .sgpr_count:     %d" % m["sgpr"],
            "    .sgpr_spill_count: 0",
            "    .symbol:         %s.kd" % m["name"],
            "    .uniform_work_group_size: 1",
            "    .uses_dynamic_stack: false",
            "    .vgpr_count:     %d" % m["vgpr"],
            "    .vgpr_spill_count: 0",
            "    .wavefront_size: 64",
        ]
    out += ["amdhsa.target:   amdgcn-amd-amdhsa--gfx950", "amdhsa.version:", "  - 1", "  - 2", "...", "", "\t.end_amdgpu_metadata", ""]
    return "\n".join(out)
