"""The hand-allocated gfx950 code object's SOURCE is generated (tools/gen_hier_isa.py -> pyhillfit_amd/csrc/generated/): the committed files
must be what the generator emits, must assemble for gfx950 with the toolchain's assembler (no GPU needed), and the generator's own
bookkeeping must hold: 256 registers or fewer (two wavefronts per SIMD), no scratch, LDS that lets two workgroups share a CU.  CPU only."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

GEN = os.path.join(REPO, "tools", "gen_hier_isa.py")


def test_committed_assembly_is_what_the_generator_emits():
    out = subprocess.run([sys.executable, GEN, "--check"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


def test_generated_kernel_fits_two_wavefronts_per_simd_and_assembles(tmp_path):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    sys.path.insert(0, os.path.join(REPO, "tools", "isa"))
    import gen_hier_isa_main as G
    (lines, meta), info = G.main_kernel()
    assert meta["vgpr"] <= 256 and info["vgpr_high_water"] <= 256            # 512 registers per SIMD lane: two wavefronts
    assert meta["sgpr"] <= 108
    assert 2 * ((info["lds_bytes_per_workgroup"] + 511) // 512 * 512) <= 160 * 1024      # two 256-thread workgroups per CU
    text = "\n".join(lines)
    assert "scratch_" not in text and ".amdhsa_private_segment_fixed_size 0" in text       # nothing spills: there is no scratch at all
    it = info["count_iteration"]
    assert 1500 < it["valu_f64"] < 2000 and it["valu_int"] < 700 and it["lds"] < 350, it   # the iteration's static mix (phf_hier_model.h fixes the fp64 part)
    src = os.path.join(REPO, "pyhillfit_amd", "csrc", "generated", "phf_hier3_gfx950.s")
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm assembler here")
    obj = str(tmp_path / "k.o")
    r = subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", src, "-o", obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    names = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", obj], capture_output=True, text=True).stdout
    for k in ("phf_hier3_advance", "phf_hier3_advance_s222", "phf_hier3_advance_s554", "phf_sl3_advance", "phf_isa_unit_exp_fast", "phf_isa_unit_philox7"):
        assert k in names


@pytest.mark.parametrize("ne,shape", [(3, (2, 2, 2)), (3, (5, 5, 4)), (3, (3, 3, 3)), (3, (8, 8, 8)), (3, (7, 5, 1)), (4, (4, 4, 4, 1)), (5, (4, 4, 4, 1, 1)),
                                      (6, (4, 4, 4, 1, 1, 1))])
def test_generator_places_other_point_shapes_within_the_same_budget(ne, shape):
    """the target phase is emitted per point shape (a half's pairs, then its odd point; halves side by side where both have a pair): every
    shape of up to eight points per experiment fits the registers and the LDS of two workgroups per CU — the allocator raises otherwise; so do
    four, five and six experiments with their scratch tiers (six: streamed element by element; generated, measured, not shipped as a body)"""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    sys.path.insert(0, os.path.join(REPO, "tools", "isa"))
    import gen_hier_isa_main as G
    (lines, meta), info = G.main_kernel(ne, shape)
    assert info["vgpr_high_water"] <= 256 and 2 * info["lds_bytes_per_workgroup"] <= 160 * 1024
    assert "scratch_" not in "\n".join(lines)


def test_kernel_table_is_the_same_in_generator_header_and_host_package():
    """the (experiments, point shape) kernels exist in three places — the generator's list, the generated header's table (what the library
    looks a launch's shape up in) and pyhillfit_amd.hierarchical.ISA_SHAPES (what the host groups pairs by): one set of shape codes"""
    import re
    sys.path.insert(0, os.path.join(REPO, "tools"))
    sys.path.insert(0, os.path.join(REPO, "tools", "isa"))
    import gen_hier_isa_main as G
    from pyhillfit_amd import hierarchical as H
    gen = {(ne, G.shape_code(shape)) for ne, shape in G.HIER_KERNELS}
    assert len(gen) == len(G.HIER_KERNELS) and None not in {c for _, c in gen}
    for ne, shape in G.HIER_KERNELS:
        assert G.shape_code(shape) == H.shape_code(shape), shape
    with open(os.path.join(REPO, "pyhillfit_amd", "csrc", "generated", "phf_hier3_isa_layout.h")) as f:
        hdr = f.read()
    table = {(int(a), int(b)) for a, b in re.findall(r"^\s*\{(\d+), (\d+), \d+, \"phf_hier", hdr, flags=re.M)}
    assert table == gen == H.ISA_SHAPES
    assert ("#define PHF_ISA_FUSED_BODIES %d" % len(gen)) in hdr
    assert H.shape_code((4, 4, 4)) == 4 and H.shape_code((4, 4, 4, 1)) == 0x14 and H.shape_code((4, 4, 4, 1, 1)) == 0x40011444
    assert H.shape_code((17, 1)) == 0 and H.shape_code((1,)) == 0 and H.shape_code((3, 2, 1, 1, 1, 1, 1, 1)) == 0


def test_builder_inserts_waits_and_hazard_nops():
    """tools/isa/gfx950_asm.py: the s_waitcnt / s_nop insertion the generated code relies on (the assembler inserts none)"""
    sys.path.insert(0, os.path.join(REPO, "tools", "isa"))
    import gfx950_asm as A
    k = A.Kernel("t", first_sgpr=4)
    a, b, c = k.vd(3)
    addr = k.v1()
    k.ds_read(a, addr, 0)
    k.ds_read(b, addr, 8)
    k.fma(c, a, a, 1.0)                      # needs the FIRST read only: one may stay in flight
    k.rcp_est(c, b)                          # needs the second
    k.fma(a, c, c, c)                        # reads a transcendental's result: one wait state
    m = k.sd()
    k.cmp("lt", m, a, b)
    k.cnd64(c, 0.0, a, m)                    # reads an SGPR a VALU wrote: two wait states
    s = k.sd()
    k.s_load(s, A.Reg("s", 0, 2), 0)
    k.ds_read(b, addr, 16)
    k.fma(a, b, b, b)                        # a scalar load is in flight: only lgkmcnt(0) says anything
    k.finalize()
    text = [l.strip() for l in k.lines]
    i = text.index("v_fma_f64 v[4:5], v[0:1], v[0:1], 1.0")
    assert text[i - 1] == "s_waitcnt lgkmcnt(1)"
    j = text.index("v_rcp_f64_e32 v[4:5], v[2:3]")
    assert text[j - 1] == "s_waitcnt lgkmcnt(0)" and text[j + 1] == "s_nop 0"
    q = [t for t in text if t.startswith("v_cmp_lt_f64")][0]
    assert text[text.index(q) + 1] == "s_nop 1"
    assert text[-2] == "s_waitcnt lgkmcnt(0)"
    with pytest.raises(AssertionError):
        k.fma(a, A.Reg("s", 10, 2), b, A.Reg("s", 12, 2))      # two different SGPR operands: the constant-bus limit of VOP3
