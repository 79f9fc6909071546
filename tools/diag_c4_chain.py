"""Diagnostic: is a fused C4 step bound by work or by its longest serial chain (a block's 2 000 iterations)?  The fused grid with subsets of the groups.
-> profiles/r05/c4_fused_launch.txt"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(R, "tools", "diag_c4_fused.py")).read()
src = src[:src.index('run("fused: every assembly group')]
exec(compile(src, "diag", "exec"))
run("fused: Ne = 6 only (48 blocks: lone wavefronts)", lambda h, f: f and h.n_expts == 6)
run("fused: Ne = 5 only (192 blocks)", lambda h, f: f and h.n_expts == 5)
run("fused: Ne = 3 only", lambda h, f: f and h.n_expts == 3)
run("fused: Ne = 3 + Ne = 6", lambda h, f: f and h.n_expts in (3, 6))
run("fused: Ne = 3 + Ne = 5", lambda h, f: f and h.n_expts in (3, 5))
run("fused: Ne = 3 + Ne = 4", lambda h, f: f and h.n_expts in (3, 4))
run("fused: everything", lambda h, f: True)
