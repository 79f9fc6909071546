"""Diagnostic: a fused C4 step against the quantum of the one queue (the library's choice: ~205 iterations for 3 312 blocks x 2 000 iterations)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(R, "tools", "diag_c4_fused.py")).read()
src = src[:src.index('run("fused: every assembly group')]
for q in (0, 60, 75, 100, 300):
    os.environ["PHF_DIAG_QUANTUM"] = str(q)
    exec(compile(src.replace('quantum = int(os.environ.get("PHF_DIAG_QUANTUM", "0"))', 'quantum = %d' % q), "diag", "exec"))
    run("everything, quantum %s" % (q or "library's"), lambda h, f: True)
