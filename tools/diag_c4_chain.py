import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import runpy
src = open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "diag_c4_fused.py")).read()
src = src[:src.index('run("fused: every assembly group')]
exec(compile(src, "diag", "exec"))
run("fused: Ne = 3 + Ne = 5", lambda h, f: f and h.n_expts in (3, 5))
run("fused: Ne = 3 + Ne = 4", lambda h, f: f and h.n_expts in (3, 4))
run("fused: Ne = 3 + ONE Ne = 5 group (4+4+4+4+4: 1 pair)", lambda h, f: f and (h.n_expts == 3 or (h.n_expts == 5 and h.Q == 1 and h.points.packed.points_per_expt == 4)))
run("fused: Ne = 3", lambda h, f: f and h.n_expts == 3)
