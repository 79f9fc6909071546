"""Load the *reference* PyHillFit python files in memory (generator-side only).

TEST INFRASTRUCTURE — used only by tests/golden/make_golden.py, in the build
container where /root/reference exists.  Nothing here runs on the GPU box and
no reference text is ever written to disk: the Python-2 sources are converted
with the stdlib ``lib2to3`` in memory and exec'd into throw-away module objects.
"""
import ast
import os
import sys
import types
import warnings

REF_ROOT = os.environ.get("PHF_REFERENCE_ROOT", "/root/reference")
REF_PY = os.path.join(REF_ROOT, "python")


def _to_py3(src, name):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
        fixers = refactor.get_fixers_from_package("lib2to3.fixes")
        tool = refactor.RefactoringTool(fixers)
        if not src.endswith("\n"):
            src += "\n"
        return str(tool.refactor_string(src, name))


def load_doseresponse():
    """reference python/doseresponse.py as a live module (all functions unmodified)."""
    path = os.path.join(REF_PY, "doseresponse.py")
    with open(path) as f:
        src3 = _to_py3(f.read(), "doseresponse.py")
    mod = types.ModuleType("ref_doseresponse")
    mod.__dict__["sys"] = sys  # the reference forgets `import sys` (doseresponse.py:315)
    exec(compile(src3, path, "exec"), mod.__dict__)
    return mod


def lift_functions(filename, names, glb):
    """exec only the named top-level ``def``s of a reference script into ``glb``."""
    path = os.path.join(REF_PY, filename)
    with open(path) as f:
        src3 = _to_py3(f.read(), filename)
    tree = ast.parse(src3)
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in wanted}
    if missing:
        raise RuntimeError("reference functions not found: %s" % sorted(missing))
    code = compile(ast.Module(body=wanted, type_ignores=[]), path, "exec")
    exec(code, glb)
    return glb


class Py2Int(int):
    """int whose ``/`` floors, restoring Python-2 semantics for PyHillTemp.py:70-71,109."""

    def __truediv__(self, other):
        return Py2Int(int(self) // int(other))

    def __rtruediv__(self, other):
        return Py2Int(int(other) // int(self))


def lift_statements(filename, func_name, ranges, glb):
    """exec consecutive top-level statements of the BODY of one reference function, unmodified, in `glb`.

    ranges: list of (first, last) pairs; `first` / `last` are predicates on the Python-3 source text of a statement
    (ast.unparse) choosing the first and last statement of a run.  Used for the Metropolis loops that the reference inlines
    in run_single_level / run_hierarchical between its CMA-ES search and its plotting code (python/PyHillFit.py:748-751,
    787-864 and :431-511): the functions cannot be called (cma absent, matplotlib API drift), their loop statements can."""
    path = os.path.join(REF_PY, filename)
    with open(path) as f:
        src3 = _to_py3(f.read(), filename)
    tree = ast.parse(src3)
    func = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == func_name)
    texts = [ast.unparse(s) for s in func.body]
    picked = []
    for first, last in ranges:
        i0 = next(i for i, t in enumerate(texts) if first(t))
        i1 = next(i for i in range(i0, len(texts)) if last(texts[i]))
        picked.append((func.body[i0].lineno, func.body[i1].end_lineno))
        code = compile(ast.Module(body=func.body[i0:i1 + 1], type_ignores=[]), path, "exec")
        exec(code, glb)
    return picked
