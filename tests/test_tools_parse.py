"""CPU: every script under tools/ and oracle/ (the GPU probes, the rocprof summarisers, the golden generators) and the root entry
points compile, and every name a probe uses at module level is defined before its use -- they run on the GPU box or next to
/root/reference only, so nothing else in the CPU suite would notice a broken one before a GPU call is spent on it."""
import ast
import builtins
import glob
import os
import py_compile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "oracle", "*.py")) +
                 [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")])


@pytest.mark.parametrize("path", SCRIPTS, ids=[os.path.relpath(p, ROOT) for p in SCRIPTS])
def test_script_compiles_and_module_level_names_are_bound(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "x.pyc"), doraise=True)
    tree = ast.parse(open(path).read())
    bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}

    def bind(node):
        for t in ast.walk(node):
            if isinstance(t, ast.Name) and isinstance(t.ctx, ast.Store):
                bound.add(t.id)

    for node in tree.body:                      # top-level statements in order: loads must follow their bindings
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            bound.update((a.asname or a.name).split(".")[0] for a in node.names)
            continue
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            bound.add(node.name)                # bodies run later: not checked
            continue
        bind(node)                              # (loop variables, with-targets, assignments of this statement)
        inner = set()
        for t in ast.walk(node):                # names introduced inside nested scopes of the statement
            if isinstance(t, (ast.FunctionDef, ast.Lambda)):
                a = t.args
                inner.update(x.arg for x in a.args + a.kwonlyargs + a.posonlyargs)
                if a.vararg:
                    inner.add(a.vararg.arg)
                if a.kwarg:
                    inner.add(a.kwarg.arg)
                if isinstance(t, ast.FunctionDef):
                    bound.add(t.name)
            if isinstance(t, ast.comprehension):
                for n in ast.walk(t.target):
                    if isinstance(n, ast.Name):
                        inner.add(n.id)
            if isinstance(t, ast.ExceptHandler) and t.name:
                inner.add(t.name)
        missing = sorted({t.id for t in ast.walk(node) if isinstance(t, ast.Name) and isinstance(t.ctx, ast.Load)} - bound - inner)
        assert not missing, f"{os.path.relpath(path, ROOT)}:{node.lineno}: {missing} used before any binding"
