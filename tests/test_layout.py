"""Repository rules that can be checked without a GPU:
  * the oracle is test infrastructure: nothing under mt3_amd/ imports it; bench.py touches it only inside
    `cpu_baseline`, __graft_entry__.py only inside `smoke`/`build`;
  * the product has no CPU fallback: loading the library fails loudly when it is missing, and no module of the
    package catches that to continue on numpy/torch;
  * nothing that runs on the GPU box reads /root/reference;
  * bench.py's argument defaults and JSON keys follow the driver contract."""
import ast
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mt3_amd")


def _py_files(d):
    for base, _, files in os.walk(d):
        if "__pycache__" in base:
            continue
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(base, f)


def _imports(tree):
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield node, a.name
        elif isinstance(node, ast.ImportFrom):
            yield node, node.module or ""


def test_package_never_imports_the_oracle():
    for path in _py_files(PKG):
        tree = ast.parse(open(path).read())
        for _, mod in _imports(tree):
            assert mod.split(".")[0] != "oracle", "%s imports the oracle" % path
        assert "/root/reference" not in open(path).read(), "%s reads the reference tree" % path
    for path in list(_py_files(os.path.join(PKG, "csrc"))) + [os.path.join(PKG, "csrc", f)
                                                              for f in os.listdir(os.path.join(PKG, "csrc"))]:
        assert "oracle/" not in open(path, errors="ignore").read()


def _functions_importing_oracle(path):
    tree = ast.parse(open(path).read())
    out = set()
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for _, mod in _imports(fn):
            if mod.split(".")[0] == "oracle":
                out.add(fn.name)
    top = [mod for node, mod in _imports(ast.Module(body=[n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))],
                                                    type_ignores=[]))]
    assert not any(m.split(".")[0] == "oracle" for m in top), "%s imports the oracle at module level" % path
    return out


def test_oracle_is_only_the_checker_in_bench_and_entry():
    assert _functions_importing_oracle(os.path.join(ROOT, "bench.py")) == {"cpu_baseline"}
    assert _functions_importing_oracle(os.path.join(ROOT, "__graft_entry__.py")) <= {"smoke", "build"}
    for f in ("bench.py", "__graft_entry__.py"):
        src = open(os.path.join(ROOT, f)).read()
        # /root/reference may only appear in build() (which skips it when absent), never in smoke()/bench
        if f == "bench.py":
            assert "/root/reference" not in src


def test_no_cpu_fallback_when_the_library_is_missing(tmp_path, monkeypatch):
    from mt3_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libmt3hip.so"))
    with pytest.raises((OSError, RuntimeError, _lib.Mt3Error)):
        _lib.load()
    # and no module swallows that error to carry on without the library
    for path in _py_files(PKG):
        src = open(path).read()
        for m in re.finditer(r"except\s+(\(?[\w., ]*\)?)\s*(as \w+)?:\s*\n\s+(.*)", src):
            caught, body = m.group(1), m.group(3)
            assert not ("OSError" in caught and "numpy" in body), path


def test_bench_contract_statics():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"',
                '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"', '"workload"',
                '"roofline"', '"bound"', '"achieved"', '"peak"', '"frac"', '"traffic"', '"cpu_baseline"', '"cores"',
                '"kind"', '"sample"'):
        assert key in src, key
    tree = ast.parse(src)
    defaults = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
            name = node.args[0].value
            for kw in node.keywords:
                if kw.arg == "default" and isinstance(kw.value, ast.Constant):
                    defaults[name] = kw.value.value
    assert defaults["--gpus"] == 1 and defaults["--batch"] == 256 and defaults["--decode-steps"] == 1024
    assert defaults["--decoding"] == "greedy"            # BASELINE configs[2] names greedy decode
    assert defaults["--steps"] >= 1 and defaults["--warmup"] >= 1
    import json
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert base["metric"] in src


def test_bench_cpu_baseline_leg_runs_and_reports_the_contract_fields():
    """`bench.py --cpu-baseline-only` (the leg the GPU bench spawns as a subprocess) on a tiny sample."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-only", "--cpu-segments", "1",
                        "--decode-steps", "4"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("CPU_BASELINE ")]
    assert r.returncode == 0 and line, r.stderr[-500:]
    d = json.loads(line[-1][len("CPU_BASELINE "):])
    assert d["unit"] == "audio-s/s" and d["kind"] == "port" and d["value"] > 0 and 1 <= d["cores"] <= 16
    assert "segments" in d["sample"]
