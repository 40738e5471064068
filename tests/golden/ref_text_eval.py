"""Evaluate the reference's RHS *from its own source text* (fixture generation only).

The reference's RHS is C++ written against ``casadi::SX`` (CasADi is not installed here and
no stand-in header is written).  To obtain reference-anchored golden vectors anyway, this
module reads ``aux_states.hpp`` and ``ode.hpp`` **from /root/reference at run time**,
rewrites every statement token-wise into a Python statement (``p(80)`` -> ``p[80]``,
``exp`` -> ``math.exp`` ...) and executes them with IEEE doubles.  Nothing from the
reference is stored in this repository: only the numeric outputs are committed (as
``tests/golden/*.npz``) by ``make_golden.py``.

Runs only in the build container (needs /root/reference); never imported by tests.
"""
from __future__ import annotations

import math
import re
from pathlib import Path

import numpy as np

MODELS = Path("/root/reference/gl_gym/environments/models")

_FUNCS = {
    "exp": "math.exp", "sqrt": "math.sqrt", "cos": "math.cos", "tanh": "math.tanh",
    "fabs": "math.fabs", "pow": "math.pow", "fmax": "max", "fmin": "min",
}


def _strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    # pre-processor lines and the stray '#//...' line
    src = "\n".join(l for l in src.split("\n") if not l.lstrip().startswith("#"))
    return src


def _expr(e: str) -> str:
    e = " ".join(e.split())
    e = re.sub(r"\b([xudp])\((\d+)\)", r"\1[\2]", e)          # casadi element access
    e = re.sub(r"\b(a|dxdt)\((\d+)\)", r"\1[\2]", e)
    e = e.replace("M_PI", "math.pi")
    e = e.replace("||", " or ")
    for k, v in _FUNCS.items():
        e = re.sub(rf"(?<![\w.]){k}\(", v + "(", e)
    return e


def _helper_functions(src: str) -> str:
    out = []
    for m in re.finditer(r"inline\s+SX\s+(\w+)\s*\((.*?)\)\s*\{(.*?)\n\}", src, flags=re.S):
        name, args, body = m.group(1), m.group(2), m.group(3)
        argn = [a.split()[-1].lstrip("&") for a in args.split(",")]
        lines = [f"def {name}({', '.join(argn)}):"]
        for st in body.split(";"):
            st = " ".join(st.split())
            if not st:
                continue
            mf = re.match(r"const float (\w+) = (.*)", st)
            if mf:  # a C `float` constant: float32-rounded, then promoted
                lines.append(f"    {mf.group(1)} = float(np.float32({mf.group(2)}))")
                continue
            st = re.sub(r"^(const\s+)?(double|SX)\s+", "", st)
            lines.append("    " + _expr(st))
        out.append("\n".join(lines))
    return "\n\n".join(out)


def _body(src: str, start_pat: str, end_pat: str) -> str:
    i = re.search(start_pat, src).end()
    j = re.search(end_pat, src[i:]).start() + i
    return src[i:j]


def build():
    aux_src = _strip_comments((MODELS / "aux_states.hpp").read_text(errors="replace"))
    ode_src = _strip_comments((MODELS / "ode.hpp").read_text(errors="replace"))

    py = ["import math", "import numpy as np", "def if_else(c, a, b):\n    return a if c else b", ""]
    py.append(_helper_functions(aux_src))

    upd = _body(aux_src, r"std::vector<SX>\s+a\(239\)\s*;", r"return\s+vertcat\(a\)")
    py.append("def update(x, u, d, p):\n    a = [0.0] * 239")
    n_stmt = 0
    for st in upd.split(";"):
        st = " ".join(st.split())
        if not st:
            continue
        assert re.match(r"a\[\d+\]\s*=", st), st
        py.append("    " + _expr(st))
        n_stmt += 1
    assert n_stmt == 239, n_stmt
    py.append("    return a\n")

    ode = _body(ode_src, r"SX\s+ODE\s*\(.*?\)\s*\{", r"return\s+dxdt\s*;")
    py.append("def ODE(x, u, d, p):\n    a = update(x, u, d, p)\n    dxdt = [0.0] * 28")
    n_dx = 0
    for st in ode.split(";"):
        st = " ".join(st.split())
        if not st or st.startswith("SX "):
            continue
        assert re.match(r"dxdt\(\d+\)\s*=", st), st
        py.append("    " + _expr(st))
        n_dx += 1
    assert n_dx == 28, n_dx
    py.append("    return dxdt, a\n")

    # ODE_pipe (ode.hpp:126-263): same statement rewriting; its four `SX name = ...` locals become plain assignments
    pipe = _body(ode_src, r"SX\s+ODE_pipe\s*\(.*?\)\s*\{", r"return\s+dxdt\s*;")
    py.append("def ODE_pipe(x, u, d, p):\n    a = update(x, u, d, p)\n    dxdt = [0.0] * 28")
    n_dx = 0
    for st in pipe.split(";"):
        st = " ".join(st.replace("\\", " ").split())
        if not st or st.startswith("SX a =") or st.startswith("SX dxdt"):
            continue
        if st.startswith("SX "):
            st = st[3:]
            assert re.match(r"t\w*Pipe(On|Off)\s*=", st), st
        else:
            assert re.match(r"dxdt\(\d+\)\s*=", st), st
            n_dx += 1
        py.append("    " + _expr(st))
    assert n_dx == 28, n_dx
    py.append("    return dxdt, a\n")

    code = "\n".join(py)
    ns: dict = {}
    exec(compile(code, "<reference-text>", "exec"), ns)
    return ns


_NS = None


def ref_rhs(x, u, d, p):
    """(dx[28], aux[239]) from the reference's own expressions, IEEE double."""
    global _NS
    if _NS is None:
        _NS = build()
    x = [float(v) for v in x]
    u = [float(v) for v in u]
    d = [float(v) for v in d]
    p = [float(v) for v in p]
    dx, a = _NS["ODE"](x, u, d, p)
    return np.array(dx, dtype=np.float64), np.array(a, dtype=np.float64)


def ref_rhs_pipe(x, u, d, p):
    """(dx[28], aux[239]) of ODE_pipe (d has 14 entries: 10 tPipe, 11 tGroPipe, 12 pipeSwitchOff, 13 groPipeSwitchOff)."""
    global _NS
    if _NS is None:
        _NS = build()
    dx, a = _NS["ODE_pipe"]([float(v) for v in x], [float(v) for v in u], [float(v) for v in d], [float(v) for v in p])
    return np.array(dx, dtype=np.float64), np.array(a, dtype=np.float64)


if __name__ == "__main__":
    ns = build()
    print("helpers:", [k for k in ns if callable(ns[k]) and not k.startswith("_")])
