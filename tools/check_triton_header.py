#!/usr/bin/env python3
"""Step 0 of a real deployment: diff include/tritonbackend_hps.h against the REAL Triton headers.

    python tools/check_triton_header.py <path/to/tritonbackend.h> <path/to/tritonserver.h> [--ours include/tritonbackend_hps.h]

Neither Triton's headers nor tritonserver exist in the build image (the reference fetches them over the network:
/root/reference/hps_backend/CMakeLists.txt:82-100), so include/tritonbackend_hps.h RESTATES the part of Triton's public C API
the plugin imports (45 functions, 6 enums, 2 flags, the API version).  This script parses the real headers of the Triton release the
plugin is about to be loaded into and reports every difference that would matter at the ABI:

  * a function the plugin imports that the real headers do not declare,
  * a return type or a parameter type that differs (parameter NAMES do not matter; `struct X*` == `X*`),
  * an enumerator or flag the plugin uses whose VALUE differs,
  * the backend API version: the real major must equal ours, the real minor must be >= ours (hps.cc:64-82 checks the same at load).

Exit status 0: no difference.  1: differences (listed on stdout).  2: a header could not be read.
Pure text processing (no compiler needed); tested in the CPU suite against small synthetic headers (tests/test_check_triton_header.py).
"""
from __future__ import annotations

import argparse
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

# decoration of the real headers that carries no type information
_NOISE = re.compile(r"\b(TRITONSERVER_DECLSPEC|TRITONBACKEND_DECLSPEC|TRITONBACKEND_ISPEC|TRITONREPOAGENT_DECLSPEC|HPS_TRITON_EXPORT|"
                    r"extern|__declspec\s*\([^)]*\)|__attribute__\s*\(\([^)]*(\([^)]*\))?[^)]*\)\))")


def strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _norm_type(t: str) -> str:
    t = _NOISE.sub(" ", t)
    t = re.sub(r"\bstruct\s+", "", t)
    t = re.sub(r"\benum\s+", "", t)
    t = re.sub(r"\s+", " ", t).strip()
    t = re.sub(r"\s*\*\s*", "*", t)
    # `const T` and `T const` are the same type; a top-level const of a by-value parameter is not part of the ABI
    t = re.sub(r"^(\w+) const\b", r"const \1", t)
    if "*" not in t and t.startswith("const "):
        t = t[len("const "):]
    return t


def _param_type(p: str) -> str:
    p = p.strip()
    if p in ("", "void"):
        return ""
    p = re.sub(r"\s+", " ", p)
    # drop the parameter name: the last identifier, unless it is the only word or part of the type (`unsigned int`)
    m = re.match(r"^(.*?[\*\s])([A-Za-z_]\w*)(\s*\[\s*\])?$", p)
    if m and m.group(1).strip() and m.group(2) not in ("int", "char", "long", "short", "unsigned", "float", "double", "bool", "size_t"):
        p = m.group(1) + ("*" if m.group(3) else "")
    return _norm_type(p)


def parse_functions(text: str) -> dict:
    """name -> (return type, [parameter types]) for every TRITONSERVER_* / TRITONBACKEND_* function declaration."""
    text = strip_comments(text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(TRITON(?:SERVER|BACKEND)_\w+)\s*\(([^;{}()]*)\)\s*;", text):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        if "typedef" in ret:
            continue
        ptypes = [_param_type(p) for p in params.split(",")]
        ptypes = [p for p in ptypes if p != ""]
        out[name] = (_norm_type(ret), ptypes)
    return out


def _eval_int(expr: str, known: dict) -> int | None:
    expr = expr.strip().rstrip("uUlL")
    expr = re.sub(r"\b([A-Za-z_]\w*)\b", lambda m: str(known[m.group(1)]) if m.group(1) in known else m.group(0), expr)
    expr = re.sub(r"(\d+)[uUlL]+", r"\1", expr)
    if not re.fullmatch(r"[\d\sxXa-fA-F\(\)\+\-\*\|&<>~]+", expr):
        return None
    try:
        return int(eval(expr, {"__builtins__": {}}, {}))   # digits and operators only (checked above)
    except Exception:  # noqa: BLE001
        return None


def parse_constants(text: str) -> dict:
    """name -> integer for every enumerator of every enum and every object-like #define with an integer value."""
    raw = text
    text = strip_comments(text)
    vals = {}
    for m in re.finditer(r"\benum\s+\w*\s*\{([^}]*)\}", text, flags=re.S):
        nxt = 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, expr = item.split("=", 1)
                v = _eval_int(expr, vals)
                if v is None:
                    continue
                nxt = v
            else:
                name = item
            vals[name.strip()] = nxt
            nxt += 1
    for m in re.finditer(r"^[ \t]*#[ \t]*define[ \t]+(\w+)[ \t]+(.+?)[ \t]*$", strip_comments(raw), flags=re.M):
        v = _eval_int(m.group(2), vals)
        if v is not None:
            vals.setdefault(m.group(1), v)
    return vals


def compare(ours_text: str, real_texts: list) -> list:
    problems = []
    ours_f, ours_c = parse_functions(ours_text), parse_constants(ours_text)
    real_f, real_c = {}, {}
    for t in real_texts:
        real_f.update(parse_functions(t))
        real_c.update(parse_constants(t))
    exports = {"TRITONBACKEND_Initialize", "TRITONBACKEND_Finalize", "TRITONBACKEND_ModelInitialize", "TRITONBACKEND_ModelFinalize",
               "TRITONBACKEND_ModelInstanceInitialize", "TRITONBACKEND_ModelInstanceFinalize", "TRITONBACKEND_ModelInstanceExecute"}
    for name, (ret, params) in sorted(ours_f.items()):
        if name not in real_f:
            problems.append(f"{name}: not declared by the real headers" + (" (an entry point the backend exports)" if name in exports else ""))
            continue
        rret, rparams = real_f[name]
        if rret != ret:
            problems.append(f"{name}: return type '{ret}' here, '{rret}' in the real header")
        if len(rparams) != len(params):
            problems.append(f"{name}: {len(params)} parameters here, {len(rparams)} in the real header ({rparams})")
            continue
        for i, (a, b) in enumerate(zip(params, rparams)):
            if a != b:
                problems.append(f"{name}: parameter {i + 1} is '{a}' here, '{b}' in the real header")
    for name, v in sorted(ours_c.items()):
        if not name.startswith(("TRITONSERVER_", "TRITONBACKEND_")) or name.endswith("_H_"):
            continue
        if name in ("TRITONBACKEND_API_VERSION_MAJOR", "TRITONBACKEND_API_VERSION_MINOR"):
            continue
        if name not in real_c:
            problems.append(f"{name} = {v}: not defined by the real headers")
        elif real_c[name] != v:
            problems.append(f"{name}: {v} here, {real_c[name]} in the real header")
    maj, mnr = ours_c.get("TRITONBACKEND_API_VERSION_MAJOR"), ours_c.get("TRITONBACKEND_API_VERSION_MINOR")
    rmaj, rmnr = real_c.get("TRITONBACKEND_API_VERSION_MAJOR"), real_c.get("TRITONBACKEND_API_VERSION_MINOR")
    if rmaj is None or rmnr is None:
        problems.append("TRITONBACKEND_API_VERSION_MAJOR / _MINOR: not found in the real headers")
    elif rmaj != maj or rmnr < mnr:
        problems.append(f"backend API version: built against {maj}.{mnr}, the real header is {rmaj}.{rmnr} "
                        "(TRITONBACKEND_Initialize refuses to load unless major is equal and minor >= ours)")
    return problems


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("tritonbackend_h")
    ap.add_argument("tritonserver_h")
    ap.add_argument("--ours", default=str(ROOT / "include" / "tritonbackend_hps.h"))
    a = ap.parse_args(argv)
    try:
        ours = Path(a.ours).read_text()
        real = [Path(a.tritonbackend_h).read_text(), Path(a.tritonserver_h).read_text()]
    except OSError as e:
        print(f"cannot read a header: {e}")
        return 2
    problems = compare(ours, real)
    nf = len(parse_functions(ours))
    if problems:
        print(f"{len(problems)} difference(s) between {a.ours} and the real headers:")
        for p in problems:
            print("  - " + p)
        return 1
    print(f"ok: {nf} function declarations and {len([k for k in parse_constants(ours) if k.startswith('TRITON')])} constants of "
          f"{a.ours} agree with the real headers")
    return 0


if __name__ == "__main__":
    sys.exit(main())
