#!/usr/bin/env python3
"""Generate seqwin_amd/_abi.py (ctypes restype / argtypes of every entry point) from include/seqwin_hip.h.

    python3 scripts/gen_abi.py            # rewrite seqwin_amd/_abi.py
    python3 scripts/gen_abi.py --check    # exit 1 if the committed table differs from the header (tests/test_abi_cpu.py)

One declaration per function in the header, plain C types only (that is the point of the boundary), so a regular expression
is enough of a parser.  Pointers to anything but char are void pointers on the Python side (numpy buffers, opaque handles,
byref(c_uint64) out-parameters all convert to c_void_p).
"""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "seqwin_hip.h"
OUT = ROOT / "seqwin_amd" / "_abi.py"

SCALARS = {"int": "c_int", "uint64_t": "c_uint64", "uint32_t": "c_uint32", "size_t": "c_size_t", "double": "c_double",
           "float": "c_float", "unsigned": "c_uint", "unsigned int": "c_uint", "int64_t": "c_int64"}


def ctype_of(decl: str) -> str:
    d = decl.strip()
    d = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*$", "", d).strip() if not d.endswith("*") and " " in d else d   # drop the parameter name
    stars = d.count("*")
    base = re.sub(r"\bconst\b|\*", " ", d)
    base = " ".join(base.split())
    if base == "sw_log_fn":
        return "LOG_FN"
    if stars == 0:
        if base not in SCALARS:
            raise SystemExit(f"unknown scalar type in {decl!r}")
        return SCALARS[base]
    if base == "char":
        return "c_char_p" if stars == 1 else "POINTER(c_char_p)"
    return "c_void_p"


def parse():
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(sw_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        if ret in ("void",):
            res = "None"
        else:
            res = ctype_of(ret + " x") if "*" not in ret else ctype_of(ret)
        argl = [] if args in ("void", "") else [ctype_of(a) for a in args.split(",")]
        protos[name] = (res, argl)
    return protos


def render(protos) -> str:
    lines = ['"""ctypes prototypes of libseqwin_hip.so -- GENERATED from include/seqwin_hip.h by scripts/gen_abi.py; do not edit."""',
             "from ctypes import (CFUNCTYPE, POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint, c_uint32, c_uint64,  # noqa: F401",
             "                    c_void_p)", "",
             "LOG_FN = CFUNCTYPE(None, c_char_p, c_char_p)   # sw_log_fn", "", "PROTOTYPES = {"]
    for name in sorted(protos):
        res, argl = protos[name]
        lines.append(f'    "{name}": ({res}, [{", ".join(argl)}]),')
    lines += ["}", ""]
    return "\n".join(lines)


if __name__ == "__main__":
    txt = render(parse())
    if "--check" in sys.argv:
        if not OUT.exists() or OUT.read_text() != txt:
            print("seqwin_amd/_abi.py is out of date: run python3 scripts/gen_abi.py")
            sys.exit(1)
        sys.exit(0)
    OUT.write_text(txt)
    print(f"{OUT}: {len(parse())} prototypes")
