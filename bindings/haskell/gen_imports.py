#!/usr/bin/env python3
"""Emit the `foreign import ccall` block of GridHip.hs from include/gridhip.h, one import per prototype, so that the
Haskell binding cannot drift from the C ABI.  `python bindings/haskell/gen_imports.py` prints the block;
tests/test_haskell_shim.py checks that GridHip.hs contains exactly this block.  (There is no GHC in the build image:
the shim is source only; this keeps at least names, arity and C types mechanical.)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HEADER = os.path.join(ROOT, "include", "gridhip.h")

# blocking entry points that start host threads or wait on the device for long: `safe` so that the Haskell RTS keeps
# running other capabilities (src/Hdf5.hs uses `unsafe` throughout because its calls are short)
SAFE = {"gridhip_comm_convgrid2", "gridhip_comm_create", "gridhip_comm_create_rank", "gridhip_comm_destroy"}

TYPES = [
    (r"^(const )?gridhip_ctx \*\*$", "Ptr (Ptr Ctx)"),
    (r"^(const )?gridhip_ctx \*$", "Ptr Ctx"),
    (r"^gridhip_plan \*\*$", "Ptr (Ptr Plan)"),
    (r"^gridhip_plan \*$", "Ptr Plan"),
    (r"^gridhip_comm \*\*$", "Ptr (Ptr Comm)"),
    (r"^(const )?gridhip_comm \*$", "Ptr Comm"),
    (r"^double \*const \*$", "Ptr (Ptr CDouble)"),
    (r"^(const )?double \*$", "Ptr CDouble"),
    (r"^(const )?int64_t \*$", "Ptr Int64"),
    (r"^(const )?int \*$", "Ptr CInt"),
    (r"^const char \*$", "CString"),
    (r"^void \*\*$", "Ptr (Ptr ())"),
    (r"^(const )?void \*$", "Ptr ()"),
    (r"^int64_t$", "Int64"),
    (r"^int$", "CInt"),
    (r"^double$", "CDouble"),
]


def hs_type(c):
    c = re.sub(r"\s+", " ", c.strip())
    c = c.replace(" *", " *").replace("* *", "**")
    for pat, hs in TYPES:
        if re.match(pat, c):
            return hs
    raise ValueError(f"no Haskell type for C type '{c}'")


def prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    out = []
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[ \*]+)(gridhip_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)  # type, parameter name
                params.append((mm.group(1).strip(), mm.group(2)))
        out.append((ret, name, params))
    return out


def hs_name(cname):
    return "c_" + cname[len("gridhip_"):]


def block():
    lines = []
    for ret, name, params in prototypes():
        safety = "safe" if name in SAFE else "unsafe"
        sig = [hs_type(t) for t, _ in params] + ["IO " + (hs_type(ret) if " " not in hs_type(ret) else "(" + hs_type(ret) + ")")]
        lines.append(f"-- {ret} {name}({', '.join(n for _, n in params)})")
        lines.append(f'foreign import ccall {safety} "{name}"')
        lines.append(f"  {hs_name(name)} :: " + " -> ".join(sig))
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    sys.stdout.write(block())
