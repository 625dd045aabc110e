"""The Haskell binding (bindings/haskell/GridHip.hs) cannot be compiled here - there is no GHC in the image - so
what can be checked mechanically is: every `foreign import ccall` names a symbol of include/gridhip.h, with the
arity and the C types of its prototype; every prototype has an import; the import block is the one the generator
emits from the header; and the wrappers the module exports are defined and only call imports that exist."""
import importlib.util
import os
import re

from conftest import ROOT

HS = os.path.join(ROOT, "bindings", "haskell", "GridHip.hs")
GEN = os.path.join(ROOT, "bindings", "haskell", "gen_imports.py")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_imports", GEN)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _imports(src):
    """name -> (safety, haskell identifier, [argument types ..., result])"""
    out = {}
    for m in re.finditer(r'foreign import ccall (unsafe|safe) "([a-z0-9_]+)"\s+([a-z][A-Za-z0-9_\']*)\s*::\s*([^\n]+)', src):
        safety, cname, hname, sig = m.groups()
        depth, parts, cur = 0, [], ""
        for tok in re.split(r"(\(|\)|->)", sig):
            if tok == "(":
                depth += 1
            elif tok == ")":
                depth -= 1
            if tok == "->" and depth == 0:
                parts.append(cur.strip())
                cur = ""
            else:
                cur += tok
        parts.append(cur.strip())
        out[cname] = (safety, hname, parts)
    return out


def test_every_import_matches_a_prototype_and_every_prototype_has_an_import():
    gen = _gen()
    protos = {name: (ret, params) for ret, name, params in gen.prototypes()}
    imps = _imports(open(HS).read())
    assert sorted(imps) == sorted(protos), (sorted(set(protos) - set(imps)), sorted(set(imps) - set(protos)))
    from gridhip import _lib
    assert sorted(protos) == sorted(_lib.SIGNATURES)  # the same set the ctypes table and the library carry
    for name, (ret, params) in protos.items():
        safety, hname, parts = imps[name]
        assert hname == "c_" + name[len("gridhip_"):]
        assert len(parts) - 1 == len(params), f"{name}: {len(parts) - 1} Haskell arguments, {len(params)} in the header"
        for (ctype, pname), hs in zip(params, parts[:-1]):
            assert hs == gen.hs_type(ctype), f"{name}({pname}): {hs} vs {ctype}"
        want = gen.hs_type(ret)
        assert parts[-1] in (f"IO {want}", f"IO ({want})"), f"{name}: returns {parts[-1]}, header says {ret}"
        assert safety == ("safe" if name in gen.SAFE else "unsafe")


def test_import_block_is_the_generated_one():
    src = open(HS).read()
    a = src.index("-- BEGIN GENERATED IMPORTS\n") + len("-- BEGIN GENERATED IMPORTS\n")
    b = src.index("-- END GENERATED IMPORTS")
    assert src[a:b] == _gen().block(), "run bindings/haskell/gen_imports.py and paste its output between the markers"


def test_exported_wrappers_exist_and_use_imported_names_only():
    src = open(HS).read()
    head = src[src.index("module GridHip"):src.index(") where")]
    exported = [e for e in re.findall(r"\b([a-z][A-Za-z0-9]*)\b", re.sub(r"--[^\n]*", "", head))
                if e not in ("module", "where")]
    for need in ("gridIO", "convgridIO", "convgrid2IO", "degrid2IO", "awgridIO", "doImagingIO", "withNode",
                 "convgrid2NodeIO", "withGridHip", "simpleImagingIO", "convImagingIO", "wCacheImagingIO", "awImagingIO"):
        assert need in exported
    body = src[src.index("-- END GENERATED IMPORTS"):]
    for e in exported:
        assert re.search(rf"^{e} ::", body, flags=re.M), f"exported {e} has no type signature / definition"
    imps = {h for _, h, _ in _imports(src).values()}
    used = set(re.findall(r"\b(c_[a-z0-9_]+)\b", body))
    assert used <= imps, used - imps
    # the gridders the patch swaps in are all bound
    assert {"c_grid", "c_convgrid", "c_convgrid2", "c_degrid2", "c_awgrid", "c_do_imaging", "c_comm_convgrid2"} <= used
    patch = open(os.path.join(ROOT, "bindings", "haskell", "Gridding.patch.md")).read()
    for fn in re.findall(r"GH\.([A-Za-z0-9]+)", patch):
        assert fn in exported or fn in ("GridHip", "WCacheImaging"), fn
