"""Regenerates tests/golden/jutul_api.json: for every Jutul name the binding jutul.jl_amd/julia/JutulHIP.jl imports, overloads
or calls (`import Jutul: ...`, `Jutul.name`), WHERE the reference defines it and with WHICH call shape (positional arity range,
keyword names, keyword splat).  Runs only in the build container (needs /root/reference).  The JSON holds signatures as DATA
(names, counts, file:line) -- no reference source text.  tests/test_julia_binding_static.py checks the binding against it and,
when /root/reference is present, that the file is up to date.
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
from _julia_static import find_methods, match_bracket, split_top, strip_comments  # noqa: E402

REF = "/root/reference"


def binding_names():
    src = strip_comments(open(os.path.join(ROOT, "jutul.jl_amd", "julia", "JutulHIP.jl")).read())
    m = re.search(r"import Jutul:(.*?)\n\n", src, flags=re.S)
    imported = [x.strip() for x in m.group(1).replace("\n", " ").split(",") if x.strip()]
    qualified = sorted(set(re.findall(r"\bJutul\.([A-Za-z_]\w*!?)", src)))
    return imported, qualified


def scan():
    imported, qualified = binding_names()
    wanted = set(imported) | set(qualified)
    api = {n: {"methods": [], "types": []} for n in sorted(wanted)}
    for sub in ("src", "ext"):
        for dp, _, fns in os.walk(os.path.join(REF, sub)):
            for fn in sorted(fns):
                if not fn.endswith(".jl"):
                    continue
                path = os.path.join(dp, fn)
                rel = os.path.relpath(path, REF)
                src = open(path, errors="replace").read()
                for name, line, sig in find_methods(src, wanted):
                    api[name]["methods"].append({"at": f"{rel}:{line}", "min": sig["min"], "max": sig["max"],
                                                 "keywords": sig["keywords"], "kwsplat": sig["kwsplat"], "types": sig["types"]})
                s = strip_comments(src)
                for m in re.finditer(r"(?m)^[ \t]*(?:export[ \t]+)?(mutable struct|struct|abstract type|const|macro)[ \t]+([A-Za-z_]\w*)", s):
                    if m.group(2) in wanted:
                        ent = {"at": f"{rel}:{s.count(chr(10), 0, m.start()) + 1}", "kind": m.group(1)}
                        if s[m.end():m.end() + 1] == "{":          # number of type parameters of a parametric type
                            ent["nparams"] = len(split_top(s[m.end() + 1:match_bracket(s, m.end())], ","))
                        elif "struct" in m.group(1) or "type" in m.group(1):
                            ent["nparams"] = 0
                        sup = re.match(r"(?:\{[^\n]*?\})?[ \t]*<:[ \t]*([\w.]+)", s[m.end():m.end() + 300])
                        if sup:
                            ent["supertype"] = sup.group(1).split(".")[-1]
                        api[m.group(2)]["types"].append(ent)
    # fields of the reference structs whose fields the binding reads (krylov.config, cfg.max_iterations, model.context, ...)
    fields = {}
    for want, rel in (("GenericKrylov", "src/linsolve/krylov.jl"), ("IterativeSolverConfig", "src/linsolve/utils.jl"),
                      ("SimulationModel", "src/core_types/core_types.jl")):
        s = strip_comments(open(os.path.join(REF, rel), errors="replace").read())
        m = re.search(r"(?m)^[ \t]*(?:mutable[ \t]+)?struct[ \t]+" + want + r"\b[^\n]*\n", s)
        names = []
        for ln in s[m.end():].split("\n"):
            if re.match(r"^[ \t]*(function|end)\b", ln):
                break
            f = re.match(r"^[ \t]*(?:const[ \t]+)?([A-Za-z_]\w*)[ \t]*(::.*)?$", ln)
            if f:
                names.append(f.group(1))
        fields[want] = names
    # names re-exported from packages Jutul depends on (not defined under /root/reference/src): recorded as such
    external = {"OrderedDict": "OrderedCollections (Project.toml dependency, re-exported through `using`)"}
    for n, why in external.items():
        if n in api and not api[n]["methods"] and not api[n]["types"]:
            api[n]["external"] = why
    return {"_meta": {"reference": "sintefmath/Jutul.jl as vendored under /root/reference", "imported": imported, "qualified": qualified},
            "struct_fields": fields,
            "names": api}


if __name__ == "__main__":
    out = scan()
    json.dump(out, open(os.path.join(HERE, "jutul_api.json"), "w"), indent=1, sort_keys=True)
    missing = [n for n, v in out["names"].items() if not v["methods"] and not v["types"] and "external" not in v]
    print(len(out["names"]), "names;", sum(len(v["methods"]) for v in out["names"].values()), "methods; undefined in the reference:", missing)
