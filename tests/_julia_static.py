"""Static reader of Julia source, just enough for tests/test_julia_binding_static.py and tests/golden/make_jutul_api.py:
method signatures (`function name(args; kw)` and `name(args) = ...`), block balance, `ccall` / `@jh` call sites.
No Julia is executed (there is none in the image); this is a lexer with bracket matching, not a parser.
"""
import re

OPEN, CLOSE = "([{", ")]}"


def strip_comments(src):
    """Remove `# ...` comments and `#= ... =#` blocks, keep strings and line structure."""
    out, i, n = [], 0, len(src)
    in_str = None
    while i < n:
        c = src[i]
        if in_str:
            out.append(c)
            if c == "\\" and i + 1 < n:
                out.append(src[i + 1]); i += 2; continue
            if src.startswith(in_str, i):
                out.append(src[i + 1:i + len(in_str)]); i += len(in_str); in_str = None; continue
            i += 1; continue
        if src.startswith('"""', i):
            in_str = '"""'; out.append('"""'); i += 3; continue
        if c == '"':
            in_str = '"'; out.append(c); i += 1; continue
        if c == "'" and i + 2 < n and (src[i + 2] == "'" or (src[i + 1] == "\\" and i + 3 < n and src[i + 3] == "'")):
            j = i + (3 if src[i + 2] == "'" else 4)   # character literal, not the adjoint operator
            out.append(src[i:j]); i = j; continue
        if src.startswith("#=", i):
            j = src.find("=#", i + 2); j = n if j < 0 else j + 2
            out.append("\n" * src.count("\n", i, j)); i = j; continue
        if c == "#":
            j = src.find("\n", i); j = n if j < 0 else j
            i = j; continue
        out.append(c); i += 1
    return "".join(out)


def match_bracket(src, i):
    """index of the bracket closing src[i] (one of ([{), strings skipped"""
    depth, n = 0, len(src)
    while i < n:
        c = src[i]
        if c == '"':
            q = '"""' if src.startswith('"""', i) else '"'
            i += len(q)
            while i < n and not src.startswith(q, i):
                i += 2 if src[i] == "\\" else 1
            i += len(q); continue
        if c in OPEN:
            depth += 1
        elif c in CLOSE:
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise ValueError("unbalanced bracket")


def split_top(s, seps=","):
    """split at separators that are outside every bracket and string"""
    parts, depth, cur, i, n = [], 0, [], 0, len(s)
    while i < n:
        c = s[i]
        if c == '"':
            j = i + 1
            while j < n and s[j] != '"':
                j += 2 if s[j] == "\\" else 1
            cur.append(s[i:j + 1]); i = j + 1; continue
        if c in OPEN:
            depth += 1
        elif c in CLOSE:
            depth -= 1
        if depth == 0 and c in seps:
            parts.append("".join(cur)); cur = []
        else:
            cur.append(c)
        i += 1
    parts.append("".join(cur))
    return [p.strip() for p in parts]


def parse_arglist(inner):
    """'(a, b::T = 1, c...; k = 2, kw...)' body -> dict(min, max, keywords, kwsplat, positional)"""
    semi = split_top(inner, ";")
    pos = [p for p in split_top(semi[0], ",") if p]
    kws = [p for part in semi[1:] for p in split_top(part, ",") if p]
    npos_min = npos_max = 0
    names = []
    vararg = False
    for p in pos:
        head = split_top(p, "=")[0] if "=" in p else p
        has_default = len(split_top(p, "=")) > 1
        if head.rstrip().endswith("..."):
            vararg = True
            names.append(head.strip())
            continue
        names.append(head.strip())
        npos_max += 1
        if not has_default:
            npos_min += 1
    keywords, kwsplat = [], False
    for k in kws:
        head = split_top(k, "=")[0].strip()
        if head.endswith("..."):
            kwsplat = True
        else:
            keywords.append(re.split(r"::", head)[0].strip())
    types = []
    for h in names:
        parts = split_top(h.rstrip(".").rstrip("."), ":")          # `x::T` splits into ['x', '', 'T'] at top level
        types.append("::".join(parts[2:]).strip() if len(parts) >= 3 else "")
    return {"min": npos_min, "max": None if vararg else npos_max, "keywords": keywords, "kwsplat": kwsplat, "positional": names,
            "types": types}


NAME = r"(?:[A-Za-z_][\w]*\.)*[A-Za-z_][\w!]*"


def find_methods(src, wanted=None):
    """[(name, line, sig dict)] for every `function name(...)` and `name(...) = ` / `name(...) where ... =` at line starts"""
    s = strip_comments(src)
    out = []
    for m in re.finditer(r"(?m)^[ \t]*(?:@inline[ \t]+|@noinline[ \t]+)?(function[ \t]+)?(" + NAME + r")[ \t]*(\{[^}\n]*\})?\(", s):
        is_fn, name = bool(m.group(1)), m.group(2)
        short = name.split(".")[-1]
        if wanted is not None and short not in wanted:
            continue
        op = m.end() - 1
        try:
            cl = match_bracket(s, op)
        except ValueError:
            continue
        if not is_fn:
            rest = s[cl + 1:cl + 200]
            if not re.match(r"[ \t]*(?:::[^\n=]*?)?(?:where[^\n=]*?)?=(?!=)", rest):
                continue
        line = s.count("\n", 0, m.start(2)) + 1
        try:
            sig = parse_arglist(s[op + 1:cl])
        except Exception:  # noqa: BLE001
            continue
        out.append((short, line, sig))
    return out


BLOCK_OPEN = {"function", "struct", "if", "for", "while", "let", "do", "begin", "try", "module", "macro", "quote", "baremodule"}


def block_balance(src):
    """[(line, message)] of block-structure faults: every function/struct/if/for/while/let/do/begin/try/module/macro/quote
    (and `mutable struct`, `abstract type`/`primitive type` ... end) needs its `end`; `end` inside [...] is an index, not a closer."""
    s = strip_comments(src)
    # blank out strings (keep length and newlines)
    def blank(m):
        return re.sub(r"[^\n]", " ", m.group(0))
    s = re.sub(r'"""(?:\\.|[^\\])*?"""', blank, s, flags=re.S)
    s = re.sub(r'"(?:\\.|[^"\\\n])*"', blank, s)
    faults, stack = [], []
    square = 0
    paren_ctx = []   # bracket stack to know whether we are inside (...) of a call (generators: `for` inside brackets has no end)
    for m in re.finditer(r"[\[\]\(\)\{\}]|:?\b[A-Za-z_]\w*\b", s):
        t = m.group(0)
        line = s.count("\n", 0, m.start()) + 1
        if t in "[({":
            paren_ctx.append(t); square += t == "["; continue
        if t in "])}":
            if paren_ctx:
                o = paren_ctx.pop(); square -= o == "["
            else:
                faults.append((line, f"unmatched {t}"))
            continue
        if t.startswith(":"):          # a quoted symbol such as :end / :for
            continue
        prev = s[max(0, m.start() - 1):m.start()]
        if prev == ".":                # field access such as x.end
            continue
        if t == "type":
            before = s[max(0, m.start() - 12):m.start()]
            if re.search(r"\b(abstract|primitive)\s+$", before):
                stack.append(("type", line))
            continue
        if t in BLOCK_OPEN:
            if paren_ctx and t in ("for", "if"):      # generator / comprehension / ternary-free filter inside brackets
                continue
            if t == "struct" and stack and stack[-1][0] == "mutable":
                stack.pop()
            stack.append((t, line)); continue
        if t == "mutable":
            continue
        if t == "end":
            if square > 0 and paren_ctx and paren_ctx[-1] == "[":
                continue               # a[end]
            if not stack:
                faults.append((line, "`end` without an opener")); continue
            stack.pop()
    for t, line in stack:
        faults.append((line, f"`{t}` is never closed"))
    if paren_ctx:
        faults.append((0, f"unclosed brackets: {paren_ctx}"))
    return faults


def top_level_definitions(src):
    """[(first line, last line, header)] of the top-level `function` / `struct` / `macro` blocks inside the module"""
    s = strip_comments(src)
    lines = s.split("\n")
    out, i = [], 0
    while i < len(lines):
        ln = lines[i]
        if re.match(r"^(function|mutable struct|struct|macro|abstract type)\b", ln):
            if re.match(r"^abstract type\b.*\bend\s*$", ln):
                out.append((i + 1, i + 1, ln.strip())); i += 1; continue
            j = i + 1
            while j < len(lines) and not re.match(r"^end\b", lines[j]):
                j += 1
            out.append((i + 1, j + 1, ln.strip())); i = j + 1; continue
        i += 1
    return out
