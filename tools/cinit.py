"""Minimal reader for C static-initialiser tables.

Used only by the table extraction tools: the reference's tuning tables, windows and
codebooks exist only as C initialisers (lib/window.c, lib/modes/*.h, lib/books/**.h,
lib/masking.h, lib/psy.c), so they are read as TEXT here and re-emitted as binary
VPK packs.  No reference code is compiled or executed.

parse_file(path) -> dict name -> Decl(ctype, dims, value)
  value is a nested python list; leaves are int / float / str (identifier or
  '&identifier' / cast-stripped expression).
"""
import re
from collections import namedtuple

Decl = namedtuple("Decl", "ctype dims value")

_TOKEN = re.compile(r"""
    (?P<num>[-+]?(?:0[xX][0-9a-fA-F]+|(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?)[fFlLuU]*)
  | (?P<id>&?\s*[A-Za-z_][A-Za-z_0-9]*)
  | (?P<str>"(?:[^"\\]|\\.)*")
  | (?P<punc>[{}\[\]=;,()*])
""", re.X)


def preprocess(text, defined=()):
    """Strip comments, resolve #if/#ifdef with nothing defined (scalar path), drop other directives."""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = []
    stack = []  # each: [active_before, taken]
    active = True
    for line in text.split("\n"):
        s = line.strip()
        if s.startswith("#"):
            m = re.match(r"#\s*(\w+)\s*(.*)", s)
            d, rest = (m.group(1), m.group(2).strip()) if m else ("", "")
            if d in ("ifdef", "ifndef", "if"):
                if d == "ifdef":
                    cond = rest.split()[0] in defined
                elif d == "ifndef":
                    cond = rest.split()[0] not in defined
                else:
                    cond = rest.strip() in ("1",) or bool(re.match(r"defined\s*\(?\s*(\w+)", rest)
                                                           and re.match(r"defined\s*\(?\s*(\w+)", rest).group(1) in defined)
                stack.append((active, cond))
                active = active and cond
            elif d == "else":
                prev, cond = stack[-1]
                stack[-1] = (prev, not cond)
                active = prev and not cond
            elif d == "elif":
                prev, cond = stack[-1]
                stack[-1] = (prev, False if cond else False)
                active = False
            elif d == "endif":
                prev, _ = stack.pop()
                active = prev
            # other directives (#include, #define, #pragma) are dropped
            continue
        if active:
            out.append(line)
    return "\n".join(out)


class FLit(float):
    """A float literal that remembers its source text (for exact decimal->float32 rounding)."""
    def __new__(cls, text, is_f32=False):
        o = float.__new__(cls, text)
        o.text = text
        o.is_f32 = is_f32  # literal carried an F suffix (C type float, single rounding)
        return o

    def __neg__(self):
        return FLit(self.text[1:] if self.text.startswith("-") else "-" + self.text.lstrip("+"), self.is_f32)


def _num(tok):
    t = tok.rstrip("fFlLuU")
    if re.match(r"[-+]?0[xX]", t):
        return int(t, 16)
    if re.match(r"[-+]?\d+$", t):
        return int(t)
    return FLit(t, tok[-1] in "fF")


def to_f32(v):
    """Value a C compiler stores when this literal initialises a `float`: an F-suffixed
    literal is rounded decimal -> binary32 once (ties-to-even); an unsuffixed one is a
    double constant narrowed to float (two roundings)."""
    import numpy as np
    from fractions import Fraction
    if not isinstance(v, FLit) or not v.is_f32:
        return np.float32(float(v))
    exact = Fraction(v.text)
    x = np.float32(float(v))
    best = x
    for cand in (np.nextafter(x, np.float32(-np.inf)), np.nextafter(x, np.float32(np.inf))):
        dc = abs(Fraction(float(cand)) - exact)
        db = abs(Fraction(float(best)) - exact)
        if dc < db or (dc == db and (int(np.float32(cand).view(np.uint32)) & 1) == 0):
            best = cand
    return np.float32(best)


def _tokens(text):
    pos = 0
    n = len(text)
    while pos < n:
        if text[pos].isspace():
            pos += 1
            continue
        m = _TOKEN.match(text, pos)
        if not m:
            # unknown char (e.g. '.', '-', '>' in expressions) -> single char token
            yield ("punc", text[pos])
            pos += 1
            continue
        kind = m.lastgroup
        yield (kind, m.group(kind))
        pos = m.end()


def _parse_value(toks, i):
    """Parse one initialiser element starting at toks[i]; returns (value, next_i)."""
    kind, tok = toks[i]
    if kind == "punc" and tok == "{":
        i += 1
        items = []
        while True:
            kind, tok = toks[i]
            if kind == "punc" and tok == "}":
                return items, i + 1
            v, i = _parse_value(toks, i)
            items.append(v)
            kind, tok = toks[i]
            if kind == "punc" and tok == ",":
                i += 1
    # scalar expression: consume up to the next ',' or '}' at depth 0, strip casts
    parts = []
    depth = 0
    while True:
        kind, tok = toks[i]
        if kind == "punc":
            if tok == "(":
                depth += 1
            elif tok == ")":
                depth -= 1
            elif depth == 0 and tok in ",}":
                break
        parts.append((kind, tok))
        i += 1
    # strip casts like (char *) / (long *) / (static_codebook *)
    flat = []
    j = 0
    while j < len(parts):
        k, t = parts[j]
        if k == "punc" and t == "(":
            # find matching ')'
            d, e = 1, j + 1
            while d:
                if parts[e] == ("punc", "("):
                    d += 1
                elif parts[e] == ("punc", ")"):
                    d -= 1
                e += 1
            inner = parts[j + 1:e - 1]
            if inner and all(k2 in ("id",) or t2 == "*" for k2, t2 in inner) and any(t2 == "*" for _, t2 in inner):
                j = e  # pointer cast: drop
                continue
            if inner and all(k2 == "id" for k2, _ in inner) and inner[-1][1] in (
                    "char", "long", "int", "float", "double", "short"):
                j = e  # arithmetic cast: drop
                continue
            flat.extend(inner)
            j = e
            continue
        flat.append((k, t))
        j += 1
    if len(flat) == 1:
        k, t = flat[0]
        if k == "num":
            return _num(t), i
        if k == "id":
            return re.sub(r"\s+", "", t), i
        if k == "str":
            return t, i
    if len(flat) == 2 and flat[0] == ("punc", "-") and flat[1][0] == "num":
        return -_num(flat[1][1]), i
    if len(flat) == 2 and flat[0] == ("punc", "&") and flat[1][0] == "id":
        return "&" + flat[1][1], i
    # general constant expression of numbers: evaluate
    expr = "".join(t.rstrip("fFlLuU") if k == "num" else t for k, t in flat)
    try:
        return eval(expr, {"__builtins__": {}}), i
    except Exception:
        return expr, i


_DECL = re.compile(r"""
    (?:static\s+)?(?:const\s+)?
    (?P<ctype>(?:unsigned\s+)?[A-Za-z_][A-Za-z_0-9]*(?:\s+[A-Za-z_][A-Za-z_0-9]*)*?)
    \s*(?P<ptr>\*?\s*(?:const\s+)?)
    (?P<name>[A-Za-z_][A-Za-z_0-9]*)
    \s*(?P<dims>(?:\[[^\]]*\]\s*)*)
    =\s*\{
""", re.X)


def parse_text(text):
    text = preprocess(text)
    decls = {}
    pos = 0
    while True:
        m = _DECL.search(text, pos)
        if not m:
            break
        # find end of initialiser (matching brace) then ';'
        start = m.end() - 1
        depth = 0
        k = start
        while True:
            c = text[k]
            if c == "{":
                depth += 1
            elif c == "}":
                depth -= 1
                if depth == 0:
                    break
            k += 1
        body = text[start:k + 1]
        toks = list(_tokens(body))
        value, _ = _parse_value(toks, 0)
        ctype = re.sub(r"\b(static|const)\b", "", m.group("ctype")).strip()
        if m.group("ptr").strip().startswith("*"):
            ctype += " *"
        dims = re.findall(r"\[([^\]]*)\]", m.group("dims"))
        decls[m.group("name")] = Decl(ctype, dims, value)
        pos = k + 1
    return decls


def parse_file(path):
    with open(path, "r", errors="replace") as f:
        return parse_text(f.read())
