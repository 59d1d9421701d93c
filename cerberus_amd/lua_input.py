"""Reader for the reference's input files: an AMReX ParmParse `*.inputs` file whose `mfp.lua` entry is a Lua
script (Source/MFP_config.cpp:10-22,40-356,462-475; defaults Source/default.lua:12-52).

No Lua interpreter exists in this image and the reference's vendored Lua may not be copied, so this module
carries its own small tree-walking interpreter for the part of Lua 5.4 that the shipped problem definitions use:
numbers, strings, tables, arithmetic / comparison / logical operators, local and global variables, functions and
closures, `if / while / repeat / for` (numeric and generic), multiple assignment and returns, and the library
entries `math.*`, `table.insert / remove / sort / concat / unpack`, `ipairs`, `pairs`, `next`, `type`,
`tostring`, `tonumber`, `select`, `print`, `error`, `assert`, `pcall`, `string.format / rep / len / sub`,
`dofile`.  Coroutines, metatables, goto, integer division semantics of Lua integers (all numbers are floats here,
as they are once they reach the C++ side through sol2) and the debug library are outside it.

`load_inputs(path)` returns the problem dict `cerberus_amd.config.Problem` takes: geometry keys from the
ParmParse part, the Lua globals of SURVEY.md Appendix A, `states` / `actions` as nested dicts with Lua functions
wrapped as vectorised Python callables f(dat).  Objects outside the built path (particle states, `elliptic`,
`hydro_tracer`, embedded boundaries, refinement) are dropped and listed under `_ignored`.
Start-up only: nothing here is on the hot path.
"""
import math
import os
import re

import numpy as np


class LuaError(RuntimeError):
    pass


# --------------------------------------------------------------------------------------------- lexer
_TOKEN = re.compile(r"""
    (?P<ws>\s+)
  | (?P<lcomment>--\[(?P<lc_eq>=*)\[)
  | (?P<comment>--[^\n]*)
  | (?P<number>0[xX][0-9a-fA-F]+(?:\.[0-9a-fA-F]*)?(?:[pP][+-]?\d+)? | (?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)
  | (?P<name>[A-Za-z_][A-Za-z_0-9]*)
  | (?P<lstring>\[(?P<ls_eq>=*)\[)
  | (?P<string>"(?:\\.|[^"\\])*"|'(?:\\.|[^'\\])*')
  | (?P<op>\.\.\.|\.\.|==|~=|<=|>=|//|::|<<|>>|[-+*/%^\#&~|<>=(){}\[\];:,.])
""", re.X | re.S)
_KEYWORDS = {"and", "break", "do", "else", "elseif", "end", "false", "for", "function", "goto", "if", "in", "local",
             "nil", "not", "or", "repeat", "return", "then", "true", "until", "while"}
_ESC = {"n": "\n", "t": "\t", "r": "\r", "\\": "\\", '"': '"', "'": "'", "a": "\a", "b": "\b", "f": "\f", "v": "\v", "\n": "\n"}


def _unescape(s):
    out, i = [], 0
    while i < len(s):
        c = s[i]
        if c == "\\" and i + 1 < len(s):
            n = s[i + 1]
            if n in _ESC:
                out.append(_ESC[n])
                i += 2
            elif n.isdigit():
                j = i + 1
                while j < len(s) and j < i + 4 and s[j].isdigit():
                    j += 1
                out.append(chr(int(s[i + 1:j])))
                i = j
            elif n == "x":
                out.append(chr(int(s[i + 2:i + 4], 16)))
                i += 4
            elif n == "z":
                i += 2
                while i < len(s) and s[i].isspace():
                    i += 1
            else:
                out.append(n)
                i += 2
        else:
            out.append(c)
            i += 1
    return "".join(out)


def tokenize(src):
    toks, pos, line = [], 0, 1
    n = len(src)
    while pos < n:
        m = _TOKEN.match(src, pos)
        if not m:
            raise LuaError(f"line {line}: unexpected character {src[pos]!r}")
        kind = m.lastgroup
        text = m.group(kind)
        if kind in ("lcomment", "lstring"):
            eq = m.group("lc_eq") if kind == "lcomment" else m.group("ls_eq")
            close = "]" + eq + "]"
            end = src.find(close, m.end())
            if end < 0:
                raise LuaError(f"line {line}: unfinished long {'comment' if kind == 'lcomment' else 'string'}")
            body = src[m.end():end]
            if kind == "lstring":
                toks.append(("string", body[1:] if body.startswith("\n") else body, line))
            line += src.count("\n", pos, end + len(close))
            pos = end + len(close)
            continue
        line_here = line
        line += text.count("\n")
        pos = m.end()
        if kind in ("ws", "comment"):
            continue
        if kind == "number":
            v = float.fromhex(text) if text[:2].lower() == "0x" else float(text)
            toks.append(("number", v, line_here))
        elif kind == "name":
            toks.append(("kw" if text in _KEYWORDS else "name", text, line_here))
        elif kind == "string":
            toks.append(("string", _unescape(text[1:-1]), line_here))
        else:
            toks.append(("op", text, line_here))
    toks.append(("eof", None, line))
    return toks


# --------------------------------------------------------------------------------------------- parser
_BINPRI = {"or": (1, 1), "and": (2, 2), "<": (3, 3), ">": (3, 3), "<=": (3, 3), ">=": (3, 3), "~=": (3, 3), "==": (3, 3),
           "..": (9, 8), "+": (10, 10), "-": (10, 10), "*": (11, 11), "/": (11, 11), "//": (11, 11), "%": (11, 11),
           "^": (14, 13)}
_UNARY_PRI = 12


class Parser:
    def __init__(self, src):
        self.t = tokenize(src)
        self.i = 0

    def peek(self):
        return self.t[self.i]

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def check(self, kind, val=None):
        k, v, _ = self.t[self.i]
        return k == kind and (val is None or v == val)

    def accept(self, kind, val=None):
        if self.check(kind, val):
            return self.next()
        return None

    def expect(self, kind, val=None):
        tok = self.accept(kind, val)
        if tok is None:
            k, v, ln = self.peek()
            raise LuaError(f"line {ln}: expected {val or kind}, found {v!r}")
        return tok

    def block(self):
        stmts = []
        while True:
            k, v, _ = self.peek()
            if k == "eof" or (k == "kw" and v in ("end", "else", "elseif", "until")):
                break
            if k == "kw" and v == "return":
                self.next()
                exprs = []
                if not (self.check("eof") or self.check("op", ";") or (self.peek()[0] == "kw" and self.peek()[1] in ("end", "else", "elseif", "until"))):
                    exprs = self.exprlist()
                self.accept("op", ";")
                stmts.append(("return", exprs))
                break
            s = self.statement()
            if s is not None:
                stmts.append(s)
        return stmts

    def statement(self):
        k, v, ln = self.peek()
        if k == "op" and v == ";":
            self.next()
            return None
        if k == "kw":
            if v == "if":
                self.next()
                clauses = []
                cond = self.expr()
                self.expect("kw", "then")
                clauses.append((cond, self.block()))
                orelse = None
                while True:
                    if self.accept("kw", "elseif"):
                        cond = self.expr()
                        self.expect("kw", "then")
                        clauses.append((cond, self.block()))
                    elif self.accept("kw", "else"):
                        orelse = self.block()
                        self.expect("kw", "end")
                        break
                    else:
                        self.expect("kw", "end")
                        break
                return ("if", clauses, orelse)
            if v == "while":
                self.next()
                cond = self.expr()
                self.expect("kw", "do")
                body = self.block()
                self.expect("kw", "end")
                return ("while", cond, body)
            if v == "do":
                self.next()
                body = self.block()
                self.expect("kw", "end")
                return ("do", body)
            if v == "repeat":
                self.next()
                body = self.block()
                self.expect("kw", "until")
                return ("repeat", body, self.expr())
            if v == "for":
                self.next()
                n1 = self.expect("name")[1]
                if self.accept("op", "="):
                    a = self.expr()
                    self.expect("op", ",")
                    b = self.expr()
                    c = self.expr() if self.accept("op", ",") else None
                    self.expect("kw", "do")
                    body = self.block()
                    self.expect("kw", "end")
                    return ("fornum", n1, a, b, c, body)
                names = [n1]
                while self.accept("op", ","):
                    names.append(self.expect("name")[1])
                self.expect("kw", "in")
                exprs = self.exprlist()
                self.expect("kw", "do")
                body = self.block()
                self.expect("kw", "end")
                return ("forin", names, exprs, body)
            if v == "function":
                self.next()
                target = ("name", self.expect("name")[1])
                is_method = False
                while True:
                    if self.accept("op", "."):
                        target = ("index", target, ("const", self.expect("name")[1]))
                    elif self.accept("op", ":"):
                        target = ("index", target, ("const", self.expect("name")[1]))
                        is_method = True
                        break
                    else:
                        break
                fn = self.funcbody(is_method)
                return ("assign", [target], [fn])
            if v == "local":
                self.next()
                if self.accept("kw", "function"):
                    name = self.expect("name")[1]
                    return ("localfunc", name, self.funcbody(False))
                names = [self.expect("name")[1]]
                if self.accept("op", "<"):      # attribs <const> / <close>
                    self.expect("name")
                    self.expect("op", ">")
                while self.accept("op", ","):
                    names.append(self.expect("name")[1])
                    if self.accept("op", "<"):
                        self.expect("name")
                        self.expect("op", ">")
                exprs = self.exprlist() if self.accept("op", "=") else []
                return ("local", names, exprs)
            if v == "break":
                self.next()
                return ("break",)
            if v == "goto":
                raise LuaError(f"line {ln}: goto is not supported")
        # expression statement: call or assignment
        e = self.suffixedexp()
        if self.check("op", "=") or self.check("op", ","):
            targets = [e]
            while self.accept("op", ","):
                targets.append(self.suffixedexp())
            self.expect("op", "=")
            exprs = self.exprlist()
            for tg in targets:
                if tg[0] not in ("name", "index"):
                    raise LuaError(f"line {ln}: cannot assign to this expression")
            return ("assign", targets, exprs)
        if e[0] not in ("call", "method"):
            raise LuaError(f"line {ln}: syntax error (expression is not a statement)")
        return ("callstat", e)

    def funcbody(self, is_method):
        self.expect("op", "(")
        params, vararg = (["self"] if is_method else []), False
        if not self.check("op", ")"):
            while True:
                if self.accept("op", "..."):
                    vararg = True
                    break
                params.append(self.expect("name")[1])
                if not self.accept("op", ","):
                    break
        self.expect("op", ")")
        body = self.block()
        self.expect("kw", "end")
        return ("function", params, vararg, body)

    def exprlist(self):
        out = [self.expr()]
        while self.accept("op", ","):
            out.append(self.expr())
        return out

    def primaryexp(self):
        k, v, ln = self.next()
        if k == "name":
            return ("name", v)
        if k == "op" and v == "(":
            e = self.expr()
            self.expect("op", ")")
            return ("paren", e)
        raise LuaError(f"line {ln}: unexpected symbol {v!r}")

    def suffixedexp(self):
        e = self.primaryexp()
        while True:
            k, v, _ = self.peek()
            if k == "op" and v == ".":
                self.next()
                e = ("index", e, ("const", self.expect("name")[1]))
            elif k == "op" and v == "[":
                self.next()
                idx = self.expr()
                self.expect("op", "]")
                e = ("index", e, idx)
            elif k == "op" and v == ":":
                self.next()
                name = self.expect("name")[1]
                e = ("method", e, name, self.callargs())
            elif (k == "op" and v in ("(", "{")) or k == "string":
                e = ("call", e, self.callargs())
            else:
                return e

    def callargs(self):
        k, v, _ = self.peek()
        if k == "string":
            self.next()
            return [("const", v)]
        if k == "op" and v == "{":
            return [self.table()]
        self.expect("op", "(")
        args = [] if self.check("op", ")") else self.exprlist()
        self.expect("op", ")")
        return args

    def table(self):
        self.expect("op", "{")
        items = []   # ("pos", expr) | ("key", kexpr, vexpr)
        while not self.check("op", "}"):
            if self.check("op", "["):
                self.next()
                kx = self.expr()
                self.expect("op", "]")
                self.expect("op", "=")
                items.append(("key", kx, self.expr()))
            elif self.check("name") and self.t[self.i + 1][0] == "op" and self.t[self.i + 1][1] == "=":
                name = self.next()[1]
                self.next()
                items.append(("key", ("const", name), self.expr()))
            else:
                items.append(("pos", self.expr()))
            if not (self.accept("op", ",") or self.accept("op", ";")):
                break
        self.expect("op", "}")
        return ("table", items)

    def simpleexp(self):
        k, v, _ = self.peek()
        if k == "number" or k == "string":
            self.next()
            return ("const", v)
        if k == "kw":
            if v == "nil":
                self.next()
                return ("const", None)
            if v == "true":
                self.next()
                return ("const", True)
            if v == "false":
                self.next()
                return ("const", False)
            if v == "function":
                self.next()
                return self.funcbody(False)
        if k == "op" and v == "...":
            self.next()
            return ("vararg",)
        if k == "op" and v == "{":
            return self.table()
        return self.suffixedexp()

    def expr(self, limit=0):
        k, v, _ = self.peek()
        if (k == "kw" and v == "not") or (k == "op" and v in ("-", "#", "~")):
            self.next()
            left = ("unop", v, self.expr(_UNARY_PRI))
        else:
            left = self.simpleexp()
        while True:
            k, v, _ = self.peek()
            if (k == "op" or k == "kw") and v in _BINPRI and _BINPRI[v][0] > limit:
                self.next()
                right = self.expr(_BINPRI[v][1])
                left = ("binop", v, left, right)
            else:
                return left


def parse(src):
    p = Parser(src)
    b = p.block()
    if not p.check("eof"):
        raise LuaError(f"line {p.peek()[2]}: unexpected {p.peek()[1]!r}")
    return b


# ---------------------------------------------------------------------------------------- run time
class LuaTable:
    __slots__ = ("d",)

    def __init__(self):
        self.d = {}

    def get(self, k):
        if isinstance(k, float) and k.is_integer():
            k = int(k)
        return self.d.get(k)

    def set(self, k, v):
        if k is None:
            raise LuaError("table index is nil")
        if isinstance(k, float) and k.is_integer():
            k = int(k)
        if v is None:
            self.d.pop(k, None)
        else:
            self.d[k] = v

    def length(self):
        n = 0
        while (n + 1) in self.d:
            n += 1
        return n


class LuaFunction:
    __slots__ = ("params", "vararg", "body", "env", "interp")

    def __init__(self, params, vararg, body, env, interp):
        self.params, self.vararg, self.body, self.env, self.interp = params, vararg, body, env, interp

    def __call__(self, *args):
        return self.interp.call(self, list(args))


class _Break(Exception):
    pass


class _Return(Exception):
    def __init__(self, values):
        self.values = values


class Scope:
    __slots__ = ("vars", "parent")

    def __init__(self, parent=None):
        self.vars, self.parent = {}, parent

    def find(self, name):
        s = self
        while s is not None:
            if name in s.vars:
                return s
            s = s.parent
        return None


def _truthy(v):
    return v is not None and v is not False


def _tonumber(v, what="arithmetic"):
    if isinstance(v, bool) or v is None:
        raise LuaError(f"attempt to perform {what} on a {lua_type(v)} value")
    if isinstance(v, (int, float)):
        return float(v)
    if isinstance(v, str):
        try:
            return float(v) if not v.strip().lower().startswith("0x") else float.fromhex(v.strip())
        except ValueError:
            pass
    raise LuaError(f"attempt to perform {what} on a {lua_type(v)} value")


def lua_type(v):
    if v is None:
        return "nil"
    if isinstance(v, bool):
        return "boolean"
    if isinstance(v, (int, float)):
        return "number"
    if isinstance(v, str):
        return "string"
    if isinstance(v, LuaTable):
        return "table"
    return "function"


def lua_tostring(v):
    if v is None:
        return "nil"
    if v is True:
        return "true"
    if v is False:
        return "false"
    if isinstance(v, (int, float)):
        f = float(v)
        if f.is_integer() and abs(f) < 1e15:
            return f"{f:.1f}"
        return repr(f) if not math.isinf(f) else ("inf" if f > 0 else "-inf")
    if isinstance(v, str):
        return v
    return f"{lua_type(v)}: 0x{id(v):x}"


# What MFP::set_lua_script puts in front of the user's script (MFP_config.cpp:14-21 prepends Source/default.lua): the
# utility functions of default.lua:117-250 a problem definition may call or override - same names and results,
# written here for this interpreter (get_args / get_num_args need Lua's debug library and are left out) - and the
# empty preprocess() hook that load_inputs / load_lua call after the script, as the reference appends "preprocess()".
PRELUDE = r"""
function spairs(t, order)
  local keys = {}
  for k in pairs(t) do keys[#keys + 1] = k end
  if order then table.sort(keys, function(a, b) return order(t, a, b) end) else table.sort(keys) end
  local i = 0
  return function()
    i = i + 1
    if keys[i] then return keys[i], t[keys[i]] end
  end
end
function len(T)
  local n = 0
  for _ in pairs(T) do n = n + 1 end
  return n
end
function get_sorted_keys(T)
  local keys = {}
  local n = 0
  for k, v in spairs(T) do n = n + 1; keys[n] = k end
  return keys
end
function expand(v, opt)
  if type(v) ~= 'table' then return { v, v or opt } end
  return v
end
function exists(v) if not v then return 0 end return 1 end
function is_function(v) if type(v) == 'function' then return 1 end return 0 end
function is_table(v) if type(v) == 'table' then return 1 end return 0 end
function get_ipairs(v)
  local out = {}
  for _, value in ipairs(v) do table.insert(out, value) end
  return out
end
function preprocess() end
"""


class Interpreter:
    def __init__(self, base_dir="."):
        self.base_dir = base_dir
        self.globals = LuaTable()
        self.output = []
        self._install_stdlib()
        self.run(PRELUDE)

    def run_user_script(self, src):
        """the user's script followed by preprocess() (MFP_config.cpp:14-21)"""
        self.run(src)
        hook = self.globals.get("preprocess")
        if hook is not None:
            self.call_value(hook, [])

    # -- library ----------------------------------------------------------------------------------
    def _install_stdlib(self):
        G = self.globals

        def tbl(d):
            t = LuaTable()
            for k, v in d.items():
                t.set(k, v)
            return t

        def m1(f):
            def g(x, *_):
                try:
                    return f(_tonumber(x))
                except OverflowError:      # C's libm returns inf where Python raises
                    return math.inf
                except ValueError:
                    return math.nan
            return g

        def lua_max(*a):
            return max(_tonumber(x) for x in a)

        def lua_min(*a):
            return min(_tonumber(x) for x in a)

        def lua_log(x, b=None):
            x = _tonumber(x)
            if x == 0:
                return -math.inf
            return math.log(x) if b is None else math.log(x, _tonumber(b))

        def lua_fmod(a, b):
            return math.fmod(_tonumber(a), _tonumber(b))

        rng = np.random.RandomState(0)

        def lua_random(m=None, n=None):
            if m is None:
                return float(rng.random_sample())
            lo, hi = (1, int(_tonumber(m))) if n is None else (int(_tonumber(m)), int(_tonumber(n)))
            return float(rng.randint(lo, hi + 1))

        def lua_sqrt(x):
            x = _tonumber(x)
            return math.sqrt(x) if x >= 0 else math.nan

        def lua_atan(y, x=1.0):
            return math.atan2(_tonumber(y), _tonumber(x))

        G.set("math", tbl({
            "sqrt": lua_sqrt, "abs": m1(abs), "sin": m1(math.sin), "cos": m1(math.cos), "tan": m1(math.tan),
            "asin": m1(math.asin), "acos": m1(math.acos), "atan": lua_atan, "exp": m1(math.exp), "log": lua_log,
            "floor": m1(lambda x: float(math.floor(x))), "ceil": m1(lambda x: float(math.ceil(x))), "max": lua_max,
            "min": lua_min, "fmod": lua_fmod, "pow": lambda a, b: _tonumber(a) ** _tonumber(b), "pi": math.pi,
            "huge": math.inf, "random": lua_random, "randomseed": lambda *a: rng.seed(int(_tonumber(a[0])) if a else 0),
            "tanh": m1(math.tanh), "sinh": m1(math.sinh), "cosh": m1(math.cosh), "maxinteger": float(2 ** 63 - 1),
            "mininteger": float(-2 ** 63), "tointeger": lambda x: (float(x) if float(x).is_integer() else None),
            "modf": lambda x: [float(math.trunc(_tonumber(x))), _tonumber(x) - math.trunc(_tonumber(x))],
            "rad": m1(math.radians), "deg": m1(math.degrees),
        }))

        # `PolySpline` is registered by the reference's embedded-boundary module (modules/eb/MFP_polyspline.cpp,
        # MFP_config.cpp) for geometry scripts.  EB is outside this path (its tables are listed as ignored), but the
        # shipped inputs build splines at load time: the constructor and the builders record their arguments, only
        # a query — which nothing on this path makes — fails.
        def spline_new(*_):
            sp = LuaTable()
            sp.set("segments", LuaTable())

            def add(kind):
                def f(self_, pts=None, *_a):
                    seg = self_.get("segments")
                    seg.set(seg.length() + 1, tbl({"kind": kind, "points": pts}))
                return f

            def query(*_a):
                raise LuaError("PolySpline:query: embedded boundaries are outside the hyperbolic-update path")
            sp.set("addLine", add("line"))
            sp.set("addSpline", add("spline"))
            sp.set("query", query)
            return sp
        G.set("PolySpline", tbl({"new": spline_new}))

        def svg_new(path=None, *_):   # modules/eb/MFP_read_geom.cpp: same role, the file is not read here
            sv = LuaTable()
            sv.set("file", path)

            def query(*_a):
                raise LuaError("ReadSVG:query: embedded boundaries are outside the hyperbolic-update path")
            sv.set("query", query)
            return sv
        G.set("ReadSVG", tbl({"new": svg_new}))

        def t_insert(t, a, b=None, *rest):
            n = t.length()
            if b is None and not rest:
                t.set(n + 1, a)
            else:
                pos = int(_tonumber(a))
                for i in range(n, pos - 1, -1):
                    t.set(i + 1, t.get(i))
                t.set(pos, b)

        def t_remove(t, pos=None):
            n = t.length()
            if n == 0:
                return None
            pos = n if pos is None else int(_tonumber(pos))
            v = t.get(pos)
            for i in range(pos, n):
                t.set(i, t.get(i + 1))
            t.set(n, None)
            return v

        def t_sort(t, comp=None):
            import functools
            n = t.length()
            vals = [t.get(i) for i in range(1, n + 1)]
            if comp is None:
                vals.sort()
            else:
                def lt(a, b):
                    r = self.call_value(comp, [a, b])
                    return bool(r) and _truthy(r[0])

                def cmp(a, b):
                    return -1 if lt(a, b) else (1 if lt(b, a) else 0)
                vals.sort(key=functools.cmp_to_key(cmp))
            for i, v in enumerate(vals):
                t.set(i + 1, v)

        def t_concat(t, sep="", i=1, j=None):
            j = t.length() if j is None else int(_tonumber(j))
            return sep.join(lua_tostring(t.get(k)) for k in range(int(_tonumber(i)), j + 1))

        def t_unpack(t, i=1, j=None):
            j = t.length() if j is None else int(_tonumber(j))
            return [t.get(k) for k in range(int(_tonumber(i)), j + 1)]

        G.set("table", tbl({"insert": t_insert, "remove": t_remove, "sort": t_sort, "concat": t_concat, "unpack": t_unpack}))
        G.set("unpack", t_unpack)

        def lua_next(t, k=None):
            keys = list(t.d.keys())
            if k is None:
                idx = 0
            else:
                if isinstance(k, float) and k.is_integer():
                    k = int(k)
                idx = keys.index(k) + 1
            if idx >= len(keys):
                return [None]
            kk = keys[idx]
            return [float(kk) if isinstance(kk, int) and not isinstance(kk, bool) else kk, t.d[kk]]

        def lua_ipairs(t):
            def it(tt, i):
                i = int(_tonumber(i)) + 1
                v = tt.get(i)
                return [None] if v is None else [float(i), v]
            return [it, t, 0.0]

        def lua_pairs(t):
            return [lua_next, t, None]

        def lua_tonumber(v, base=None):
            try:
                if base is not None:
                    return float(int(str(v).strip(), int(_tonumber(base))))
                return _tonumber(v)
            except (LuaError, ValueError):
                return None

        def lua_select(n, *a):
            if n == "#":
                return float(len(a))
            n = int(_tonumber(n))
            return list(a[n - 1:]) if n > 0 else list(a[n:])

        def lua_error(msg=None, *_):
            raise LuaError(lua_tostring(msg))

        def lua_assert(v=None, msg="assertion failed!", *rest):
            if not _truthy(v):
                raise LuaError(lua_tostring(msg))
            return [v, msg] + list(rest)

        def lua_pcall(f, *a):
            try:
                return [True] + list(self.call_value(f, list(a)))
            except LuaError as e:
                return [False, str(e)]

        def lua_print(*a):
            self.output.append("\t".join(lua_tostring(x) for x in a))

        def lua_dofile(name):
            path = name if os.path.isabs(name) else os.path.join(self.base_dir, name)
            with open(path) as f:
                return self.run(f.read())

        def s_format(fmt, *a):
            a = list(a)
            out, i, k = [], 0, 0
            while i < len(fmt):
                if fmt[i] == "%" and i + 1 < len(fmt):
                    m = re.match(r"%[-+ #0]*\d*(?:\.\d+)?([a-zA-Z%])", fmt[i:])
                    if not m:
                        raise LuaError("invalid format")
                    spec, conv = m.group(0), m.group(1)
                    if conv == "%":
                        out.append("%")
                    else:
                        v = a[k]
                        k += 1
                        if conv in "di":
                            out.append(spec[:-1].replace("i", "d") % int(_tonumber(v)) if spec[:-1] != "%" else str(int(_tonumber(v))))
                        elif conv in "eEfgG":
                            out.append(spec % _tonumber(v))
                        elif conv == "s":
                            out.append(spec % lua_tostring(v))
                        elif conv == "q":
                            out.append('"' + lua_tostring(v) + '"')
                        elif conv in "xX":
                            out.append(spec % int(_tonumber(v)))
                        else:
                            raise LuaError(f"format '%{conv}' is not supported")
                    i += len(spec)
                else:
                    out.append(fmt[i])
                    i += 1
            return "".join(out)

        def s_sub(s, i, j=-1):
            n = len(s)
            i, j = int(_tonumber(i)), int(_tonumber(j))
            i = max(i if i > 0 else n + i + 1, 1)
            j = min(j if j >= 0 else n + j + 1, n)
            return s[i - 1:j] if i <= j else ""

        G.set("string", tbl({"format": s_format, "rep": lambda s, n, sep="": sep.join([s] * int(_tonumber(n))),
                             "len": lambda s: float(len(s)), "sub": s_sub, "upper": lambda s: s.upper(),
                             "lower": lambda s: s.lower()}))
        for name, fn in (("ipairs", lua_ipairs), ("pairs", lua_pairs), ("next", lua_next), ("type", lambda v=None: lua_type(v)),
                         ("tostring", lambda v=None: lua_tostring(v)), ("tonumber", lua_tonumber), ("select", lua_select),
                         ("error", lua_error), ("assert", lua_assert), ("pcall", lua_pcall), ("print", lua_print),
                         ("dofile", lua_dofile), ("rawget", lambda t, k: t.get(k)), ("rawset", lambda t, k, v: t.set(k, v)),
                         ("rawlen", lambda t: float(t.length() if isinstance(t, LuaTable) else len(t))),
                         ("rawequal", lambda a, b: a is b or a == b)):
            G.set(name, fn)
        G.set("_G", G)
        G.set("_VERSION", "Lua 5.4 (cerberus_amd subset)")

    # -- calls ------------------------------------------------------------------------------------
    def call_value(self, f, args):
        """calls a Lua or host function; returns the list of results"""
        if isinstance(f, LuaFunction):
            return self.call(f, args)
        if callable(f):
            r = f(*args)
            if r is None:
                return []
            return r if isinstance(r, list) else [r]
        raise LuaError(f"attempt to call a {lua_type(f)} value")

    def call(self, fn, args):
        scope = Scope(fn.env)
        for i, p in enumerate(fn.params):
            scope.vars[p] = args[i] if i < len(args) else None
        if fn.vararg:
            scope.vars["..."] = args[len(fn.params):]
        try:
            self.exec_block(fn.body, scope)
        except _Return as r:
            return r.values
        return []

    # -- statements -------------------------------------------------------------------------------
    def run(self, src):
        try:
            self.exec_block(parse(src), Scope(None))
        except _Return as r:
            return r.values
        return []

    def exec_block(self, stmts, scope):
        for s in stmts:
            self.exec(s, scope)

    def exec(self, s, scope):
        op = s[0]
        if op == "assign":
            vals = self.eval_list(s[2], scope, len(s[1]))
            for tg, v in zip(s[1], vals):
                self.assign(tg, v, scope)
        elif op == "local":
            vals = self.eval_list(s[2], scope, len(s[1]))
            for n, v in zip(s[1], vals):
                scope.vars[n] = v
        elif op == "callstat":
            self.eval_multi(s[1], scope)
        elif op == "if":
            for cond, body in s[1]:
                if _truthy(self.eval(cond, scope)):
                    self.exec_block(body, Scope(scope))
                    return
            if s[2] is not None:
                self.exec_block(s[2], Scope(scope))
        elif op == "return":
            raise _Return(self.eval_list(s[1], scope, None))
        elif op == "localfunc":
            scope.vars[s[1]] = None
            scope.vars[s[1]] = self.eval(s[2], scope)
        elif op == "fornum":
            a, b = _tonumber(self.eval(s[2], scope), "'for'"), _tonumber(self.eval(s[3], scope), "'for'")
            c = 1.0 if s[4] is None else _tonumber(self.eval(s[4], scope), "'for'")
            if c == 0:
                raise LuaError("'for' step is zero")
            x = a
            try:
                while (c > 0 and x <= b) or (c < 0 and x >= b):
                    inner = Scope(scope)
                    inner.vars[s[1]] = x
                    self.exec_block(s[5], inner)
                    x += c
            except _Break:
                pass
        elif op == "forin":
            vals = self.eval_list(s[2], scope, 3)
            f, st, ctl = vals[0], vals[1], vals[2]
            try:
                while True:
                    r = self.call_value(f, [st, ctl])
                    r = r + [None] * (len(s[1]) - len(r))
                    if r[0] is None:
                        break
                    ctl = r[0]
                    inner = Scope(scope)
                    for n, v in zip(s[1], r):
                        inner.vars[n] = v
                    self.exec_block(s[3], inner)
            except _Break:
                pass
        elif op == "while":
            try:
                while _truthy(self.eval(s[1], scope)):
                    self.exec_block(s[2], Scope(scope))
            except _Break:
                pass
        elif op == "repeat":
            try:
                while True:
                    inner = Scope(scope)
                    self.exec_block(s[1], inner)
                    if _truthy(self.eval(s[2], inner)):
                        break
            except _Break:
                pass
        elif op == "do":
            self.exec_block(s[1], Scope(scope))
        elif op == "break":
            raise _Break()
        else:
            raise LuaError(f"unknown statement {op}")

    def assign(self, target, v, scope):
        if target[0] == "name":
            s = scope.find(target[1])
            if s is not None:
                s.vars[target[1]] = v
            else:
                self.globals.set(target[1], v)
        else:
            obj = self.eval(target[1], scope)
            if not isinstance(obj, LuaTable):
                raise LuaError(f"attempt to index a {lua_type(obj)} value")
            obj.set(self.eval(target[2], scope), v)

    # -- expressions ------------------------------------------------------------------------------
    def eval_list(self, exprs, scope, want):
        """values of an expression list with Lua's multi-value expansion of the last expression; padded or
        truncated to `want` values when given"""
        out = []
        for i, e in enumerate(exprs):
            if i == len(exprs) - 1 and e[0] in ("call", "method", "vararg"):
                out.extend(self.eval_multi(e, scope))
            else:
                out.append(self.eval(e, scope))
        if want is not None:
            out = out[:want] + [None] * (want - len(out))
        return out

    def eval_multi(self, e, scope):
        op = e[0]
        if op == "call":
            f = self.eval(e[1], scope)
            args = self.eval_list(e[2], scope, None)
            if f is None:
                raise LuaError(f"attempt to call a nil value ({self._describe(e[1])})")
            return self.call_value(f, args)
        if op == "method":
            obj = self.eval(e[1], scope)
            f = self.index(obj, e[2])
            if f is None:
                raise LuaError(f"attempt to call a nil value (method '{e[2]}')")
            return self.call_value(f, [obj] + self.eval_list(e[3], scope, None))
        if op == "vararg":
            s = scope.find("...")
            return list(s.vars["..."]) if s else []
        return [self.eval(e, scope)]

    @staticmethod
    def _describe(e):
        if e[0] == "name":
            return f"global '{e[1]}'"
        if e[0] == "index" and e[2][0] == "const":
            return f"field '{e[2][1]}'"
        return "?"

    def index(self, obj, key):
        if isinstance(obj, LuaTable):
            return obj.get(key)
        if isinstance(obj, str):
            return self.globals.get("string").get(key)
        raise LuaError(f"attempt to index a {lua_type(obj)} value (key {key!r})")

    def eval(self, e, scope):
        op = e[0]
        if op == "const":
            return e[1]
        if op == "name":
            s = scope.find(e[1])
            return s.vars[e[1]] if s is not None else self.globals.get(e[1])
        if op == "index":
            return self.index(self.eval(e[1], scope), self.eval(e[2], scope))
        if op in ("call", "method", "vararg"):
            r = self.eval_multi(e, scope)
            return r[0] if r else None
        if op == "paren":
            return self.eval(e[1], scope)
        if op == "function":
            return LuaFunction(e[1], e[2], e[3], scope, self)
        if op == "table":
            t = LuaTable()
            n = 0
            for k, item in enumerate(e[1]):
                if item[0] == "pos":
                    if k == len(e[1]) - 1 and item[1][0] in ("call", "method", "vararg"):
                        for v in self.eval_multi(item[1], scope):
                            n += 1
                            t.set(n, v)
                    else:
                        n += 1
                        t.set(n, self.eval(item[1], scope))
                else:
                    t.set(self.eval(item[1], scope), self.eval(item[2], scope))
            return t
        if op == "binop":
            o = e[1]
            if o == "and":
                a = self.eval(e[2], scope)
                return self.eval(e[3], scope) if _truthy(a) else a
            if o == "or":
                a = self.eval(e[2], scope)
                return a if _truthy(a) else self.eval(e[3], scope)
            a, b = self.eval(e[2], scope), self.eval(e[3], scope)
            if o == "==":
                return self._eq(a, b)
            if o == "~=":
                return not self._eq(a, b)
            if o in ("<", "<=", ">", ">="):
                if isinstance(a, str) and isinstance(b, str):
                    pass
                elif isinstance(a, (int, float)) and isinstance(b, (int, float)) and not isinstance(a, bool) and not isinstance(b, bool):
                    pass
                else:
                    raise LuaError(f"attempt to compare {lua_type(a)} with {lua_type(b)}")
                return a < b if o == "<" else a <= b if o == "<=" else a > b if o == ">" else a >= b
            if o == "..":
                if not isinstance(a, (str, int, float)) or not isinstance(b, (str, int, float)) or isinstance(a, bool) or isinstance(b, bool):
                    raise LuaError(f"attempt to concatenate a {lua_type(a) if not isinstance(a, (str, int, float)) else lua_type(b)} value")
                return lua_tostring(a) + lua_tostring(b)
            x, y = _tonumber(a), _tonumber(b)
            if o == "+":
                return x + y
            if o == "-":
                return x - y
            if o == "*":
                return x * y
            if o == "/":
                if y == 0:
                    return math.nan if (x == 0 or math.isnan(x)) else math.copysign(math.inf, x) * math.copysign(1.0, y)
                return x / y
            if o == "^":
                try:
                    return math.pow(x, y)
                except (OverflowError, ValueError):
                    return math.inf if x > 0 else math.nan
            if o == "%":
                if y == 0:
                    return math.nan
                r = math.fmod(x, y)
                return r + y if r != 0 and (r < 0) != (y < 0) else r
            if o == "//":
                return math.floor(x / y) if y != 0 else (math.nan if x == 0 else math.copysign(math.inf, x) * math.copysign(1.0, y))
            raise LuaError(f"operator {o} is not supported")
        if op == "unop":
            v = self.eval(e[2], scope)
            if e[1] == "not":
                return not _truthy(v)
            if e[1] == "-":
                return -_tonumber(v)
            if e[1] == "#":
                if isinstance(v, str):
                    return float(len(v))
                if isinstance(v, LuaTable):
                    return float(v.length())
                raise LuaError(f"attempt to get length of a {lua_type(v)} value")
            raise LuaError(f"operator {e[1]} is not supported")
        raise LuaError(f"unknown expression {op}")

    @staticmethod
    def _eq(a, b):
        if isinstance(a, bool) or isinstance(b, bool):
            return a is b
        if isinstance(a, (int, float)) and isinstance(b, (int, float)):
            return float(a) == float(b)
        if isinstance(a, str) and isinstance(b, str):
            return a == b
        return a is b


# ------------------------------------------------------------------------------------ ParmParse file
def parse_inputs(text):
    """AMReX ParmParse subset: `key = v1 v2 ...` with `#` comments; a value in double quotes may span lines
    (that is how `mfp.lua` is given).  Returns {key: [tokens]}."""
    out, i, n = {}, 0, len(text)
    key = None
    while i < n:
        c = text[i]
        if c == "#":
            while i < n and text[i] != "\n":
                i += 1
        elif c == '"':
            j = text.find('"', i + 1)
            if j < 0:
                raise ValueError("unterminated string in inputs file")
            if key is None:
                raise ValueError("string without a key in inputs file")
            out[key].append(text[i + 1:j])
            i = j + 1
        elif c.isspace():
            i += 1
        elif c == "=":
            i += 1
        else:
            j = i
            while j < n and not text[j].isspace() and text[j] not in '=#"':
                j += 1
            word = text[i:j]
            k = j
            while k < n and text[k] in " \t":
                k += 1
            if k < n and text[k] == "=":
                key = word
                out[key] = []
                i = k + 1
            else:
                if key is None:
                    raise ValueError(f"value '{word}' without a key in inputs file")
                out[key].append(word)
                i = j
    return out


# ---------------------------------------------------------------------------------- problem dict
_ON_PATH_STATES = ("hydro", "field")
_ON_PATH_ACTIONS = ("CTU", "plasma5", "Lorentz", "collisions_rambo")
_GLOBAL_KEYS = ("cfl", "time_integration_scheme", "force_dt", "lightspeed", "beta", "skin_depth", "Larmor", "Debye",
                "ref_length", "ref_density", "ref_mass", "ref_temp", "verbosity")


def wrap_function(fn, interp):
    """A Lua value function f(dat) (modules/MFP_optional_func.cpp:34-53: dat.x, dat.y, dat.z, dat.t) as the
    vectorised callable of config.Problem: evaluated cell by cell (start-up only)."""
    def call(dat):
        x = np.asarray(dat["x"], dtype=np.float64)
        y = np.broadcast_to(np.asarray(dat.get("y", 0.0), dtype=np.float64), x.shape)
        z = np.broadcast_to(np.asarray(dat.get("z", 0.0), dtype=np.float64), x.shape)
        t = float(dat.get("t", 0.0))
        out = np.empty(x.shape)
        of, xf, yf, zf = out.reshape(-1), x.reshape(-1), y.reshape(-1), z.reshape(-1)
        arg = LuaTable()
        arg.set("t", t)
        for i in range(xf.size):
            arg.d["x"], arg.d["y"], arg.d["z"] = float(xf[i]), float(yf[i]), float(zf[i])
            r = interp.call_value(fn, [arg])
            if not r or not isinstance(r[0], (int, float)) or isinstance(r[0], bool):
                raise LuaError("a value function must return a number")
            of[i] = r[0]
        return out
    call.lua_function = fn
    # A function may depend on time if its own text mentions the key 't' - or if a function it calls does (a helper or
    # an upvalue can reach dat.t out of sight): called Lua functions are followed through the closure's scopes and the
    # globals; a callee that cannot be resolved to a Lua function or a library member counts as time dependent.  Its
    # boundary tables are then refreshed every step: erring on that side costs a re-tabulation, the other side leaves
    # boundary data stale.
    call.uses_time = _may_use_time(fn, interp, set()) if isinstance(fn, LuaFunction) else True
    return call


_LIBRARY_TABLES = ("math", "string", "table")


def _callees(node, out):
    if isinstance(node, tuple):
        if node and node[0] in ("call", "method"):
            out.append(node)
        for x in node:
            _callees(x, out)
    elif isinstance(node, list):
        for x in node:
            _callees(x, out)


def _may_use_time(fn, interp, seen):
    if id(fn) in seen:
        return False
    seen.add(id(fn))
    if _mentions(fn.body, "t"):
        return True
    calls = []
    _callees(fn.body, calls)
    for c in calls:
        if c[0] == "method":
            return True                                   # obj:method(...): not followed
        callee = c[1]
        if isinstance(callee, tuple) and callee[0] == "index" and isinstance(callee[1], tuple) and callee[1][0] == "name" \
                and callee[1][1] in _LIBRARY_TABLES:
            continue                                      # math.* / string.* / table.* members
        if not (isinstance(callee, tuple) and callee[0] == "name"):
            return True
        name = callee[1]
        sc = fn.env.find(name) if isinstance(fn.env, Scope) else None
        target = sc.vars[name] if sc is not None else interp.globals.get(name)
        if isinstance(target, LuaFunction):
            if _may_use_time(target, interp, seen):
                return True
        elif callable(target) and name in ("tostring", "tonumber", "type", "ipairs", "pairs", "select", "print", "assert", "error"):
            continue
        else:
            return True                                   # a parameter, a table member, nil: unknown at this point
    return False


def _mentions(node, key):
    if isinstance(node, tuple):
        if len(node) == 2 and node[0] == "const" and node[1] == key:
            return True
        return any(_mentions(x, key) for x in node)
    if isinstance(node, list):
        return any(_mentions(x, key) for x in node)
    return False


def to_python(v, interp, fn_wrap=True):
    """LuaTable -> dict (string keys) or list (keys 1..n only); functions -> callables"""
    if isinstance(v, LuaTable):
        n = v.length()
        if n and len(v.d) == n:
            return [to_python(v.get(i), interp, fn_wrap) for i in range(1, n + 1)]
        return {(k if isinstance(k, str) else float(k)): to_python(x, interp, fn_wrap) for k, x in v.d.items()}
    if isinstance(v, LuaFunction):
        return wrap_function(v, interp) if fn_wrap else v
    return v


def problem_from_lua(interp, geometry, action_map=None):
    """Builds the dict of config.Problem from the Lua globals (after the script has run) and the ParmParse
    geometry {dim, n_cell, prob_lo, prob_hi, is_periodic}.  action_map renames action types that have an
    on-path stand-in ({'BraginskiiCTU': 'CTU'} for config 1)."""
    G = interp.globals
    cfg = dict(geometry)
    ignored = []
    for k in _GLOBAL_KEYS:
        v = G.get(k)
        if v is not None:
            cfg[k] = v
    states, actions = {}, {}
    st = G.get("states")
    if not isinstance(st, LuaTable) or not st.d:
        raise LuaError("the script defines no states")
    for name, sd in st.d.items():
        d = to_python(sd, interp)
        if d.get("type") not in _ON_PATH_STATES:
            ignored.append(f"states.{name} (type '{d.get('type')}')")
            continue
        for k in ("refinement", "particles", "eb_divergence", "project_divergence"):
            if k in d:
                ignored.append(f"states.{name}.{k}")
                d.pop(k)
        states[name] = d
    ac = G.get("actions")
    for name, ad in (ac.d.items() if isinstance(ac, LuaTable) else []):
        d = to_python(ad, interp)
        t = d.get("type")
        if action_map and t in action_map:
            ignored.append(f"actions.{name}: type '{t}' -> '{action_map[t]}'")
            d["type"] = t = action_map[t]
            d = {k: v for k, v in d.items() if k in ("type", "states", "corner_transport", "solver", "cross_sections", "species")}
            d.setdefault("corner_transport", True)
        if t == "elliptic" and int(d.get("projection", 0)) == 1 and d.get("state") in states:
            # divergence projection (cmfp_elliptic_project_bc): periodic directions and outflow / symmetry / inflow
            # faces; a no-op in 1-D as in the reference (MFP_elliptic.cpp:640-650)
            actions[name] = d
            continue
        if t not in _ON_PATH_ACTIONS:
            ignored.append(f"actions.{name} (type '{t}')")
            continue
        d["states"] = [s for s in d.get("states", [])]
        missing = [s for s in d["states"] if s not in states]
        if missing:
            raise LuaError(f"action '{name}' names states that are not Eulerian states on the built path: {missing}")
        actions[name] = d
    eb = G.get("embedded_boundaries")
    if isinstance(eb, LuaTable) and eb.d:
        ignored.append("embedded_boundaries")
    cfg["states"], cfg["actions"], cfg["_ignored"] = states, actions, ignored
    return cfg


def load_inputs(path, n_cell=None, action_map=None, max_level0_only=True):
    """Reads `path` (a *.inputs file), runs its `mfp.lua` script (dofile relative to the file's directory) and
    returns (cfg, info): cfg for config.Problem; info = {'stop_time', 'max_step', 'ignored', 'output'}.
    n_cell overrides amr.n_cell (the shipped inputs give a coarse level 0 refined by AMR; this path is single
    level)."""
    with open(path) as f:
        pp = parse_inputs(f.read())
    base = os.path.dirname(os.path.abspath(path))

    def nums(key, default=None):
        if key not in pp:
            if default is None:
                raise ValueError(f"inputs file has no '{key}'")
            return default
        return [float(v) for v in pp[key]]

    cells = [int(v) for v in (n_cell if n_cell is not None else nums("amr.n_cell"))]
    # AMREX_SPACEDIM is a build-time constant of the reference; here it is the length of amr.n_cell
    dim = len(cells)
    geometry = {"dim": dim, "n_cell": cells, "prob_lo": nums("geometry.prob_lo")[:dim], "prob_hi": nums("geometry.prob_hi")[:dim],
                "is_periodic": [int(v) for v in nums("geometry.is_periodic", [0.0] * dim)[:dim]]}
    if int(nums("geometry.coord_sys", [0.0])[0]) != 0:
        raise ValueError("only Cartesian coordinates (geometry.coord_sys = 0)")
    interp = Interpreter(base)
    script = "\n".join(pp.get("mfp.lua", []))
    if not script.strip():
        raise ValueError("inputs file has no mfp.lua script")
    interp.run_user_script(script)
    cfg = problem_from_lua(interp, geometry, action_map)
    info = {"stop_time": nums("stop_time", [-1.0])[0], "max_step": int(nums("max_step", [-1.0])[0]),
            "ignored": cfg["_ignored"], "output": interp.output, "interpreter": interp}
    return cfg, info


def load_lua(text, geometry, base_dir=".", action_map=None):
    """The same for a Lua script given as text."""
    interp = Interpreter(base_dir)
    interp.run_user_script(text)
    return problem_from_lua(interp, geometry, action_map), interp
