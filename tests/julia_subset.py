"""A small interpreter for the numeric subset of Julia that julia/PhiloxRNG.jl and julia/amc_tables.jl are written in.

TEST INFRASTRUCTURE.  There is no Julia in this image or on the GPU box, so the Julia side of the boundary cannot be
executed; what this module does is run the ARITHMETIC of julia/PhiloxRNG.jl -- the draw schedule offered to the
reference's `R=` hook (src/metropolis.jl:245,263) -- from its actual source text, with Julia's own rules where they
differ from Python's or C's:
  * operator precedence (shifts above `*`; `&` and `%` with `*`; `|` and xor with `+`; comparisons below all of them),
  * fixed-width integers with Julia's literal typing (0xF is UInt8, 0xFFF UInt16, 0x00ffffff UInt32, 3 Int64),
    promotion (UInt32 with UInt64 -> UInt64, Int64 with UInt64 -> UInt64, UInt16 with Int64 -> Int64), wrap-around
    arithmetic, `x % T` as truncation and `T(x)` as a CHECKED conversion (InexactError when the value does not fit),
  * 1-based indexing, tuples and destructuring, `a ? b : c`, `cond || error(...)`, `for _ in 1:10`, if / elseif,
  * `mutable struct T{P,Q}` with converting constructors, methods selected by arity and by literal type parameters
    (`rng::PhiloxRNG{SEED,1}` ... `where {SEED}`),
  * Float64 arithmetic as IEEE doubles with a correctly rounded `fma` (exact rational arithmetic, one rounding).
It is NOT a Julia implementation: anything outside this subset raises JuliaSubsetError, and passing here says nothing about
parts of the language it does not model (module system, Random's sampler dispatch, compilation).  tests/test_julia_philox.py
uses it to compare the stub's words, uniforms, normals and call order with the CPU oracle, bit for bit.
"""
from __future__ import annotations

import math
import os
import re
import struct
from fractions import Fraction


class JuliaSubsetError(Exception):
    pass


class JuliaError(Exception):
    """error(...) or an InexactError raised by the interpreted program."""


# ---- values ----------------------------------------------------------------------------------------------------------
class JInt:
    __slots__ = ("v", "bits", "signed")

    def __init__(self, v, bits=64, signed=True):
        m = 1 << bits
        v %= m
        if signed and v >= m >> 1:
            v -= m
        self.v, self.bits, self.signed = v, bits, signed

    @property
    def tname(self):
        return ("Int" if self.signed else "UInt") + str(self.bits)

    def __repr__(self):
        return f"{self.tname}({self.v})"


INT_TYPES = {f"{'U' if not s else ''}Int{b}": (b, s) for b in (8, 16, 32, 64) for s in (True, False)}
INT_TYPES["Int"] = (64, True)
INT_TYPES["UInt"] = (64, False)


def convert_int(tname, x, checked=True):
    bits, signed = INT_TYPES[tname]
    if isinstance(x, JInt):
        val = x.v
    elif isinstance(x, bool):
        val = int(x)
    elif isinstance(x, float):
        if x != math.floor(x):
            raise JuliaError(f"InexactError: {tname}({x})")
        val = int(x)
    else:
        raise JuliaSubsetError(f"cannot convert {x!r} to {tname}")
    lo, hi = (-(1 << (bits - 1)), (1 << (bits - 1)) - 1) if signed else (0, (1 << bits) - 1)
    if checked and not (lo <= val <= hi):
        raise JuliaError(f"InexactError: {tname}({val})")
    return JInt(val, bits, signed)


def promote(a: JInt, b: JInt):
    if a.bits == b.bits:
        return a.bits, a.signed and b.signed          # Int64 with UInt64 -> UInt64
    big = a if a.bits > b.bits else b
    return big.bits, big.signed


def fma(a, b, c):
    for v in (a, b, c):
        if v != v or v in (math.inf, -math.inf):
            return a * b + c
    exact = Fraction(a) * Fraction(b) + Fraction(c)
    if exact == 0:
        # sign of an exact zero: (+0) unless both the product and c are negative zeros
        prod_neg = (math.copysign(1.0, a) * math.copysign(1.0, b)) < 0
        return -0.0 if (prod_neg and math.copysign(1.0, c) < 0 and a * b == 0 and c == 0) else 0.0
    return float(exact)


class Struct:
    def __init__(self, tdef, params, values):
        self.tdef, self.params = tdef, params
        self.fields = dict(zip([f for f, _ in tdef["fields"]], values))


class Range:
    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi


# ---- tokenizer -------------------------------------------------------------------------------------------------------
TOKEN = re.compile(r"""
    (?P<ws>[ \t]+) | (?P<comment>\#[^\n]*) | (?P<nl>\n) |
    (?P<hexfloat>0x[0-9a-fA-F]*\.?[0-9a-fA-F]*p[+-]?\d+) |
    (?P<hexint>0x[0-9a-fA-F]+) |
    (?P<float>\d+\.\d*(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+) |
    (?P<int>\d+) |
    (?P<str>"(?:[^"\\]|\\.)*") |
    (?P<id>[A-Za-z_][A-Za-z_0-9]*!?) |
    (?P<op><<=|>>=|>>>|<<|>>|==|!=|<=|>=|&&|\|\||\+=|-=|\*=|&=|\|=|⊻=|<:|::|[-+*/%^⊻&|<>=?:,;()\[\]{}.!])
""", re.X)


def tokenize(text):
    out, pos, depth = [], 0, 0
    while pos < len(text):
        m = TOKEN.match(text, pos)
        if not m:
            raise JuliaSubsetError(f"cannot tokenize at {text[pos:pos + 30]!r}")
        pos = m.end()
        kind = m.lastgroup
        if kind in ("ws", "comment"):
            continue
        tok = m.group(kind)
        if kind == "op" and tok in "([{":
            depth += 1
        elif kind == "op" and tok in ")]}":
            depth -= 1
        if kind == "nl":
            if depth == 0:
                out.append(("nl", "\n"))
            continue
        out.append((kind, tok))
    out.append(("eof", ""))
    return out


# ---- parser (expressions: precedence climbing with Julia's table) ----------------------------------------------------
BINARY = [  # low -> high
    ("||",), ("&&",), ("==", "!=", "<", "<=", ">", ">="), (":",), ("+", "-", "|", "⊻"), ("*", "/", "%", "&"), ("<<", ">>"),
]


class Parser:
    def __init__(self, tokens):
        self.t, self.i = tokens, 0
        self.in_ternary = 0

    def peek(self, k=0):
        return self.t[self.i + k]

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def accept(self, val):
        if self.peek()[1] == val and self.peek()[0] in ("op", "id"):
            self.i += 1
            return True
        return False

    def expect(self, val):
        if not self.accept(val):
            raise JuliaSubsetError(f"expected {val!r}, got {self.peek()}")

    def skip_nl(self):
        while self.peek()[0] == "nl" or self.peek()[1] == ";":
            self.i += 1

    # -- statements
    def block(self, terminators=("end",)):
        body = []
        self.skip_nl()
        while not (self.peek()[0] == "id" and self.peek()[1] in terminators) and self.peek()[0] != "eof":
            body.append(self.statement())
            self.skip_nl()
        return body

    def statement(self):
        kind, tok = self.peek()
        if kind == "id":
            if tok == "module":
                self.next(); self.next()
                body = self.block()
                self.expect("end")
                return ("block", body)
            if tok in ("using", "export", "import"):
                while self.peek()[0] not in ("nl", "eof"):
                    self.next()
                return ("nop",)
            if tok == "include":
                self.next(); self.expect("(")
                path = self.next()[1][1:-1]
                self.expect(")")
                return ("include", path)
            if tok == "const":
                self.next()
                name = self.next()[1]
                self.expect("=")
                return ("assign", ("name", name), self.expr())
            if tok == "mutable" or tok == "struct":
                return self.struct_def()
            if tok == "function":
                return self.function_def()
            if tok == "return":
                self.next()
                if self.peek()[0] in ("nl", "eof") or self.peek()[1] in (";", "end"):
                    return ("return", None)
                return ("return", self.expr_list())
            if tok == "for":
                self.next()
                var = self.next()[1]
                self.expect("in")
                it = self.expr()
                body = self.block()
                self.expect("end")
                return ("for", var, it, body)
            if tok == "if":
                self.next()
                arms = []
                cond = self.expr()
                body = self.block(("elseif", "else", "end"))
                arms.append((cond, body))
                orelse = []
                while True:
                    if self.accept("elseif"):
                        c = self.expr()
                        arms.append((c, self.block(("elseif", "else", "end"))))
                    elif self.accept("else"):
                        orelse = self.block()
                    else:
                        break
                self.expect("end")
                return ("if", arms, orelse)
        if kind == "str":                       # docstring in front of a definition
            self.next()
            return ("nop",)
        # short-form method definition?  name(params) [where {...}] = expr
        save = self.i
        sig = self.try_signature()
        if sig is not None and self.accept("="):
            return ("function", sig, [("return", self.expr_list())])
        self.i = save
        lhs = self.expr_list()
        for op in ("=", "+=", "-=", "*=", "&=", "|=", "⊻="):
            if self.accept(op):
                rhs = self.expr_list()
                if op != "=":
                    rhs = ("bin", op[:-1], lhs, rhs)
                return ("assign", lhs, rhs)
        return ("expr", lhs)

    def struct_def(self):
        if self.accept("mutable"):
            pass
        self.expect("struct")
        name = self.next()[1]
        params = []
        if self.accept("{"):
            while not self.accept("}"):
                params.append(self.next()[1])
                self.accept(",")
        if self.accept("<:"):
            self.postfix()
        fields = []
        self.skip_nl()
        while not self.accept("end"):
            fname = self.next()[1]
            self.expect("::")
            fields.append((fname, self.next()[1]))
            self.skip_nl()
        return ("struct", name, params, fields)

    def try_signature(self):
        """name[{P,...}](param[::Type], ...) [where {P,...}]  ->  dict, or None when this is not a signature."""
        if self.peek()[0] != "id":
            return None
        name = self.next()[1]
        while self.accept("."):                                  # Random.rand
            name = self.next()[1]
        tparams = None
        if self.accept("{"):
            tparams = []
            while not self.accept("}"):
                tparams.append(self.next()[1])
                self.accept(",")
        if not self.accept("("):
            return None
        params = []
        while not self.accept(")"):
            pname = None
            if self.peek()[1] != "::":
                pname = self.next()[1]
            ptype = None
            if self.accept("::"):
                ptype = self.type_expr()
            params.append((pname, ptype))
            if not self.accept(","):
                if self.peek()[1] != ")":
                    return None
        where = []
        if self.accept("where"):
            self.expect("{")
            while not self.accept("}"):
                where.append(self.next()[1])
                self.accept(",")
        return dict(name=name, tparams=tparams, params=params, where=where)

    def type_expr(self):
        name = self.next()[1]
        while self.accept("."):
            name = self.next()[1]
        args = None
        if self.accept("{"):
            args = []
            while not self.accept("}"):
                if self.peek()[0] in ("int",):
                    args.append(int(self.next()[1]))
                else:
                    args.append(self.type_expr())
                self.accept(",")
        return (name, args)

    def function_def(self):
        self.expect("function")
        sig = self.try_signature()
        if sig is None:
            raise JuliaSubsetError("unsupported function signature")
        body = self.block()
        self.expect("end")
        return ("function", sig, body)

    # -- expressions
    def expr_list(self):
        first = self.expr()
        if self.peek()[1] == "," and self.peek()[0] == "op":
            items = [first]
            while self.accept(","):
                items.append(self.expr())
            return ("tuple", items)
        return first

    def expr(self):
        return self.ternary()

    def ternary(self):
        cond = self.binary(0)
        if self.accept("?"):
            self.in_ternary += 1                         # a ':' in the first branch ends it; it is not a range
            a = self.ternary()
            self.in_ternary -= 1
            self.expect(":")
            b = self.ternary()
            return ("ternary", cond, a, b)
        return cond

    def binary(self, level):
        if level == len(BINARY):
            return self.unary()
        ops = BINARY[level]
        if ops == (":",):
            lhs = self.binary(level + 1)
            if self.peek()[1] == ":" and self.peek()[0] == "op" and not self.in_ternary:
                self.next()
                return ("range", lhs, self.binary(level + 1))
            return lhs
        lhs = self.binary(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in ops:
            op = self.next()[1]
            rhs = self.binary(level + 1)
            lhs = ("bin", op, lhs, rhs)
        return lhs

    def unary(self):
        if self.peek()[0] == "op" and self.peek()[1] in ("-", "+", "!"):
            op = self.next()[1]
            return ("un", op, self.unary())
        return self.power()

    def power(self):
        base = self.postfix()
        if self.accept("^"):
            return ("bin", "^", base, self.unary())     # right-associative, exponent may carry a sign
        return base

    def postfix(self):
        node = self.atom()
        while True:
            if self.peek()[1] == "(" and self.peek()[0] == "op":
                self.next()
                args = []
                while not self.accept(")"):
                    args.append(self.expr())
                    self.accept(",")
                node = ("call", node, args)
            elif self.peek()[1] == "[" and self.peek()[0] == "op":
                self.next()
                if node[0] == "name" and node[1] == "Float64":            # typed array literal
                    items = []
                    while not self.accept("]"):
                        items.append(self.expr())
                        self.accept(",")
                    node = ("array", items)
                else:
                    idx = self.expr()
                    self.expect("]")
                    node = ("index", node, idx)
            elif self.peek()[1] == "." and self.peek()[0] == "op":
                self.next()
                node = ("field", node, self.next()[1])
            elif self.peek()[1] == "{" and self.peek()[0] == "op" and node[0] == "name":
                self.next()
                params = []
                while not self.accept("}"):
                    params.append(self.expr())
                    self.accept(",")
                node = ("curly", node[1], params)
            else:
                return node

    def atom(self):
        kind, tok = self.next()
        if kind == "int":
            return ("lit", JInt(int(tok), 64, True))
        if kind == "hexint":
            digits = len(tok) - 2
            bits = 8 if digits <= 2 else 16 if digits <= 4 else 32 if digits <= 8 else 64
            if digits > 16:
                raise JuliaSubsetError("UInt128 literal")
            return ("lit", JInt(int(tok, 16), bits, False))
        if kind == "float":
            return ("lit", float(tok))
        if kind == "hexfloat":
            return ("lit", float.fromhex(tok))
        if kind == "str":
            return ("lit", tok[1:-1])
        if kind == "id":
            if tok == "true":
                return ("lit", True)
            if tok == "false":
                return ("lit", False)
            return ("name", tok)
        if kind == "op" and tok == "(":
            first = self.expr()
            if self.accept(","):
                items = [first]
                while not self.accept(")"):
                    items.append(self.expr())
                    self.accept(",")
                return ("tuple", items)
            self.expect(")")
            return first
        raise JuliaSubsetError(f"unexpected token {kind} {tok!r}")


# ---- evaluator -------------------------------------------------------------------------------------------------------
class Return(Exception):
    def __init__(self, value):
        self.value = value


class Interpreter:
    def __init__(self, base_dir):
        self.base_dir = base_dir
        self.globals = {}
        self.methods = {}
        self.structs = {}

    def load(self, path):
        text = open(os.path.join(self.base_dir, path), encoding="utf-8").read()
        for st in Parser(tokenize(text)).block(()):
            self.exec(st, self.globals)

    # -- statements
    def exec(self, st, env):
        k = st[0]
        if k == "nop":
            return
        if k == "block":
            for s in st[1]:
                self.exec(s, env)
        elif k == "include":
            self.load(st[1])
        elif k == "struct":
            self.structs[st[1]] = dict(name=st[1], params=st[2], fields=st[3])
        elif k == "function":
            self.methods.setdefault(st[1]["name"], []).append((st[1], st[2]))
        elif k == "assign":
            self.assign(st[1], self.eval(st[2], env), env)
        elif k == "expr":
            self.eval(st[1], env)
        elif k == "return":
            raise Return(None if st[1] is None else self.eval(st[1], env))
        elif k == "for":
            it = self.eval(st[2], env)
            if not isinstance(it, Range):
                raise JuliaSubsetError("for over something that is not a range")
            for v in range(it.lo.v, it.hi.v + 1):
                env[st[1]] = JInt(v)
                for s in st[3]:
                    self.exec(s, env)
        elif k == "if":
            for cond, body in st[1]:
                if self.truth(self.eval(cond, env)):
                    for s in body:
                        self.exec(s, env)
                    return
            for s in st[2]:
                self.exec(s, env)
        else:
            raise JuliaSubsetError(f"statement {k}")

    def assign(self, target, value, env):
        if target[0] == "name":
            env[target[1]] = value
        elif target[0] == "tuple":
            if not isinstance(value, tuple) or len(value) != len(target[1]):
                raise JuliaError("destructuring: wrong number of values")
            for t, v in zip(target[1], value):
                self.assign(t, v, env)
        elif target[0] == "field":
            obj = self.eval(target[1], env)
            ftype = dict(obj.tdef["fields"])[target[2]]
            obj.fields[target[2]] = convert_int(ftype, value) if ftype in INT_TYPES else value
        else:
            raise JuliaSubsetError(f"assignment target {target[0]}")

    @staticmethod
    def truth(v):
        if not isinstance(v, bool):
            raise JuliaError(f"TypeError: non-boolean ({v!r}) used in boolean context")
        return v

    # -- expressions
    def eval(self, e, env):
        k = e[0]
        if k == "lit":
            return e[1]
        if k == "name":
            if e[1] in env:
                return env[e[1]]
            if e[1] in self.globals:
                return self.globals[e[1]]
            if e[1] in INT_TYPES or e[1] in ("Float64",) or e[1] in self.structs:
                return ("type", e[1])
            raise JuliaError(f"UndefVarError: {e[1]}")
        if k == "tuple":
            return tuple(self.eval(x, env) for x in e[1])
        if k == "array":
            return [self.eval(x, env) for x in e[1]]
        if k == "range":
            return Range(self.eval(e[1], env), self.eval(e[2], env))
        if k == "ternary":
            return self.eval(e[2] if self.truth(self.eval(e[1], env)) else e[3], env)
        if k == "un":
            v = self.eval(e[2], env)
            if e[1] == "!":
                return not self.truth(v)
            if isinstance(v, float):
                return -v if e[1] == "-" else v
            if isinstance(v, JInt):
                return JInt(-v.v if e[1] == "-" else v.v, v.bits, v.signed)
            raise JuliaSubsetError("unary on " + repr(v))
        if k == "bin":
            op = e[1]
            if op == "&&":
                return self.truth(self.eval(e[2], env)) and self.truth(self.eval(e[3], env))
            if op == "||":
                return self.truth(self.eval(e[2], env)) or self.truth(self.eval(e[3], env))
            a = self.eval(e[2], env)
            if op == "%" and e[3][0] == "name" and e[3][1] in INT_TYPES:          # x % UInt32: truncation
                return convert_int(e[3][1], a, checked=False)
            return self.binop(op, a, self.eval(e[3], env))
        if k == "index":
            obj, idx = self.eval(e[1], env), self.eval(e[2], env)
            if not isinstance(idx, JInt) or not isinstance(obj, (tuple, list)):
                raise JuliaSubsetError("indexing")
            if not 1 <= idx.v <= len(obj):
                raise JuliaError(f"BoundsError: index {idx.v} of {len(obj)}")
            return obj[idx.v - 1]
        if k == "field":
            return self.eval(e[1], env).fields[e[2]]
        if k == "curly":
            return ("ptype", e[1], [self.eval(p, env) for p in e[2]])
        if k == "call":
            return self.call(e[1], [self.eval(a, env) for a in e[2]], env)
        raise JuliaSubsetError(f"expression {k}")

    def binop(self, op, a, b):
        if op in ("==", "!=", "<", "<=", ">", ">="):
            x = a.v if isinstance(a, JInt) else a
            y = b.v if isinstance(b, JInt) else b
            return {"==": x == y, "!=": x != y, "<": x < y, "<=": x <= y, ">": x > y, ">=": x >= y}[op]
        if isinstance(a, JInt) and isinstance(b, JInt):
            if op in ("<<", ">>"):
                if op == "<<":
                    return JInt(a.v << b.v if b.v < 2 * a.bits else 0, a.bits, a.signed)
                return JInt(a.v >> b.v, a.bits, a.signed)            # arithmetic for signed, logical for unsigned (v >= 0)
            if op == "/":
                return a.v / b.v
            if op == "^":
                if b.v < 0:
                    raise JuliaError("DomainError: integer to a negative power")
                bits, signed = a.bits, a.signed
                return JInt(a.v ** b.v, bits, signed)
            bits, signed = promote(a, b)
            x, y = JInt(a.v, bits, signed).v, JInt(b.v, bits, signed).v
            if op == "%":
                if y == 0:
                    raise JuliaError("DivideError")
                r = abs(x) % abs(y)
                return JInt(-r if x < 0 else r, bits, signed)
            val = {"+": x + y, "-": x - y, "*": x * y, "&": x & y, "|": x | y, "⊻": x ^ y}[op]
            return JInt(val, bits, signed)
        fa = float(a.v) if isinstance(a, JInt) else a
        fb = float(b.v) if isinstance(b, JInt) else b
        if isinstance(fa, float) and isinstance(fb, float):
            if op == "^":
                return fa ** (b.v if isinstance(b, JInt) else fb)
            if op == "/":
                if fb == 0.0:
                    return math.nan if fa == 0.0 or fa != fa else math.copysign(math.inf, fa) * math.copysign(1.0, fb)
                return fa / fb
            if op in ("+", "-", "*"):
                return {"+": fa + fb, "-": fa - fb, "*": fa * fb}[op]
        raise JuliaSubsetError(f"operator {op} on {a!r}, {b!r}")

    # -- calls
    def call(self, fnode, args, env):
        if fnode[0] == "curly":                                              # PhiloxRNG{42,1}(...)
            params = [self.eval(p, env) for p in fnode[2]]
            return self.construct(fnode[1], params, args)
        if fnode[0] == "field":                                              # Random.rand(...)
            return self.call(("name", fnode[2]), args, env)
        name = fnode[1]
        if name in INT_TYPES:
            return convert_int(name, args[0])
        if name == "Float64":
            x = args[0]
            return float(x.v) if isinstance(x, JInt) else float(x)
        if name == "reinterpret":
            ttype, x = args
            if ttype == ("type", "Float64"):
                return struct.unpack("<d", struct.pack("<Q", JInt(x.v, 64, False).v))[0]
            if ttype == ("type", "UInt64"):
                return JInt(struct.unpack("<Q", struct.pack("<d", x))[0], 64, False)
            raise JuliaSubsetError("reinterpret to " + repr(ttype))
        if name == "fma":
            return fma(*[float(v.v) if isinstance(v, JInt) else v for v in args])
        if name == "sqrt":
            x = args[0]
            if x < 0:
                raise JuliaError("DomainError: sqrt of a negative number")
            return math.sqrt(x)                                              # -0.0 -> -0.0, like Julia
        if name == "divrem":
            a, b = args
            bits, signed = promote(a, b)
            x, y = JInt(a.v, bits, signed).v, JInt(b.v, bits, signed).v
            q = abs(x) // abs(y) * (1 if (x < 0) == (y < 0) else -1)
            return (JInt(q, bits, signed), JInt(x - q * y, bits, signed))
        if name == "error":
            raise JuliaError(str(args[0]) if args else "error")
        if name in self.methods:
            return self.invoke(name, args)
        if name in self.structs:
            return self.construct(name, None, args)
        raise JuliaError(f"UndefVarError: {name}")

    def construct(self, name, params, args):
        tdef = self.structs[name]
        for sig, body in self.methods.get(name, []):                         # outer constructors first, by arity
            if len(sig["params"]) == len(args) and len(args) != len(tdef["fields"]):
                local = dict(zip(sig["tparams"] or [], params or []))
                local.update({p[0]: a for p, a in zip(sig["params"], args) if p[0]})
                return self.run(body, local)
        if len(args) != len(tdef["fields"]):
            raise JuliaError(f"MethodError: no constructor {name} with {len(args)} arguments")
        vals = [convert_int(ft, a) if ft in INT_TYPES else a for (_, ft), a in zip(tdef["fields"], args)]
        return Struct(tdef, list(params or []), vals)

    def matches(self, sig, args, bind):
        if len(sig["params"]) != len(args):
            return False
        for (pname, ptype), a in zip(sig["params"], args):
            if ptype is None:
                continue
            tname, targs = ptype
            if isinstance(a, Struct):
                if tname != a.tdef["name"]:
                    return False
                for want, have in zip(targs or [], a.params):
                    if isinstance(want, int):
                        if not isinstance(have, JInt) or have.v != want:
                            return False
                    elif isinstance(want, tuple) and want[0] in sig["where"]:
                        bind[want[0]] = have
            elif tname in INT_TYPES:
                if not (isinstance(a, JInt) and a.tname == ("Int64" if tname == "Int" else "UInt64" if tname == "UInt" else tname)):
                    return False
            elif tname == "Integer":
                if not isinstance(a, JInt):
                    return False
            elif tname == "Float64":
                if not isinstance(a, float):
                    return False
            elif tname == "NTuple":
                n, (et, _) = targs
                if not (isinstance(a, tuple) and len(a) == n and all(isinstance(v, JInt) and v.tname == et for v in a)):
                    return False
            # anything else (Type{Float64}, Random.SamplerTrivial{...}): a dispatch tag the tests pass as a placeholder
        return True

    def invoke(self, name, args):
        for sig, body in self.methods[name]:
            bind = {}
            if self.matches(sig, args, bind):
                local = dict(bind)
                local.update({p[0]: a for p, a in zip(sig["params"], args) if p[0]})
                return self.run(body, local)
        raise JuliaError(f"MethodError: no method matching {name}({', '.join(getattr(a, 'tname', type(a).__name__) for a in args)})")

    def run(self, body, local):
        try:
            for st in body:
                self.exec(st, local)
        except Return as r:
            return r.value
        return None


def load_philox_stub():
    """The interpreter with julia/PhiloxRNG.jl (and the tables it includes) loaded."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "julia", "src")
    it = Interpreter(root)
    it.load("PhiloxRNG.jl")
    return it
