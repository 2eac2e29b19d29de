"""Host mirror of Madarch.Exprs (reference madarch/madarch-exprs.ads:12-125): the expression
trees a user-defined primitive kind states its Distance, Normal and Material with.

Where the reference turns a tree into GLSL text (To_GLSL, madarch-exprs.adb:325-711) that the
driver compiles, this back end turns it into an MDH_X register program (include/madarch_hip.h)
that the HIP kernels interpret: `compile_program`.  The lowering follows the GLSL the reference
would emit, operation for operation, in the order DESIGN.md section 5 fixes.

The tree is built with the reference's names: Literal, Value_Identifier, Struct_Identifier,
Construct_Vector3, the arithmetic and comparison operators, Dot, Cross, Min, Max, Clamp, Length,
Normalize, Abs_Value, To_Float, Sign, Floor, Sqrt, Dot2, Sin, Cos, Tan, Asin, Acos, Atan, Get,
If_Then_Else, Let_In, Forward_Difference (Madarch.Exprs.Derivatives).
"""
import struct

import numpy as np

from . import values

Vector3_Kind, Float_Kind, Int_Kind = values.Vector3_Kind, values.Float_Kind, values.Int_Kind
_Bool_Kind = 3  # comparison results: only usable as an If_Then_Else condition


class Type_Inference_Error(Exception):  # exprs.ads:125
    pass


class Unsupported_Expr(Exception):
    """A node the MDH_X programs cannot express (External_Call, integer arithmetic, an expression
    that needs more than 64 registers)."""


# ---- opcodes: include/madarch_hip.h, enum MDH_X_*
X_LIT, X_MOV, X_COMP, X_POINT, X_ADD, X_SUB, X_MUL, X_DIV, X_DIVF, X_NEG, X_ABS, X_FLOOR, X_SIGN = range(13)
X_MIN, X_MAX, X_SQRT, X_POW, X_LT, X_GT, X_LE, X_GE, X_SEL, X_ITOF, X_ACOS = range(13, 24)
X_SIN, X_COS, X_TAN, X_ASIN, X_ATAN = range(24, 29)
X_REGS, X_MAX_WORDS = 64, 4096


def image_roundtrip(x):
    """A float literal as it survives Single'Image in the generated GLSL text
    (madarch-exprs.adb:330-336): 6 significant digits."""
    x = np.float32(x)
    if not np.isfinite(x):
        return x
    return np.float32(float("%.5E" % float(x)))


# ------------------------------------------------------------------------------- the tree
class Expr:
    # ---- operators (exprs.ads:51-69)
    def __add__(self, r): return Bin_Op("+", self, _lift(r))
    def __sub__(self, r): return Bin_Op("-", self, _lift(r))
    def __mul__(self, r): return Bin_Op("*", self, _lift(r))
    def __truediv__(self, r): return Bin_Op("/", self, _lift(r))
    def __pow__(self, r): return Builtin_Call("pow", [self, _lift(r)])
    def __lt__(self, r): return Bin_Op("<", self, _lift(r))
    def __gt__(self, r): return Bin_Op(">", self, _lift(r))
    def __le__(self, r): return Bin_Op("<=", self, _lift(r))
    def __ge__(self, r): return Bin_Op(">=", self, _lift(r))
    def __neg__(self): return Builtin_Call("neg", [self])

    # ---- builtins in method form, as the Ada sources use them (S.Get (Center)."-" (P).Length)
    def Dot(self, r): return Builtin_Call("dot", [self, r])
    def Cross(self, r): return Builtin_Call("cross", [self, r])
    def Min(self, r): return Builtin_Call("min", [self, _lift(r)])
    def Max(self, r): return Builtin_Call("max", [self, _lift(r)])
    def Clamp(self, lb, ub): return Builtin_Call("clamp", [self, _lift(lb), _lift(ub)])
    def Length(self): return Builtin_Call("length", [self])
    def Normalize(self): return Builtin_Call("normalize", [self])
    def Abs_Value(self): return Builtin_Call("abs", [self])
    def To_Float(self): return Builtin_Call("float", [self])
    def Sign(self): return Builtin_Call("sign", [self])
    def Floor(self): return Builtin_Call("floor", [self])
    def Sqrt(self): return Builtin_Call("sqrt", [self])
    def Dot2(self): return Builtin_Call("dot2", [self])
    def Acos(self): return Builtin_Call("acos", [self])
    def Sin(self): return Builtin_Call("sin", [self])
    def Cos(self): return Builtin_Call("cos", [self])
    def Tan(self): return Builtin_Call("tan", [self])
    def Asin(self): return Builtin_Call("asin", [self])
    def Atan(self): return Builtin_Call("atan", [self])

    def Get(self, axis):  # exprs.ads:97, axis = 0, 1, 2 (GL.X, GL.Y, GL.Z)
        return Project_Axis(self, int(axis))

    def Let_In(self, kind, name, body):  # exprs.ads:119-123: Value.Let_In (Kind, Name, In_Body)
        return Var_Body([Var_Decl(kind, name, self)], body)


class Ident(Expr):
    def __init__(self, name): self.name = name


class Lit(Expr):
    def __init__(self, v): self.v = v


class Bin_Op(Expr):
    def __init__(self, op, l, r): self.op, self.l, self.r = op, l, r


class Builtin_Call(Expr):
    def __init__(self, builtin, args): self.builtin, self.args = builtin, list(args)


class Project_Axis(Expr):
    def __init__(self, e, axis): self.e, self.axis = e, axis


class Get_Component(Expr):
    def __init__(self, prefix, comp): self.prefix, self.comp = prefix, comp


class Var_Body(Expr):
    def __init__(self, decls, body): self.decls, self.body = list(decls), body


class Condition(Expr):
    def __init__(self, c, t, e): self.c, self.t, self.e = c, t, e


class Unchecked_Call(Expr):
    def __init__(self, callee, struct_args, expr_args): self.callee, self.struct_args, self.expr_args = callee, struct_args, expr_args


class Struct_Expr:  # exprs.ads:14
    def __init__(self, name): self.name = name

    def Get(self, comp):  # exprs.ads:96
        return Get_Component(self, comp)


class Var_Decl:  # exprs.ads:106-110
    def __init__(self, kind, name, value): self.kind, self.name, self.value = kind, name, value


def _lift(x):
    if isinstance(x, Expr):
        return x
    if isinstance(x, values.Value):
        return Lit(x)
    if isinstance(x, (int, np.integer)) and not isinstance(x, bool):
        return Lit(values.Int(x))
    return Lit(values.Float(x))


def Literal(v): return Lit(v)
def Value_Identifier(n): return Ident(str(n))
def Struct_Identifier(n): return Struct_Expr(str(n))
def Construct_Vector3(x, y, z): return Builtin_Call("vec3", [x, y, z])
def Dot(l, r): return l.Dot(r)
def Cross(l, r): return l.Cross(r)
def Max(l, r): return l.Max(r)
def Clamp(e, lb, ub): return e.Clamp(lb, ub)
def Dot2(e): return e.Dot2()
def Length(e): return e.Length()
def Normalize(e): return e.Normalize()
def Sqrt(e): return e.Sqrt()
def If_Then_Else(c, thn, els): return Condition(c, thn, els)
def Create(kind, name, value): return Var_Decl(kind, name, value)
def External_Call(callee, struct_args, expr_args): return Unchecked_Call(callee, struct_args, expr_args)


def Min(a, b, c=None):  # exprs.adb:150-155: Min (A, B, C) = A.Min (B).Min (C)
    return a.Min(b) if c is None else a.Min(b).Min(c)


def Let_In(*args):
    """Let_In (Vars, In_Body) or Let_In (Value, Kind, Name, In_Body)  (exprs.ads:117-123)."""
    if len(args) == 2:
        decls, body = args
        return Var_Body([d if isinstance(d, Var_Decl) else Var_Decl(*d) for d in decls], body)
    value, kind, name, body = args
    return Var_Body([Var_Decl(kind, name, value)], body)


_fresh = [0]


def Fresh_Name(prefix):  # exprs.ads:123
    _fresh[0] += 1
    return "%s_%d" % (prefix, _fresh[0])


# ---------------------------------------------------------------------------- the lowering
class _Compiler:
    def __init__(self, comps):
        # packed instance layout of MDH_X_COMP: components in declaration order, vec3 = 3 floats
        self.comp_off, off = {}, 0
        for c in comps:
            self.comp_off[id(c)] = (off, c.kind)
            off += 3 if c.kind == Vector3_Kind else 1
        self.code = []
        self.free = list(range(X_REGS - 1, 2, -1))  # R0..R2 hold the result
        self.peak = 3

    # -- registers
    def alloc(self, n=1):
        if len(self.free) < n:
            raise Unsupported_Expr("expression needs more than %d registers" % X_REGS)
        r = [self.free.pop() for _ in range(n)]
        self.peak = max(self.peak, X_REGS - len(self.free))
        return r

    def release(self, regs, owned):
        if owned:
            self.free.extend(reversed(regs))

    def emit(self, op, dst, a=0, b=0, extra=None):
        self.code.append(op | dst << 8 | a << 16 | b << 24)
        if extra is not None:
            self.code.append(extra)

    def lit(self, bits):
        r, = self.alloc()
        self.emit(X_LIT, r, extra=bits)
        return r

    # -- expressions: returns (kind, [registers], owned)
    def expr(self, e, env):
        if isinstance(e, Lit):
            v = e.v
            if v.kind == Vector3_Kind:
                return Vector3_Kind, [self.lit(_f2i(image_roundtrip(c))) for c in v.data], True
            if v.kind == Float_Kind:
                return Float_Kind, [self.lit(_f2i(image_roundtrip(v.data)))], True
            return Int_Kind, [self.lit(int(v.data) & 0xFFFFFFFF)], True
        if isinstance(e, Ident):
            if e.name not in env:
                raise Type_Inference_Error("unbound identifier %r" % e.name)
            k, regs = env[e.name]
            return k, list(regs), False
        if isinstance(e, Get_Component):
            if id(e.comp) not in self.comp_off:
                raise Type_Inference_Error("component %r is not one of the kind's components" % e.comp.name)
            off, k = self.comp_off[id(e.comp)]
            n = 3 if k == Vector3_Kind else 1
            regs = self.alloc(n)
            for i, r in enumerate(regs):
                self.emit(X_COMP, r, off + i)
            return k, regs, True
        if isinstance(e, Project_Axis):
            k, regs, owned = self.expr(e.e, env)
            if k != Vector3_Kind:
                raise Type_Inference_Error("axis projection of a non-vector")
            r, = self.alloc()
            self.emit(X_MOV, r, regs[e.axis])
            self.release(regs, owned)
            return Float_Kind, [r], True
        if isinstance(e, Var_Body):
            env2, bound = dict(env), []
            for d in e.decls:
                k, regs, owned = self.expr(d.value, env2)
                if k == Int_Kind and d.kind == Float_Kind:
                    k, regs, owned = self.as_float(k, regs, owned)
                if k != d.kind:
                    raise Type_Inference_Error("declaration %r: kind mismatch" % d.name)
                if not owned:  # an alias of another variable: own a copy
                    cp = self.alloc(len(regs))
                    for a, b in zip(cp, regs):
                        self.emit(X_MOV, a, b)
                    regs = cp
                env2[str(d.name)] = (k, regs)
                bound.append(regs)
            k, regs, owned = self.expr(e.body, env2)
            if not owned:  # the body is one of the variables: move it out before they die
                cp = self.alloc(len(regs))
                for a, b in zip(cp, regs):
                    self.emit(X_MOV, a, b)
                regs, owned = cp, True
            for b in bound:
                self.release(b, True)
            return k, regs, owned
        if isinstance(e, Condition):
            kc, rc, oc = self.expr(e.c, env)
            if kc not in (_Bool_Kind, Int_Kind):
                raise Type_Inference_Error("Invalid value for ternary condition")  # exprs.adb:663-665
            kt, rt, ot = self.expr(e.t, env)
            ke, re_, oe = self.expr(e.e, env)
            if kt != ke:
                raise Type_Inference_Error("if expression: kind mismatch")
            out = self.alloc(len(rt))
            for d, a, b in zip(out, rt, re_):
                self.emit(X_SEL, d, rc[0], a, extra=b)
            self.release(re_, oe); self.release(rt, ot); self.release(rc, oc)
            return kt, out, True
        if isinstance(e, Bin_Op):
            return self.bin_op(e, env)
        if isinstance(e, Builtin_Call):
            return self.builtin(e, env)
        raise Unsupported_Expr("cannot compile %s" % type(e).__name__)

    def as_float(self, k, regs, owned):
        """GLSL's implicit int -> float conversion of a scalar operand."""
        if k == Int_Kind:
            r, = self.alloc()
            self.emit(X_ITOF, r, regs[0])
            self.release(regs, owned)
            return Float_Kind, [r], True
        return k, regs, owned

    def bin_op(self, e, env):
        kl, rl, ol = self.as_float(*self.expr(e.l, env))
        kr, rr, orr = self.as_float(*self.expr(e.r, env))
        if e.op in ("<", ">", "<=", ">="):
            if kl != Float_Kind or kr != Float_Kind:
                raise Type_Inference_Error("comparison of non-scalars")
            d, = self.alloc()
            self.emit({"<": X_LT, ">": X_GT, "<=": X_LE, ">=": X_GE}[e.op], d, rl[0], rr[0])
            self.release(rr, orr); self.release(rl, ol)
            return _Bool_Kind, [d], True
        if kl not in (Vector3_Kind, Float_Kind) or kr not in (Vector3_Kind, Float_Kind):
            raise Type_Inference_Error("binary operation on a condition")
        vec = kl == Vector3_Kind or kr == Vector3_Kind
        op = {"+": X_ADD, "-": X_SUB, "*": X_MUL, "/": X_DIV if vec else X_DIVF}[e.op]
        n = 3 if vec else 1
        out = self.alloc(n)
        for i in range(n):  # GLSL: component-wise, a scalar operand is broadcast
            self.emit(op, out[i], rl[i if kl == Vector3_Kind else 0], rr[i if kr == Vector3_Kind else 0])
        self.release(rr, orr); self.release(rl, ol)
        return (Vector3_Kind if vec else Float_Kind), out, True

    def dot(self, a, b):
        """(ax bx + ay by) + az bz into a fresh register."""
        t0, t1, d = self.alloc(3)
        self.emit(X_MUL, t0, a[0], b[0]); self.emit(X_MUL, t1, a[1], b[1]); self.emit(X_ADD, t0, t0, t1)
        self.emit(X_MUL, t1, a[2], b[2]); self.emit(X_ADD, d, t0, t1)
        self.release([t0, t1], True)
        return d

    def builtin(self, e, env):
        b = e.builtin
        args = [self.expr(a, env) for a in e.args]

        def done(kind, out):
            for k, regs, owned in reversed(args):
                self.release(regs, owned)
            return kind, out, True

        def want(i, kind):
            k, regs, owned = args[i]
            if kind == Float_Kind and k == Int_Kind:
                args[i] = self.as_float(k, regs, owned)
            elif k != kind:
                raise Type_Inference_Error("builtin %r: argument %d has the wrong kind" % (b, i + 1))
            return args[i][1]

        if b == "vec3":
            x, y, z = want(0, Float_Kind), want(1, Float_Kind), want(2, Float_Kind)
            out = self.alloc(3)
            for d, s in zip(out, (x[0], y[0], z[0])):
                self.emit(X_MOV, d, s)
            return done(Vector3_Kind, out)
        if b == "dot":
            return done(Float_Kind, [self.dot(want(0, Vector3_Kind), want(1, Vector3_Kind))])
        if b == "dot2":
            v = want(0, Vector3_Kind)
            return done(Float_Kind, [self.dot(v, v)])
        if b == "length":
            v = want(0, Vector3_Kind)
            d = self.dot(v, v)
            self.emit(X_SQRT, d, d)
            return done(Float_Kind, [d])
        if b == "normalize":  # v / length (v), math_utils.ads:77-83
            v = want(0, Vector3_Kind)
            d = self.dot(v, v)
            self.emit(X_SQRT, d, d)
            out = self.alloc(3)
            for i in range(3):
                self.emit(X_DIV, out[i], v[i], d)
            self.release([d], True)
            return done(Vector3_Kind, out)
        if b == "cross":
            a, c = want(0, Vector3_Kind), want(1, Vector3_Kind)
            out = self.alloc(3)
            t, = self.alloc()
            for i, (p, q) in enumerate(((1, 2), (2, 0), (0, 1))):  # a_p c_q - a_q c_p
                self.emit(X_MUL, out[i], a[p], c[q]); self.emit(X_MUL, t, a[q], c[p]); self.emit(X_SUB, out[i], out[i], t)
            self.release([t], True)
            return done(Vector3_Kind, out)
        if b in ("neg", "abs", "floor"):
            k, regs, _ = args[0]
            if k not in (Vector3_Kind, Float_Kind):
                raise Type_Inference_Error("builtin %r of a non-float" % b)
            out = self.alloc(len(regs))
            for d, s in zip(out, regs):
                self.emit({"neg": X_NEG, "abs": X_ABS, "floor": X_FLOOR}[b], d, s)
            return done(k, out)
        if b in ("sign", "sqrt", "acos", "sin", "cos", "tan", "asin", "atan"):
            v = want(0, Float_Kind)
            d, = self.alloc()
            self.emit({"sign": X_SIGN, "sqrt": X_SQRT, "acos": X_ACOS, "sin": X_SIN, "cos": X_COS, "tan": X_TAN,
                       "asin": X_ASIN, "atan": X_ATAN}[b], d, v[0])
            return done(Float_Kind, [d])
        if b in ("min", "max"):  # GLSL min / max: component-wise, a scalar operand is broadcast
            (kx, x, _), (ky, y, _) = args[0], args[1]
            if kx == Int_Kind:
                x = want(0, Float_Kind); kx = Float_Kind
            if ky == Int_Kind:
                y = want(1, Float_Kind); ky = Float_Kind
            if kx not in (Vector3_Kind, Float_Kind) or ky not in (Vector3_Kind, Float_Kind):
                raise Type_Inference_Error("builtin %r of a non-float" % b)
            vec = kx == Vector3_Kind or ky == Vector3_Kind
            out = self.alloc(3 if vec else 1)
            for i, d in enumerate(out):
                self.emit(X_MIN if b == "min" else X_MAX, d, x[i if kx == Vector3_Kind else 0], y[i if ky == Vector3_Kind else 0])
            return done(Vector3_Kind if vec else Float_Kind, out)
        if b == "pow" and isinstance(e.args[1], Lit) and e.args[1].v.kind in (Float_Kind, Int_Kind) \
                and float(e.args[1].v.data) == int(e.args[1].v.data) and 2 <= int(e.args[1].v.data) <= 16:
            # a literal integer power is repeated multiplication, squaring from the top bit
            x = want(0, Float_Kind)
            n = int(e.args[1].v.data)
            d, = self.alloc()
            self.emit(X_MOV, d, x[0])
            for bit in bin(n)[3:]:
                self.emit(X_MUL, d, d, d)
                if bit == "1":
                    self.emit(X_MUL, d, d, x[0])
            return done(Float_Kind, [d])
        if b == "pow":
            x, y = want(0, Float_Kind), want(1, Float_Kind)
            d, = self.alloc()
            self.emit(X_POW, d, x[0], y[0])
            return done(Float_Kind, [d])
        if b == "clamp":  # min (max (x, lb), ub)
            x, lo, hi = want(0, Float_Kind), want(1, Float_Kind), want(2, Float_Kind)
            d, = self.alloc()
            self.emit(X_MAX, d, x[0], lo[0]); self.emit(X_MIN, d, d, hi[0])
            return done(Float_Kind, [d])
        if b == "float":  # float (int), float (bool) -> 1.0 / 0.0, float (float)
            k, regs, _ = args[0]
            if k == Vector3_Kind:
                raise Type_Inference_Error("To_Float of a vector")
            d, = self.alloc()
            self.emit(X_ITOF if k == Int_Kind else X_MOV, d, regs[0])
            return done(Float_Kind, [d])
        raise Unsupported_Expr("builtin %r" % b)


def _f2i(x):
    return struct.unpack("<I", struct.pack("<f", float(np.float32(x))))[0]


def compile_program(expr, comps, result_kind, point_name=None, args=None):
    """The MDH_X words of `expr` for a kind with components `comps`.  Arguments (MDH_X_POINT floats)
    are bound to names: `point_name` is the vector at floats 0..2; `args` = [(name, kind, first float)]
    for the general case (a light's Sample: pos 0, normal 3, dir 6, dist 9).  The result lands in R0
    (R0..R2 for a vector)."""
    c = _Compiler(comps)
    env = {}
    if point_name is not None:
        args = [(point_name, Vector3_Kind, 0)] + list(args or [])
    for name, kind, first in args or []:
        regs = c.alloc(3 if kind == Vector3_Kind else 1)
        for i, r in enumerate(regs):
            c.emit(X_POINT, r, first + i)
        env[name] = (kind, regs)
    k, regs, owned = c.expr(expr, env)
    if result_kind == Float_Kind and k == Int_Kind:
        k, regs, owned = c.as_float(k, regs, owned)
    if k != result_kind:
        raise Type_Inference_Error("expression has kind %d, expected %d" % (k, result_kind))
    for i, r in enumerate(regs):
        c.emit(X_MOV, i, r)
    if len(c.code) > X_MAX_WORDS:
        raise Unsupported_Expr("program longer than %d words" % X_MAX_WORDS)
    return [w - (1 << 32) if w >= (1 << 31) else w for w in c.code]


# ------------------------------------------------------- Madarch.Exprs.Derivatives
def Forward_Difference(Exp, Param, Point, Epsilon=0.000001):
    """madarch-exprs-derivatives.adb:12-46: (f (p + h e_a) - f (p)) for a = x, y, z as a vector,
    f = `Exp` with its free identifier `Param` bound to the point."""
    eps = np.float32(Epsilon)
    h = [Literal(values.Vector3(v)) for v in ((eps, 0.0, 0.0), (0.0, eps, 0.0), (0.0, 0.0, eps))]
    f_p, f_x, f_y, f_z = (Value_Identifier(n) for n in ("f_p", "f_x", "f_y", "f_z"))
    return Let_In(
        [Var_Decl(Float_Kind, "f_p", Let_In(Point, Vector3_Kind, Param, Exp)),
         Var_Decl(Float_Kind, "f_x", Let_In(Point + h[0], Vector3_Kind, Param, Exp) - f_p),
         Var_Decl(Float_Kind, "f_y", Let_In(Point + h[1], Vector3_Kind, Param, Exp) - f_p),
         Var_Decl(Float_Kind, "f_z", Let_In(Point + h[2], Vector3_Kind, Param, Exp) - f_p)],
        Construct_Vector3(f_x, f_y, f_z))
