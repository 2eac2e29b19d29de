/* orc_exprs.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * See orc_exprs.h.  PARITY UNPINNED.
 *
 * The interpreter keeps the cost structure of the reference on purpose (it is
 * the "Madarch.Exprs Ada CPU evaluator" baseline of BASELINE.json config 1):
 * one heap node per context binding (madarch-exprs.adb:15-39), name lookup by
 * string comparison up the parent chain (:49-81), linear component search
 * (madarch-entities.adb:9-20), one dynamic dispatch per tree node.
 */
#include "orc_exprs.h"

#include <stdlib.h>
#include <string.h>

__thread const char *orc_eval_error = 0;
__thread uint64_t orc_eval_nodes = 0;
static int g_ada_div = 1;

void orc_exprs_set_ada_div(int on) { g_ada_div = on; }

/* ---------------------------------------------------------------- values */

static value val_v(v3 v) { value r; memset(&r, 0, sizeof r); r.kind = VK_VEC3; r.v = v; return r; }
static value val_f(float f) { value r; memset(&r, 0, sizeof r); r.kind = VK_FLOAT; r.f = f; return r; }
static value val_i(int32_t i) { value r; memset(&r, 0, sizeof r); r.kind = VK_INT; r.i = i; return r; }

static value fail(const char *msg) { orc_eval_error = msg; return val_f(NAN); }

/* madarch-values.adb:17-23 */
static int check_kinds(value l, value r)
{
   if (l.kind != r.kind) { orc_eval_error = "Incompatible kinds"; return 0; }
   return 1;
}

/* Single'Min / Single'Max */
static float ada_min(float a, float b) { return a < b ? a : b; }
static float ada_max(float a, float b) { return a > b ? a : b; }

/* madarch-values.adb:37-48 */
static value v_add(value l, value r)
{
   if (!check_kinds(l, r)) return fail(orc_eval_error);
   switch (l.kind) {
   case VK_VEC3: return val_v(add(l.v, r.v));
   case VK_FLOAT: return val_f(l.f + r.f);
   default: return val_i(l.i + r.i);
   }
}
/* madarch-values.adb:50-61 */
static value v_sub(value l, value r)
{
   if (!check_kinds(l, r)) return fail(orc_eval_error);
   switch (l.kind) {
   case VK_VEC3: return val_v(sub(l.v, r.v));
   case VK_FLOAT: return val_f(l.f - r.f);
   default: return val_i(l.i - r.i);
   }
}
/* madarch-values.adb:63-103: the only mixed-kind operator */
static value v_mul(value l, value r)
{
   switch (l.kind) {
   case VK_VEC3:
      switch (r.kind) {
      case VK_VEC3: return val_v(mul(l.v, r.v));
      case VK_FLOAT: return val_v(scale(l.v, r.f));
      default: return val_v(scale(l.v, (float)r.i));
      }
   case VK_FLOAT:
      switch (r.kind) {
      case VK_VEC3: return val_v(V3(l.f * r.v.x, l.f * r.v.y, l.f * r.v.z));
      case VK_FLOAT: return val_f(l.f * r.f);
      default: return val_f(l.f * (float)r.i);
      }
   default:
      switch (r.kind) {
      case VK_VEC3: return val_v(V3((float)l.i * r.v.x, (float)l.i * r.v.y, (float)l.i * r.v.z));
      case VK_FLOAT: return val_f((float)l.i * r.f);
      default: return val_i(l.i * r.i);
      }
   }
}
/* madarch-values.adb:105-116: Float and Int "/" compute L + R in the reference */
static value v_div(value l, value r)
{
   if (!check_kinds(l, r)) return fail(orc_eval_error);
   switch (l.kind) {
   case VK_VEC3: return val_v(vdiv(l.v, r.v));
   case VK_FLOAT: return val_f(g_ada_div ? l.f + r.f : l.f / r.f);
   default: return val_i(g_ada_div ? l.i + r.i : (r.i ? l.i / r.i : 0));
   }
}
/* madarch-values.adb:118-129 */
static value v_pow(value l, value r)
{
   if (!check_kinds(l, r)) return fail(orc_eval_error);
   switch (l.kind) {
   case VK_VEC3: return fail("'**' not applicable to vector3.");
   case VK_FLOAT: return val_f(pow_(l.f, r.f));
   default: {
      int32_t acc = 1;
      for (int32_t k = 0; k < r.i; ++k) acc *= l.i;
      return val_i(acc);
   }
   }
}
/* madarch-values.adb:131-184: comparisons give Int 0/1 */
static value v_cmp(int op, value l, value r)
{
   if (!check_kinds(l, r)) return fail(orc_eval_error);
   int b;
   switch (l.kind) {
   case VK_FLOAT:
      b = op == 0 ? l.f < r.f : op == 1 ? l.f > r.f : op == 2 ? l.f <= r.f : l.f >= r.f;
      return val_i(b);
   case VK_INT:
      b = op == 0 ? l.i < r.i : op == 1 ? l.i > r.i : op == 2 ? l.i <= r.i : l.i >= r.i;
      return val_i(b);
   default: return fail("comparison not applicable to vector3.");
   }
}
static value v_dot(value l, value r)
{
   if (!check_kinds(l, r)) return fail(orc_eval_error);
   if (l.kind != VK_VEC3) return fail("Dot is only allowed on vectors");
   return val_f(dot(l.v, r.v));
}
static value v_cross(value l, value r)
{
   if (!check_kinds(l, r)) return fail(orc_eval_error);
   if (l.kind != VK_VEC3) return fail("Cross is only allowed on vectors");
   return val_v(cross(l.v, r.v));
}
/* madarch-values.adb:208-234 */
static value v_min(value l, value r)
{
   if (!check_kinds(l, r)) return fail(orc_eval_error);
   switch (l.kind) {
   case VK_VEC3: return val_v(V3(ada_min(l.v.x, r.v.x), ada_min(l.v.y, r.v.y), ada_min(l.v.z, r.v.z)));
   case VK_FLOAT: return val_f(ada_min(l.f, r.f));
   default: return val_i(l.i < r.i ? l.i : r.i);
   }
}
static value v_max(value l, value r)
{
   if (!check_kinds(l, r)) return fail(orc_eval_error);
   switch (l.kind) {
   case VK_VEC3: return val_v(V3(ada_max(l.v.x, r.v.x), ada_max(l.v.y, r.v.y), ada_max(l.v.z, r.v.z)));
   case VK_FLOAT: return val_f(ada_max(l.f, r.f));
   default: return val_i(l.i > r.i ? l.i : r.i);
   }
}
/* madarch-values.adb:236-242 */
static value v_clamp(value v, value lb, value ub)
{
   if (!check_kinds(v, lb) || !check_kinds(lb, ub)) return fail(orc_eval_error);
   return v_min(v_max(v, lb), ub);
}
/* madarch-values.adb:244-254: vector negation is -1.0 * V */
static value v_neg(value v)
{
   switch (v.kind) {
   case VK_VEC3: return val_v(V3(-1.0f * v.v.x, -1.0f * v.v.y, -1.0f * v.v.z));
   case VK_FLOAT: return val_f(-v.f);
   default: return val_i(-v.i);
   }
}

/* ----------------------------------------------------------------- trees */

typedef enum { N_IDENT, N_LIT, N_BINOP, N_BUILTIN, N_PROJECT, N_GETCOMP, N_LET, N_COND, N_CALL } node_kind;

/* madarch-exprs.ads:182-183 */
typedef enum { B_ADD, B_SUB, B_MUL, B_DIV, B_LT, B_GT, B_LTE, B_GTE } binop_kind;

/* madarch-exprs.ads:196-204 */
typedef enum {
   F_DOT, F_CROSS, F_ABS, F_FLOOR, F_MIN, F_MAX, F_CLAMP, F_POW, F_NEG, F_SIN, F_COS, F_TAN, F_ASIN,
   F_ACOS, F_ATAN, F_SQRT, F_DOT2, F_LEN, F_NORM, F_SIGN, F_FLOAT, F_VEC3
} builtin_kind;

typedef struct { value_kind kind; const char *name; const expr *val; } var_decl;

struct expr {
   node_kind k;
   const char *name;        /* ident; get_component suffix; callee   */
   const char *prefix;      /* get_component struct name             */
   value lit;
   int op;                  /* binop_kind | builtin_kind | axis      */
   int nargs;
   const expr *args[3];
   int ndecl;
   var_decl decls[8];
   const expr *body;
};

static expr *mk(node_kind k)
{
   expr *e = (expr *)calloc(1, sizeof(expr));
   e->k = k;
   return e;
}
static const expr *ident(const char *n) { expr *e = mk(N_IDENT); e->name = n; return e; }
static const expr *lit(value v) { expr *e = mk(N_LIT); e->lit = v; return e; }
static const expr *bin(binop_kind op, const expr *l, const expr *r)
{
   expr *e = mk(N_BINOP); e->op = op; e->nargs = 2; e->args[0] = l; e->args[1] = r; return e;
}
static const expr *call1(builtin_kind f, const expr *a)
{
   expr *e = mk(N_BUILTIN); e->op = f; e->nargs = 1; e->args[0] = a; return e;
}
static const expr *call2(builtin_kind f, const expr *a, const expr *b)
{
   expr *e = mk(N_BUILTIN); e->op = f; e->nargs = 2; e->args[0] = a; e->args[1] = b; return e;
}
static const expr *call3(builtin_kind f, const expr *a, const expr *b, const expr *c)
{
   expr *e = mk(N_BUILTIN); e->op = f; e->nargs = 3; e->args[0] = a; e->args[1] = b; e->args[2] = c; return e;
}
static const expr *proj(const expr *a, int axis)
{
   expr *e = mk(N_PROJECT); e->op = axis; e->nargs = 1; e->args[0] = a; return e;
}
static const expr *getcomp(const char *st, const char *comp)
{
   expr *e = mk(N_GETCOMP); e->prefix = st; e->name = comp; return e;
}
static const expr *let1(const expr *val, value_kind kind, const char *name, const expr *body)
{
   expr *e = mk(N_LET); e->ndecl = 1; e->decls[0].kind = kind; e->decls[0].name = name;
   e->decls[0].val = val; e->body = body; return e;
}
static const expr *letn(int n, const var_decl *d, const expr *body)
{
   expr *e = mk(N_LET); e->ndecl = n;
   for (int i = 0; i < n; ++i) e->decls[i] = d[i];
   e->body = body; return e;
}
static const expr *cond(const expr *c, const expr *t, const expr *f)
{
   expr *e = mk(N_COND); e->nargs = 3; e->args[0] = c; e->args[1] = t; e->args[2] = f; return e;
}

/* ------------------------------------------------ context (exprs.adb:15-81) */

typedef struct ctx_node {
   struct ctx_node *parent;
   const char *name;
   int is_entity;
   const entity *ent;
   value val;
} ctx_node;

static ctx_node *ctx_append_val(ctx_node *parent, const char *name, value v)
{
   ctx_node *n = (ctx_node *)malloc(sizeof(ctx_node));
   n->parent = parent; n->name = name; n->is_entity = 0; n->ent = 0; n->val = v;
   return n;
}
static ctx_node *ctx_append_ent(ctx_node *parent, const char *name, const entity *e)
{
   ctx_node *n = (ctx_node *)malloc(sizeof(ctx_node));
   n->parent = parent; n->name = name; n->is_entity = 1; n->ent = e; memset(&n->val, 0, sizeof n->val);
   return n;
}

static value eval(const expr *e, ctx_node *ctx);

static value eval_builtin(const expr *e, ctx_node *ctx)
{
   value a[3];
   for (int i = 0; i < e->nargs; ++i) a[i] = eval(e->args[i], ctx); /* exprs.adb:433-435 */
   value f = a[0];
   switch ((builtin_kind)e->op) {
   case F_NEG: return v_neg(f);
   case F_LEN: return f.kind == VK_VEC3 ? val_f(length(f.v)) : fail("Cannot take length");
   case F_NORM: return f.kind == VK_VEC3 ? val_v(normalize(f.v)) : fail("Cannot normalize");
   case F_ABS:
      return f.kind == VK_VEC3 ? val_v(vabs(f.v)) : f.kind == VK_FLOAT ? val_f(fabsf(f.f)) : val_i(f.i < 0 ? -f.i : f.i);
   case F_SIGN: return f.kind == VK_FLOAT ? val_f(sign_(f.f)) : fail("sign not applicable.");
   case F_FLOOR:
      return f.kind == VK_VEC3 ? val_v(vfloor(f.v)) : f.kind == VK_FLOAT ? val_f(floorf(f.f)) : fail("floor not applicable to int");
#define ELEM(fn)                                                        \
   (f.kind == VK_FLOAT ? val_f((float)fn((double)f.f))                  \
    : f.kind == VK_INT ? val_f((float)fn((double)(float)f.i))           \
                       : fail("Cannot apply elementary function to vector3."))
   case F_SIN: return ELEM(sin);
   case F_COS: return ELEM(cos);
   case F_TAN: return ELEM(tan);
   case F_ASIN: return ELEM(asin);
   case F_ACOS: return ELEM(acos);
   case F_ATAN: return ELEM(atan);
#undef ELEM
   case F_SQRT:
      return f.kind == VK_FLOAT ? val_f(sqrtf(f.f)) : f.kind == VK_INT ? val_f(sqrtf((float)f.i)) : fail("Cannot apply Sqrt to vector3.");
   case F_DOT2: return f.kind == VK_VEC3 ? val_f(dot2(f.v)) : fail("cannot apply dot2");
   case F_DOT: return v_dot(a[0], a[1]);
   case F_CROSS: return v_cross(a[0], a[1]);
   case F_MIN: return v_min(a[0], a[1]);
   case F_MAX: return v_max(a[0], a[1]);
   case F_CLAMP: return v_clamp(a[0], a[1], a[2]);
   case F_POW: return v_pow(a[0], a[1]);
   case F_FLOAT: /* exprs.adb:478-485 */
      return f.kind == VK_VEC3 ? fail("Invalid cast.") : f.kind == VK_FLOAT ? f : val_f((float)f.i);
   case F_VEC3: return val_v(V3(a[0].f, a[1].f, a[2].f));
   }
   return fail("bad builtin");
}

static value eval(const expr *e, ctx_node *ctx)
{
   ++orc_eval_nodes;
   switch (e->k) {
   case N_IDENT: /* exprs.adb:66-81 */
      for (ctx_node *c = ctx; c; c = c->parent)
         if (strcmp(c->name, e->name) == 0) return c->val;
      return fail("Key not in eval context values");
   case N_LIT: return e->lit;
   case N_BINOP: { /* exprs.adb:360-382 */
      value l = eval(e->args[0], ctx);
      value r = eval(e->args[1], ctx);
      switch ((binop_kind)e->op) {
      case B_ADD: return v_add(l, r);
      case B_SUB: return v_sub(l, r);
      case B_MUL: return v_mul(l, r);
      case B_DIV: return v_div(l, r);
      case B_LT: return v_cmp(0, l, r);
      case B_GT: return v_cmp(1, l, r);
      case B_LTE: return v_cmp(2, l, r);
      case B_GTE: return v_cmp(3, l, r);
      }
      return fail("bad binop");
   }
   case N_BUILTIN: return eval_builtin(e, ctx);
   case N_PROJECT: { /* exprs.adb:547-556 */
      value v = eval(e->args[0], ctx);
      if (v.kind != VK_VEC3) return fail("Cannot project component.");
      return val_f(e->op == 0 ? v.v.x : e->op == 1 ? v.v.y : v.v.z);
   }
   case N_GETCOMP: { /* exprs.adb:49-64,579-583 and entities.adb:9-20 */
      for (ctx_node *c = ctx; c; c = c->parent)
         if (strcmp(c->name, e->prefix) == 0) {
            const entity *ent = c->ent;
            for (int i = 0; i < ent->n; ++i)
               if (strcmp(ent->names[i], e->name) == 0) return ent->vals[i];
            return fail("Entity does not have given component.");
         }
      return fail("Key not in eval context entities");
   }
   case N_LET: { /* exprs.adb:604-620 */
      ctx_node *nc = ctx;
      for (int i = 0; i < e->ndecl; ++i) nc = ctx_append_val(nc, e->decls[i].name, eval(e->decls[i].val, nc));
      value r = eval(e->body, nc);
      while (nc != ctx) { ctx_node *p = nc->parent; free(nc); nc = p; }
      return r;
   }
   case N_COND: { /* exprs.adb:658-671 */
      value c = eval(e->args[0], ctx);
      if (c.kind != VK_INT) return fail("Invalid value for ternary condition");
      return c.i == 0 ? eval(e->args[2], ctx) : eval(e->args[1], ctx);
   }
   case N_CALL: return fail("Cannot evaluate unchecked call"); /* exprs.adb:715-716 */
   }
   return fail("bad node");
}

/* --------------------------------------------- the built-in primitive trees */

#define AX 0
#define AY 1
#define AZ 2

/* madarch-primitives-spheres.ads:13-17 */
static const expr *sphere_dist(const char *s, const expr *p)
{
   return bin(B_SUB, call1(F_LEN, bin(B_SUB, getcomp(s, "center"), p)), getcomp(s, "radius"));
}
static const expr *sphere_normal(const char *s, const expr *p)
{
   return call1(F_NORM, bin(B_SUB, p, getcomp(s, "center")));
}
/* madarch-primitives-planes.ads:13-17 */
static const expr *plane_dist(const char *s, const expr *p)
{
   return bin(B_ADD, call2(F_DOT, getcomp(s, "normal"), p), getcomp(s, "offset"));
}
static const expr *plane_normal(const char *s, const expr *p)
{
   (void)p;
   return getcomp(s, "normal");
}
/* madarch-primitives-boxes.adb:7-15 */
static const expr *box_dist(const char *s, const expr *p)
{
   const expr *q = ident("q");
   const expr *zero_v = lit(val_v(V3(0, 0, 0)));
   const expr *zero_f = lit(val_f(0.0f));
   const expr *body = bin(
      B_ADD, call1(F_LEN, call2(F_MAX, q, zero_v)),
      call2(F_MIN, call2(F_MAX, proj(q, AX), call2(F_MAX, proj(q, AY), proj(q, AZ))), zero_f));
   return let1(bin(B_SUB, call1(F_ABS, bin(B_SUB, getcomp(s, "center"), p)), getcomp(s, "side")), VK_VEC3, "q", body);
}
/* madarch-primitives-boxes.adb:5,17-41 */
static const expr *box_normal(const char *s, const expr *p)
{
   const expr *d = ident("d"), *rx = ident("rx"), *ry = ident("ry"), *rz = ident("rz");
   const expr *e = lit(val_f(0.002f));
#define TF(a, b) call1(F_FLOAT, bin(B_GT, a, bin(B_SUB, b, e)))
   const expr *nd = call3(
      F_VEC3, bin(B_MUL, bin(B_MUL, TF(rx, ry), TF(rx, rz)), call1(F_SIGN, proj(d, AX))),
      bin(B_MUL, bin(B_MUL, TF(ry, rx), TF(ry, rz)), call1(F_SIGN, proj(d, AY))),
      bin(B_MUL, bin(B_MUL, TF(rz, rx), TF(rz, ry)), call1(F_SIGN, proj(d, AZ))));
#undef TF
   return let1(
      bin(B_DIV, bin(B_SUB, p, getcomp(s, "center")), getcomp(s, "side")), VK_VEC3, "d",
      let1(call1(F_ABS, proj(d, AX)), VK_FLOAT, "rx",
           let1(call1(F_ABS, proj(d, AY)), VK_FLOAT, "ry",
                let1(call1(F_ABS, proj(d, AZ)), VK_FLOAT, "rz", call1(F_NORM, nd)))));
}
/* madarch-primitives-triangles.adb:16-48 */
static const expr *triangle_dist(const char *s, const expr *p)
{
   const expr *v21 = ident("V21"), *v32 = ident("V32"), *v13 = ident("V13");
   const expr *p1 = ident("P1"), *p2 = ident("P2"), *p3 = ident("P3"), *nor = ident("Nor");
   const expr *f0 = lit(val_f(0.0f)), *f1 = lit(val_f(1.0f)), *f2 = lit(val_f(2.0f));
#define SGN(v, q) call1(F_SIGN, call2(F_DOT, call2(F_CROSS, v, nor), q))
   const expr *c = bin(B_LT, bin(B_ADD, bin(B_ADD, SGN(v21, p1), SGN(v32, p2)), SGN(v13, p3)), f2);
#undef SGN
#define EDGE(v, q)                                                                                    \
   call1(F_DOT2, bin(B_SUB, bin(B_MUL, v, call3(F_CLAMP, bin(B_DIV, call2(F_DOT, v, q), call1(F_DOT2, v)), f0, f1)), q))
   const expr *thn = call2(F_MIN, call2(F_MIN, EDGE(v21, p1), EDGE(v32, p2)), EDGE(v13, p3));
#undef EDGE
   const expr *els = bin(B_DIV, bin(B_MUL, call2(F_DOT, nor, p1), call2(F_DOT, nor, p1)), call1(F_DOT2, nor));
   var_decl d[7] = {
      {VK_VEC3, "V21", bin(B_SUB, getcomp(s, "v2"), getcomp(s, "v1"))},
      {VK_VEC3, "V32", bin(B_SUB, getcomp(s, "v3"), getcomp(s, "v2"))},
      {VK_VEC3, "V13", bin(B_SUB, getcomp(s, "v1"), getcomp(s, "v3"))},
      {VK_VEC3, "P1", bin(B_SUB, p, getcomp(s, "v1"))},
      {VK_VEC3, "P2", bin(B_SUB, p, getcomp(s, "v2"))},
      {VK_VEC3, "P3", bin(B_SUB, p, getcomp(s, "v3"))},
      {VK_VEC3, "Nor", call2(F_CROSS, v21, v13)},
   };
   return letn(7, d, call1(F_SQRT, cond(c, thn, els)));
}
/* madarch-exprs-derivatives.adb:12-45, epsilon 1e-6 */
static const expr *forward_difference(const expr *e, const char *param, const expr *point)
{
   const float eps = 0.000001f;
   const expr *hx = lit(val_v(V3(eps, 0, 0))), *hy = lit(val_v(V3(0, eps, 0))), *hz = lit(val_v(V3(0, 0, eps)));
   const expr *fp = ident("f_p");
   var_decl d[4] = {
      {VK_FLOAT, "f_p", let1(point, VK_VEC3, param, e)},
      {VK_FLOAT, "f_x", bin(B_SUB, let1(bin(B_ADD, point, hx), VK_VEC3, param, e), fp)},
      {VK_FLOAT, "f_y", bin(B_SUB, let1(bin(B_ADD, point, hy), VK_VEC3, param, e), fp)},
      {VK_FLOAT, "f_z", bin(B_SUB, let1(bin(B_ADD, point, hz), VK_VEC3, param, e), fp)},
   };
   return letn(4, d, call3(F_VEC3, ident("f_x"), ident("f_y"), ident("f_z")));
}
/* madarch-primitives-triangles.adb:50-56 */
static const expr *triangle_normal(const char *s, const expr *p)
{
   return call1(F_NORM, forward_difference(triangle_dist(s, ident("DX")), "DX", p));
}

static const expr *g_dist[4], *g_norm[4];

static void build_once(void)
{
   static int done = 0;
   if (done) return;
   /* Primitives.Eval_Dist builds the trees on idents "prim" and "x"
    * (madarch-primitives.adb:9-19) */
   const expr *x = ident("x");
   g_dist[0] = sphere_dist("prim", x);
   g_norm[0] = sphere_normal("prim", x);
   g_dist[1] = plane_dist("prim", x);
   g_norm[1] = plane_normal("prim", x);
   g_dist[2] = box_dist("prim", x);
   g_norm[2] = box_normal("prim", x);
   g_dist[3] = triangle_dist("prim", x);
   g_norm[3] = triangle_normal("prim", x);
   done = 1;
}

__attribute__((constructor)) static void orc_exprs_init(void) { build_once(); }

const expr *orc_prim_dist_expr(int type) { build_once(); return g_dist[type & 3]; }
const expr *orc_prim_normal_expr(int type) { build_once(); return g_norm[type & 3]; }

/* madarch-primitives.adb:67-79 */
value orc_eval_from_point(const expr *e, const entity *ent, v3 point)
{
   ctx_node *c = ctx_append_ent(0, "prim", ent);
   c = ctx_append_val(c, "x", val_v(point));
   value r = eval(e, c);
   while (c) { ctx_node *p = c->parent; free(c); c = p; }
   return r;
}

float orc_eval_dist(int type, const entity *ent, v3 point)
{
   value r = orc_eval_from_point(orc_prim_dist_expr(type), ent, point);
   if (r.kind != VK_FLOAT) { orc_eval_error = "Unexpected value kind."; return NAN; }
   return r.f;
}

v3 orc_eval_normal(int type, const entity *ent, v3 point)
{
   value r = orc_eval_from_point(orc_prim_normal_expr(type), ent, point);
   if (r.kind != VK_VEC3) { orc_eval_error = "Unexpected value kind."; return V3(NAN, NAN, NAN); }
   return r.v;
}
