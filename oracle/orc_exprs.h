/* orc_exprs.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Restatement of the reference's CPU evaluator: the expression-tree DSL of
 * /root/reference/madarch/madarch-exprs.ads:127-280 with its `Eval`
 * interpretation (madarch-exprs.adb:322-716) over the tagged-union values of
 * madarch-values.adb:37-360, and the four built-in primitive kinds built as
 * trees exactly as madarch-primitives-{spheres,planes}.ads and
 * madarch-primitives-{boxes,triangles}.adb build them.
 * PARITY UNPINNED (no reference fixture exists).
 */
#ifndef ORC_EXPRS_H
#define ORC_EXPRS_H

#include "orc_math.h"

/* madarch-values.ads:8 */
typedef enum { VK_VEC3 = 0, VK_FLOAT = 1, VK_INT = 2 } value_kind;

typedef struct {
   value_kind kind;
   v3 v;
   float f;
   int32_t i;
} value;

/* an entity = component name -> value, searched linearly (madarch-entities.adb:9-20) */
typedef struct {
   int n;
   const char *names[8];
   value vals[8];
} entity;

typedef struct expr expr; /* madarch-exprs.ads:127 Expr_Node */

/* evaluation error flag (the Ada code raises Program_Error) */
extern __thread const char *orc_eval_error;

/* the cached trees of Primitives.Eval_Dist / Eval_Normal
 * (madarch-primitives.adb:90-137); type: 0 Sphere, 1 Plane, 2 Box, 3 Triangle */
const expr *orc_prim_dist_expr(int type);
const expr *orc_prim_normal_expr(int type);

/* Primitives.Eval_Expr_From_Point (madarch-primitives.adb:67-79): binds
 * "prim" and "x" in a fresh context and evaluates */
value orc_eval_from_point(const expr *e, const entity *ent, v3 point);

/* Eval_Dist / Eval_Normal (madarch-primitives.adb:90-137) */
float orc_eval_dist(int type, const entity *ent, v3 point);
v3 orc_eval_normal(int type, const entity *ent, v3 point);

/* 1: "/" on two floats computes L + R as madarch-values.adb:112 does (default);
 * 0: true division (what the generated GLSL does) */
void orc_exprs_set_ada_div(int on);

/* number of tree nodes evaluated since the last reset (per thread) */
extern __thread uint64_t orc_eval_nodes;

#endif
