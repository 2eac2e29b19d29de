/* madarch_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement (plain C, scalar fp32) of the hot path of Roldak/Madarch:
 * Madarch.Renderers.Render and the GLSL it dispatches.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (madarch_amd/, libmadarch_hip.so) never does.
 *
 * PARITY UNPINNED: the reference (Ada 2012 + GLSL 4.30 over OpenGLAda) cannot
 * be built or run in this pipeline (no GNAT, no GL) and holds no test, golden
 * vector or numeric fixture for this path (SURVEY.md sections 4 and 8c).  The
 * oracle is pinned only by (i) the std140 offsets SURVEY.md section 8d derives
 * from the reference's layout rules, (ii) an independent numpy restatement of
 * the closed-form pieces (tests/test_oracle_pins.py) and (iii) the committed
 * fixtures under tests/golden/ that were generated from this file.
 *
 * Every function cites the reference file:line it follows; all paths are
 * relative to /root/reference/madarch/.  It exports the same operations as
 * include/madarch_hip.h with the prefix orc_ so that the same host code can
 * drive either.  Operation order, rounding and the few places where GLSL
 * leaves behaviour to the driver are fixed in orc_math.h and DESIGN.md.
 */
#define _GNU_SOURCE
#include "../include/madarch_hip.h"
#include "orc_exprs.h"
#include "orc_math.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAX_KINDS 8
#define MAX_MATERIALS 20 /* glsl/materials.glsl:9 */

enum { PK_SPHERE = 0, PK_PLANE = 1, PK_BOX = 2, PK_TRIANGLE = 3, PK_CUSTOM = 4 }; /* PK_CUSTOM: user-defined kind (MDH_X programs, include/madarch_hip.h) */
enum { LK_POINT = 0, LK_SPOT = 1, LK_CUSTOM = 2 };

/* oracle-only options (>= 100) */
enum { ORC_OPT_SDF_MODE = 100 /* 0 closed form, 1 Madarch.Exprs tree evaluator */, ORC_OPT_THREADS = 101 };

static __thread char g_err[256];
static int seterr(int code, const char *msg)
{
   snprintf(g_err, sizeof g_err, "%s", msg);
   return code;
}
const char *orc_last_error(void) { return g_err; }
const char *orc_version(void) { return "madarch-oracle 1 (CPU restatement; parity unpinned)"; }

/* ------------------------------------------------------------------------
 * std140 layout calculator = GPU_Types (support/gpu_types-base.ads:21-37,
 * gpu_types-structs.adb:11-38, gpu_types-fixed_arrays.adb:17-39)
 * ---------------------------------------------------------------------- */
static int pad_to(int x, int a) { while (x % a) ++x; return x; } /* gpu_types.adb:2-8 */
static int base_align(int kind) { return kind == MDH_VEC3 ? 16 : 4; }
static int base_size(int kind) { return kind == MDH_VEC3 ? 12 : 4; }

typedef struct {
   int type;      /* PK_* or LK_* */
   int max_count; /* declared */
   int ncomp;
   char comp_name[8][24];
   int comp_kind[8];
   int comp_off[8];
   int elem_size; /* Struct.Size */
   int stride;    /* pad16(elem_size) */
   int count_off, array_off;
   /* resolved field offsets inside an element */
   int f_a, f_b, f_c, f_mat;
   /* user-defined kinds: the three MDH_X programs and, for MDH_X_COMP, the byte offset inside an
    * element of every float of the packed instance (components in order, vec3 = 3 floats) */
   int32_t *x_code[3];
   int x_len[3];
   int x_float_off[24], inst_floats;
} kind_t;

static int comp_offset(const kind_t *k, const char *name, int kind)
{
   for (int i = 0; i < k->ncomp; ++i)
      if (strcmp(k->comp_name[i], name) == 0 && k->comp_kind[i] == kind) return k->comp_off[i];
   return -1;
}

/* gpu_types-structs.adb:11-38 for one entity struct */
static void layout_kind(kind_t *k)
{
   int off = 0;
   for (int i = 0; i < k->ncomp; ++i) {
      off = pad_to(off, base_align(k->comp_kind[i]));
      k->comp_off[i] = off;
      off += base_size(k->comp_kind[i]);
   }
   k->elem_size = off;
   k->stride = pad_to(off, 16); /* gpu_types-fixed_arrays.adb:22-24 */
}

typedef struct {
   int w, h, c;
   int unorm8, flush_nan;
   float *data;
} tex_t;

struct orc_renderer {
   int W, H;
   int npk, nlk;
   kind_t pk[MAX_KINDS], lk[MAX_KINDS];
   int prim_base[MAX_KINDS]; /* flat index base = sum of earlier DECLARED counts (scenes.adb:656-666) */
   float max_dist;
   mdh_partitioning part;
   float part_gpu_diag; /* Single'Image round trip of Length(spacing) (scenes.adb:1140-1143) */
   float pg_spacing[3], pg_offset[3]; /* spacing/offset as the GLSL text carries them (scenes.adb:807-811) */
   mdh_probe_settings probes;
   mdh_volumetrics vol;
   uint8_t *scene_ubo;
   int scene_ubo_size, total_light_off;
   uint8_t materials_ubo[16 + 32 * MAX_MATERIALS]; /* renderers.adb:77-89: count @0, array @16 stride 32 */
   int last_material_index;                        /* renderers.adb:130 */
   int host_count[MAX_KINDS];                      /* All_Primitives vector lengths (renderers.ads:128) */
   float cam_pos[3], cam_m[9];
   int *part_table; /* [cell][npk + index_count] */
   int part_cells, part_warnings;
   tex_t tex[4];
   float *fb;
   uint8_t *front; /* orc_swap_buffers */
   int64_t swaps;
   int32_t *gb_index, *gb_steps;
   float *gb_t;
   int opt_atlas, opt_mode, opt_ao, opt_gbuffer, opt_rank, opt_world, opt_timing, opt_ada_div, opt_irr_all, opt_window, opt_spec, opt_hyst, opt_mips;
   tex_t rad_mips[16]; /* MDH_OPT_RADIANCE_MIPS: levels 1 .. radiance_lods of the radiance atlas ([0] unused) */
   int opt_sdf_mode, opt_threads;
   uint64_t sdf_evals; /* closest_primitive[_info] calls of the last pass set */
   /* the work of the last run of each pass -- rays started (raycast, raycast_hit_position /
      raycast_visibility, softshadows), SDF evaluations inside their march loops, SDF evaluations in all (+ the
      occlusion taps, lighting.glsl:51-69); a -DMDH_DIAG build of the kernels reports the same three (orc_work_counters) */
   uint64_t work[MDH_PASS_COUNT][3];
};
typedef struct orc_renderer orc_renderer;

static __thread uint64_t t_sdf_evals, t_rays, t_steps;

/* Single'Image prints 6 significant digits; literals that travel through the
 * generated GLSL text lose the rest (scenes.adb:21-24,1200-1201; renderers.adb:119-134) */
static float image_roundtrip(float x)
{
   char buf[64];
   snprintf(buf, sizeof buf, "%.5E", (double)x);
   return strtof(buf, 0);
}

/* ------------------------------------------------------------------------ UBO views */
static inline float ubo_f(const orc_renderer *r, int off) { float f; memcpy(&f, r->scene_ubo + off, 4); return f; }
static inline int ubo_i(const orc_renderer *r, int off) { int32_t i; memcpy(&i, r->scene_ubo + off, 4); return i; }
static inline v3 ubo_v3(const orc_renderer *r, int off) { return V3(ubo_f(r, off), ubo_f(r, off + 4), ubo_f(r, off + 8)); }

typedef struct { v3 albedo; float metallic, roughness; } material_t;
/* glsl/materials.glsl:1-10 with Material_Type offsets (renderers.adb:77-81): albedo 0, metallic 12, roughness 16 */
static material_t get_material(const orc_renderer *r, int id)
{
   material_t m;
   const uint8_t *p = r->materials_ubo + 16 + 32 * id;
   float f[5];
   memcpy(f, p, 20);
   m.albedo = V3(f[0], f[1], f[2]);
   m.metallic = f[3];
   m.roughness = f[4];
   return m;
}

/* ------------------------------------------------------------------------
 * built-in SDFs, closed form with the operation order of the generated GLSL
 * ---------------------------------------------------------------------- */
/* madarch-primitives-spheres.ads:13-14: length(center - p) - radius */
static inline float sd_sphere(v3 c, float rad, v3 p) { return length(sub(c, p)) - rad; }
/* madarch-primitives-planes.ads:13-14: dot(normal, p) + offset */
static inline float sd_plane(v3 n, float o, v3 p) { return dot(n, p) + o; }
/* madarch-primitives-boxes.adb:7-15 */
static inline float sd_box(v3 c, v3 s, v3 p)
{
   v3 q = sub(vabs(sub(c, p)), s);
   return length(vmaxs(q, 0.0f)) + fmin_(fmax_(q.x, fmax_(q.y, q.z)), 0.0f);
}
/* madarch-primitives-triangles.adb:16-48; `ada_div` selects Values."/" (L + R) */
static inline float tri_div(float a, float b, int ada_div) { return ada_div ? a + b : a / b; }
static float sd_triangle(v3 a, v3 b, v3 c, v3 p, int ada_div)
{
   v3 v21 = sub(b, a), v32 = sub(c, b), v13 = sub(a, c);
   v3 p1 = sub(p, a), p2 = sub(p, b), p3 = sub(p, c);
   v3 nor = cross(v21, v13);
   float s = (sign_(dot(cross(v21, nor), p1)) + sign_(dot(cross(v32, nor), p2))) + sign_(dot(cross(v13, nor), p3));
   float r;
   if (s < 2.0f) {
      float e1 = dot2(sub(scale(v21, clamp_(tri_div(dot(v21, p1), dot2(v21), ada_div), 0.0f, 1.0f)), p1));
      float e2 = dot2(sub(scale(v32, clamp_(tri_div(dot(v32, p2), dot2(v32), ada_div), 0.0f, 1.0f)), p2));
      float e3 = dot2(sub(scale(v13, clamp_(tri_div(dot(v13, p3), dot2(v13), ada_div), 0.0f, 1.0f)), p3));
      r = fmin_(fmin_(e1, e2), e3);
   } else {
      r = tri_div(dot(nor, p1) * dot(nor, p1), dot2(nor), ada_div);
   }
   return sqrtf(r);
}

/* madarch-primitives-boxes.adb:5,17-41 */
static v3 nrm_box(v3 c, v3 s, v3 p)
{
   const float e = 0.002f;
   v3 d = vdiv(sub(p, c), s);
   float rx = fabsf(d.x), ry = fabsf(d.y), rz = fabsf(d.z);
   v3 n = V3(((float)(rx > ry - e) * (float)(rx > rz - e)) * sign_(d.x),
             ((float)(ry > rx - e) * (float)(ry > rz - e)) * sign_(d.y),
             ((float)(rz > rx - e) * (float)(rz > ry - e)) * sign_(d.z));
   return normalize(n);
}
/* madarch-primitives-triangles.adb:50-56 + madarch-exprs-derivatives.adb:12-45 */
static v3 nrm_triangle(v3 a, v3 b, v3 c, v3 p, int ada_div)
{
   const float eps = 0.000001f;
   float fp = sd_triangle(a, b, c, p, ada_div);
   float fx = sd_triangle(a, b, c, add(p, V3(eps, 0, 0)), ada_div) - fp;
   float fy = sd_triangle(a, b, c, add(p, V3(0, eps, 0)), ada_div) - fp;
   float fz = sd_triangle(a, b, c, add(p, V3(0, 0, eps)), ada_div) - fp;
   return normalize(V3(fx, fy, fz));
}

/* entity view of element i of kind k, for the tree evaluator */
static void make_entity(const orc_renderer *r, const kind_t *k, int i, entity *e)
{
   int base = k->array_off + k->stride * i;
   e->n = k->ncomp;
   for (int c = 0; c < k->ncomp; ++c) {
      e->names[c] = k->comp_name[c];
      memset(&e->vals[c], 0, sizeof(value));
      e->vals[c].kind = (value_kind)k->comp_kind[c];
      if (k->comp_kind[c] == MDH_VEC3) e->vals[c].v = ubo_v3(r, base + k->comp_off[c]);
      else if (k->comp_kind[c] == MDH_FLOAT) e->vals[c].f = ubo_f(r, base + k->comp_off[c]);
      else e->vals[c].i = ubo_i(r, base + k->comp_off[c]);
   }
}

/* The MDH_X register programs of a user-defined kind (include/madarch_hip.h): the oracle's own
 * interpreter, a plain switch over the instruction list.  which: 0 Distance, 1 Normal, 2 Material. */
/* args: the MDH_X_POINT floats -- 0..2 the point (pos), 3..5 normal, 6..8 dir, 9 dist (light Sample) */
static void orc_xrun_args(const orc_renderer *r, const kind_t *k, int which, int inst, const float args[10], int ada_div, float out[3]);
static void orc_xrun(const orc_renderer *r, const kind_t *k, int which, int inst, v3 x, int ada_div, float out[3])
{
   float args[10] = {x.x, x.y, x.z, 0, 0, 0, 0, 0, 0, 0};
   orc_xrun_args(r, k, which, inst, args, ada_div, out);
}
static void orc_xrun_args(const orc_renderer *r, const kind_t *k, int which, int inst, const float args[10], int ada_div, float out[3])
{
   float R[MDH_X_REGS];
   memset(R, 0, sizeof R);
   const int32_t *code = k->x_code[which];
   const int n = k->x_len[which], base = k->array_off + k->stride * inst;
   for (int pc = 0; pc < n; ++pc) {
      const uint32_t w = (uint32_t)code[pc];
      const int op = w & 255, d = (w >> 8) & 63, a = (w >> 16) & 255, b = (w >> 24) & 63;
      const float va = R[a & 63], vb = R[b];
      float res = 0.0f;
      switch (op) {
      case MDH_X_LIT: memcpy(&res, &code[++pc], 4); break;
      case MDH_X_MOV: res = va; break;
      case MDH_X_COMP: res = ubo_f(r, base + k->x_float_off[a]); break;
      case MDH_X_POINT: res = args[a < 10 ? a : 9]; break;
      case MDH_X_ADD: res = va + vb; break;
      case MDH_X_SUB: res = va - vb; break;
      case MDH_X_MUL: res = va * vb; break;
      case MDH_X_DIV: res = va / vb; break;
      case MDH_X_DIVF: res = ada_div ? va + vb : va / vb; break;
      case MDH_X_NEG: res = -va; break;
      case MDH_X_ABS: res = fabsf(va); break;
      case MDH_X_FLOOR: res = floorf(va); break;
      case MDH_X_SIGN: res = sign_(va); break;
      case MDH_X_MIN: res = fmin_(va, vb); break;
      case MDH_X_MAX: res = fmax_(va, vb); break;
      case MDH_X_SQRT: res = sqrtf(va); break;
      case MDH_X_POW: res = pow_(va, vb); break;
      case MDH_X_LT: res = va < vb ? 1.0f : 0.0f; break;
      case MDH_X_GT: res = va > vb ? 1.0f : 0.0f; break;
      case MDH_X_LE: res = va <= vb ? 1.0f : 0.0f; break;
      case MDH_X_GE: res = va >= vb ? 1.0f : 0.0f; break;
      case MDH_X_SEL: { const float vc = R[code[++pc] & 63]; res = va != 0.0f ? vb : vc; break; }
      case MDH_X_ITOF: { int32_t iv; memcpy(&iv, &va, 4); res = (float)iv; break; }
      case MDH_X_ACOS: res = acos_(va); break;
      case MDH_X_SIN: res = sin_(va); break;
      case MDH_X_COS: res = cos_(va); break;
      case MDH_X_TAN: res = tan_(va); break;
      case MDH_X_ASIN: res = asin_(va); break;
      case MDH_X_ATAN: res = atan_(va); break;
      default: break;
      }
      R[d] = res;
   }
   out[0] = R[0]; out[1] = R[1]; out[2] = R[2];
}

/* dist_to_<Kind>(prims[i], x) as emitted by scenes.adb:417-455 */
static inline float prim_dist(const orc_renderer *r, const kind_t *k, int i, v3 x)
{
   if (k->type == PK_CUSTOM) {
      float o[3];
      orc_xrun(r, k, 0, i, x, 0, o);
      return o[0];
   }
   if (r->opt_sdf_mode == 1) {
      entity e;
      make_entity(r, k, i, &e);
      return orc_eval_dist(k->type, &e, x);
   }
   int b = k->array_off + k->stride * i;
   switch (k->type) {
   case PK_SPHERE: return sd_sphere(ubo_v3(r, b + k->f_a), ubo_f(r, b + k->f_b), x);
   case PK_PLANE: return sd_plane(ubo_v3(r, b + k->f_a), ubo_f(r, b + k->f_b), x);
   case PK_BOX: return sd_box(ubo_v3(r, b + k->f_a), ubo_v3(r, b + k->f_b), x);
   default: return sd_triangle(ubo_v3(r, b + k->f_a), ubo_v3(r, b + k->f_b), ubo_v3(r, b + k->f_c), x, 0);
   }
}
/* <Kind>_normal(prims[i], x) as emitted by scenes.adb:457-495 */
static inline v3 prim_normal(const orc_renderer *r, const kind_t *k, int i, v3 x)
{
   if (k->type == PK_CUSTOM) {
      float o[3];
      orc_xrun(r, k, 1, i, x, 0, o);
      return V3(o[0], o[1], o[2]);
   }
   if (r->opt_sdf_mode == 1) {
      entity e;
      make_entity(r, k, i, &e);
      return orc_eval_normal(k->type, &e, x);
   }
   int b = k->array_off + k->stride * i;
   switch (k->type) {
   case PK_SPHERE: return normalize(sub(x, ubo_v3(r, b + k->f_a))); /* spheres.ads:16-17 */
   case PK_PLANE: return ubo_v3(r, b + k->f_a);                      /* planes.ads:16-17 */
   case PK_BOX: return nrm_box(ubo_v3(r, b + k->f_a), ubo_v3(r, b + k->f_b), x);
   default: return nrm_triangle(ubo_v3(r, b + k->f_a), ubo_v3(r, b + k->f_b), ubo_v3(r, b + k->f_c), x, 0);
   }
}

/* closest_primitive (scenes.adb:602-629) */
static float closest_primitive(const orc_renderer *r, v3 x)
{
   ++t_sdf_evals;
   float closest = r->max_dist;
   for (int k = 0; k < r->npk; ++k) {
      int n = ubo_i(r, r->pk[k].count_off);
      for (int i = 0; i < n; ++i) closest = fmin_(closest, prim_dist(r, &r->pk[k], i, x));
   }
   return closest;
}
/* closest_primitive_info (scenes.adb:631-674) */
static float closest_primitive_info(const orc_renderer *r, v3 x, int *index)
{
   ++t_sdf_evals;
   float closest = r->max_dist;
   for (int k = 0; k < r->npk; ++k) {
      int n = ubo_i(r, r->pk[k].count_off);
      for (int i = 0; i < n; ++i) {
         float d = prim_dist(r, &r->pk[k], i, x);
         if (d < closest) { closest = d; *index = r->prim_base[k] + i; }
      }
   }
   return closest;
}
/* primitive_info (scenes.adb:676-729): kind by successive subtraction of the DECLARED counts */
static void primitive_info(const orc_renderer *r, int index, v3 pos, v3 *normal, int *material_id)
{
   for (int k = 0; k < r->npk; ++k) {
      const kind_t *kk = &r->pk[k];
      if (index < kk->max_count) {
         *normal = prim_normal(r, kk, index, pos);
         if (kk->type == PK_CUSTOM) {
            float o[3];
            orc_xrun(r, kk, 2, index, pos, 0, o);
            memcpy(material_id, &o[0], 4);
         } else
            *material_id = ubo_i(r, kk->array_off + kk->stride * index + kk->f_mat);
         return;
      }
      index -= kk->max_count;
   }
   *normal = V3(0, 0, 0);
   *material_id = 0;
}

/* ------------------------------------------------------------ space partition */
/* partitioning index (scenes.adb:799-837).  The reference clamps to `dims`
 * (not dims-1); an index past the table reads an empty cell here. */
static int partition_cell(const orc_renderer *r, v3 x, int *fallback)
{
   const mdh_partitioning *p = &r->part;
   v3 off = V3(r->pg_offset[0], r->pg_offset[1], r->pg_offset[2]);
   v3 sp = V3(r->pg_spacing[0], r->pg_spacing[1], r->pg_spacing[2]);
   v3 fx = vfloor(vdiv(sub(x, off), sp));
   v3 cfx = V3(clamp_(fx.x, 0.0f, (float)p->grid_dimensions[0]), clamp_(fx.y, 0.0f, (float)p->grid_dimensions[1]),
               clamp_(fx.z, 0.0f, (float)p->grid_dimensions[2]));
   *fallback = 0;
   if (p->border_behavior == 0) fx = cfx;
   else if (fx.x != cfx.x || fx.y != cfx.y || fx.z != cfx.z) { *fallback = 1; return -1; }
   float yz = (float)(p->grid_dimensions[1] * p->grid_dimensions[2]);
   float zz = (float)p->grid_dimensions[2];
   return (int)((fx.x * yz + fx.y * zz) + fx.z);
}
/* partitioning_closest (scenes.adb:839-958) */
static float partitioning_closest(const orc_renderer *r, v3 x)
{
   if (!r->part.enable) return closest_primitive(r, x); /* scenes.adb:1256-1262 */
   int fb;
   int cell = partition_cell(r, x, &fb);
   if (fb) return closest_primitive(r, x);
   ++t_sdf_evals;
   float closest = r->max_dist;
   if (cell < 0 || cell >= r->part_cells) return closest;
   const int *rec = r->part_table + (size_t)cell * (r->npk + r->part.index_count);
   int i = 0;
   for (int k = 0; k < r->npk; ++k) {
      int size = i + rec[k];
      for (; i < size && i < r->part.index_count; ++i)
         closest = fmin_(closest, prim_dist(r, &r->pk[k], rec[r->npk + i], x));
      i = size;
   }
   return closest;
}
/* partitioning_closest_info (scenes.adb:960-1118) */
static float partitioning_closest_info(const orc_renderer *r, v3 x, int *index)
{
   if (!r->part.enable) return closest_primitive_info(r, x, index);
   int fb;
   int cell = partition_cell(r, x, &fb);
   if (fb) return closest_primitive_info(r, x, index);
   ++t_sdf_evals;
   float closest = r->max_dist;
   if (cell < 0 || cell >= r->part_cells) return closest;
   const int *rec = r->part_table + (size_t)cell * (r->npk + r->part.index_count);
   int i = 0;
   for (int k = 0; k < r->npk; ++k) {
      int size = i + rec[k];
      for (; i < size && i < r->part.index_count; ++i) {
         int pi = rec[r->npk + i];
         float d = prim_dist(r, &r->pk[k], pi, x);
         if (d < closest) { closest = d; *index = r->prim_base[k] + pi; }
      }
      i = size;
   }
   return closest;
}

/* Primitives.Eval_Dist on the host copy (madarch-primitives.adb:90-108): always the tree evaluator */
static float host_eval_dist(const orc_renderer *r, int k, int i, v3 p)
{
   if (r->pk[k].type == PK_CUSTOM) { /* the CPU builders evaluate with Madarch.Values: "/" on floats adds */
      float o[3];
      orc_xrun(r, &r->pk[k], 0, i, p, 1, o);
      return o[0];
   }
   entity e;
   make_entity(r, &r->pk[k], i, &e);
   return orc_eval_dist(r->pk[k].type, &e, p);
}

static void write_cell(orc_renderer *r, int cell, int cand_n[MAX_KINDS], int cand[MAX_KINDS][256])
{
   int *rec = r->part_table + (size_t)cell * (r->npk + r->part.index_count);
   int written = 0;
   for (int k = 0; k < r->npk; ++k) {
      int n = cand_n[k];
      if (written + n > r->part.index_count) { /* renderers.adb:593-606 only warns; clamped here */
         ++r->part_warnings;
         n = r->part.index_count - written;
      }
      rec[k] = n;
      for (int j = 0; j < n; ++j) rec[r->npk + written + j] = cand[k][j];
      written += n;
   }
}

/* Update_Partitioning_CPU (madarch-renderers.adb:551-755) */
static void update_partitioning_cpu(orc_renderer *r, int optimized)
{
   const mdh_partitioning *p = &r->part;
   v3 sp = V3(p->grid_spacing[0], p->grid_spacing[1], p->grid_spacing[2]);
   v3 off = V3(p->grid_offset[0], p->grid_offset[1], p->grid_offset[2]);
   float cell_diag = length(sp);
   for (int X = 0; X < p->grid_dimensions[0]; ++X)
      for (int Y = 0; Y < p->grid_dimensions[1]; ++Y)
         for (int Z = 0; Z < p->grid_dimensions[2]; ++Z) {
            int cell = X * p->grid_dimensions[1] * p->grid_dimensions[2] + Y * p->grid_dimensions[2] + Z;
            v3 grid_pos = add(mul(V3((float)X, (float)Y, (float)Z), sp), off);
            v3 center = add(grid_pos, scale(sp, 0.5f));
            float closest = 1.0e10f;
            for (int k = 0; k < r->npk; ++k)
               for (int i = 0; i < r->host_count[k]; ++i) {
                  float d = host_eval_dist(r, k, i, center);
                  if (d < closest) closest = d;
               }
            /* precandidates, kinds in scene order (the reference iterates a hashed map) */
            int pre_k[512], pre_i[512], npre = 0;
            for (int k = 0; k < r->npk; ++k)
               for (int i = 0; i < r->host_count[k]; ++i) {
                  float d = host_eval_dist(r, k, i, center);
                  if (d < closest + cell_diag && npre < 512) { pre_k[npre] = k; pre_i[npre] = i; ++npre; }
               }
            int cand_n[MAX_KINDS] = {0};
            int cand[MAX_KINDS][256];
            if (optimized) { /* Find_Candidates, renderers.adb:669-723 */
               char accepted[512] = {0};
               for (int sx = 1; sx <= 3; ++sx)
                  for (int sy = 1; sy <= 3; ++sy)
                     for (int sz = 1; sz <= 3; ++sz) {
                        v3 offv = vdiv(sub(V3((float)sx, (float)sy, (float)sz), v3s(1.0f)), sub(v3s(3.0f), v3s(1.0f)));
                        v3 pt = add(mul(offv, sp), grid_pos);
                        float c = 1.0e10f;
                        int ci = -1;
                        for (int q = 0; q < npre; ++q) {
                           float d = host_eval_dist(r, pre_k[q], pre_i[q], pt);
                           if (d < c) { c = d; ci = q; }
                        }
                        if (ci >= 0 && !accepted[ci]) {
                           accepted[ci] = 1;
                           int k = pre_k[ci];
                           if (cand_n[k] < 256) cand[k][cand_n[k]++] = pre_i[ci];
                        }
                     }
            } else { /* Assume_Candidates, renderers.adb:725-732 */
               for (int q = 0; q < npre; ++q) {
                  int k = pre_k[q];
                  if (cand_n[k] < 256) cand[k][cand_n[k]++] = pre_i[q];
               }
            }
            write_cell(r, cell, cand_n, cand);
         }
}

/* Update_Partitioning_GPU = compute_scene_partitioning.glsl:7-21 with
 * partitioning_compute_grid_cell (scenes.adb:1120-1187); dispatch dims/2
 * groups of 2x2x2 (renderers.adb:539-549) */
static void update_partitioning_gpu(orc_renderer *r)
{
   const mdh_partitioning *p = &r->part;
   int gx = 2 * (p->grid_dimensions[0] / 2), gy = 2 * (p->grid_dimensions[1] / 2), gz = 2 * (p->grid_dimensions[2] / 2);
   v3 sp = V3(r->pg_spacing[0], r->pg_spacing[1], r->pg_spacing[2]);
   v3 off = V3(r->pg_offset[0], r->pg_offset[1], r->pg_offset[2]);
   int saved = r->opt_sdf_mode;
   r->opt_sdf_mode = 0;
   for (int X = 0; X < gx; ++X)
      for (int Y = 0; Y < gy; ++Y)
         for (int Z = 0; Z < gz; ++Z) {
            int cell = X * gy * gz + Y * gz + Z;
            if (cell >= r->part_cells) continue;
            v3 vc = add(V3((float)X, (float)Y, (float)Z), v3s(0.5f));
            v3 center = add(mul(vc, sp), off);
            float thr = closest_primitive(r, center) + r->part_gpu_diag;
            int cand_n[MAX_KINDS] = {0};
            int cand[MAX_KINDS][256];
            for (int k = 0; k < r->npk; ++k) {
               int n = ubo_i(r, r->pk[k].count_off);
               for (int i = 0; i < n; ++i)
                  if (prim_dist(r, &r->pk[k], i, center) < thr && cand_n[k] < 256) cand[k][cand_n[k]++] = i;
            }
            write_cell(r, cell, cand_n, cand);
         }
   r->opt_sdf_mode = saved;
}

/* ------------------------------------------------------------------ raymarching */
/* glsl/raymarching.glsl:4-23 */
static float softshadows(const orc_renderer *r, v3 from, v3 dir, float min_dist, float max_dist, float k)
{
   float res = 1.0f;
   float prev_dist = 1e20f;
   ++t_rays;
   for (float total = min_dist; total < max_dist;) {
      float dist = partitioning_closest(r, add(from, scale(dir, total)));
      ++t_steps;
      if (dist < ORC_EPSILON) return 0.0f;
      float y = dist * dist / (2.0f * prev_dist);
      float d = sqrtf(dist * dist - y * y);
      res = fmin_(res, k * d / fmax_(0.0f, total - y));
      prev_dist = dist;
      total += dist;
   }
   return res;
}
/* glsl/raymarching.glsl:25-37; `steps` counts the SDF evaluations */
static int raycast(const orc_renderer *r, v3 from, v3 dir, int *index, v3 *coll, float *t_out, int *steps)
{
   int n = 0;
   ++t_rays;
   for (float total = 0.0f; total < r->max_dist;) {
      float dist = partitioning_closest_info(r, add(from, scale(dir, total)), index);
      ++n;
      ++t_steps;
      if (dist < ORC_EPSILON) {
         *coll = add(from, scale(dir, total));
         if (t_out) *t_out = total;
         if (steps) *steps = n;
         return 1;
      }
      total += dist;
   }
   if (steps) *steps = n;
   return 0;
}
/* glsl/raymarching.glsl:39-51 */
static int raycast_hit_position(const orc_renderer *r, v3 from, v3 dir, float max_dist, v3 *coll)
{
   ++t_rays;
   for (float total = 0.0f; total < max_dist;) {
      float dist = partitioning_closest(r, add(from, scale(dir, total)));
      ++t_steps;
      if (dist < ORC_EPSILON) {
         *coll = add(from, scale(dir, total));
         return 1;
      }
      total += dist;
   }
   return 0;
}
/* glsl/raymarching.glsl:53-56 */
static float raycast_visibility(const orc_renderer *r, v3 from, v3 dir, float max_dist)
{
   v3 dummy;
   return 1.0f - (float)raycast_hit_position(r, from, dir, max_dist, &dummy);
}

/* ----------------------------------------------------------------------- lights */
/* sample_<Light> (scenes.adb:497-549) + sample_light (scenes.adb:731-764) */
static v3 sample_light(const orc_renderer *r, int index, v3 pos, v3 normal, v3 *dir, float *dist)
{
   for (int k = 0; k < r->nlk; ++k) {
      const kind_t *lk = &r->lk[k];
      int n = ubo_i(r, lk->count_off); /* runtime count, scenes.adb:737-751 */
      if (index < n) {
         if (lk->type == LK_CUSTOM) { /* the generated sample_<Light> (scenes.adb:497-549) around the kind's programs */
            float o[3], args[10] = {pos.x, pos.y, pos.z, normal.x, normal.y, normal.z, 0, 0, 0, 0};
            orc_xrun_args(r, lk, 1, index, args, 0, o); /* Position */
            *dir = sub(V3(o[0], o[1], o[2]), pos);
            *dist = length(*dir);
            *dir = divs(*dir, *dist);
            args[6] = dir->x; args[7] = dir->y; args[8] = dir->z; args[9] = *dist;
            orc_xrun_args(r, lk, 0, index, args, 0, o); /* Sample */
            return V3(o[0], o[1], o[2]);
         }
         int b = lk->array_off + lk->stride * index;
         v3 lpos = ubo_v3(r, b + lk->f_a);
         *dir = sub(lpos, pos);
         *dist = length(*dir);
         *dir = divs(*dir, *dist);
         if (lk->type == LK_POINT) /* madarch-lights-point_lights.ads:20-22 */
            return divs(ubo_v3(r, b + lk->f_b), (*dist * *dist) * 0.03f);
         /* madarch-lights-spot_lights.adb:5-24 */
         float attenuation = 1.0f / ((*dist * *dist) * 0.03f);
         float theta = acos_(fmax_(dot(neg(*dir), ubo_v3(r, b + lk->f_b)), 0.0f));
         float ratio = clamp_(theta / ubo_f(r, b + lk->f_c), 0.0f, 1.0f);
         float visible = 1.0f - pow8_(ratio);
         return scale(scale(ubo_v3(r, b + lk->f_mat), fmin_(attenuation, 1.5f)), visible);
      }
      index -= n;
   }
   *dir = V3(0, 0, 0);
   *dist = 0.0f;
   return V3(0, 0, 0);
}

/* -------------------------------------------------------------------------- BRDF */
/* glsl/cook_torrance_brdf.glsl:35-52 (and :1-33) */
static void cook_torrance(v3 N, v3 V, v3 L, float NdotL, v3 albedo, float metallic, float roughness, v3 *kD, v3 *kS)
{
   v3 Hv = normalize(add(V, L));
   float NdotV = fmax_(dot(N, V), 0.0f);
   v3 F0 = V3(mix_(0.04f, albedo.x, metallic), mix_(0.04f, albedo.y, metallic), mix_(0.04f, albedo.z, metallic));
   /* distribution_GGX :5-16 */
   float a = roughness * roughness;
   float a2 = a * a;
   float NdotH = fmax_(dot(N, Hv), 0.0f);
   float NdotH2 = NdotH * NdotH;
   float denom = NdotH2 * (a2 - 1.0f) + 1.0f;
   denom = ORC_PI * denom * denom;
   float NDF = a2 / denom;
   /* geometry_smith :18-33 */
   float rr = roughness + 1.0f;
   float kk = (rr * rr) / 8.0f;
   float ggx2 = NdotV / (NdotV * (1.0f - kk) + kk);
   float ggx1 = NdotL / (NdotL * (1.0f - kk) + kk);
   float G = ggx1 * ggx2;
   /* fresnel_schlick :1-3 */
   float p5 = pow5_(1.001f - fmax_(dot(Hv, V), 0.0f));
   v3 F = V3(F0.x + (1.0f - F0.x) * p5, F0.y + (1.0f - F0.y) * p5, F0.z + (1.0f - F0.z) * p5);
   v3 numerator = scale(F, NDF * G);
   float denominator = 4.0f * NdotV * NdotL;
   float dm = fmax_(denominator, 0.001f);
   *kD = scale(V3(1.0f - F.x, 1.0f - F.y, 1.0f - F.z), 1.0f - metallic);
   *kS = vmins(divs(numerator, dm), 1.0f);
}

/* glsl/lighting.glsl:1-40 */
static v3 compute_direct_lighting(const orc_renderer *r, v3 pos, v3 normal, v3 dir, v3 albedo, float metallic,
                                  float roughness, int direct_specular)
{
   v3 N = normal, V = neg(dir), Lo = V3(0, 0, 0);
   int total = ubo_i(r, r->total_light_off);
   for (int i = 0; i < total; ++i) {
      v3 L;
      float L_dist;
      v3 radiance = sample_light(r, i, pos, N, &L, &L_dist);
      float NdotL = fmax_(dot(N, L), 0.0f);
      v3 kD, kS;
      cook_torrance(N, V, L, NdotL, albedo, metallic, roughness, &kD, &kS);
      float shadows = 0.0f;
      if (NdotL > ORC_EPSILON)
         shadows = softshadows(r, add(pos, scale(scale(normal, ORC_MIN_STEP), 5.0f)), L, 0.0f, L_dist, 64.0f);
      if (!direct_specular) kS = V3(0, 0, 0);
      v3 brdf = add(divs(mul(kD, albedo), ORC_PI), kS);
      Lo = add(Lo, scale(scale(mul(brdf, radiance), NdotL), shadows));
   }
   return Lo;
}
/* glsl/lighting.glsl:42-49 */
static v3 compute_indirect_lighting(v3 irradiance, v3 radiance, v3 V, v3 N, v3 L, v3 albedo, float metallic, float roughness)
{
   v3 kD, kS;
   float NdotL = fmax_(dot(N, L), 0.0f);
   cook_torrance(N, V, L, NdotL, albedo, metallic, roughness, &kD, &kS);
   return add(divs(mul(kD, irradiance), ORC_PI), scale(mul(kS, radiance), NdotL));
}
/* glsl/lighting.glsl:51-69 */
static float compute_ambient_occlusion(const orc_renderer *r, v3 pos, v3 normal, int steps)
{
   if (steps <= 0) return 1.0f;
   const float ao_step_size = 0.1f;
   float ao_sum = 0.0f, max_ao_sum = 0.0f, factor = 1.0f;
   for (int i = 0; i < steps; ++i) {
      v3 p = add(pos, scale(scale(normal, (float)(i + 1)), ao_step_size));
      ao_sum += factor * partitioning_closest(r, p);
      max_ao_sum += factor * (float)(i + 1) * ao_step_size;
      factor = factor * 0.5f; /* 1.0 / pow(2.0, i), exact */
   }
   return 0.6f + 0.4f * ao_sum / max_ao_sum;
}

/* -------------------------------------------------------------------- probe utils */
/* glsl/probe_utils.glsl:19-56 */
static int coord_to_probe_id(const orc_renderer *r, v2 nc)
{
   int px = (int)(nc.x * (float)r->probes.probe_count[0]);
   int py = (int)(nc.y * (float)r->probes.probe_count[1]);
   return py * r->probes.probe_count[0] + px;
}
static iv3 probe_id_to_grid_position(const orc_renderer *r, int id)
{
   int xy = r->probes.grid_dimensions[0] * r->probes.grid_dimensions[1], xc = r->probes.grid_dimensions[0];
   iv3 g;
   g.z = id / xy;
   g.y = (id - g.z * xy) / xc;
   g.x = id - g.z * xy - g.y * xc;
   return g;
}
static v3 grid_to_world(const orc_renderer *r, iv3 g)
{
   return V3((float)g.x * r->probes.grid_spacing[0], (float)g.y * r->probes.grid_spacing[1], (float)g.z * r->probes.grid_spacing[2]);
}
static v3 probe_spacing(const orc_renderer *r) { return V3(r->probes.grid_spacing[0], r->probes.grid_spacing[1], r->probes.grid_spacing[2]); }
static iv3 world_to_grid(const orc_renderer *r, v3 p)
{
   v3 f = vfloor(vdiv(p, probe_spacing(r)));
   iv3 g = {(int)f.x, (int)f.y, (int)f.z};
   return g;
}
static int grid_to_probe_id(const orc_renderer *r, iv3 g)
{
   return g.z * r->probes.grid_dimensions[0] * r->probes.grid_dimensions[1] + g.y * r->probes.grid_dimensions[0] + g.x;
}
static v2 probe_id_to_coord(const orc_renderer *r, int id)
{
   int y = id / r->probes.probe_count[0];
   int x = id - y * r->probes.probe_count[0];
   return V2((float)x / (float)r->probes.probe_count[0], (float)y / (float)r->probes.probe_count[1]);
}
/* glsl/probe_utils.glsl:58-92 */
static float sign_not_zero(float v) { return v >= 0.0f ? 1.0f : -1.0f; }
static v2 float32x3_to_oct(v3 v)
{
   float s = 1.0f / ((fabsf(v.x) + fabsf(v.y)) + fabsf(v.z));
   v2 p = V2(v.x * s, v.y * s);
   if (v.z <= 0.0f) return V2((1.0f - fabsf(p.y)) * sign_not_zero(p.x), (1.0f - fabsf(p.x)) * sign_not_zero(p.y));
   return p;
}
static v3 oct_to_float32x3(v2 e)
{
   v3 v = V3(e.x, e.y, (1.0f - fabsf(e.x)) - fabsf(e.y));
   if (v.z < 0.0f) {
      float nx = (1.0f - fabsf(v.y)) * sign_not_zero(v.x);
      float ny = (1.0f - fabsf(v.x)) * sign_not_zero(v.y);
      v.x = nx;
      v.y = ny;
   }
   return normalize(v);
}
static v2 coord_to_ray_id(const orc_renderer *r, v2 nc)
{
   return V2(fract_(nc.x * (float)r->probes.probe_count[0]), fract_(nc.y * (float)r->probes.probe_count[1]));
}
static v3 ray_id_to_ray_dir(v2 id) { return oct_to_float32x3(V2(id.x * 2.0f - 1.0f, id.y * 2.0f - 1.0f)); }
static v2 ray_dir_to_ray_id(v3 d)
{
   v2 raw = float32x3_to_oct(d);
   return V2((raw.x + 1.0f) * 0.5f, (raw.y + 1.0f) * 0.5f);
}

/* ------------------------------------------------------------------------ textures */
/* GL_MIRRORED_REPEAT (support/render_passes.adb:111-112) */
static int mirror(int i, int n)
{
   int m = i % (2 * n);
   if (m < 0) m += 2 * n;
   return m >= n ? 2 * n - 1 - m : m;
}
/* GL_LINEAR, single level (render_passes.adb:113-114); up to 4 channels */
static void tex_sample(const tex_t *t, float cx, float cy, float *out)
{
   float px = cx * (float)t->w - 0.5f, py = cy * (float)t->h - 0.5f;
   float fx0 = floorf(px), fy0 = floorf(py);
   float fx = px - fx0, fy = py - fy0;
   int x0 = mirror((int)fx0, t->w), x1 = mirror((int)fx0 + 1, t->w);
   int y0 = mirror((int)fy0, t->h), y1 = mirror((int)fy0 + 1, t->h);
   float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
   const float *a = t->data + ((size_t)y0 * t->w + x0) * t->c, *b = t->data + ((size_t)y0 * t->w + x1) * t->c;
   const float *c = t->data + ((size_t)y1 * t->w + x0) * t->c, *d = t->data + ((size_t)y1 * t->w + x1) * t->c;
   for (int k = 0; k < t->c; ++k) out[k] = ((a[k] * w00 + b[k] * w10) + c[k] * w01) + d[k] * w11;
}
/* store with the texture's format: RGB8 clamps to [0,1] and rounds to 8 bits */
static float unorm8_level(float x) /* 0 .. 255 */
{
   if (x != x) return 0.0f;
   return rintf(clamp_(x, 0.0f, 1.0f) * 255.0f);
}
static float unorm8(float x) { return unorm8_level(x) / 255.0f; }
static void tex_store(tex_t *t, int x, int y, const float *v)
{
   float *p = t->data + ((size_t)y * t->w + x) * t->c;
   /* fp32 atlases (a build extension) flush NaN to 0 like the RGB8 conversion does */
   for (int k = 0; k < t->c; ++k) p[k] = t->unorm8 ? unorm8(v[k]) : (t->flush_nan && v[k] != v[k]) ? 0.0f : v[k];
}

/* ------------------------------------------------------------- probes: sampling */
/* glsl/render_probes.glsl:6-69 */
static v3 sample_irradiance(const orc_renderer *r, v3 pos, v3 normal)
{
   iv3 gp = world_to_grid(r, pos);
   v3 irradiance = V3(0, 0, 0);
   float total_weight = 0.0f;
   v3 sp = probe_spacing(r);
   v3 alpha = sub(vdiv(pos, sp), V3((float)gp.x, (float)gp.y, (float)gp.z));
   float ires = (float)r->probes.irradiance_resolution;
   float irr_min = 0.5f / ires, irr_max = 1.0f - irr_min; /* probe_utils.glsl:11-12 */
   for (int i = 0; i < 8; ++i) {
      iv3 o = {i & 1, (i >> 1) & 1, (i >> 2) & 1};
      iv3 q = {iclamp_(gp.x + o.x, 0, r->probes.grid_dimensions[0] - 1), iclamp_(gp.y + o.y, 0, r->probes.grid_dimensions[1] - 1),
               iclamp_(gp.z + o.z, 0, r->probes.grid_dimensions[2] - 1)};
      v3 hit_to_probe = sub(grid_to_world(r, q), pos);
      float probe_distance = length(hit_to_probe);
      v3 dir_to_probe = divs(hit_to_probe, probe_distance);
      float weight = 1.0f;
      float angle = (dot(dir_to_probe, normal) + 1.0f) * 0.5f;
      weight *= angle * angle + 0.2f;
      weight *= raycast_visibility(r, add(pos, scale(scale(normal, ORC_MIN_STEP), 5.0f)), dir_to_probe,
                                   probe_distance - ORC_MIN_STEP * 5.0f);
      const float crush = 0.2f;
      if (weight < crush) weight *= weight * weight * (1.0f / (crush * crush));
      v3 tri = V3(mix_(1.0f - alpha.x, alpha.x, (float)o.x), mix_(1.0f - alpha.y, alpha.y, (float)o.y),
                  mix_(1.0f - alpha.z, alpha.z, (float)o.z));
      weight *= tri.x * tri.y * tri.z;
      int probe_id = grid_to_probe_id(r, q);
      v2 base = probe_id_to_coord(r, probe_id);
      v2 rid = ray_dir_to_ray_id(normal);
      rid = V2(clamp_(rid.x, irr_min, irr_max), clamp_(rid.y, irr_min, irr_max));
      v2 coord = V2(base.x + rid.x / (float)r->probes.probe_count[0], base.y + rid.y / (float)r->probes.probe_count[1]);
      float tx[4];
      tex_sample(&r->tex[MDH_TEX_IRRADIANCE], coord.x, coord.y, tx);
      irradiance = add(irradiance, scale(vsqrt(V3(tx[0], tx[1], tx[2])), weight));
      total_weight += weight;
   }
   /* render_probes.glsl:65: 0/0 when every probe is occluded; fixed as 0 */
   if (total_weight == 0.0f) return V3(0, 0, 0);
   irradiance = divs(irradiance, total_weight);
   return mul(irradiance, irradiance);
}

/* MDH_OPT_RADIANCE_MIPS (not the reference's behaviour: its atlases have ONE level, render_passes.adb:113, so textureLod
   reads level 0 whatever lod says; the switch builds what probe_utils.glsl:17 and render_probes.glsl:84-86,131,197 were
   written for).  Level l of the radiance atlas = the 2x2 box filter of level l - 1 over the whole atlas image --
   ((a + b) + (c + d)) * 0.25 per channel in fp32, texels taken as stored and the result stored in the atlas's own format --
   down to one texel per probe (radiance_lods = int (log2 (radiance_resolution)) levels; the resolution must be a power
   of two, so that no box crosses a probe's tile). */
static void tex_alloc(tex_t *t, int w, int h, int c, int unorm);
static int radiance_lods(const orc_renderer *r)
{
   int lods = 0;
   while ((2 << lods) <= r->probes.radiance_resolution) ++lods;
   return lods;
}
static void build_radiance_mips(orc_renderer *r)
{
   const int lods = radiance_lods(r);
   const tex_t *src = &r->tex[MDH_TEX_RADIANCE];
   for (int l = 1; l <= lods; ++l) {
      tex_t *dst = &r->rad_mips[l];
      if (dst->w != src->w / 2 || dst->h != src->h / 2 || !dst->data) tex_alloc(dst, src->w / 2, src->h / 2, 3, 0);
      dst->unorm8 = r->tex[MDH_TEX_RADIANCE].unorm8; dst->flush_nan = r->tex[MDH_TEX_RADIANCE].flush_nan;
      for (int y = 0; y < dst->h; ++y)
         for (int x = 0; x < dst->w; ++x) {
            const float *a = src->data + ((size_t)(2 * y) * src->w + 2 * x) * 3, *b = a + 3;
            const float *c = src->data + ((size_t)(2 * y + 1) * src->w + 2 * x) * 3, *d = c + 3;
            float v[3];
            for (int k = 0; k < 3; ++k) v[k] = ((a[k] + b[k]) + (c[k] + d[k])) * 0.25f;
            tex_store(dst, x, y, v);
         }
      src = dst;
   }
}
/* textureLod (radiance_data, coord, lod) with GL_LINEAR_MIPMAP_LINEAR: lod clamped to the chain, the two nearest levels
   sampled bilinearly and mixed by the fraction (the upper one only when the fraction is not 0); level 0 without the switch */
static void sample_radiance_lod(const orc_renderer *r, float cx, float cy, float lod, float *out)
{
   if (!r->opt_mips) { tex_sample(&r->tex[MDH_TEX_RADIANCE], cx, cy, out); return; }
   const float top = (float)radiance_lods(r);
   const float d = clamp_(lod, 0.0f, top);
   const int l0 = (int)d;
   const float f = d - (float)l0;
   float lo[4], hi[4];
   tex_sample(l0 == 0 ? &r->tex[MDH_TEX_RADIANCE] : &r->rad_mips[l0], cx, cy, lo);
   if (!(f > 0.0f)) { out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; return; }
   tex_sample(&r->rad_mips[l0 + 1], cx, cy, hi);
   for (int k = 0; k < 3; ++k) out[k] = lo[k] * (1.0f - f) + hi[k] * f; /* mix () */
}

/* glsl/render_probes.glsl:138-209 (M_COMPUTE_INDIRECT_SPECULAR == 2, M_ADD_INDIRECT_SPECULAR == 1) */
static v3 sample_radiance_no_specular(const orc_renderer *r, v3 pos, v3 normal, v3 dir)
{
   int prim_index = -1;
   v3 spec_pos;
   v3 from = add(pos, scale(scale(normal, ORC_MIN_STEP), 5.0f));
   if (!raycast(r, from, dir, &prim_index, &spec_pos, 0, 0)) return V3(0, 0, 0);
   v3 spec_normal;
   int spec_mat;
   primitive_info(r, prim_index, spec_pos, &spec_normal, &spec_mat);
   iv3 gp = world_to_grid(r, spec_pos);
   float max_weight = -2.0f;
   iv3 best_q = {0, 0, 0};
   v3 best_pts = V3(0, 0, 1);
   for (int i = 0; i < 8; ++i) {
      iv3 o = {i & 1, (i >> 1) & 1, (i >> 2) & 1};
      iv3 q = {iclamp_(gp.x + o.x, 0, r->probes.grid_dimensions[0] - 1), iclamp_(gp.y + o.y, 0, r->probes.grid_dimensions[1] - 1),
               iclamp_(gp.z + o.z, 0, r->probes.grid_dimensions[2] - 1)};
      v3 probe_to_spec = sub(spec_pos, grid_to_world(r, q));
      float distance = length(probe_to_spec);
      probe_to_spec = divs(probe_to_spec, distance);
      float weight = dot(probe_to_spec, neg(spec_normal));
      weight *= raycast_visibility(r, add(spec_pos, scale(scale(spec_normal, ORC_MIN_STEP), 5.0f)), neg(probe_to_spec),
                                   distance - ORC_MIN_STEP * 5.0f);
      if (weight > max_weight) { max_weight = weight; best_q = q; best_pts = probe_to_spec; }
   }
   int probe_id = grid_to_probe_id(r, best_q);
   v2 base = probe_id_to_coord(r, probe_id);
   float rres = (float)r->probes.radiance_resolution;
   float rmin = 0.5f / rres, rmax = 1.0f - rmin; /* probe_utils.glsl:14-15 */
   v2 rid = ray_dir_to_ray_id(best_pts);
   rid = V2(clamp_(rid.x, rmin, rmax), clamp_(rid.y, rmin, rmax));
   v2 coord = V2(base.x + rid.x / (float)r->probes.probe_count[0], base.y + rid.y / (float)r->probes.probe_count[1]);
   float tx[4];
   sample_radiance_lod(r, coord.x, coord.y, 1.0f, tx); /* textureLod(.., 1.0): on a single level = lod 0 (MDH_OPT_RADIANCE_MIPS: level 1) */
   v3 radiance = V3(tx[0], tx[1], tx[2]);
   material_t m = get_material(r, spec_mat);
   radiance = add(radiance, compute_direct_lighting(r, spec_pos, spec_normal, dir, V3(0, 0, 0), m.metallic, m.roughness, 1));
   return radiance;
}

/* glsl/render_probes.glsl:71-136 (M_COMPUTE_INDIRECT_SPECULAR == 1).  radiance_lods = int(log2(radiance_resolution))
   (probe_utils.glsl:17) is taken as the position of the highest set bit; the atlases have one level, so
   textureLod(.., lod) is level 0 and lod only narrows the clamp of the tap; a total weight of 0
   (trilinear factors of a point outside the grid can cancel) gives 0, as for the irradiance. */
static v3 sample_radiance_with_specular(const orc_renderer *r, v3 pos, v3 normal, v3 dir, float roughness)
{
   v3 spec_pos;
   v3 from = add(pos, scale(scale(normal, ORC_MIN_STEP), 5.0f));
   if (!raycast_hit_position(r, from, dir, r->max_dist, &spec_pos)) return V3(0, 0, 0);
   v3 pos_to_spec_pos = sub(spec_pos, pos);
   iv3 gp = world_to_grid(r, pos);
   v3 alpha = sub(vdiv(pos, probe_spacing(r)), V3((float)gp.x, (float)gp.y, (float)gp.z));
   int lods = 0;
   while ((2 << lods) <= r->probes.radiance_resolution) ++lods;
   float lod = mix_(0.0f, (float)lods, roughness * 2.0f);
   int new_res = r->probes.radiance_resolution / (int)(lod + 1.0f);
   float rmin = 0.5f / (float)new_res, rmax = 1.0f - rmin;
   float total_weight = 0.0f;
   v3 radiance = V3(0, 0, 0);
   for (int i = 0; i < 8; ++i) {
      iv3 o = {i & 1, (i >> 1) & 1, (i >> 2) & 1};
      iv3 q = {iclamp_(gp.x + o.x, 0, r->probes.grid_dimensions[0] - 1), iclamp_(gp.y + o.y, 0, r->probes.grid_dimensions[1] - 1),
               iclamp_(gp.z + o.z, 0, r->probes.grid_dimensions[2] - 1)};
      v3 probe_to_pos = sub(pos, grid_to_world(r, q));
      v3 probe_to_spec = add(probe_to_pos, pos_to_spec_pos);
      float distance = length(probe_to_spec);
      probe_to_spec = divs(probe_to_spec, distance);
      float weight = fmax_(softshadows(r, spec_pos, neg(probe_to_spec), ORC_MIN_STEP * 5.0f, distance - ORC_MIN_STEP * 5.0f, 0.5f), 0.001f);
      v3 tri = V3(mix_(1.0f - alpha.x, alpha.x, (float)o.x), mix_(1.0f - alpha.y, alpha.y, (float)o.y),
                  mix_(1.0f - alpha.z, alpha.z, (float)o.z));
      weight *= tri.x * tri.y * tri.z;
      v2 base = probe_id_to_coord(r, grid_to_probe_id(r, q));
      v2 rid = ray_dir_to_ray_id(probe_to_spec);
      rid = V2(clamp_(rid.x, rmin, rmax), clamp_(rid.y, rmin, rmax));
      v2 coord = V2(base.x + rid.x / (float)r->probes.probe_count[0], base.y + rid.y / (float)r->probes.probe_count[1]);
      float tx[4];
      sample_radiance_lod(r, coord.x, coord.y, lod, tx);
      radiance = add(radiance, scale(V3(tx[0], tx[1], tx[2]), weight));
      total_weight += weight;
   }
   if (total_weight == 0.0f) return V3(0, 0, 0);
   return divs(radiance, total_weight);
}

/* glsl/render_probes.glsl:211-244 (M_COMPUTE_INDIRECT_SPECULAR == 3): the reflection's hit shaded in full */
static v3 compute_indirect_lighting(v3 irradiance, v3 radiance, v3 V, v3 N, v3 L, v3 albedo, float metallic, float roughness);
static v3 compute_indirect_specular(const orc_renderer *r, v3 pos, v3 normal, v3 dir, int direct_specular)
{
   int prim_index = -1;
   v3 spec_pos;
   v3 from = add(pos, scale(scale(normal, ORC_MIN_STEP), 5.0f));
   if (!raycast(r, from, dir, &prim_index, &spec_pos, 0, 0)) {
      float s = dir.y * 0.7f;
      return V3(0.30f - s, 0.36f - s, 0.60f - s);
   }
   v3 spec_normal;
   int spec_mat;
   primitive_info(r, prim_index, spec_pos, &spec_normal, &spec_mat);
   material_t m = get_material(r, spec_mat);
   v3 direct = compute_direct_lighting(r, spec_pos, spec_normal, dir, m.albedo, m.metallic, m.roughness, direct_specular);
   v3 irradiance = sample_irradiance(r, spec_pos, spec_normal);
   v3 specular_dir = reflect(dir, spec_normal);
   v3 indirect = compute_indirect_lighting(irradiance, V3(0, 0, 0), neg(dir), spec_normal, specular_dir, m.albedo, m.metallic, m.roughness);
   return add(indirect, direct);
}

/* -------------------------------------------------------------------- volumetrics */
#define TAU_SCATTERING 0.1f /* glsl/volumetrics.glsl:12 */
/* glsl/volumetrics.glsl:21-30 */
static float henvey_greenstein_phase(v3 in_dir, v3 out_dir)
{
   float cos_angle = dot(in_dir, out_dir);
   float t2 = TAU_SCATTERING * TAU_SCATTERING;
   float result = 1.0f - t2;
   result /= 4.0f * ORC_PI * pow1_5_(1.0f + t2 - 2.0f * TAU_SCATTERING * cos_angle);
   return result;
}
/* glsl/volumetrics.glsl:34-54; `hit` = 0 fixes len = max_dist */
static v3 render_volumetrics(const orc_renderer *r, v3 L, v3 from, v3 to, int hit, v2 frag_pos)
{
   v2 tc = V2((frag_pos.x + 1.0f) * 0.5f, (frag_pos.y + 1.0f) * 0.5f);
   float len = hit ? length(sub(to, from)) : r->max_dist;
   float closest = r->max_dist;
   v3 fog = V3(0, 0, 0);
   float sx = 1.0f / (float)r->vol.scattering_resolution[0], sy = 1.0f / (float)r->vol.scattering_resolution[1];
   for (int x = -1; x <= 1; ++x)
      for (int y = -1; y <= 1; ++y) {
         float d[4];
         tex_sample(&r->tex[MDH_TEX_SCATTERING], tc.x + (float)x * sx, tc.y + (float)y * sy, d);
         float dist = fabsf(d[3] - len);
         if (dist < closest) { closest = dist; fog = V3(d[0], d[1], d[2]); }
      }
   return add(scale(L, exp_(-len * TAU_SCATTERING)), fog);
}

/* ------------------------------------------------------------ pixel_color_probes */
typedef struct {
   int direct_specular;   /* M_COMPUTE_DIRECT_SPECULAR   */
   int indirect_specular; /* M_COMPUTE_INDIRECT_SPECULAR (0 .. 3) */
   int ao_steps;          /* M_AMBIENT_OCCLUSION_STEPS   */
   int volumetrics;       /* M_RENDER_VOLUMETRICS        */
   int mode;              /* MDH_OPT_SCREEN_MODE         */
} pass_cfg;

/* glsl/render_probes.glsl:246-291 */
static v3 pixel_color_probes(const orc_renderer *r, const pass_cfg *cfg, v3 from, v3 dir, v2 frag_pos, int *gb_index,
                             float *gb_t, int *gb_steps)
{
   int prim_index = -1, steps = 0;
   float t = 0.0f;
   v3 pos = V3(0, 0, 0), result;
   int hit = raycast(r, from, dir, &prim_index, &pos, &t, &steps);
   if (gb_index) { *gb_index = hit ? prim_index : -1; *gb_t = hit ? t : 0.0f; *gb_steps = steps; }
   if (hit) {
      int material_id;
      v3 normal;
      primitive_info(r, prim_index, pos, &normal, &material_id);
      if (cfg->mode == 1) return add(scale(normal, 0.5f), v3s(0.5f));
      material_t m = get_material(r, material_id);
      v3 direct = compute_direct_lighting(r, pos, normal, dir, m.albedo, m.metallic, m.roughness, cfg->direct_specular);
      if (cfg->mode == 2) {
         result = scale(direct, compute_ambient_occlusion(r, pos, normal, cfg->ao_steps));
      } else {
         v3 irradiance = sample_irradiance(r, pos, normal);
         v3 specular_col = V3(0, 0, 0);
         v3 specular_dir = reflect(dir, normal);
         if (cfg->indirect_specular > 0 && m.roughness < 0.75f) { /* render_probes.glsl:264-272 */
            if (cfg->indirect_specular == 1) specular_col = sample_radiance_with_specular(r, pos, normal, specular_dir, m.roughness);
            else if (cfg->indirect_specular == 2) specular_col = sample_radiance_no_specular(r, pos, normal, specular_dir);
            else specular_col = compute_indirect_specular(r, pos, normal, specular_dir, cfg->direct_specular);
         }
         v3 indirect = compute_indirect_lighting(irradiance, specular_col, neg(dir), normal, specular_dir, m.albedo, m.metallic, m.roughness);
         float ao = compute_ambient_occlusion(r, pos, normal, cfg->ao_steps);
         result = scale(add(direct, indirect), ao);
      }
   } else {
      float s = dir.y * 0.7f;
      result = V3(0.30f - s, 0.36f - s, 0.60f - s);
   }
   if (cfg->volumetrics) result = render_volumetrics(r, result, from, pos, hit, frag_pos);
   return result;
}

/* camera (glsl/draw_screen.glsl:21-24); M column-major */
static v3 mat_mul(const float *m, v3 v)
{
   return V3((m[0] * v.x + m[3] * v.y) + m[6] * v.z, (m[1] * v.x + m[4] * v.y) + m[7] * v.z, (m[2] * v.x + m[5] * v.y) + m[8] * v.z);
}
static void camera_ray(const orc_renderer *r, float u, float v, v3 *origin, v3 *dir)
{
   v3 frag = V3(u, v, 0.0f);
   v3 d = normalize(sub(frag, V3(0.0f, 0.0f, -1.5f)));
   *dir = mat_mul(r->cam_m, d);
   *origin = add(mat_mul(r->cam_m, frag), V3(r->cam_pos[0], r->cam_pos[1], r->cam_pos[2]));
}
/* texel/pixel centre in [-1,1]: (2 i + 1) / n - 1 */
static float centre(int i, int n) { return (float)(2 * i + 1) / (float)n - 1.0f; }

/* ------------------------------------------------------------------------- passes */
static int probe_total(const orc_renderer *r) { return r->probes.probe_count[0] * r->probes.probe_count[1]; }
static void own_probes(const orc_renderer *r, int *begin, int *end)
{
   int P = probe_total(r);
   *begin = (int)((int64_t)P * r->opt_rank / r->opt_world);
   *end = (int)((int64_t)P * (r->opt_rank + 1) / r->opt_world);
}
static int nthreads(const orc_renderer *r)
{
#ifdef _OPENMP
   return r->opt_threads > 0 ? r->opt_threads : omp_get_max_threads();
#else
   (void)r;
   return 1;
#endif
}

/* compute_probe_radiance.glsl:16-27 */
static void pass_radiance(orc_renderer *r)
{
   tex_t *t = &r->tex[MDH_TEX_RADIANCE];
   pass_cfg cfg = {0, 0, 0, 0, 0}; /* renderers.adb:115-117; AO/volumetrics macros undefined (Q12) */
   int pb, pe;
   own_probes(r, &pb, &pe);
   uint64_t evals = 0, rays = 0, steps = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads(r)) reduction(+ : evals, rays, steps)
   for (int j = 0; j < t->h; ++j) {
      t_sdf_evals = 0; t_rays = 0; t_steps = 0;
      for (int i = 0; i < t->w; ++i) {
         v2 nc = V2((centre(i, t->w) + 1.0f) * 0.5f, (centre(j, t->h) + 1.0f) * 0.5f);
         int probe_id = coord_to_probe_id(r, nc);
         if (probe_id < pb || probe_id >= pe) continue;
         v3 world = grid_to_world(r, probe_id_to_grid_position(r, probe_id));
         v3 ray_dir = ray_id_to_ray_dir(coord_to_ray_id(r, nc));
         v3 c = pixel_color_probes(r, &cfg, world, ray_dir, nc, 0, 0, 0);
         float o[3] = {c.x, c.y, c.z};
         tex_store(t, i, j, o);
      }
      evals += t_sdf_evals; rays += t_rays; steps += t_steps;
   }
   r->sdf_evals += evals;
   r->work[MDH_PASS_RADIANCE][0] = rays; r->work[MDH_PASS_RADIANCE][1] = steps; r->work[MDH_PASS_RADIANCE][2] = evals;
}

/* update_probe_irradiance.glsl:8-43 */
static void pass_irradiance(orc_renderer *r)
{
   tex_t *t = &r->tex[MDH_TEX_IRRADIANCE];
   const tex_t *rad = &r->tex[MDH_TEX_RADIANCE];
   int pb, pe;
   own_probes(r, &pb, &pe);
   if (r->opt_world > 1 && r->opt_irr_all) { pb = 0; pe = probe_total(r); } /* MDH_OPT_IRRADIANCE_ALL */
   int rres = r->probes.radiance_resolution;
   float pcx = (float)r->probes.probe_count[0], pcy = (float)r->probes.probe_count[1];
   v2 step = V2(1.0f / pcx / (float)rres, 1.0f / pcy / (float)rres);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads(r))
   for (int j = 0; j < t->h; ++j)
      for (int i = 0; i < t->w; ++i) {
         v2 nc = V2((centre(i, t->w) + 1.0f) * 0.5f, (centre(j, t->h) + 1.0f) * 0.5f);
         int probe_id = coord_to_probe_id(r, nc);
         if (probe_id < pb || probe_id >= pe) continue;
         v3 irr_dir = ray_id_to_ray_dir(coord_to_ray_id(r, nc));
         v2 rad_coord = probe_id_to_coord(r, probe_id);
         v3 irradiance = V3(0, 0, 0);
         float total_weight = 0.0f;
         for (int y = 0; y < rres; ++y)
            for (int x = 0; x < rres; ++x) {
               v2 offset = V2((float)x * step.x, (float)y * step.y);
               v2 c = V2(clamp_(rad_coord.x + offset.x, step.x, 1.0f - step.x), clamp_(rad_coord.y + offset.y, step.y, 1.0f - step.y));
               float tx[4];
               tex_sample(rad, c.x, c.y, tx);
               v3 rad_dir = ray_id_to_ray_dir(coord_to_ray_id(r, c));
               float w = fmax_(dot(irr_dir, rad_dir), 0.0f);
               irradiance = add(irradiance, scale(V3(tx[0], tx[1], tx[2]), w));
               total_weight += w;
            }
         irradiance = divs(irradiance, total_weight);
         if (r->opt_hyst > 0) { /* MDH_OPT_HYSTERESIS_PERMILLE (not in the reference): mix (fresh, previous, h) */
            const float h = (float)r->opt_hyst / 1000.0f;
            const float *old = t->data + ((size_t)j * t->w + i) * t->c;
            irradiance = V3(mix_(irradiance.x, old[0], h), mix_(irradiance.y, old[1], h), mix_(irradiance.z, old[2], h));
         }
         float o[3] = {irradiance.x, irradiance.y, irradiance.z};
         tex_store(t, i, j, o);
      }
}

/* compute_frustrum_visibility.glsl:8-42 */
static void pass_visibility(orc_renderer *r)
{
   tex_t *t = &r->tex[MDH_TEX_VISIBILITY];
   float dz = (float)r->vol.visibility_resolution[2];
   float vstep = image_roundtrip(r->vol.visibility_step_size);
   uint64_t evals = 0, rays = 0, steps = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads(r)) reduction(+ : evals, rays, steps)
   for (int j = 0; j < t->h; ++j) {
      t_sdf_evals = 0; t_rays = 0; t_steps = 0;
      for (int i = 0; i < t->w; ++i) {
         float px = centre(i, t->w), py = centre(j, t->h);
         float norm_height = (py + 1.0f) * 0.5f;
         float tex_height = norm_height * dz;
         float depth = floorf(tex_height);
         float fract_height = tex_height - depth;
         float frag_height = fract_height * 2.0f - 1.0f;
         v3 origin, dir;
         camera_ray(r, px, frag_height, &origin, &dir);
         v3 pos = add(origin, scale(scale(dir, depth), vstep));
         v3 result = V3(0, 0, 0);
         int total = ubo_i(r, r->total_light_off);
         for (int l = 0; l < total; ++l) { /* sample_lights :8-19 */
            v3 L;
            float L_dist;
            v3 radiance = sample_light(r, l, pos, V3(1, 0, 0), &L, &L_dist);
            float visibility = raycast_visibility(r, pos, L, L_dist);
            v3 L_in = scale(radiance, exp_(-L_dist * TAU_SCATTERING) * visibility);
            result = add(result, scale(scale(L_in, TAU_SCATTERING), henvey_greenstein_phase(L, dir)));
         }
         float o[3] = {result.x, result.y, result.z};
         tex_store(t, i, j, o);
      }
      evals += t_sdf_evals; rays += t_rays; steps += t_steps;
   }
   r->sdf_evals += evals;
   r->work[MDH_PASS_VISIBILITY][0] = rays; r->work[MDH_PASS_VISIBILITY][1] = steps; r->work[MDH_PASS_VISIBILITY][2] = evals;
}

/* accumulate_scattering.glsl:9-48 */
static void pass_scattering(orc_renderer *r)
{
   tex_t *t = &r->tex[MDH_TEX_SCATTERING];
   const tex_t *vis = &r->tex[MDH_TEX_VISIBILITY];
   float dz = (float)r->vol.visibility_resolution[2];
   float vstep = image_roundtrip(r->vol.visibility_step_size);
   float sstep = image_roundtrip(r->vol.scattering_step_size);
   float max_depth = vstep * dz; /* volumetrics.glsl:3-4 */
   uint64_t evals = 0, rays = 0, steps = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads(r)) reduction(+ : evals, rays, steps)
   for (int j = 0; j < t->h; ++j) {
      t_sdf_evals = 0; t_rays = 0; t_steps = 0;
      for (int i = 0; i < t->w; ++i) {
         float px = centre(i, t->w), py = centre(j, t->h);
         v3 from, dir;
         camera_ray(r, px, py, &from, &dir);
         v2 norm_pos = V2(0.5f * (px + 1.0f), 0.5f * (py + 1.0f));
         v3 to = add(from, scale(dir, max_depth));
         int idx = -1;
         raycast(r, from, dir, &idx, &to, 0, 0);
         float len = fmin_(length(sub(to, from)), max_depth);
         v3 L = V3(0, 0, 0);
         for (float f = 0.0f; f < len; f += sstep) {
            float rel = floorf(f / vstep); /* sample_visibility :9-15 */
            float tx[4];
            tex_sample(vis, norm_pos.x, (norm_pos.y + rel) / dz, tx);
            L = add(L, scale(V3(tx[0], tx[1], tx[2]), exp_(-f * TAU_SCATTERING)));
         }
         L = scale(L, sstep);
         float o[4] = {L.x, L.y, L.z, len};
         tex_store(t, i, j, o);
      }
      evals += t_sdf_evals; rays += t_rays; steps += t_steps;
   }
   r->sdf_evals += evals;
   r->work[MDH_PASS_SCATTERING][0] = rays; r->work[MDH_PASS_SCATTERING][1] = steps; r->work[MDH_PASS_SCATTERING][2] = evals;
}

/* 8x8 screen tiles dealt round-robin over the ranks of a sharded run */
static int tile_owner(const orc_renderer *r, int px, int py)
{
   int tiles_x = (r->W + 7) / 8;
   return ((py / 8) * tiles_x + (px / 8)) % r->opt_world;
}

/* draw_screen.glsl:20-30 */
static void pass_screen(orc_renderer *r)
{
   if (r->opt_mips && r->opt_mode == 0) build_radiance_mips(r); /* (every screen pass: the levels of the atlas it is about to read) */
   pass_cfg cfg = {1, r->opt_spec, r->opt_ao, r->vol.enabled ? 1 : 0, r->opt_mode}; /* renderers.adb:136-143; MDH_OPT_INDIRECT_SPECULAR */
   if (cfg.mode != 0) cfg.volumetrics = 0;
   uint64_t evals = 0, rays = 0, steps = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads(r)) reduction(+ : evals, rays, steps)
   for (int j = 0; j < r->H; ++j) {
      t_sdf_evals = 0; t_rays = 0; t_steps = 0;
      for (int i = 0; i < r->W; ++i) {
         float *o = r->fb + ((size_t)j * r->W + i) * 3;
         if (r->opt_world > 1 && tile_owner(r, i, j) != r->opt_rank) { o[0] = o[1] = o[2] = 0.0f; continue; }
         float u = centre(i, r->W), v = -centre(j, r->H); /* row 0 = top */
         v3 origin, dir;
         camera_ray(r, u, v, &origin, &dir);
         size_t px = (size_t)j * r->W + i;
         v3 c = pixel_color_probes(r, &cfg, origin, dir, V2(u, v), r->opt_gbuffer ? &r->gb_index[px] : 0,
                                   r->opt_gbuffer ? &r->gb_t[px] : 0, r->opt_gbuffer ? &r->gb_steps[px] : 0);
         if (cfg.mode != 1) /* draw_screen.glsl:29 */
            c = V3(pow_(c.x / (c.x + 1.0f), 0.4545f), pow_(c.y / (c.y + 1.0f), 0.4545f), pow_(c.z / (c.z + 1.0f), 0.4545f));
         o[0] = c.x; o[1] = c.y; o[2] = c.z;
      }
      evals += t_sdf_evals; rays += t_rays; steps += t_steps;
   }
   r->sdf_evals += evals;
   r->work[MDH_PASS_SCREEN][0] = rays; r->work[MDH_PASS_SCREEN][1] = steps; r->work[MDH_PASS_SCREEN][2] = evals;
}

/* ---------------------------------------------------------------------- C ABI */
static const char *PRIM_NAMES[4] = {"Sphere", "Plane", "Box", "Triangle"};
static const char *LIGHT_NAMES[2] = {"PointLight", "SpotLight"};

static int resolve_kind(kind_t *k, const mdh_kind_decl *d, int is_light)
{
   /* A kind's behaviour is its expressions (madarch-primitives.ads:24-30, madarch-lights.ads:20-24), never its name:
      one that brings programs runs them, whatever it is called; the hand-written built-in paths are taken only by
      a kind that brings none and carries one of the library's own names (the reference's own six kinds). */
   k->type = -1;
   int n = is_light ? 2 : 4;
   const int has_programs = d->dist_code || d->normal_code || d->material_code;
   if (!has_programs)
      for (int t = 0; t < n; ++t)
         if (strcmp(d->name, is_light ? LIGHT_NAMES[t] : PRIM_NAMES[t]) == 0) k->type = t;
   int custom = has_programs && d->dist_code && d->normal_code && (is_light || d->material_code);
   if (custom) k->type = is_light ? (int)LK_CUSTOM : (int)PK_CUSTOM;
   if (k->type < 0 || d->n_components > 8 || d->n_components < 1) return 0;
   k->max_count = d->max_count;
   k->ncomp = d->n_components;
   for (int i = 0; i < d->n_components; ++i) {
      snprintf(k->comp_name[i], sizeof k->comp_name[i], "%s", d->components[i].name);
      k->comp_kind[i] = d->components[i].kind;
   }
   layout_kind(k);
   k->f_a = k->f_b = k->f_c = k->f_mat = -1;
   if (custom) {
      const int32_t *src[3] = {d->dist_code, d->normal_code, d->material_code};
      const int len[3] = {d->dist_len, d->normal_len, d->material_len};
      k->inst_floats = 0;
      for (int c = 0; c < k->ncomp; ++c)
         for (int j = 0; j < (k->comp_kind[c] == MDH_VEC3 ? 3 : 1); ++j) k->x_float_off[k->inst_floats++] = k->comp_off[c] + 4 * j;
      for (int q = 0; q < (is_light ? 2 : 3); ++q) {
         if (len[q] < 1 || len[q] > MDH_X_MAX_WORDS) return 0;
         for (int pc = 0; pc < len[q]; ++pc) { /* the operands this interpreter indexes with */
            const uint32_t w = (uint32_t)src[q][pc];
            const int op = w & 255, a = (w >> 16) & 255;
            if (op >= MDH_X_OPS || (op == MDH_X_COMP && a >= k->inst_floats)) return 0;
            if (op == MDH_X_LIT || op == MDH_X_SEL) { if (++pc >= len[q]) return 0; }
         }
         k->x_code[q] = (int32_t *)malloc(sizeof(int32_t) * (size_t)len[q]);
         memcpy(k->x_code[q], src[q], sizeof(int32_t) * (size_t)len[q]);
         k->x_len[q] = len[q];
      }
      return 1;
   }
   if (!is_light) {
      k->f_mat = comp_offset(k, "material_id", MDH_INT);
      switch (k->type) {
      case PK_SPHERE: k->f_a = comp_offset(k, "center", MDH_VEC3); k->f_b = comp_offset(k, "radius", MDH_FLOAT); k->f_c = 0; break;
      case PK_PLANE: k->f_a = comp_offset(k, "normal", MDH_VEC3); k->f_b = comp_offset(k, "offset", MDH_FLOAT); k->f_c = 0; break;
      case PK_BOX: k->f_a = comp_offset(k, "center", MDH_VEC3); k->f_b = comp_offset(k, "side", MDH_VEC3); k->f_c = 0; break;
      default: k->f_a = comp_offset(k, "v1", MDH_VEC3); k->f_b = comp_offset(k, "v2", MDH_VEC3); k->f_c = comp_offset(k, "v3", MDH_VEC3); break;
      }
   } else {
      k->f_a = comp_offset(k, "position", MDH_VEC3);
      if (k->type == LK_POINT) { k->f_b = comp_offset(k, "color", MDH_VEC3); k->f_c = 0; k->f_mat = 0; }
      else { k->f_b = comp_offset(k, "direction", MDH_VEC3); k->f_c = comp_offset(k, "aperture", MDH_FLOAT); k->f_mat = comp_offset(k, "color", MDH_VEC3); }
   }
   return k->f_a >= 0 && k->f_b >= 0 && k->f_c >= 0 && k->f_mat >= 0;
}

static void tex_alloc(tex_t *t, int w, int h, int c, int unorm)
{
   t->w = w; t->h = h; t->c = c; t->unorm8 = unorm;
   free(t->data);
   t->data = (float *)calloc((size_t)w * h * c, sizeof(float));
}

int32_t orc_create(int32_t width, int32_t height, const mdh_scene_desc *scene, const mdh_probe_settings *probes,
                   const mdh_volumetrics *vol, int32_t device, orc_renderer **out)
{
   (void)device;
   if (!scene || !probes || !vol || !out || width <= 0 || height <= 0) return seterr(MDH_E_INVALID, "bad argument");
   if (scene->n_prim_kinds > MAX_KINDS || scene->n_light_kinds > MAX_KINDS) return seterr(MDH_E_INVALID, "too many kinds");
   /* Setup_Probe_Layout, renderers.adb:54-65 */
   if (probes->grid_dimensions[0] * probes->grid_dimensions[1] * probes->grid_dimensions[2] != probes->probe_count[0] * probes->probe_count[1])
      return seterr(MDH_E_PROBE_MISMATCH, "Probe_Count should match grid dimensions.");
   orc_renderer *r = (orc_renderer *)calloc(1, sizeof *r);
   r->W = width; r->H = height;
   r->npk = scene->n_prim_kinds; r->nlk = scene->n_light_kinds;
   /* Compute_Scene_GPU_Type, scenes.adb:1268-1345: {int count; Kind array[max]} per kind, then total_light_count */
   int off = 0, base = 0;
   for (int k = 0; k < r->npk; ++k) {
      if (!resolve_kind(&r->pk[k], &scene->prim_kinds[k], 0)) { free(r); return seterr(MDH_E_UNSUPPORTED_KIND, "unsupported primitive kind"); }
      off = pad_to(off, 4); r->pk[k].count_off = off; off += 4;
      off = pad_to(off, 16); r->pk[k].array_off = off; off += r->pk[k].stride * r->pk[k].max_count;
      r->prim_base[k] = base; base += r->pk[k].max_count;
   }
   for (int k = 0; k < r->nlk; ++k) {
      if (!resolve_kind(&r->lk[k], &scene->light_kinds[k], 1)) { free(r); return seterr(MDH_E_UNSUPPORTED_KIND, "unsupported light kind"); }
      off = pad_to(off, 4); r->lk[k].count_off = off; off += 4;
      off = pad_to(off, 16); r->lk[k].array_off = off; off += r->lk[k].stride * r->lk[k].max_count;
   }
   off = pad_to(off, 4); r->total_light_off = off; off += 4;
   r->scene_ubo_size = off;
   r->scene_ubo = (uint8_t *)calloc(1, (size_t)off + 16);
   r->max_dist = image_roundtrip(scene->max_dist);
   r->part = scene->partitioning;
   if (r->part.enable) {
      for (int a = 0; a < 3; ++a) { r->pg_spacing[a] = image_roundtrip(r->part.grid_spacing[a]); r->pg_offset[a] = image_roundtrip(r->part.grid_offset[a]); }
      r->part_cells = r->part.grid_dimensions[0] * r->part.grid_dimensions[1] * r->part.grid_dimensions[2];
      r->part_table = (int *)calloc((size_t)r->part_cells * (r->npk + r->part.index_count), sizeof(int));
      v3 sp = V3(scene->partitioning.grid_spacing[0], scene->partitioning.grid_spacing[1], scene->partitioning.grid_spacing[2]);
      r->part_gpu_diag = image_roundtrip(length(sp)); /* the CPU builders use the settings record, not the GLSL text */
   }
   r->probes = *probes;
   r->vol = *vol;
   r->cam_m[0] = r->cam_m[4] = r->cam_m[8] = 1.0f; /* renderers.adb:225-226 */
   r->opt_ao = 3; r->opt_world = 1; r->opt_ada_div = 1; r->opt_irr_all = 1; r->opt_window = 2; r->opt_spec = 2;
   tex_alloc(&r->tex[MDH_TEX_RADIANCE], probes->radiance_resolution * probes->probe_count[0], probes->radiance_resolution * probes->probe_count[1], 3, 1);
   tex_alloc(&r->tex[MDH_TEX_IRRADIANCE], probes->irradiance_resolution * probes->probe_count[0], probes->irradiance_resolution * probes->probe_count[1], 3, 1);
   r->tex[MDH_TEX_RADIANCE].flush_nan = r->tex[MDH_TEX_IRRADIANCE].flush_nan = 1;
   tex_alloc(&r->tex[MDH_TEX_VISIBILITY], vol->visibility_resolution[0], vol->visibility_resolution[1] * vol->visibility_resolution[2], 3, 0);
   tex_alloc(&r->tex[MDH_TEX_SCATTERING], vol->scattering_resolution[0], vol->scattering_resolution[1], 4, 0);
   r->fb = (float *)calloc((size_t)width * height * 3, sizeof(float));
   r->gb_index = (int32_t *)calloc((size_t)width * height, sizeof(int32_t));
   r->gb_steps = (int32_t *)calloc((size_t)width * height, sizeof(int32_t));
   r->gb_t = (float *)calloc((size_t)width * height, sizeof(float));
   *out = r;
   return MDH_OK;
}

int32_t orc_destroy(orc_renderer *r)
{
   if (!r) return MDH_OK;
   for (int i = 0; i < 4; ++i) free(r->tex[i].data);
   for (int i = 0; i < 16; ++i) free(r->rad_mips[i].data);
   for (int k = 0; k < MAX_KINDS; ++k)
      for (int q = 0; q < 3; ++q) { free(r->pk[k].x_code[q]); free(r->lk[k].x_code[q]); }
   free(r->scene_ubo); free(r->part_table); free(r->fb); free(r->front); free(r->gb_index); free(r->gb_steps); free(r->gb_t);
   free(r);
   return MDH_OK;
}

int32_t orc_set_option(orc_renderer *r, int32_t option, int32_t value)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   switch (option) {
   case MDH_OPT_ATLAS_FORMAT:
      r->opt_atlas = value;
      r->tex[MDH_TEX_RADIANCE].unorm8 = r->tex[MDH_TEX_IRRADIANCE].unorm8 = (value == 0);
      break;
   case MDH_OPT_SCREEN_MODE: r->opt_mode = value; break;
   case MDH_OPT_AO_STEPS: r->opt_ao = value; break;
   case MDH_OPT_GBUFFER: r->opt_gbuffer = value; break;
   case MDH_OPT_RANK: r->opt_rank = value; break;
   case MDH_OPT_WORLD: if (value < 1) return seterr(MDH_E_INVALID, "world < 1"); r->opt_world = value; break;
   case MDH_OPT_TIMING: r->opt_timing = value; break;
   case MDH_OPT_ADA_EVAL_DIV: r->opt_ada_div = value; break;
   case MDH_OPT_FRAME_OVERLAP: break; /* scheduling only: nothing to restate */
   case MDH_OPT_JIT: break;           /* how the kernels run the MDH_X programs: nothing to restate */
   case MDH_OPT_IRRADIANCE_ALL: r->opt_irr_all = value ? 1 : 0; break;
   case MDH_OPT_WINDOW: /* where the pixels are converted: no effect on them */
      if (value < 0 || value > 2) return seterr(MDH_E_INVALID, "bad value");
      r->opt_window = value;
      break;
   case MDH_OPT_INDIRECT_SPECULAR:
      if (value < 0 || value > 3) return seterr(MDH_E_INVALID, "indirect specular mode is 0 .. 3");
      r->opt_spec = value;
      break;
   case MDH_OPT_HYSTERESIS_PERMILLE:
      if (value < 0 || value > 999) return seterr(MDH_E_INVALID, "hysteresis is 0 .. 999 per mille");
      r->opt_hyst = value;
      break;
   case MDH_OPT_RADIANCE_ORDER: break; /* which lane computes a texel, not what it holds: nothing to restate */
   case MDH_OPT_SCREEN_ORDER: break;   /* the order the tiles are started in: nothing to restate */
   case MDH_OPT_SCREEN_SPLIT: break;   /* how many wavefronts draw a tile: nothing to restate */
   case MDH_OPT_NUMERICS: if (value != 0) return seterr(MDH_E_STATE, "the oracle's numerics are the contract"); break;
   case MDH_OPT_RADIANCE_MIPS:
      if (value && (r->probes.radiance_resolution & (r->probes.radiance_resolution - 1)) != 0)
         return seterr(MDH_E_INVALID, "radiance mips need a power-of-two radiance resolution");
      r->opt_mips = value ? 1 : 0;
      break;
   case ORC_OPT_SDF_MODE: r->opt_sdf_mode = value; break;
   case ORC_OPT_THREADS: r->opt_threads = value; break;
   default: return seterr(MDH_E_INVALID, "unknown option");
   }
   return MDH_OK;
}
int32_t orc_get_option(orc_renderer *r, int32_t option, int32_t *value)
{
   if (!r || !value) return seterr(MDH_E_INVALID, "bad argument");
   switch (option) {
   case MDH_OPT_ATLAS_FORMAT: *value = r->opt_atlas; break;
   case MDH_OPT_SCREEN_MODE: *value = r->opt_mode; break;
   case MDH_OPT_AO_STEPS: *value = r->opt_ao; break;
   case MDH_OPT_GBUFFER: *value = r->opt_gbuffer; break;
   case MDH_OPT_RANK: *value = r->opt_rank; break;
   case MDH_OPT_WORLD: *value = r->opt_world; break;
   case MDH_OPT_TIMING: *value = r->opt_timing; break;
   case MDH_OPT_ADA_EVAL_DIV: *value = r->opt_ada_div; break;
   case MDH_OPT_FRAME_OVERLAP: *value = 0; break;
   case MDH_OPT_JIT: *value = 0; break;
   case MDH_OPT_IRRADIANCE_ALL: *value = r->opt_irr_all; break;
   case MDH_OPT_WINDOW: *value = r->opt_window; break;
   case MDH_OPT_INDIRECT_SPECULAR: *value = r->opt_spec; break;
   case MDH_OPT_HYSTERESIS_PERMILLE: *value = r->opt_hyst; break;
   case MDH_OPT_RADIANCE_MIPS: *value = r->opt_mips; break;
   case MDH_OPT_RADIANCE_ORDER: *value = 0; break;
   case MDH_OPT_SCREEN_ORDER: *value = 0; break;
   case MDH_OPT_SCREEN_SPLIT: *value = 0; break;
   case MDH_OPT_NUMERICS: *value = 0; break;
   case ORC_OPT_SDF_MODE: *value = r->opt_sdf_mode; break;
   case ORC_OPT_THREADS: *value = nthreads(r); break;
   default: return seterr(MDH_E_INVALID, "unknown option");
   }
   return MDH_OK;
}

/* Set_Material, renderers.adb:349-367 */
int32_t orc_set_material(orc_renderer *r, int32_t id0, const float albedo[3], float metallic, float roughness)
{
   if (!r || !albedo) return seterr(MDH_E_INVALID, "bad argument");
   if (id0 < 0 || id0 >= MAX_MATERIALS) return seterr(MDH_E_INDEX, "material index out of range");
   uint8_t *p = r->materials_ubo + 16 + 32 * id0;
   memcpy(p, albedo, 12);
   memcpy(p + 12, &metallic, 4);
   memcpy(p + 16, &roughness, 4);
   if (id0 >= r->last_material_index) r->last_material_index = id0 + 1;
   return MDH_OK;
}
/* Add_Material, renderers.adb:369-377 */
int32_t orc_add_material(orc_renderer *r, const float albedo[3], float metallic, float roughness, int32_t *out_id0)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   int id = r->last_material_index;
   int rc = orc_set_material(r, id, albedo, metallic, roughness);
   if (rc == MDH_OK && out_id0) *out_id0 = id;
   return rc;
}
static int write_entity(orc_renderer *r, const kind_t *k, int index1, const void *blob, int nbytes)
{
   if (!blob || nbytes != k->elem_size) return seterr(MDH_E_INVALID, "blob size does not match the std140 element size");
   if (index1 < 1 || index1 > k->max_count) return seterr(MDH_E_INDEX, "index out of the declared range");
   memcpy(r->scene_ubo + k->array_off + k->stride * (index1 - 1), blob, (size_t)nbytes);
   return MDH_OK;
}
/* Set_Primitive, renderers.adb:379-398 */
int32_t orc_set_primitive(orc_renderer *r, int32_t kind_ix, int32_t index1, const void *blob, int32_t nbytes)
{
   if (!r || kind_ix < 0 || kind_ix >= r->npk) return seterr(MDH_E_INVALID, "bad kind index");
   if (index1 < 1 || index1 > r->host_count[kind_ix]) return seterr(MDH_E_INDEX, "index past the primitives added");
   return write_entity(r, &r->pk[kind_ix], index1, blob, nbytes);
}
/* Add_Primitive, renderers.adb:435-456 */
int32_t orc_add_primitive(orc_renderer *r, int32_t kind_ix, const void *blob, int32_t nbytes, int32_t *out_count)
{
   if (!r || kind_ix < 0 || kind_ix >= r->npk) return seterr(MDH_E_INVALID, "bad kind index");
   int count = r->host_count[kind_ix] + 1;
   int rc = write_entity(r, &r->pk[kind_ix], count, blob, nbytes);
   if (rc != MDH_OK) return rc;
   r->host_count[kind_ix] = count;
   int32_t c = count;
   memcpy(r->scene_ubo + r->pk[kind_ix].count_off, &c, 4);
   if (out_count) *out_count = count;
   return MDH_OK;
}
/* Set_Light, renderers.adb:458-483 */
int32_t orc_set_light(orc_renderer *r, int32_t index1, int32_t light_kind_ix, const void *blob, int32_t nbytes)
{
   if (!r || light_kind_ix < 0 || light_kind_ix >= r->nlk) return seterr(MDH_E_INVALID, "bad light kind index");
   int rc = write_entity(r, &r->lk[light_kind_ix], index1, blob, nbytes);
   if (rc != MDH_OK) return rc;
   int32_t c = index1;
   memcpy(r->scene_ubo + r->lk[light_kind_ix].count_off, &c, 4);
   memcpy(r->scene_ubo + r->total_light_off, &c, 4);
   return MDH_OK;
}
int32_t orc_set_camera_position(orc_renderer *r, const float p[3])
{
   if (!r || !p) return seterr(MDH_E_INVALID, "bad argument");
   memcpy(r->cam_pos, p, 12);
   return MDH_OK;
}
int32_t orc_set_camera_orientation(orc_renderer *r, const float m[9])
{
   if (!r || !m) return seterr(MDH_E_INVALID, "bad argument");
   memcpy(r->cam_m, m, 36);
   return MDH_OK;
}
/* Update_Partitioning, renderers.adb:757-775 */
int32_t orc_update_partitioning(orc_renderer *r, int32_t method)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (!r->part.enable) return MDH_OK;
   r->part_warnings = 0;
   orc_exprs_set_ada_div(1);
   if (method == 0) update_partitioning_cpu(r, 1);
   else if (method == 1) update_partitioning_cpu(r, 0);
   else if (method == 2) update_partitioning_gpu(r);
   else return seterr(MDH_E_INVALID, "bad method");
   return MDH_OK;
}
int32_t orc_render_pass(orc_renderer *r, int32_t pass)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   switch (pass) {
   case MDH_PASS_RADIANCE: pass_radiance(r); break;
   case MDH_PASS_IRRADIANCE: pass_irradiance(r); break;
   case MDH_PASS_VISIBILITY: pass_visibility(r); break;
   case MDH_PASS_SCATTERING: pass_scattering(r); break;
   case MDH_PASS_SCREEN: pass_screen(r); break;
   default: return seterr(MDH_E_INVALID, "bad pass");
   }
   return MDH_OK;
}
/* Render, renderers.adb:302-321 */
int32_t orc_render(orc_renderer *r)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   r->sdf_evals = 0;
   if (r->opt_mode == 0) {
      pass_radiance(r);
      pass_irradiance(r);
      if (r->vol.enabled) { pass_visibility(r); pass_scattering(r); }
   }
   pass_screen(r);
   return MDH_OK;
}
/* the three-step form of Render (include/madarch_hip.h: mdh_frame_begin / _probe_pass / _end);
 * here the steps simply run one after the other */
int32_t orc_frame_begin(orc_renderer *r)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   r->sdf_evals = 0;
   return MDH_OK;
}
int32_t orc_frame_probe_pass(orc_renderer *r, int32_t pass)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (pass != MDH_PASS_RADIANCE && pass != MDH_PASS_IRRADIANCE) return seterr(MDH_E_INVALID, "not a probe pass");
   if (r->opt_mode != 0) return MDH_OK;
   return orc_render_pass(r, pass);
}
int32_t orc_frame_end(orc_renderer *r)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (r->opt_mode == 0 && r->vol.enabled) { pass_visibility(r); pass_scattering(r); }
   pass_screen(r);
   return MDH_OK;
}
int32_t orc_finish(orc_renderer *r) { (void)r; return MDH_OK; }

int32_t orc_read_framebuffer(orc_renderer *r, float *rgb_out)
{
   if (!r || !rgb_out) return seterr(MDH_E_INVALID, "bad argument");
   memcpy(rgb_out, r->fb, (size_t)r->W * r->H * 3 * sizeof(float));
   return MDH_OK;
}
/* Swap_Buffers (renderers.adb:320) into the RGBA8 default framebuffer of the reference's window: clamp to
 * [0, 1], nearest of 256 levels (OpenGL 4.3 core 2.3.5.1; rintf = ties to even), NaN -> 0, alpha 255 */
int32_t orc_swap_buffers(orc_renderer *r)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   size_t px = (size_t)r->W * r->H;
   if (!r->front) r->front = (uint8_t *)malloc(px * 4);
   if (!r->front) return seterr(MDH_E_INVALID, "out of memory");
   for (size_t i = 0; i < px; ++i) {
      r->front[4 * i] = (uint8_t)unorm8_level(r->fb[3 * i]); /* the conversion of the RGB8 atlas store above */
      r->front[4 * i + 1] = (uint8_t)unorm8_level(r->fb[3 * i + 1]);
      r->front[4 * i + 2] = (uint8_t)unorm8_level(r->fb[3 * i + 2]);
      r->front[4 * i + 3] = 255;
   }
   r->swaps += 1;
   return MDH_OK;
}
int32_t orc_front_buffer(orc_renderer *r, const uint8_t **rgba, int64_t *swap_count)
{
   if (!r || !rgba) return seterr(MDH_E_INVALID, "bad argument");
   if (r->swaps == 0) return seterr(MDH_E_STATE, "no swap yet");
   *rgba = r->front;
   if (swap_count) *swap_count = r->swaps;
   return MDH_OK;
}
int32_t orc_read_gbuffer(orc_renderer *r, int32_t *index_out, float *t_out, int32_t *steps_out)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   size_t n = (size_t)r->W * r->H;
   if (index_out) memcpy(index_out, r->gb_index, n * 4);
   if (t_out) memcpy(t_out, r->gb_t, n * 4);
   if (steps_out) memcpy(steps_out, r->gb_steps, n * 4);
   return MDH_OK;
}
int32_t orc_read_texture(orc_renderer *r, int32_t tex, float *out, int32_t *w, int32_t *h, int32_t *c)
{
   if (r && tex > MDH_TEX_RADIANCE_MIP0 && tex <= MDH_TEX_RADIANCE_MIP0 + 15) { /* level tex - MDH_TEX_RADIANCE_MIP0 of the radiance atlas */
      if (!r->opt_mips || tex - MDH_TEX_RADIANCE_MIP0 > radiance_lods(r)) return seterr(MDH_E_INVALID, "no such level (MDH_OPT_RADIANCE_MIPS)");
      build_radiance_mips(r);
      const tex_t *m = &r->rad_mips[tex - MDH_TEX_RADIANCE_MIP0];
      if (w) *w = m->w;
      if (h) *h = m->h;
      if (c) *c = m->c;
      if (out) memcpy(out, m->data, (size_t)m->w * m->h * m->c * sizeof(float));
      return MDH_OK;
   }
   if (!r || tex < 0 || tex > 3) return seterr(MDH_E_INVALID, "bad argument");
   const tex_t *t = &r->tex[tex];
   if (w) *w = t->w;
   if (h) *h = t->h;
   if (c) *c = t->c;
   if (out) memcpy(out, t->data, (size_t)t->w * t->h * t->c * sizeof(float));
   return MDH_OK;
}
int32_t orc_write_texture(orc_renderer *r, int32_t tex, const float *in, int32_t w, int32_t h, int32_t c)
{
   if (!r || tex < 0 || tex > 3 || !in) return seterr(MDH_E_INVALID, "bad argument");
   tex_t *t = &r->tex[tex];
   if (w != t->w || h != t->h || c != t->c) return seterr(MDH_E_INVALID, "texture shape mismatch");
   for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) tex_store(t, x, y, in + ((size_t)y * w + x) * c);
   return MDH_OK;
}
/* probe-major slices [probe][res][res][3] of an atlas, for the exchange step of a sharded run */
static int atlas_slice(orc_renderer *r, int tex, int pb, int n, float *buf, int write)
{
   if (!r || (tex != MDH_TEX_RADIANCE && tex != MDH_TEX_IRRADIANCE) || !buf) return seterr(MDH_E_INVALID, "bad argument");
   int res = tex == MDH_TEX_RADIANCE ? r->probes.radiance_resolution : r->probes.irradiance_resolution;
   tex_t *t = &r->tex[tex];
   if (pb < 0 || n < 0 || pb + n > probe_total(r)) return seterr(MDH_E_INDEX, "probe range");
   for (int p = 0; p < n; ++p) {
      int id = pb + p, ty = id / r->probes.probe_count[0], tx = id - ty * r->probes.probe_count[0];
      for (int y = 0; y < res; ++y)
         for (int x = 0; x < res; ++x) {
            float *a = t->data + ((size_t)(ty * res + y) * t->w + (tx * res + x)) * 3;
            float *b = buf + (((size_t)p * res + y) * res + x) * 3;
            if (write) memcpy(a, b, 12); else memcpy(b, a, 12);
         }
   }
   return MDH_OK;
}
int32_t orc_read_atlas_slice(orc_renderer *r, int32_t tex, int32_t probe_begin, int32_t n_probes, float *out)
{
   return atlas_slice(r, tex, probe_begin, n_probes, out, 0);
}
int32_t orc_write_atlas_slice(orc_renderer *r, int32_t tex, int32_t probe_begin, int32_t n_probes, const float *in)
{
   return atlas_slice(r, tex, probe_begin, n_probes, (float *)in, 1);
}

/* Eval_Distance_To, renderers.adb:499-526 (batched) */
int32_t orc_eval_distance_to(orc_renderer *r, int32_t n, const float *pts, const int32_t *kind_ixs, int32_t n_kinds,
                             float *normals_out, float *dist_out)
{
   if (!r || !pts || !kind_ixs || !dist_out || n < 0) return seterr(MDH_E_INVALID, "bad argument");
   orc_exprs_set_ada_div(r->opt_ada_div);
   for (int q = 0; q < n; ++q) {
      v3 p = V3(pts[3 * q], pts[3 * q + 1], pts[3 * q + 2]);
      float closest = 1.0e10f;
      v3 normal = V3(0, 0, 0);
      for (int a = 0; a < n_kinds; ++a) {
         int k = kind_ixs[a];
         if (k < 0 || k >= r->npk) return seterr(MDH_E_INVALID, "bad kind index");
         for (int i = 0; i < r->host_count[k]; ++i) {
            if (r->pk[k].type == PK_CUSTOM) { /* the kind's own expressions, with Madarch.Values."/" when asked */
               float o[3];
               orc_xrun(r, &r->pk[k], 0, i, p, r->opt_ada_div, o);
               if (o[0] < closest) { closest = o[0]; orc_xrun(r, &r->pk[k], 1, i, p, r->opt_ada_div, o); normal = V3(o[0], o[1], o[2]); }
               continue;
            }
            entity e;
            make_entity(r, &r->pk[k], i, &e);
            float d = orc_eval_dist(r->pk[k].type, &e, p);
            if (d < closest) { closest = d; normal = orc_eval_normal(r->pk[k].type, &e, p); }
         }
      }
      dist_out[q] = closest;
      if (normals_out) { normals_out[3 * q] = normal.x; normals_out[3 * q + 1] = normal.y; normals_out[3 * q + 2] = normal.z; }
   }
   orc_exprs_set_ada_div(1);
   return MDH_OK;
}

int32_t orc_pass_time(orc_renderer *r, int32_t pass, double *ms, int64_t *launches)
{
   (void)r; (void)pass;
   if (ms) *ms = 0.0;
   if (launches) *launches = 0;
   return MDH_OK;
}
int32_t orc_reset_pass_times(orc_renderer *r) { (void)r; return MDH_OK; }

/* Scenes.Get_Primitives_Location / Get_Lights_Location, scenes.adb:1435-1462 */
int32_t orc_scene_layout(orc_renderer *r, int32_t is_light, int32_t kind_ix, int32_t *count_off, int32_t *array_off,
                         int32_t *stride, int32_t *elem_size)
{
   if (!r || kind_ix < 0 || kind_ix >= (is_light ? r->nlk : r->npk)) return seterr(MDH_E_INVALID, "bad kind index");
   const kind_t *k = is_light ? &r->lk[kind_ix] : &r->pk[kind_ix];
   if (count_off) *count_off = k->count_off;
   if (array_off) *array_off = k->array_off;
   if (stride) *stride = k->stride;
   if (elem_size) *elem_size = k->elem_size;
   return MDH_OK;
}
int32_t orc_scene_buffer_size(orc_renderer *r, int32_t *size, int32_t *total_light_off)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (size) *size = r->scene_ubo_size;
   if (total_light_off) *total_light_off = r->total_light_off;
   return MDH_OK;
}
int32_t orc_read_scene_buffer(orc_renderer *r, void *out, int32_t nbytes)
{
   if (!r || !out || nbytes > r->scene_ubo_size) return seterr(MDH_E_INVALID, "bad argument");
   memcpy(out, r->scene_ubo, (size_t)nbytes);
   return MDH_OK;
}
int32_t orc_read_partitioning(orc_renderer *r, int32_t *out, int32_t n_ints)
{
   if (!r || !out) return seterr(MDH_E_INVALID, "bad argument");
   if (!r->part.enable) return seterr(MDH_E_STATE, "partitioning disabled");
   int total = r->part_cells * (r->npk + r->part.index_count);
   if (n_ints != total) return seterr(MDH_E_INVALID, "size mismatch");
   memcpy(out, r->part_table, (size_t)total * 4);
   return MDH_OK;
}

/* oracle-only probes for the unit tests ------------------------------------ */
uint64_t orc_sdf_evals(orc_renderer *r) { return r ? r->sdf_evals : 0; }
/* rays, march steps and SDF evaluations of the last run of `pass` */
int32_t orc_work_counters(orc_renderer *r, int32_t pass, uint64_t out[3])
{
   if (!r || !out || pass < 0 || pass >= MDH_PASS_COUNT) return seterr(MDH_E_INVALID, "bad argument");
   memcpy(out, r->work[pass], sizeof r->work[pass]);
   return MDH_OK;
}
int32_t orc_partition_warnings(orc_renderer *r) { return r ? r->part_warnings : 0; }

/* closest_primitive_info / partitioning_closest_info at n points */
int32_t orc_probe_closest(orc_renderer *r, int32_t n, const float *pts, int32_t use_partitioning, float *dist, int32_t *index)
{
   for (int q = 0; q < n; ++q) {
      v3 p = V3(pts[3 * q], pts[3 * q + 1], pts[3 * q + 2]);
      int idx = -1;
      dist[q] = use_partitioning ? partitioning_closest_info(r, p, &idx) : closest_primitive_info(r, p, &idx);
      index[q] = idx;
   }
   return MDH_OK;
}
/* raycast at n rays: hit, index, t, steps */
int32_t orc_probe_raycast(orc_renderer *r, int32_t n, const float *org, const float *dir, int32_t *hit, int32_t *index, float *t, int32_t *steps)
{
   for (int q = 0; q < n; ++q) {
      int idx = -1, st = 0;
      float tt = 0.0f;
      v3 c;
      hit[q] = raycast(r, V3(org[3 * q], org[3 * q + 1], org[3 * q + 2]), V3(dir[3 * q], dir[3 * q + 1], dir[3 * q + 2]), &idx, &c, &tt, &st);
      index[q] = hit[q] ? idx : -1;
      t[q] = hit[q] ? tt : 0.0f;
      steps[q] = st;
   }
   return MDH_OK;
}
int32_t orc_probe_softshadow(orc_renderer *r, int32_t n, const float *org, const float *dir, const float *tmax, float k, float *out)
{
   for (int q = 0; q < n; ++q)
      out[q] = softshadows(r, V3(org[3 * q], org[3 * q + 1], org[3 * q + 2]), V3(dir[3 * q], dir[3 * q + 1], dir[3 * q + 2]), 0.0f, tmax[q], k);
   return MDH_OK;
}
/* sample_irradiance (render_probes.glsl:6-69) at n points with normals, from the current irradiance atlas */
int32_t orc_probe_sample_irradiance(orc_renderer *r, int32_t n, const float *pos, const float *nrm, float *out)
{
   for (int q = 0; q < n; ++q) {
      v3 c = sample_irradiance(r, V3(pos[3 * q], pos[3 * q + 1], pos[3 * q + 2]), V3(nrm[3 * q], nrm[3 * q + 1], nrm[3 * q + 2]));
      out[3 * q] = c.x; out[3 * q + 1] = c.y; out[3 * q + 2] = c.z;
   }
   return MDH_OK;
}
/* the three indirect-specular bodies of render_probes.glsl:71-244 (mode 1, 2 or 3) at n points: pos, normal, the
   REFLECTED direction and the shaded point's roughness (mode 1), from the current atlases */
int32_t orc_probe_specular(orc_renderer *r, int32_t mode, int32_t n, const float *pos, const float *nrm, const float *dir, const float *roughness, float *out)
{
   if (r->opt_mips) build_radiance_mips(r);
   for (int q = 0; q < n; ++q) {
      v3 P = V3(pos[3 * q], pos[3 * q + 1], pos[3 * q + 2]), N = V3(nrm[3 * q], nrm[3 * q + 1], nrm[3 * q + 2]), D = V3(dir[3 * q], dir[3 * q + 1], dir[3 * q + 2]);
      v3 c = mode == 1 ? sample_radiance_with_specular(r, P, N, D, roughness[q]) : (mode == 2 ? sample_radiance_no_specular(r, P, N, D) : compute_indirect_specular(r, P, N, D, 1));
      out[3 * q] = c.x; out[3 * q + 1] = c.y; out[3 * q + 2] = c.z;
   }
   return MDH_OK;
}
/* render_volumetrics (volumetrics.glsl:34-54) on n pixels: surface colour, ray origin, hit position (hit != 0), fragment */
int32_t orc_probe_render_volumetrics(orc_renderer *r, int32_t n, const float *L, const float *from, const float *to, const int32_t *hit, const float *frag, float *out)
{
   for (int q = 0; q < n; ++q) {
      v3 c = render_volumetrics(r, V3(L[3 * q], L[3 * q + 1], L[3 * q + 2]), V3(from[3 * q], from[3 * q + 1], from[3 * q + 2]), V3(to[3 * q], to[3 * q + 1], to[3 * q + 2]), hit[q],
                                V2(frag[2 * q], frag[2 * q + 1]));
      out[3 * q] = c.x; out[3 * q + 1] = c.y; out[3 * q + 2] = c.z;
   }
   return MDH_OK;
}
/* single SDF / normal of a built-in kind from raw parameters (a: vec3, b: vec3 or scalar in b[0], c: vec3) */
float orc_sdf(int32_t type, const float *a, const float *b, const float *c, const float *p)
{
   v3 A = V3(a[0], a[1], a[2]), P = V3(p[0], p[1], p[2]);
   switch (type) {
   case PK_SPHERE: return sd_sphere(A, b[0], P);
   case PK_PLANE: return sd_plane(A, b[0], P);
   case PK_BOX: return sd_box(A, V3(b[0], b[1], b[2]), P);
   default: return sd_triangle(A, V3(b[0], b[1], b[2]), V3(c[0], c[1], c[2]), P, 0);
   }
}
void orc_sdf_normal(int32_t type, const float *a, const float *b, const float *c, const float *p, float *out)
{
   v3 A = V3(a[0], a[1], a[2]), P = V3(p[0], p[1], p[2]), n;
   switch (type) {
   case PK_SPHERE: n = normalize(sub(P, A)); break;
   case PK_PLANE: n = A; break;
   case PK_BOX: n = nrm_box(A, V3(b[0], b[1], b[2]), P); break;
   default: n = nrm_triangle(A, V3(b[0], b[1], b[2]), V3(c[0], c[1], c[2]), P, 0); break;
   }
   out[0] = n.x; out[1] = n.y; out[2] = n.z;
}
void orc_oct_encode(const float *v, float *out) { v2 r = ray_dir_to_ray_id(V3(v[0], v[1], v[2])); out[0] = r.x; out[1] = r.y; }
void orc_oct_decode(const float *id, float *out) { v3 d = ray_id_to_ray_dir(V2(id[0], id[1])); out[0] = d.x; out[1] = d.y; out[2] = d.z; }
void orc_cook_torrance(const float *N, const float *V, const float *L, const float *albedo, float metallic, float roughness, float *kD, float *kS)
{
   v3 n = V3(N[0], N[1], N[2]), v = V3(V[0], V[1], V[2]), l = V3(L[0], L[1], L[2]), d, s;
   cook_torrance(n, v, l, fmax_(dot(n, l), 0.0f), V3(albedo[0], albedo[1], albedo[2]), metallic, roughness, &d, &s);
   kD[0] = d.x; kD[1] = d.y; kD[2] = d.z; kS[0] = s.x; kS[1] = s.y; kS[2] = s.z;
}
/* sample_light at a point: radiance, dir, dist */
void orc_probe_light(orc_renderer *r, int32_t index, const float *pos, float *radiance, float *dir, float *dist)
{
   v3 d;
   v3 c = sample_light(r, index, V3(pos[0], pos[1], pos[2]), V3(0, 1, 0), &d, dist);
   radiance[0] = c.x; radiance[1] = c.y; radiance[2] = c.z; dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}
/* tree evaluator on raw parameters: dist and normal of one built-in primitive */
float orc_exprs_sdf(int32_t type, const float *a, const float *b, const float *c, const float *p, int32_t ada_div, float *normal_out)
{
   entity e;
   memset(&e, 0, sizeof e);
   static const char *names[4][3] = {{"center", "radius", 0}, {"normal", "offset", 0}, {"center", "side", 0}, {"v1", "v2", "v3"}};
   e.n = type == PK_TRIANGLE ? 3 : 2;
   e.names[0] = names[type][0]; e.vals[0].kind = VK_VEC3; e.vals[0].v = V3(a[0], a[1], a[2]);
   e.names[1] = names[type][1];
   if (type == PK_SPHERE || type == PK_PLANE) { e.vals[1].kind = VK_FLOAT; e.vals[1].f = b[0]; }
   else { e.vals[1].kind = VK_VEC3; e.vals[1].v = V3(b[0], b[1], b[2]); }
   if (type == PK_TRIANGLE) { e.names[2] = names[type][2]; e.vals[2].kind = VK_VEC3; e.vals[2].v = V3(c[0], c[1], c[2]); }
   orc_exprs_set_ada_div(ada_div);
   v3 P = V3(p[0], p[1], p[2]);
   float d = orc_eval_dist(type, &e, P);
   if (normal_out) { v3 n = orc_eval_normal(type, &e, P); normal_out[0] = n.x; normal_out[1] = n.y; normal_out[2] = n.z; }
   orc_exprs_set_ada_div(1);
   return d;
}
/* the fp32 transcendental algorithms of orc_math.h, for the accuracy tests */
float orc_acos(float x) { return acos_(x); }
float orc_exp(float x) { return exp_(x); }
float orc_pow(float x, float y) { return pow_(x, y); }
float orc_sin(float x) { return sin_(x); }
float orc_cos(float x) { return cos_(x); }
float orc_tan(float x) { return tan_(x); }
float orc_asin(float x) { return asin_(x); }
float orc_atan(float x) { return atan_(x); }
