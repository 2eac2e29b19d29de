// mdh_device.h -- hand-written gfx950 device code of the Madarch render path.
//
// What the reference expresses as GLSL (static shaders under madarch/glsl/ plus
// the scene code Madarch.Scenes generates, madarch-scenes.adb:1189-1266) is
// written here directly for CDNA4: one lane per ray, 64 rays per wavefront,
// scene tables (primitives, lights, materials) staged once per workgroup into
// LDS and read with wave-uniform addresses (broadcast reads, no bank conflicts),
// the uniform scene header in SGPRs through the kernel argument block.  The
// sphere-tracing loops have no step cap, as in the reference: every iteration
// advances by >= epsilon, so a ray ends within max_dist / epsilon iterations and
// a wave leaves a loop as soon as the ballot of live lanes is empty.
//
// Arithmetic contract (DESIGN.md "Numerics"): fp32, one IEEE operation per source
// operation in the order written (built with -ffp-contract=off, correctly
// rounded division and square root), min/max = IEEE minNum/maxNum.  Reference
// file:line citations are relative to /root/reference/madarch/.
#pragma once

#ifndef MDH_JIT // (hiprtc brings the HIP device builtins itself and has no system headers; nothing below needs more)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#define MDH_DEV static __device__ __forceinline__
// Triangle code is kept out of line: it is cold in every benchmark scene and register hungry
#ifndef MDH_TRI
#define MDH_TRI static __device__ __noinline__
#endif

#define MDH_MAX_KINDS 8
#define MDH_MAX_LIGHT_KINDS 4

enum { PK_SPHERE = 0, PK_PLANE = 1, PK_BOX = 2, PK_TRIANGLE = 3, PK_CUSTOM = 4 }; // PK_CUSTOM: a user-defined kind (MDH_X programs)
enum { LK_POINT = 0, LK_SPOT = 1, LK_CUSTOM = 2 }; // LK_CUSTOM: a user-defined light kind (MDH_X Sample and Position programs)

// ---------------------------------------------------------------- kernel argument blocks
// Uniform scene header.  The part the march loops need on every step lives in SGPRs (kernel
// arguments); the per-kind bookkeeping that only the cold paths read (arg-min index, normals,
// light dispatch) sits in an int block at the head of the LDS table -- SGPRs are the scarce
// resource of these kernels (they ran at the 102-SGPR limit with everything passed by value).
struct KScene {
   // the primitives by TYPE, for the order-free min of closest_primitive
   int tcount[4];    // runtime count of Sphere / Plane / Box / Triangle
   int tslot[4];     // first float4 of that type's geometry in the table
   // Planes whose normal is exactly +-e_x, +-e_y or +-e_z (the walls of a room): for finite p,
   // dot(n, p) + o is bit-identical to (+-p_axis) + o, and since rounding is monotonic the min over
   // all planes of one direction is (+-p_axis) + min(o).  axis_off = that min offset for
   // +x, -x, +y, -y, +z, -z (+inf where there is none); gplane_* = the remaining general planes.
   float axis_off[6];
   int n_axis;       // number of planes folded into axis_off (0: skip the block)
   int gplane_count, gplane_slot;
   float max_dist;
   int total_lights; // total_light_count (scenes.adb:594)
   int mat_slot;     // first float4 of the materials (2 per material)
   int u8_slot;      // float4 index of the 256-entry k / 255 table
   int table_f4;     // float4 count of the whole table
   // space partition (scenes.adb:799-1118)
   int part_enable, part_border, part_index_count, part_cells;
   int part_dims[3];
   float part_sp[3], part_off[3];
   int part_sp_pow2; // all three spacings are powers of two: part_inv_sp holds their exact reciprocals (x / 2^k == x * 2^-k)
   float part_inv_sp[3];
   const float4 *table;   // HBM image of the table (staged to LDS by every workgroup)
   const int *part_table; // [cell][nk + index_count], then the same lists as bits: [cell][part_mask_words] at int part_mask_off
   int part_mask_off, part_mask_words; // (partitioning_closest_bits below)
   int part_bits_f4; // float4 count of the bits when every workgroup stages them into LDS behind the scene table (0: read from memory)
   // Small scenes (no user-defined kinds, at most 64 declared primitives, at most 32 of a kind -- every scene of the
   // reference's examples): a cell's bits are ONE 8-byte load and every built-in TYPE's candidates one shift and mask of
   // it -- part_tbit = the first bit of the type's kind, part_tmask = its declared count as a mask (0: no such kind) --
   // so the lookup is straight-line code without the table's kind headers (partitioning_closest_bits).
   int part_small;
   unsigned part_tbit[4], part_tmask[4];
   float part_fdims[3], part_fyz; // (float)part_dims[a] and (float)(part_dims[1] * part_dims[2]): the same conversions, once on the host
   // MDH_SDF_SGPR: the first sphere and the first box of the table once more, as kernel arguments -- scalar operands of the
   // brute-force scan (closest_primitive) instead of three LDS broadcast reads into twelve vector registers per evaluation
   float first_sphere[4], first_box[8];
};
// int block at table[0..]: per-kind data in SCENE order (the order the flat primitive index
// and the arg-min tie-break follow, scenes.adb:656-666) and the light kinds
enum {
   H_NK = 0,
   H_NL = 1,
   H_KTYPE = 8,    // [8] PK_*
   H_KCOUNT = 16,  // [8] runtime count (prim_<K>_count, scenes.adb:560-565)
   H_KBASE = 24,   // [8] flat index base = sum of earlier DECLARED counts
   H_KMAX = 32,    // [8] declared count
   H_KSLOT = 40,   // [8] first float4 of the kind's geometry
   H_KMAT = 48,    // [8] first int of the kind's material ids (built-in kinds)
   H_KSTRIDE = 56, // [8] float4 per instance
   // user-defined kinds: first int and length of the MDH_X programs (include/madarch_hip.h)
   H_XDIST = 64, H_XDISTN = 72, H_XNRM = 80, H_XNRMN = 88, H_XMAT = 96, H_XMATN = 104,
   H_LTYPE = 112,  // [4]
   H_LCOUNT = 116, // [4]
   H_LSLOT = 120,  // [4]
   H_LSTRIDE = 124, // [4] float4 per instance
   // user-defined light kinds: Sample and Position programs
   H_XLSAMPLE = 128, H_XLSAMPLEN = 132, H_XLPOS = 136, H_XLPOSN = 140,
   // the typed arg-min (closest_primitive_info): flat index of the plane behind each folded axis offset (-1: none),
   // flat index base of each built-in TYPE, and whether the scene allows it
   H_AXIS_IDX = 144, H_TBASE = 150, H_FASTINFO = 154,
   // per kind {type, first float4, flat index base, declared count}: what the space partition's lookup needs of a kind,
   // in one 16-byte LDS read
   H_KQUAD = 156,  // [8][4]
   H_INTS = 188    // 47 float4
};

struct KProbes {
   int pcx, pcy;      // probe_count
   int gx, gy, gz;    // grid_dimensions
   float sx, sy, sz;  // grid_spacing
   int rres, ires;    // radiance / irradiance resolution
   int fmt;           // 0 = RGBA8 unorm texels, 1 = float4 texels
   int rshift, ishift, pcx_shift; // log2 of rres / ires / pcx when a power of two, else -1
   float inv_pcx, inv_pcy;        // 1 / pcx, 1 / pcy when a power of two (x / 2^k == x * 2^-k exactly), else 0
   void *rad;         // probe-major [probe][y][x]
   void *irr;
   int probe_begin, probe_end; // slice this rank updates
   // Wave-uniform fp32 values the pixel program needs at every atlas tap, computed once on the host with the same
   // IEEE operations (one correctly rounded division / subtraction / conversion each): as kernel arguments they are
   // scalar operands; computed in the kernel they are vector instructions on uniform data that the compiler hoists
   // into VGPRs which then stay live (or spilled to scratch) through the whole pixel program.
   float irr_lo, irr_hi; // 0.5f / ires, 1.0f - irr_lo: the clamp of an irradiance tap inside its tile (render_probes.glsl:53-57)
   float rad_lo, rad_hi; // 0.5f / rres, 1.0f - rad_lo (render_probes.glsl:190-194)
   float irr_w, irr_h, rad_w, rad_h; // (float)(pcx * ires), (float)(pcy * ires), (float)(pcx * rres), (float)(pcy * rres)
   float fpcx, fpcy;     // (float)pcx, (float)pcy
   int rad_lods;         // radiance_lods = int(log2(radiance_resolution)) (probe_utils.glsl:17): the highest set bit
   unsigned m_rres, m_ires, m_pcx; // div_magic's numbers for rres, ires and pcx (0: the atlas is too large for them)
   void *rad_mips;       // MDH_OPT_RADIANCE_MIPS: levels 1 .. rad_lods of `rad`, one behind the other (null: level 0 only)
};

struct KCamera {
   float px, py, pz;
   float m[9]; // column-major
};

struct KVolumetrics {
   int enabled;
   int vw, vh, vz; // visibility texture is vw x (vh * vz)
   int sw, sh;
   float vstep, sstep;
   float *vis;   // 3 floats per texel
   float4 *scat; // rgb + ray length
};

// The lane's index in its wavefront taken from the hardware at the point of use (two instructions, no inputs).  A
// volatile statement is neither hoisted nor merged with an earlier one: what is derived from it (the pixel of the
// screen pass's epilogue, the thread's LDS park row) is recomputed where it is needed instead of being kept in a
// VGPR -- or in scratch -- across the whole pixel program.
MDH_DEV int lane_index_fresh()
{
   int l;
   asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
   return l;
}

// -DMDH_DIAG: per loop type, count SDF evaluations (wave level) and the lanes alive in them
#ifdef MDH_DIAG
__device__ unsigned long long g_diag[16];

#define MDH_DIAG_STEP(type)                                                                 \
   do {                                                                                     \
      unsigned long long m_ = __ballot(1);                                                  \
      if ((threadIdx.x & 63) == __ffsll((long long)m_) - 1) {                               \
         atomicAdd(&g_diag[2 * (type)], 1ull);                                              \
         atomicAdd(&g_diag[2 * (type) + 1], (unsigned long long)__popcll(m_));              \
      }                                                                                     \
   } while (0)
// the work of a launch as the oracle counts it (orc_work_counters) -- slot 0 rays started,
// 1 SDF evaluations inside march loops (a step that reuses the shared first evaluation counts: it is a step of the ray),
// 2 SDF evaluations in all, 3 of them the arg-min evaluations at hit points (the kernels march without the arg-min and
// evaluate it once at the hit; the reference's raycast carries it through every step) -- lanes, not wavefronts.
__device__ unsigned long long g_work[4];
#define MDH_WORK(slot)                                                                      \
   do {                                                                                     \
      unsigned long long m_ = __ballot(1);                                                  \
      if ((threadIdx.x & 63) == __ffsll((long long)m_) - 1) atomicAdd(&g_work[slot], (unsigned long long)__popcll(m_)); \
   } while (0)
#else
#define MDH_DIAG_STEP(type) do { } while (0)
#define MDH_WORK(slot) do { } while (0)
#endif
// ------------------------------------------------------------------------------ vec math
struct f3 { float x, y, z; };
struct f2 { float x, y; };
struct i3 { int x, y, z; };

#define MDH_PI 3.14159265358f // maths.glsl:1
#define MDH_EPS 0.001f        // maths.glsl:3
#define MDH_MIN_STEP 0.05f    // raymarching.glsl:1

MDH_DEV f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
MDH_DEV f3 F3s(float s) { return F3(s, s, s); }
MDH_DEV f2 F2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
MDH_DEV f3 xyz(float4 v) { return F3(v.x, v.y, v.z); }

// min/max = IEEE minNum/maxNum.  __builtin_fminf costs a canonicalising v_max_f32 per operand in
// IEEE mode; issuing the instruction directly (MDH_ASM_MINMAX=1) removes it but measured 3 % SLOWER
// on MI355X (the opaque asm defeats the scheduler), so the builtin stays the default.
#ifndef MDH_ASM_MINMAX
#define MDH_ASM_MINMAX 0
#endif
#if MDH_ASM_MINMAX
MDH_DEV float min_(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
MDH_DEV float max_(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#else
MDH_DEV float min_(float a, float b) { return __builtin_fminf(a, b); } // v_min_f32: minNum
MDH_DEV float max_(float a, float b) { return __builtin_fmaxf(a, b); }
#endif
// The same two operations as bare instructions, for operands that reach them across a branch: there the compiler
// cannot see that a value is no signaling NaN (none exists on this path: every operand is an arithmetic
// result) and quiets it with a v_max_f32 x, x first -- six or seven of the ~75 instructions of one SDF evaluation.
#ifndef MDH_RAW_MINMAX
#define MDH_RAW_MINMAX 1
#endif
#if MDH_RAW_MINMAX
MDH_DEV float min_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
MDH_DEV float max0_raw(float a) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a)); return r; }
MDH_DEV float min0_raw(float a) { float r; asm("v_min_f32 %0, 0, %1" : "=v"(r) : "v"(a)); return r; }
#else
MDH_DEV float min_raw(float a, float b) { return min_(a, b); }
MDH_DEV float max0_raw(float a) { return max_(a, 0.0f); }
MDH_DEV float min0_raw(float a) { return min_(a, 0.0f); }
#endif

// Correctly rounded sqrt.  hipcc's expansion (v_sqrt_f32 + two fma residual tests) also rescales
// inputs below 2^-96 and re-selects 0/inf with a class test on every call; MDH_FAST_EXACT_SQRT=1
// sends those inputs down a rare branch and keeps the 9-instruction core (same result for every
// input).  Measured 5 % SLOWER on MI355X than the compiler's branch-free form: off by default.
#ifndef MDH_FAST_EXACT_SQRT
#define MDH_FAST_EXACT_SQRT 0
#endif
// The core of that expansion by itself: v_sqrt_f32, its two neighbours, two fma residuals, two selects -- the
// compiler's own nine instructions, hence its result, for every input that it does not rescale: 0, anything from
// 2^-96 up, infinities and NaN (for 0 both residual tests fail and the 0 of v_sqrt_f32 stands).  For callers that
// KNOW their operand is no positive number below 2^-96.
// MDH_FAST_NUMERICS: the LABELLED EXPERIMENT build (`make -C madarch_amd/csrc fast`, never the shipped library): what the
// path would cost under BASELINE.json's tolerance (1e-4 relative per channel) instead of this build's own contract (the
// oracle's bits) -- the hardware's v_sqrt_f32 / v_rcp_f32 / v_log_f32 / v_exp_f32 as GLSL on any GPU uses them, fused
// multiply-adds, the irradiance fold in four partial sums.  scripts/numerics_experiment.py measures the rate and counts
// the pixels that leave the tolerance.
#ifndef MDH_FAST_NUMERICS
#define MDH_FAST_NUMERICS 0
#endif
MDH_DEV float sqrt_unscaled_(float x)
{
#if MDH_FAST_NUMERICS
   return __builtin_amdgcn_sqrtf(x);
#endif
   float s = __builtin_amdgcn_sqrtf(x);
   const int si = __float_as_int(s);
   const float sd = __int_as_float(si - 1), su = __int_as_float(si + 1);
   const float vp = __builtin_fmaf(-sd, s, x), vs = __builtin_fmaf(-su, s, x);
   s = (vp <= 0.0f) ? sd : s;
   s = (vs > 0.0f) ? su : s;
   return s;
}
#ifndef MDH_SQRT_WAVE_ALL
#define MDH_SQRT_WAVE_ALL 1 // every correctly rounded root through the nine-instruction core behind a wave-wide guard (sqrt_wave_ below): the same bits; config 3 serial +1.7 %, in flight +-0
#endif
MDH_DEV float sqrt_(float x)
{
#if MDH_FAST_NUMERICS
   return __builtin_amdgcn_sqrtf(x);
#endif
#if MDH_SQRT_WAVE_ALL
   if (__ballot(x < 0x1p-96f && x > 0.0f) != 0ull) return __builtin_sqrtf(x);
   return sqrt_unscaled_(x);
#endif
#if MDH_FAST_EXACT_SQRT
   if (__builtin_expect(x < 0x1p-96f && x > 0.0f, 0)) return __builtin_sqrtf(x);
   return sqrt_unscaled_(x);
#else
   return __builtin_sqrtf(x);
#endif
}
#ifndef MDH_PART_SQRT_WAVE
#define MDH_PART_SQRT_WAVE 1
#endif
// the correctly rounded root of a wavefront's operands through the nine-instruction core when none of them is a positive
// number below 2^-96 (the only inputs hipcc's expansion rescales: sqrt_unscaled_ above), through the full expansion otherwise
MDH_DEV float sqrt_wave_(float x)
{
#if MDH_FAST_NUMERICS
   return __builtin_amdgcn_sqrtf(x);
#endif
#if MDH_PART_SQRT_WAVE
   if (__ballot(x < 0x1p-96f && x > 0.0f) != 0ull) return sqrt_(x);
   return sqrt_unscaled_(x);
#else
   return sqrt_(x);
#endif
}
// Two correctly rounded roots at once (the rooms' scan: its one sphere and its one box), the two nine-instruction cores
// interleaved: a core's comparisons each wait two slots for their select (a vector comparison's mask is not forwarded), and
// a core alone has nothing to put there -- the other core's instructions go there, and one wave-wide guard serves both.
// The same nine operations per operand: the same bits.
MDH_DEV void sqrt2_(float a, float b, float &ra, float &rb)
{
#if MDH_FAST_NUMERICS
   ra = __builtin_amdgcn_sqrtf(a); rb = __builtin_amdgcn_sqrtf(b);
   return;
#endif
   if (__ballot(min_(a, b) < 0x1p-96f) != 0ull) { ra = __builtin_sqrtf(a); rb = __builtin_sqrtf(b); return; } // (a zero goes the slow way too: never in a march)
   float sa = __builtin_amdgcn_sqrtf(a), sb = __builtin_amdgcn_sqrtf(b);
   const int ia = __float_as_int(sa), ib = __float_as_int(sb);
   const float sda = __int_as_float(ia - 1), sdb = __int_as_float(ib - 1), sua = __int_as_float(ia + 1), sub = __int_as_float(ib + 1);
   const float vpa = __builtin_fmaf(-sda, sa, a), vpb = __builtin_fmaf(-sdb, sb, b), vsa = __builtin_fmaf(-sua, sa, a), vsb = __builtin_fmaf(-sub, sb, b);
   const bool ca = vpa <= 0.0f, cb = vpb <= 0.0f, ea = vsa > 0.0f, eb = vsb > 0.0f;
   sa = ca ? sda : sa; sb = cb ? sdb : sb;
   sa = ea ? sua : sa; sb = eb ? sub : sb;
   ra = sa; rb = sb;
}
// MDH_HYBRID_NUMERICS: the second LABELLED EXPERIMENT (`make -C madarch_amd/csrc hybrid`; never the shipped
// library): every operation INSIDE a march loop and the whole primary ray (geometry buffer: index, t, steps) stay as exact
// as in the shipped build; what SHADES a point behind a hit -- normals, the BRDF, light attenuation, probe directions and
// weights, the square roots of the irradiance taps, occlusion's quotient, fog, the tone map -- uses the hardware's
// reciprocal / square root / exp2 / log2 and fused multiply-adds.  In the shipped build the s* helpers below ARE the exact
// expressions they replace.
#ifndef MDH_HYBRID_NUMERICS
#define MDH_HYBRID_NUMERICS 0
#endif
#if MDH_HYBRID_NUMERICS
#define MDH_SHADING_FP _Pragma("clang fp contract(fast)")
MDH_DEV float sdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
MDH_DEV float ssqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
#define MDH_SHADING_FP
MDH_DEV float sdiv(float a, float b) { return a / b; }
MDH_DEV float ssqrt(float x) { return sqrt_(x); }
#endif
MDH_DEV float clamp_(float x, float lo, float hi) { return min_(max_(x, lo), hi); }
MDH_DEV int iclamp_(int x, int lo, int hi) { return min(max(x, lo), hi); }
// x / d for 0 <= x < 65536 through the host's magic number m = floor (2^32 / d) + 1: mulhi (x, m) is the exact quotient
// while x * d < 2^32 (the error term x * (m d - 2^32) <= x d stays below 2^32).  m = 0: no magic, divide.
MDH_DEV int div_magic(int x, int d, unsigned m)
{
   if (m) return (int)__umulhi((unsigned)x, m);
   asm volatile("" : "+v"(d)); // (the cold path keeps its division set-up to itself: see mirror)
   return x / d;
}
MDH_DEV float sign_(float x) { return x < 0.0f ? -1.0f : (x > 0.0f ? 1.0f : 0.0f); }
MDH_DEV float fract_(float x) { return x - __builtin_floorf(x); }
MDH_DEV float mix_(float x, float y, float a) { return x * (1.0f - a) + y * a; }

MDH_DEV f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
MDH_DEV f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
MDH_DEV f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
MDH_DEV f3 operator/(f3 a, f3 b) { return F3(a.x / b.x, a.y / b.y, a.z / b.z); }
MDH_DEV f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
MDH_DEV f3 operator/(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
MDH_DEV f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }
MDH_DEV f3 abs3(f3 a) { return F3(__builtin_fabsf(a.x), __builtin_fabsf(a.y), __builtin_fabsf(a.z)); }
MDH_DEV f3 max3s(f3 a, float s) { return F3(max_(a.x, s), max_(a.y, s), max_(a.z, s)); }
MDH_DEV f3 min3s(f3 a, float s) { return F3(min_(a.x, s), min_(a.y, s), min_(a.z, s)); }
MDH_DEV f3 floor3(f3 a) { return F3(__builtin_floorf(a.x), __builtin_floorf(a.y), __builtin_floorf(a.z)); }
MDH_DEV f3 sqrt3(f3 a) { return F3(sqrt_(a.x), sqrt_(a.y), sqrt_(a.z)); }
// dot = (x*x' + y*y') + z*z'
MDH_DEV float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
MDH_DEV float dot2(f3 a) { return dot(a, a); } // maths.glsl:5-7
MDH_DEV float length(f3 a) { return sqrt_(dot2(a)); }
MDH_DEV f3 normalize(f3 a) { return a / length(a); } // support/math_utils.ads:81-83
#if MDH_HYBRID_NUMERICS
MDH_DEV f3 sdiv3(f3 a, float s) { const float r = __builtin_amdgcn_rcpf(s); return F3(a.x * r, a.y * r, a.z * r); }
MDH_DEV f3 snormalize(f3 a) { const float r = __builtin_amdgcn_rsqf(dot2(a)); return F3(a.x * r, a.y * r, a.z * r); }
#else
MDH_DEV f3 sdiv3(f3 a, float s) { return a / s; }
MDH_DEV f3 snormalize(f3 a) { return normalize(a); }
#endif
MDH_DEV f3 ssqrt3(f3 a) { return F3(ssqrt(a.x), ssqrt(a.y), ssqrt(a.z)); }
MDH_DEV f3 cross(f3 a, f3 b) { return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
MDH_DEV f3 reflect(f3 i, f3 n) { return i - n * (2.0f * dot(n, i)); }

// Transcendentals: explicit fp32 algorithms, operation for operation what the oracle fixes in
// oracle/orc_math.h (no fp64, no v_exp/v_log: those have no bit-identical CPU counterpart).
// acos: Abramowitz & Stegun 4.4.46
MDH_DEV float acos_(float x)
{
   float ax = __builtin_fabsf(x);
   float p = -0.0012624911f;
   p = p * ax + 0.0066700901f;
   p = p * ax + -0.0170881256f;
   p = p * ax + 0.0308918810f;
   p = p * ax + -0.0501743046f;
   p = p * ax + 0.0889789874f;
   p = p * ax + -0.2145988016f;
   p = p * ax + 1.5707963050f;
   float r = sqrt_(1.0f - ax) * p;
   return x < 0.0f ? 3.14159265358979f - r : r;
}
MDH_DEV float asin_(float x) { return 1.5707963267948966f - acos_(x); }
// sin / cos / tan / atan: the operation sequences of oracle/orc_math.h (Cody-Waite reduction + Cephes polynomials)
MDH_DEV void sincos_core_(float x, float &s, float &c, int &q)
{
   float kf = __builtin_rintf(x * 0.636619772367581f);
   if (!(__builtin_fabsf(kf) < 1.0e9f)) kf = 0.0f;
   float r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
   float z = r * r;
   float ps = -1.9515295891e-4f;
   ps = ps * z + 8.3321608736e-3f;
   ps = ps * z + -1.6666654611e-1f;
   s = (ps * z) * r + r;
   float pc = 2.443315711809948e-5f;
   pc = pc * z + -1.388731625493765e-3f;
   pc = pc * z + 4.166664568298827e-2f;
   c = ((pc * z) * z - 0.5f * z) + 1.0f;
   q = (int)kf & 3;
}
MDH_DEV float sin_(float x)
{
   float s, c; int q;
   sincos_core_(x, s, c, q);
   float r = (q & 1) ? c : s;
   return (q & 2) ? -r : r;
}
MDH_DEV float cos_(float x)
{
   float s, c; int q;
   sincos_core_(x, s, c, q);
   float r = (q & 1) ? s : c;
   return ((q + 1) & 2) ? -r : r;
}
MDH_DEV float tan_(float x) { return sin_(x) / cos_(x); }
MDH_DEV float atan_(float x)
{
   float ax = __builtin_fabsf(x), y = 0.0f, t = ax;
   if (ax > 2.414213562373095f) { y = 1.5707963267948966f; t = -(1.0f / ax); }
   else if (ax > 0.4142135623730950f) { y = 0.7853981633974483f; t = (ax - 1.0f) / (ax + 1.0f); }
   float z = t * t;
   float p = 8.05374449538e-2f;
   p = p * z + -1.38776856032e-1f;
   p = p * z + 1.99777106478e-1f;
   p = p * z + -3.33329491539e-1f;
   y = y + ((p * z) * t + t);
   return x < 0.0f ? -y : y;
}
MDH_DEV float exp2_(float z)
{
   if (z != z) return z;
   if (z > 128.0f) return __builtin_inff();
   if (z < -126.0f) return 0.0f;
   float n = __builtin_rintf(z);
   float u = (z - n) * 0.693147182464599609375f;
   float p = 1.0f / 5040.0f;
   p = p * u + 1.0f / 720.0f;
   p = p * u + 1.0f / 120.0f;
   p = p * u + 1.0f / 24.0f;
   p = p * u + 1.0f / 6.0f;
   p = p * u + 0.5f;
   p = p * u + 1.0f;
   p = p * u + 1.0f;
   return p * __int_as_float(((int)n + 127) << 23);
}
MDH_DEV float log2_(float x)
{
   int e = 0;
   if (x < 1.17549435e-38f) { x = x * 16777216.0f; e = -24; }
   unsigned b = (unsigned)__float_as_int(x);
   e += (int)((b >> 23) & 255u) - 127;
   float m = __int_as_float((int)((b & 0x007fffffu) | 0x3f800000u));
   if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
   float s = (m - 1.0f) / (m + 1.0f);
   float s2 = s * s;
   float p = 1.0f / 9.0f;
   p = p * s2 + 1.0f / 7.0f;
   p = p * s2 + 1.0f / 5.0f;
   p = p * s2 + 1.0f / 3.0f;
   p = p * s2 + 1.0f;
   return (float)e + ((2.0f * s) * p) * 1.44269502162933349609375f;
}
#if MDH_FAST_NUMERICS
MDH_DEV float exp_(float x) { return __builtin_amdgcn_exp2f(x * 1.44269502162933349609375f); }
#else
MDH_DEV float exp_(float x) { return exp2_(x * 1.44269502162933349609375f); }
#endif
MDH_DEV float pow_(float x, float y)
{
#if MDH_FAST_NUMERICS
   return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); // (v_log_f32 is log2; 0 -> -inf -> 0, negative -> NaN)
#endif
   if (x != x || x < 0.0f) return __builtin_nanf("");
   if (x == 0.0f) return 0.0f;
   if (x > 3.40282347e+38f) return x;
   return exp2_(y * log2_(x));
}
#if MDH_HYBRID_NUMERICS
MDH_DEV float sexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269502162933349609375f); }
MDH_DEV float spow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
#else
MDH_DEV float sexp(float x) { return exp_(x); }
MDH_DEV float spow(float x, float y) { return pow_(x, y); }
#endif
MDH_DEV float pow5_(float x) { float x2 = x * x; return (x2 * x2) * x; }                 // cook_torrance_brdf.glsl:2
MDH_DEV float pow8_(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x4; } // spot_lights.adb:18
MDH_DEV float pow1_5_(float x) { return x * sqrt_(x); }                        // volumetrics.glsl:25-28

// ---------------------------------------------------------------------- LDS scene table
// One float4 array per workgroup, staged from HBM by stage_table():
//   geometry, kind by kind, only the elements below the runtime count:
//     Sphere   {center.xyz, radius}                       1 float4
//     Plane    {normal.xyz, offset}                       1 float4
//     Box      {center.xyz, -} {side.xyz, -}              2 float4
//     Triangle {v1.xyz, -} {v2.xyz, -} {v3.xyz, -}        3 float4
//   material ids of the primitives (int32, same order)
//   PointLight {position.xyz, -} {color.xyz, -}; SpotLight {position.xyz, aperture} {direction.xyz, -} {color.xyz, -}
//   Material {albedo.xyz, metallic} {roughness, -, -, -}
//   u8 -> float table: 256 entries k / 255 (RGB8 texel decode without a division)
extern __shared__ float4 s_tab[];

MDH_DEV void stage_table(const KScene &sc)
{
   for (int i = threadIdx.x; i < sc.table_f4; i += blockDim.x) s_tab[i] = sc.table[i];
   if (sc.part_bits_f4 > 0) { // the space partition's candidate bits of every cell (partitioning_closest_bits), when they are small enough
      const float4 *bits = (const float4 *)(sc.part_table + sc.part_mask_off);
      for (int i = threadIdx.x; i < sc.part_bits_f4; i += blockDim.x) s_tab[sc.table_f4 + i] = bits[i];
   }
   __syncthreads();
}
MDH_DEV int prim_slots(int type) { return type == PK_TRIANGLE ? 3 : (type == PK_BOX ? 2 : 1); }
MDH_DEV int tab_int(int int_index) { return ((const int *)s_tab)[int_index]; }
MDH_DEV float tab_float(int float_index) { return ((const float *)s_tab)[float_index]; }
// header ints: wave-uniform LDS reads, moved to SGPRs
MDH_DEV int hdr(int i) { return __builtin_amdgcn_readfirstlane(tab_int(i)); }
// four consecutive header ints (i a multiple of 4): one LDS read
MDH_DEV int4 hdr4(int i)
{
   const int4 v = ((const int4 *)s_tab)[i >> 2];
   return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y), __builtin_amdgcn_readfirstlane(v.z), __builtin_amdgcn_readfirstlane(v.w));
}

// ---------------------------------------------------------------------------- the SDFs
// madarch-primitives-spheres.ads:13-14
MDH_DEV float sd_sphere(float4 a, f3 p) { return length(xyz(a) - p) - a.w; }
// madarch-primitives-planes.ads:13-14
MDH_DEV float sd_plane(float4 a, f3 p) { return dot(xyz(a), p) + a.w; }
// madarch-primitives-boxes.adb:7-15
MDH_DEV float sd_box(float4 a, float4 b, f3 p)
{
   f3 q = abs3(xyz(a) - p) - xyz(b);
   return length(max3s(q, 0.0f)) + min_(max_(q.x, max_(q.y, q.z)), 0.0f);
}
// madarch-primitives-triangles.adb:16-48; ADA_DIV reproduces Madarch.Values."/" (L + R)
template <bool ADA_DIV> MDH_DEV float tdiv(float a, float b) { return ADA_DIV ? a + b : a / b; }
template <bool ADA_DIV> MDH_TRI float sd_triangle(f3 a, f3 b, f3 c, f3 p)
{
   f3 v21 = b - a, v32 = c - b, v13 = a - c;
   f3 p1 = p - a, p2 = p - b, p3 = p - c;
   f3 nor = cross(v21, v13);
   float s = (sign_(dot(cross(v21, nor), p1)) + sign_(dot(cross(v32, nor), p2))) + sign_(dot(cross(v13, nor), p3));
   float r;
   if (s < 2.0f) {
      float e1 = dot2(v21 * clamp_(tdiv<ADA_DIV>(dot(v21, p1), dot2(v21)), 0.0f, 1.0f) - p1);
      float e2 = dot2(v32 * clamp_(tdiv<ADA_DIV>(dot(v32, p2), dot2(v32)), 0.0f, 1.0f) - p2);
      float e3 = dot2(v13 * clamp_(tdiv<ADA_DIV>(dot(v13, p3), dot2(v13)), 0.0f, 1.0f) - p3);
      r = min_(min_(e1, e2), e3);
   } else {
      r = tdiv<ADA_DIV>(dot(nor, p1) * dot(nor, p1), dot2(nor));
   }
   return sqrt_(r);
}
// madarch-primitives-boxes.adb:5,17-41
MDH_DEV f3 nrm_box(float4 a, float4 b, f3 p)
{
   const float e = 0.002f;
   f3 d = (p - xyz(a)) / xyz(b);
   float rx = __builtin_fabsf(d.x), ry = __builtin_fabsf(d.y), rz = __builtin_fabsf(d.z);
   f3 n = F3(((rx > ry - e ? 1.0f : 0.0f) * (rx > rz - e ? 1.0f : 0.0f)) * sign_(d.x),
             ((ry > rx - e ? 1.0f : 0.0f) * (ry > rz - e ? 1.0f : 0.0f)) * sign_(d.y),
             ((rz > rx - e ? 1.0f : 0.0f) * (rz > ry - e ? 1.0f : 0.0f)) * sign_(d.z));
   return snormalize(n);
}
// madarch-primitives-triangles.adb:50-56 + madarch-exprs-derivatives.adb:12-45
template <bool ADA_DIV> MDH_TRI f3 nrm_triangle(f3 a, f3 b, f3 c, f3 p)
{
   const float eps = 0.000001f;
   float fp = sd_triangle<ADA_DIV>(a, b, c, p);
   float fx = sd_triangle<ADA_DIV>(a, b, c, p + F3(eps, 0.0f, 0.0f)) - fp;
   float fy = sd_triangle<ADA_DIV>(a, b, c, p + F3(0.0f, eps, 0.0f)) - fp;
   float fz = sd_triangle<ADA_DIV>(a, b, c, p + F3(0.0f, 0.0f, eps)) - fp;
   return normalize(F3(fx, fy, fz));
}

// ------------------------------------------------------------- user-defined kinds: MDH_X
// The interpreter of the register programs include/madarch_hip.h specifies (one IEEE fp32
// operation per instruction).  The program is wave-uniform -- its words come out of the LDS table
// through readfirstlane, the dispatch is scalar branches -- while the instance (`ent`, a float
// index into the table) and the point are per lane.  The 64 registers are one vector value whose
// element index is uniform, which hipcc maps onto VGPRs with s_set_gpr_idx (no scratch).
#ifndef MDH_XRUN
#define MDH_XRUN static __device__ __forceinline__
#endif
// (two banks of 32: a 32-dword tuple is the widest register class gfx950 indexes dynamically)
typedef float xbank __attribute__((ext_vector_type(32)));
// Both banks are read and written UNCONDITIONALLY with the value chosen by a select: a branch on the
// bank bit makes hipcc merge two copies of each 32-register vector per instruction (measured
// 1090 ns per instruction and SIMD against 74 this way and 60 for one bank); as members of a struct
// with accessor functions the banks end up in scratch.
#define XGET(i, out)                                                                               \
   do {                                                                                            \
      const float g0_ = lo[(i) & 31], g1_ = hi[(i) & 31];                                          \
      out = ((i) & 32) ? g1_ : g0_;                                                                \
   } while (0)
#define XPUT(i, v)                                                                                 \
   do {                                                                                            \
      const float o0_ = lo[(i) & 31], o1_ = hi[(i) & 31];                                          \
      lo[(i) & 31] = ((i) & 32) ? o0_ : (v);                                                       \
      hi[(i) & 31] = ((i) & 32) ? (v) : o1_;                                                       \
   } while (0)
// x = argument floats 0..2 (the point / pos); nrm, dir, dist = floats 3..9 of a light's Sample program
template <bool ADA_DIV> MDH_XRUN f3 xrun(int code, int n, int ent, f3 x, f3 nrm = F3(0.0f, 0.0f, 0.0f), f3 dir = F3(0.0f, 0.0f, 0.0f), float dist = 0.0f)
{
   xbank lo = 0.0f, hi = 0.0f;
#pragma unroll 1
   for (int pc = 0; pc < n; ++pc) {
      const int w = hdr(code + pc);
      const int op = w & 255, d = (w >> 8) & 63, a = (w >> 16) & 255, b = (w >> 24) & 63;
      float va, vb;
      XGET(a & 63, va);
      XGET(b, vb);
      float r;
      switch (op) {
      case 0: r = __builtin_bit_cast(float, hdr(code + ++pc)); break;                  // LIT
      case 1: r = va; break;                                                          // MOV
      case 2: r = tab_float(ent + a); break;                                          // COMP
      case 3: // POINT (a is wave-uniform)
         r = a == 0 ? x.x : a == 1 ? x.y : a == 2 ? x.z : a == 3 ? nrm.x : a == 4 ? nrm.y : a == 5 ? nrm.z : a == 6 ? dir.x : a == 7 ? dir.y : a == 8 ? dir.z : dist;
         break;
      case 4: r = va + vb; break;
      case 5: r = va - vb; break;
      case 6: r = va * vb; break;
      case 7: r = va / vb; break;
      case 8: r = ADA_DIV ? va + vb : va / vb; break;                                 // DIVF
      case 9: r = -va; break;
      case 10: r = __builtin_fabsf(va); break;
      case 11: r = __builtin_floorf(va); break;
      case 12: r = sign_(va); break;
      case 13: r = min_(va, vb); break;
      case 14: r = max_(va, vb); break;
      case 15: r = sqrt_(va); break;
      case 16: r = pow_(va, vb); break;
      case 17: r = va < vb ? 1.0f : 0.0f; break;
      case 18: r = va > vb ? 1.0f : 0.0f; break;
      case 19: r = va <= vb ? 1.0f : 0.0f; break;
      case 20: r = va >= vb ? 1.0f : 0.0f; break;
      case 21: { const int c = hdr(code + ++pc) & 63; float vc; XGET(c, vc); r = va != 0.0f ? vb : vc; break; } // SEL
      case 22: r = (float)__builtin_bit_cast(int, va); break;                          // ITOF
      case 23: r = acos_(va); break;
      case 24: r = sin_(va); break;
      case 25: r = cos_(va); break;
      case 26: r = tan_(va); break;
      case 27: r = asin_(va); break;
      case 28: r = atan_(va); break;
      default: r = 0.0f; break;
      }
      XPUT(d, r);
   }
   return F3(lo[0], lo[1], lo[2]);
}
// The five entry points of user-defined kinds: kind k is wave-uniform, instance i per lane.  In the
// library build they interpret the kind's MDH_X programs; in a JIT build (MDH_JIT, the module hiprtc
// compiles for one scene, mdh_api.hip) mdh_jit_kinds.h holds the same programs as straight-line
// functions, dispatched by a scalar switch on k.
MDH_DEV int x_prim_ent(int k, int i) { return (hdr(H_KSLOT + k) + hdr(H_KSTRIDE + k) * i) * 4; }
MDH_DEV int x_light_ent(int k, int i) { return (hdr(H_LSLOT + k) + hdr(H_LSTRIDE + k) * i) * 4; }
#ifdef MDH_JIT
#include "mdh_jit_kinds.h"
#endif
template <bool ADA_DIV> MDH_DEV float xdist(int k, int i, f3 x)
{
#ifdef MDH_JIT
   return jit_prim<ADA_DIV>(0, k, x_prim_ent(k, i), x).x;
#else
   return xrun<ADA_DIV>(hdr(H_XDIST + k), hdr(H_XDISTN + k), x_prim_ent(k, i), x).x;
#endif
}
template <bool ADA_DIV> MDH_DEV f3 xnormal(int k, int i, f3 x)
{
#ifdef MDH_JIT
   return jit_prim<ADA_DIV>(1, k, x_prim_ent(k, i), x);
#else
   return xrun<ADA_DIV>(hdr(H_XNRM + k), hdr(H_XNRMN + k), x_prim_ent(k, i), x);
#endif
}
MDH_DEV int xmaterial(int k, int i)
{
#ifdef MDH_JIT
   return __builtin_bit_cast(int, jit_prim<false>(2, k, x_prim_ent(k, i), F3(0.0f, 0.0f, 0.0f)).x);
#else
   return __builtin_bit_cast(int, xrun<false>(hdr(H_XMAT + k), hdr(H_XMATN + k), x_prim_ent(k, i), F3(0.0f, 0.0f, 0.0f)).x);
#endif
}
MDH_DEV f3 xlight_position(int k, int i)
{
#ifdef MDH_JIT
   return jit_light(1, k, x_light_ent(k, i), F3(0.0f, 0.0f, 0.0f), F3(0.0f, 0.0f, 0.0f), F3(0.0f, 0.0f, 0.0f), 0.0f);
#else
   return xrun<false>(hdr(H_XLPOS + k), hdr(H_XLPOSN + k), x_light_ent(k, i), F3(0.0f, 0.0f, 0.0f));
#endif
}
MDH_DEV f3 xlight_sample(int k, int i, f3 pos, f3 normal, f3 dir, float dist)
{
#ifdef MDH_JIT
   return jit_light(0, k, x_light_ent(k, i), pos, normal, dir, dist);
#else
   return xrun<false>(hdr(H_XLSAMPLE + k), hdr(H_XLSAMPLEN + k), x_light_ent(k, i), pos, normal, dir, dist);
#endif
}

// dist_to_<Kind>(prims[i], x); `slot` = first float4 of the primitive (any lane value)
MDH_DEV float prim_dist(int type, int slot, f3 x)
{
   switch (type) {
   case PK_SPHERE: return sd_sphere(s_tab[slot], x);
   case PK_PLANE: return sd_plane(s_tab[slot], x);
   case PK_BOX: return sd_box(s_tab[slot], s_tab[slot + 1], x);
   default: return sd_triangle<false>(xyz(s_tab[slot]), xyz(s_tab[slot + 1]), xyz(s_tab[slot + 2]), x);
   }
}

#ifndef MDH_INFO_UNROLL
#define MDH_INFO_UNROLL 1
#endif
#ifndef MDH_PART_PREFETCH
#define MDH_PART_PREFETCH 1
#endif
#ifndef MDH_FAST_INFO
#define MDH_FAST_INFO 1
#endif
#ifndef MDH_PART_SMALL
#define MDH_PART_SMALL 1 // small scenes walk a cell's bits in straight-line code (KScene::part_small)
#endif
#ifndef MDH_PART_BITS
#define MDH_PART_BITS 1 // the distance-only partition lookup walks a cell's candidates as bits (partitioning_closest_bits)
#endif
#ifndef MDH_SDF_UNROLL
#define MDH_SDF_UNROLL 2
#endif
// closest_primitive (scenes.adb:602-629).  min is order-free, so the primitives are visited by
// TYPE -- planes first, they are cheap and bring `closest` down -- in four plain loops with
// wave-uniform trip counts and LDS broadcast reads.
//
// Spheres and boxes are CULLED conservatively before their square root: when a cheap lower
// bound of the distance already exceeds `closest` in every lane of the wave (one ballot), the
// rest of the evaluation is skipped; min(closest, d) would have returned `closest`, so the
// result is bit-identical.  Margins (1e-6 relative, >> the few-2^-24 rounding of the bound):
//   sphere: d = sqrt(d2) - r >= closest  <=  d2 > (closest + r)^2 (1 + 1e-6), or closest + r < 0
//   box:    d >= m (1 - 2^-23) with m = max(q.x, q.y, q.z); skip when m > closest (1 + 1e-6)
//           (for m <= 0 the distance IS m, for closest <= 0 the test m > closest is exact)
#ifndef MDH_CULL
#define MDH_CULL 1
#endif
#ifndef MDH_SDF_PREFETCH
#define MDH_SDF_PREFETCH 1
#endif
#ifndef MDH_ROOM_SQRT2
#define MDH_ROOM_SQRT2 0 // the rooms' scan evaluates its sphere and its box together, their square-root cores interleaved (sqrt2_)
#endif
#ifndef MDH_SDF_SGPR
#define MDH_SDF_SGPR 0
#endif
#ifndef MDH_SDF_SQRT_WAVE
#define MDH_SDF_SQRT_WAVE 0 // the brute-force scan's sphere and box roots through sqrt_wave_ (below)
#endif
// The first sphere and the first box of the table in registers: a march loop that evaluates the SDF many times
// reads them from LDS once (MDH_SDF_REGS) instead of once per evaluation.  (With a count of 0 the words
// belong to the next kind and are not used.)
struct SdfRegs { float4 s, b0, b1; };
MDH_DEV SdfRegs sdf_regs(const KScene &sc)
{
   SdfRegs r;
#if MDH_SDF_SGPR
   r.s = r.b0 = r.b1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f); // (unused: closest_primitive takes the kernel arguments)
   return r;
#endif
   r.s = s_tab[sc.tslot[PK_SPHERE]];
   r.b0 = s_tab[sc.tslot[PK_BOX]]; r.b1 = s_tab[sc.tslot[PK_BOX] + 1];
   return r;
}
// ROOM (MDH_PF_ROOM): the scan of a scene whose CENSUS is that of the reference's rooms -- every plane folded into the six axis
// offsets, exactly one sphere, exactly one box, no triangles, no user-defined kinds (global_illumination, light_shafts: BASELINE
// configs 3, 4, 5).  The same operations on the same operands; what goes is what a count known only at run time costs at every
// march step -- the loops' scalar bookkeeping and branches around bodies that run once or never -- and the registers their
// induction state holds: 96 -> 80 VGPRs in the screen kernel, six wavefronts per SIMD without a further spill.
template <bool CUSTOM, bool ROOM = false> MDH_DEV float closest_primitive(const KScene &sc, f3 x, const SdfRegs *regs = nullptr)
{
   float closest = sc.max_dist;
#if MDH_SDF_SGPR
   const float4 pf_s = make_float4(sc.first_sphere[0], sc.first_sphere[1], sc.first_sphere[2], sc.first_sphere[3]);
   const float4 pf_b0 = make_float4(sc.first_box[0], sc.first_box[1], sc.first_box[2], sc.first_box[3]), pf_b1 = make_float4(sc.first_box[4], sc.first_box[5], sc.first_box[6], sc.first_box[7]);
#elif MDH_SDF_PREFETCH
   // the first sphere and the first box are on their way from LDS while the planes are evaluated: a wavefront that
   // has a SIMD to itself (the tail of every pass, and whole passes of a sharded frame) has nothing else to
   // hide that latency behind.
   const float4 pf_s = regs ? regs->s : s_tab[sc.tslot[PK_SPHERE]];
   const float4 pf_b0 = regs ? regs->b0 : s_tab[sc.tslot[PK_BOX]], pf_b1 = regs ? regs->b1 : s_tab[sc.tslot[PK_BOX] + 1];
#endif
   if (ROOM || sc.n_axis > 0) { // six adds for all axis-aligned planes together
      closest = min_(closest, min_(x.x + sc.axis_off[0], -x.x + sc.axis_off[1]));
      closest = min_(closest, min_(x.y + sc.axis_off[2], -x.y + sc.axis_off[3]));
      closest = min_(closest, min_(x.z + sc.axis_off[4], -x.z + sc.axis_off[5]));
   }
#if MDH_ROOM_SQRT2
   if (ROOM) { // the one sphere and the one box together (sqrt2_ above); the box's cull against the minimum BEFORE the sphere: never less careful
      const float d2 = dot2(xyz(pf_s) - x);
      const float tsum = closest + pf_s.w;
      const bool need_s = !(tsum < 0.0f) && !(d2 > (tsum * tsum) * 1.000001f);
      const f3 q = abs3(xyz(pf_b0) - x) - xyz(pf_b1);
      const float m = max_(q.x, max_(q.y, q.z));
      const float thr = closest > 0.0f ? closest * 1.000001f : closest;
      if (!MDH_CULL || __ballot(need_s || !(m > thr)) != 0ull) {
         float rs, rb;
         sqrt2_(d2, dot2(F3(max0_raw(q.x), max0_raw(q.y), max0_raw(q.z))), rs, rb);
         closest = min_raw(closest, rs - pf_s.w);
         closest = min_raw(closest, rb + min0_raw(m));
      }
      return closest;
   }
#endif
   {
      const int n = ROOM ? 0 : sc.gplane_count, s0 = sc.gplane_slot;
#pragma unroll MDH_SDF_UNROLL
      for (int i = 0; i < n; ++i) closest = min_(closest, sd_plane(s_tab[s0 + i], x)); // (unrolled in pairs: v_min3_f32)
   }
   {
      const int n = ROOM ? 1 : sc.tcount[PK_SPHERE], s0 = sc.tslot[PK_SPHERE];
#define MDH_SPHERE_STEP(a_)                                                                  \
      do {                                                                                   \
         const float4 a = (a_);                                                              \
         const float d2 = dot2(xyz(a) - x); /* sd_sphere = sqrt(d2) - a.w (spheres.ads:13-14) */ \
         const float tsum = closest + a.w;                                                   \
         const bool need = !(tsum < 0.0f) && !(d2 > (tsum * tsum) * 1.000001f);              \
         if (!MDH_CULL || __ballot(need) != 0ull) closest = min_raw(closest, (MDH_SDF_SQRT_WAVE ? sqrt_wave_(d2) : sqrt_(d2)) - a.w); \
      } while (0)
#if MDH_SDF_PREFETCH
      if (n > 0) MDH_SPHERE_STEP(pf_s);
#pragma unroll 1
      for (int i = 1; i < n; ++i) MDH_SPHERE_STEP(s_tab[s0 + i]);
#else
#pragma unroll 1
      for (int i = 0; i < n; ++i) MDH_SPHERE_STEP(s_tab[s0 + i]);
#endif
#undef MDH_SPHERE_STEP
   }
   {
      const int n = ROOM ? 1 : sc.tcount[PK_BOX], s0 = sc.tslot[PK_BOX];
#define MDH_BOX_STEP(c_, e_)                                                                 \
      do {                                                                                   \
         const f3 q = abs3(xyz(c_) - x) - xyz(e_); /* boxes.adb:10 */                        \
         const float m = max_(q.x, max_(q.y, q.z));                                          \
         const float thr = closest > 0.0f ? closest * 1.000001f : closest;                   \
         if (!MDH_CULL || __ballot(!(m > thr)) != 0ull)                                      \
            closest = min_raw(closest, (MDH_SDF_SQRT_WAVE ? sqrt_wave_(dot2(F3(max0_raw(q.x), max0_raw(q.y), max0_raw(q.z)))) : length(F3(max0_raw(q.x), max0_raw(q.y), max0_raw(q.z)))) + min0_raw(m)); \
      } while (0)
#if MDH_SDF_PREFETCH
      if (n > 0) MDH_BOX_STEP(pf_b0, pf_b1);
#pragma unroll 1
      for (int i = 1; i < n; ++i) MDH_BOX_STEP(s_tab[s0 + 2 * i], s_tab[s0 + 2 * i + 1]);
#else
#pragma unroll 1
      for (int i = 0; i < n; ++i) MDH_BOX_STEP(s_tab[s0 + 2 * i], s_tab[s0 + 2 * i + 1]);
#endif
#undef MDH_BOX_STEP
   }
   {
      const int n = ROOM ? 0 : sc.tcount[PK_TRIANGLE], s0 = sc.tslot[PK_TRIANGLE];
#pragma unroll 1
      for (int i = 0; i < n; ++i)
         closest = min_raw(closest, sd_triangle<false>(xyz(s_tab[s0 + 3 * i]), xyz(s_tab[s0 + 3 * i + 1]), xyz(s_tab[s0 + 3 * i + 2]), x));
   }
#ifdef MDH_JIT_CLOSEST_ALL
   if (CUSTOM) closest = jit_closest_all(x, closest); // (compiled: one loop per kind, roots culled like the built-in spheres')
   else
#endif
   if (CUSTOM) { // user-defined kinds: their Distance programs, interpreted
      const int nk = hdr(H_NK);
#pragma unroll 1
      for (int k = 0; k < nk; ++k) {
         if (hdr(H_KTYPE + k) != PK_CUSTOM) continue;
         const int n = hdr(H_KCOUNT + k);
#pragma unroll 1
         for (int i = 0; i < n; ++i) closest = min_raw(closest, xdist<false>(k, i, x));
      }
   }
   return closest;
}
// closest_primitive_info (scenes.adb:631-674): kinds in SCENE order (the arg-min keeps the
// first of equal distances).  Only evaluated at hit points, so compact rather than fast.
template <bool CUSTOM> MDH_DEV float closest_primitive_info(const KScene &sc, f3 x, int &index)
{
   float closest = sc.max_dist;
#if MDH_FAST_INFO
   // The same arg-min by TYPE, with the six folded axis planes and without the per-kind dispatch: a candidate wins
   // when it is closer, or as close with a lower flat index -- what the scan in scene order keeps (the first of
   // equal distances).  Only for scenes whose planes are all folded, one per direction (H_FASTINFO).
   if (!CUSTOM && hdr(H_FASTINFO)) {
      int best = -1;
#define MDH_CAND(dist_, idx_)                                                                      \
      do {                                                                                         \
         const float d_ = (dist_);                                                                 \
         const int i_ = (idx_);                                                                    \
         if (d_ < closest || (d_ == closest && best >= 0 && i_ < best)) { closest = d_; best = i_; } \
      } while (0)
      if (sc.n_axis > 0) {
         const float c[6] = {x.x, -x.x, x.y, -x.y, x.z, -x.z};
#pragma unroll
         for (int g = 0; g < 6; ++g) {
            const int pi = hdr(H_AXIS_IDX + g);
            if (pi >= 0) MDH_CAND(c[g] + sc.axis_off[g], pi);
         }
      }
      {
         const int n = sc.tcount[PK_SPHERE], s0 = sc.tslot[PK_SPHERE], base = hdr(H_TBASE + PK_SPHERE);
#pragma unroll 1
         for (int i = 0; i < n; ++i) MDH_CAND(sd_sphere(s_tab[s0 + i], x), base + i);
      }
      {
         const int n = sc.tcount[PK_BOX], s0 = sc.tslot[PK_BOX], base = hdr(H_TBASE + PK_BOX);
#pragma unroll 1
         for (int i = 0; i < n; ++i) MDH_CAND(sd_box(s_tab[s0 + 2 * i], s_tab[s0 + 2 * i + 1], x), base + i);
      }
      {
         const int n = sc.tcount[PK_TRIANGLE], s0 = sc.tslot[PK_TRIANGLE], base = hdr(H_TBASE + PK_TRIANGLE);
#pragma unroll 1
         for (int i = 0; i < n; ++i)
            MDH_CAND(sd_triangle<false>(xyz(s_tab[s0 + 3 * i]), xyz(s_tab[s0 + 3 * i + 1]), xyz(s_tab[s0 + 3 * i + 2]), x), base + i);
      }
#undef MDH_CAND
      if (best >= 0) index = best;
      return closest;
   }
#endif
   const int nk = hdr(H_NK);
#pragma unroll 1
   for (int k = 0; k < nk; ++k) {
      const int n = hdr(H_KCOUNT + k), s0 = hdr(H_KSLOT + k), base = hdr(H_KBASE + k), type = hdr(H_KTYPE + k);
#pragma unroll MDH_INFO_UNROLL
      for (int i = 0; i < n; ++i) {
         float d = (CUSTOM && type == PK_CUSTOM) ? xdist<false>(k, i, x) : prim_dist(type, s0 + prim_slots(type) * i, x);
         if (d < closest) { closest = d; index = base + i; }
      }
   }
   return closest;
}

// material id and normal of a flat index: primitive_info (scenes.adb:676-729),
// kind found by successive subtraction of the DECLARED counts
template <bool CUSTOM> MDH_DEV void primitive_info(const KScene &sc, int index, f3 pos, f3 &normal, int &material_id)
{
   normal = F3(0.0f, 0.0f, 0.0f);
   material_id = 0;
   const int nk = hdr(H_NK);
#pragma unroll 1
   for (int k = 0; k < nk; ++k) {
      const int kmax = hdr(H_KMAX + k);
      if (index < kmax) {
         const int type = hdr(H_KTYPE + k);
         if (CUSTOM && type == PK_CUSTOM) { // the kind's Normal and Material programs on this lane's instance
            normal = xnormal<false>(k, index, pos);
            material_id = xmaterial(k, index);
            return;
         }
         const int slot = hdr(H_KSLOT + k) + prim_slots(type) * index; // per-lane LDS gather
         material_id = tab_int(hdr(H_KMAT + k) + index);
         float4 a = s_tab[slot];
         switch (type) {
         case PK_SPHERE: normal = snormalize(pos - xyz(a)); break; // spheres.ads:16-17
         case PK_PLANE: normal = xyz(a); break;                   // planes.ads:16-17
         case PK_BOX: normal = nrm_box(a, s_tab[slot + 1], pos); break;
         default: normal = nrm_triangle<false>(xyz(a), xyz(s_tab[slot + 1]), xyz(s_tab[slot + 2]), pos); break;
         }
         return;
      }
      index -= kmax;
   }
}

// ------------------------------------------------------------------- space partition
// partitioning index (scenes.adb:799-837); the reference clamps to `dims`, an index
// past the table reads an empty cell
MDH_DEV int partition_cell(const KScene &sc, f3 x, bool &fallback)
{
   const f3 rel = x - F3(sc.part_off[0], sc.part_off[1], sc.part_off[2]);
   f3 fx = floor3(sc.part_sp_pow2 ? rel * F3(sc.part_inv_sp[0], sc.part_inv_sp[1], sc.part_inv_sp[2]) : rel / F3(sc.part_sp[0], sc.part_sp[1], sc.part_sp[2]));
   f3 cfx = F3(clamp_(fx.x, 0.0f, (float)sc.part_dims[0]), clamp_(fx.y, 0.0f, (float)sc.part_dims[1]), clamp_(fx.z, 0.0f, (float)sc.part_dims[2]));
   fallback = false;
   if (sc.part_border == 0) fx = cfx;
   else if (fx.x != cfx.x || fx.y != cfx.y || fx.z != cfx.z) { fallback = true; return -1; }
   float yz = (float)(sc.part_dims[1] * sc.part_dims[2]), zz = (float)sc.part_dims[2];
   return (int)((fx.x * yz + fx.y * zz) + fx.z);
}
// (Border_Behavior = Clamp known when the kernel is built)
template <bool PSMALL = false> MDH_DEV int partition_cell_clamped(const KScene &sc, f3 x) // (PSMALL: MDH_PF_PSMALL below)
{
   const f3 rel = x - F3(sc.part_off[0], sc.part_off[1], sc.part_off[2]);
   const f3 fx = floor3((PSMALL || sc.part_sp_pow2) ? rel * F3(sc.part_inv_sp[0], sc.part_inv_sp[1], sc.part_inv_sp[2]) : rel / F3(sc.part_sp[0], sc.part_sp[1], sc.part_sp[2]));
#if MDH_PART_CELL_TRIM
   // clamp (v, 0, d) = min (max (v, 0), d) as ONE instruction: v_med3_f32 is the median of its operands and, with a NaN among
   // them, their minNum -- 0 for v = NaN, as the two-instruction form gives (max (NaN, 0) = 0; d >= 0)
   const f3 cfx = F3(__builtin_amdgcn_fmed3f(fx.x, 0.0f, sc.part_fdims[0]), __builtin_amdgcn_fmed3f(fx.y, 0.0f, sc.part_fdims[1]), __builtin_amdgcn_fmed3f(fx.z, 0.0f, sc.part_fdims[2]));
   const float yz = sc.part_fyz, zz = sc.part_fdims[2];
   // (every term and every partial sum is an integer below 2^24 for tables of fewer than 2^24 cells: the two fused forms round
   //  nothing that the four separate operations would not -- the same value with two instructions instead of four)
   if (PSMALL || sc.part_cells < (1 << 24)) return (int)__builtin_fmaf(cfx.x, yz, __builtin_fmaf(cfx.y, zz, cfx.z));
#else
   const f3 cfx = F3(clamp_(fx.x, 0.0f, sc.part_fdims[0]), clamp_(fx.y, 0.0f, sc.part_fdims[1]), clamp_(fx.z, 0.0f, sc.part_fdims[2]));
   const float yz = sc.part_fyz, zz = sc.part_fdims[2];
#endif
   return (int)((cfx.x * yz + cfx.y * zz) + cfx.z);
}
// partitioning_closest[_info] (scenes.adb:839-1118): per-lane cell record from HBM/L2,
// per-lane primitive gather from LDS
template <bool INFO, bool CUSTOM, bool FALLBACK = true, bool PSMALL = false> MDH_DEV float partitioning_lookup(const KScene &sc, f3 x, int &index)
{
   bool fb = false;
   int cell = FALLBACK ? partition_cell(sc, x, fb) : partition_cell_clamped<PSMALL>(sc, x);
   if (FALLBACK && fb) return INFO ? closest_primitive_info<CUSTOM>(sc, x, index) : closest_primitive<CUSTOM>(sc, x);
   float closest = sc.max_dist;
   if (cell < 0 || cell >= sc.part_cells) return closest;
   const int nk = hdr(H_NK);
   const int *rec = sc.part_table + (size_t)cell * (nk + sc.part_index_count);
#if MDH_PART_PREFETCH
   // every record word is asked for one step before it is needed (the count of the next kind, the next candidate
   // index): the L2 round trips of a cell overlap with the distance evaluations instead of adding up
   int i = 0, cnt = rec[0], pi_next = rec[nk];
#pragma unroll 1
   for (int k = 0; k < nk; ++k) {
      const int cnt_next = rec[k + 1 < nk ? k + 1 : k];
      const int size = i + cnt, type = hdr(H_KTYPE + k), s0 = hdr(H_KSLOT + k), base = hdr(H_KBASE + k);
      const int stop = min(size, sc.part_index_count);
      for (; i < stop; ++i) {
         const int pi = pi_next;
         pi_next = rec[nk + (i + 1 < sc.part_index_count ? i + 1 : i)]; // (inside the record; unused past the last candidate)
         float d = (CUSTOM && type == PK_CUSTOM) ? xdist<false>(k, pi, x) : prim_dist(type, s0 + prim_slots(type) * pi, x);
         if (INFO) { if (d < closest) { closest = d; index = base + pi; } }
         else closest = min_raw(closest, d);
      }
      i = size;
      cnt = cnt_next;
   }
#else
   int i = 0;
#pragma unroll 1
   for (int k = 0; k < nk; ++k) {
      const int size = i + rec[k], type = hdr(H_KTYPE + k), s0 = hdr(H_KSLOT + k), base = hdr(H_KBASE + k);
      const int stop = min(size, sc.part_index_count);
      for (; i < stop; ++i) {
         int pi = rec[nk + i];
         float d = (CUSTOM && type == PK_CUSTOM) ? xdist<false>(k, pi, x) : prim_dist(type, s0 + prim_slots(type) * pi, x);
         if (INFO) { if (d < closest) { closest = d; index = base + pi; } }
         else closest = min_raw(closest, d);
      }
      i = size;
   }
#endif
   return closest;
}
// partitioning_closest (scenes.adb:839-958) from the cell's candidates as BITS.
//
// A cell's record names its candidates as indices, kind by kind (scenes.adb:875-941): a lane that walks it pays one
// dependent L2 / L1 round trip per candidate before it can even gather the primitive from LDS, 23 ints per cell in the
// reference's simple_scene.  The minimum over the candidates does not depend on their order or on repetitions, so for
// the distance-only lookup -- every march step; the arg-min at hit points keeps the lists and their order -- the same
// set is stored once more as one bit per DECLARED primitive (bit = flat index, scenes.adb:656-666; k_partition_bits
// derives it from the lists exactly as the walk above reads them, cut at Index_Count included): 50 bits = two dwords
// per cell for simple_scene, 16 KB for the whole grid instead of 184 KB.  A lane loads its cell's dwords (one or two
// loads per step) and walks its set bits; the loop over kinds stays wave-uniform, so the type dispatch is scalar and
// lanes in the same cell gather the same primitive (an LDS broadcast).
// A lane's set bits of one built-in type, two at a time: both gathers are in flight before either distance is computed (a
// lane's last odd candidate is evaluated twice -- the minimum does not change, and the wave waits for its longest list
// anyway).  Lanes whose rays are in the same cell -- most of an 8 x 8 tile's -- gather the same primitive at the same time
// (an LDS broadcast).  `t` = the first float4 of the instance bit 0 stands for.
// ONE square root per lookup for all sphere candidates of a radius and ONE for all box candidates (round 4).
//
// simple_scene's cells name 1.8 candidates on average (at most 8), so the pair loops below mostly evaluated one sphere
// and one box TWICE each -- four correctly rounded square roots (16 instructions each) per march step, 120 of a step's 200
// vector instructions.  The square root and the IEEE addition behind it are monotonic, so the minimum commutes with them:
//   spheres of one radius r:  min_i (sqrt (d2_i) - r) = sqrt (min_i d2_i) - r     (bit for bit: fl is non-decreasing)
//   boxes:  t_i = sqrt (s_i) + u_i with s_i = |max (q_i, 0)|^2 and u_i = min (max (q_i.x, q_i.y, q_i.z), 0).  Outside a box
//           u_i = 0 and t_i = sqrt (s_i) >= 0; inside, s_i = 0 and t_i = u_i <= 0.  So min_i t_i = sqrt (min_i s_i) + min_i u_i:
//           with a point inside some box the first term is sqrt (0) = 0 and the second the deepest box; otherwise the
//           second is 0 and the first the nearest box -- the same formula as one box, the minima taken first.
//           (A box whose q is all NaN gave 0 + min (NaN, 0) = 0 before and gives s = 0, u = 0 now.)
// A lane's spheres may have several radii (ball_game: one of 1.0, ten of 0.2): a candidate whose radius differs from the
// running one first FLUSHES the running minimum through its square root (behind a ballot: skipped by the wave when no
// lane's radius changes -- never in simple_scene).  And the last root of either kind is culled like the brute-force scan's
// (closest_primitive above): skipped by the wave when no lane's bound can still lower `closest`.
#ifndef MDH_PART_MERGE_ROOTS
#define MDH_PART_MERGE_ROOTS 1
#endif
#ifndef MDH_PART_PLANE_SINGLE
#define MDH_PART_PLANE_SINGLE 1
#endif
#ifndef MDH_PART_ALIGNBIT
#define MDH_PART_ALIGNBIT 0 // (measured: the scalar branch of the funnel-shift form costs more than the 64-bit shift it replaces, -4 % on simple_scene)
#endif
#ifndef MDH_PART_CELL_TRIM
#define MDH_PART_CELL_TRIM 1
#endif
MDH_DEV float walk_planes_single(unsigned w, const float4 *t, f3 x, float closest)
{
   while (w) { // (a cell of the reference's scenes names one or two planes: pairs evaluated most of them twice)
      MDH_DIAG_STEP(6);
      const int p = __builtin_ctz(w);
      w &= w - 1u;
      closest = min_raw(closest, sd_plane(t[p], x));
   }
   return closest;
}
MDH_DEV float walk_spheres_merged(unsigned w, const float4 *t, f3 x, float closest)
{
   float S = __builtin_inff(), r_cur = 0.0f;
   while (w) {
      MDH_DIAG_STEP(6);
      const int p = __builtin_ctz(w);
      w &= w - 1u;
      const float4 a = t[p];
      const float d2 = dot2(xyz(a) - x);
      const bool other = __float_as_int(a.w) != __float_as_int(r_cur);
      if ((__ballot(other) & __ballot(S < __builtin_inff())) != 0ull) { // (wave-uniform) some lane changes its radius: its minimum so far goes through the root
         if (other) closest = min_raw(closest, sqrt_wave_(S) - r_cur); // (S = inf: sqrt = inf, no change)
      }
      S = min_raw(other ? __builtin_inff() : S, d2);
      r_cur = a.w; // (unchanged when !other)
   }
   const float tsum = closest + r_cur;
   const bool need = !(tsum < 0.0f) && !(S > (tsum * tsum) * 1.000001f); // (closest_primitive's cull; S = inf: no candidate, never needed)
   if (!MDH_CULL || __ballot(need) != 0ull) closest = min_raw(closest, sqrt_wave_(S) - r_cur);
   return closest;
}
MDH_DEV float walk_boxes_merged(unsigned w, const float4 *t, f3 x, float closest)
{
   float S = __builtin_inff(), U = 0.0f;
   while (w) {
      MDH_DIAG_STEP(6);
      const int p = __builtin_ctz(w);
      w &= w - 1u;
      const float4 a = t[2 * p], b = t[2 * p + 1];
      const f3 q = abs3(xyz(a) - x) - xyz(b); /* boxes.adb:10 */
      S = min_raw(S, dot2(F3(max0_raw(q.x), max0_raw(q.y), max0_raw(q.z))));
      U = min_raw(U, min0_raw(max_(q.x, max_(q.y, q.z))));
   }
   // sqrt (S) + U: U < 0 means S = 0 and the value is U itself; otherwise it is sqrt (S) >= 0, which cannot lower a negative
   // `closest` and cannot lower a positive one when S > closest^2 (1 + 1e-6)
   const bool need = !(U < 0.0f) && !(closest < 0.0f) && !(S > (closest * closest) * 1.000001f);
   if (!MDH_CULL || __ballot(need) != 0ull) closest = min_raw(closest, sqrt_wave_(S) + U);
   else closest = min_raw(closest, U < 0.0f ? U : closest);
   return closest;
}
template <int TYPE> MDH_DEV float walk_bits(unsigned w, const float4 *t, f3 x, float closest)
{
#if MDH_PART_MERGE_ROOTS
   if (TYPE == PK_SPHERE) return walk_spheres_merged(w, t, x, closest);
   if (TYPE == PK_BOX) return walk_boxes_merged(w, t, x, closest);
   if (TYPE == PK_PLANE && MDH_PART_PLANE_SINGLE) return walk_planes_single(w, t, x, closest);
#endif
   if (TYPE == PK_TRIANGLE) {
      while (w) { const int pi = __builtin_ctz(w); w &= w - 1u; closest = min_raw(closest, sd_triangle<false>(xyz(t[3 * pi]), xyz(t[3 * pi + 1]), xyz(t[3 * pi + 2]), x)); }
      return closest;
   }
   while (w) {
      MDH_DIAG_STEP(6);
      const int p0 = __builtin_ctz(w);
      w &= w - 1u;
      const int p1 = w ? __builtin_ctz(w) : p0;
      w &= w - 1u; // (0 & 0xffffffff = 0)
      if (TYPE == PK_SPHERE) {
         const float4 a0 = t[p0], a1 = t[p1];
         closest = min_raw(min_raw(closest, sd_sphere(a0, x)), sd_sphere(a1, x));
      } else if (TYPE == PK_PLANE) {
         const float4 a0 = t[p0], a1 = t[p1];
         closest = min_raw(min_raw(closest, sd_plane(a0, x)), sd_plane(a1, x));
      } else {
         const float4 a0 = t[2 * p0], b0 = t[2 * p0 + 1], a1 = t[2 * p1], b1 = t[2 * p1 + 1];
         closest = min_raw(min_raw(closest, sd_box(a0, b0, x)), sd_box(a1, b1, x));
      }
   }
   return closest;
}
// FALLBACK: the kernel variant of scenes whose Border_Behavior is Fallback (a point outside the grid scans every
// primitive, scenes.adb:943-957).  Scenes that clamp -- the reference's own -- run variants without that scan: inlined, its
// loops sat in every march loop of the partition kernels (a quarter of their code).
// PSMALL (MDH_PF_PSMALL): the kernel variant of scenes whose partition has the small form AND declares no triangles, with
// power-of-two grid spacings, fewer than 2^24 cells and the Clamp border (every scene of the reference's examples) -- the form
// and its census known when the kernel is built.  The same operations; what goes is the general form's loops over kinds (their
// code, and the registers their state held across every march loop: 80 -> 69 VGPRs and 48 -> 16 bytes of scratch in the screen
// kernel) and the tests of what a scene never changes.  A type without instances walks an empty word.
template <bool CUSTOM, bool FALLBACK, bool PSMALL = false> MDH_DEV float partitioning_closest_bits(const KScene &sc, f3 x)
{
   bool fb = false;
   const int cell = FALLBACK ? partition_cell(sc, x, fb) : partition_cell_clamped<PSMALL>(sc, x);
   if (FALLBACK && fb) return closest_primitive<CUSTOM>(sc, x);
   float closest = sc.max_dist;
   if (cell < 0 || cell >= sc.part_cells) return closest;
   MDH_DIAG_STEP(5); // lookups that reach a cell
   if (!CUSTOM && MDH_PART_SMALL && (PSMALL || sc.part_small)) { // (wave-uniform) straight-line code: one load, a shift and a mask per type
      typedef const unsigned long long __attribute__((address_space(1))) *GlobalPairs;
      const unsigned long long mm = ((GlobalPairs)(sc.part_table + sc.part_mask_off))[cell];
      // a type's candidates: 32 bits of the cell's 64 from bit part_tbit on -- one funnel shift (v_alignbit_b32) while the
      // type starts in the low dword, a plain shift of the high one otherwise (part_tbit is uniform: a scalar branch)
#if MDH_PART_ALIGNBIT
      const unsigned mlo = (unsigned)mm, mhi = (unsigned)(mm >> 32);
#define MDH_TYPE_BITS(T) ((sc.part_tbit[T] < 32u ? __builtin_amdgcn_alignbit(mhi, mlo, sc.part_tbit[T]) : (mhi >> (sc.part_tbit[T] - 32u))) & sc.part_tmask[T])
#else
#define MDH_TYPE_BITS(T) ((unsigned)(mm >> sc.part_tbit[T]) & sc.part_tmask[T])
#endif
      if (PSMALL || sc.part_tmask[PK_PLANE]) closest = walk_bits<PK_PLANE>(MDH_TYPE_BITS(PK_PLANE), s_tab + sc.tslot[PK_PLANE], x, closest);
      if (PSMALL || sc.part_tmask[PK_SPHERE]) closest = walk_bits<PK_SPHERE>(MDH_TYPE_BITS(PK_SPHERE), s_tab + sc.tslot[PK_SPHERE], x, closest);
      if (PSMALL || sc.part_tmask[PK_BOX]) closest = walk_bits<PK_BOX>(MDH_TYPE_BITS(PK_BOX), s_tab + sc.tslot[PK_BOX], x, closest);
      if (!PSMALL && sc.part_tmask[PK_TRIANGLE]) closest = walk_bits<PK_TRIANGLE>(MDH_TYPE_BITS(PK_TRIANGLE), s_tab + sc.tslot[PK_TRIANGLE], x, closest);
#undef MDH_TYPE_BITS
      return closest;
   }
   if (PSMALL) return closest; // (never reached: the variant is only launched for scenes of the small form)
   typedef const unsigned __attribute__((address_space(1))) *GlobalWords;
   const int nk = hdr(H_NK), nw = sc.part_mask_words;
   const bool in_lds = sc.part_bits_f4 > 0; // (wave-uniform: the whole grid's bits are staged behind the scene table)
   GlobalWords gwords = (GlobalWords)(sc.part_table + sc.part_mask_off) + (size_t)cell * nw;
   const unsigned *lwords = (const unsigned *)(s_tab + sc.table_f4) + cell * nw;
   auto word = [&](int dw) -> unsigned { return in_lds ? lwords[dw] : gwords[dw]; };
   // two consecutive dwords of the cell's bits at a time: a kind's bits start anywhere in the first
   int cur = 0;
   unsigned wlo = word(0), whi = nw > 1 ? word(1) : 0u;
#pragma unroll 1
   for (int k = 0; k < nk; ++k) {
      const int4 kq = hdr4(H_KQUAD + 4 * k); // type, first float4, flat base, declared count
      const int type = kq.x, s0 = kq.y, kmax = kq.w;
#pragma unroll 1
      for (int c = 0; c < kmax; c += 32) { // 32 of the kind's instances at a time, as ONE word whatever dwords they lie in
         const int bit0 = kq.z + c, dw = bit0 >> 5;
         if (dw != cur) { // (wave-uniform)
            wlo = dw == cur + 1 ? whi : word(dw);
            whi = dw + 1 < nw ? word(dw + 1) : 0u;
            cur = dw;
         }
         unsigned w = __builtin_amdgcn_alignbit(whi, wlo, (unsigned)(bit0 & 31));
         if (kmax - c < 32) w &= (1u << (kmax - c)) - 1u;
#ifdef MDH_DIAG
         { // candidates of all lanes (slot 7: lane-candidates; its wave count = kind visits)
            unsigned long long m_ = __ballot(1);
            int pc_ = __popc(w);
            for (int o_ = 32; o_ > 0; o_ >>= 1) pc_ += __shfl_xor(pc_, o_);
            if ((threadIdx.x & 63) == __ffsll((long long)m_) - 1) { atomicAdd(&g_diag[14], 1ull); atomicAdd(&g_diag[15], (unsigned long long)pc_); }
         }
#endif
         // the type is wave-uniform: one walk per type (walk_bits)
         if (CUSTOM && type == PK_CUSTOM) {
            while (w) { const int pi = c + __builtin_ctz(w); w &= w - 1u; closest = min_raw(closest, xdist<false>(k, pi, x)); }
         } else if (type == PK_SPHERE) closest = walk_bits<PK_SPHERE>(w, s_tab + s0 + c, x, closest);
         else if (type == PK_PLANE) closest = walk_bits<PK_PLANE>(w, s_tab + s0 + c, x, closest);
         else if (type == PK_BOX) closest = walk_bits<PK_BOX>(w, s_tab + s0 + 2 * c, x, closest);
         else closest = walk_bits<PK_TRIANGLE>(w, s_tab + s0 + 3 * c, x, closest);
      }
   }
   return closest;
}
// PART is a set of flags: bit 0 = the space partition is on, bit 1 = the scene has user-defined kinds,
#define MDH_PF_PART 1
#define MDH_PF_CUSTOM 2
#define MDH_PF_POW2 4 // bit 2 = the probe counts and both tile resolutions are powers of two (every atlas address is shifts and masks)
#define MDH_PF_FALLBACK 8 // bit 3 = the space partition's Border_Behavior is Fallback (built-in kinds; scenes with user-defined kinds keep the run-time test)
#define MDH_PF_PSMALL 32 // bit 5 = the space partition's small form and its census, known when the kernel is built (partitioning_closest_bits' PSMALL; with bit 0, never with bits 1 or 3)
#define MDH_PF_ROOM 16 // bit 4 = the census of the reference's rooms, known when the kernel is built (closest_primitive's ROOM; never with bits 0, 1 or 3)
// does this variant carry the full scan of the Fallback border?
#define MDH_PF_HAS_FALLBACK(PART) ((((PART) & MDH_PF_FALLBACK) != 0) || (((PART) & MDH_PF_CUSTOM) != 0))
template <int PART> MDH_DEV float sdf(const KScene &sc, f3 x)
{
   MDH_WORK(2);
   int dummy;
   (void)dummy;
   if (PART & MDH_PF_PART) return MDH_PART_BITS ? partitioning_closest_bits<(PART & MDH_PF_CUSTOM) != 0, MDH_PF_HAS_FALLBACK(PART), (PART & MDH_PF_PSMALL) != 0>(sc, x) : partitioning_lookup<false, (PART & MDH_PF_CUSTOM) != 0, MDH_PF_HAS_FALLBACK(PART), (PART & MDH_PF_PSMALL) != 0>(sc, x, dummy);
   return closest_primitive<(PART & MDH_PF_CUSTOM) != 0, (PART & MDH_PF_ROOM) != 0>(sc, x);
}
// the same with the first sphere and box already in registers (sdf_regs)
template <int PART> MDH_DEV float sdf(const KScene &sc, f3 x, const SdfRegs &regs)
{
   MDH_WORK(2);
   int dummy;
   (void)dummy;
   if (PART & MDH_PF_PART) return MDH_PART_BITS ? partitioning_closest_bits<(PART & MDH_PF_CUSTOM) != 0, MDH_PF_HAS_FALLBACK(PART), (PART & MDH_PF_PSMALL) != 0>(sc, x) : partitioning_lookup<false, (PART & MDH_PF_CUSTOM) != 0, MDH_PF_HAS_FALLBACK(PART), (PART & MDH_PF_PSMALL) != 0>(sc, x, dummy);
   return closest_primitive<(PART & MDH_PF_CUSTOM) != 0, (PART & MDH_PF_ROOM) != 0>(sc, x, &regs);
}
template <int PART> MDH_DEV float sdf_info(const KScene &sc, f3 x, int &index)
{
   MDH_WORK(2);
   if (PART & MDH_PF_PART) return partitioning_lookup<true, (PART & MDH_PF_CUSTOM) != 0, MDH_PF_HAS_FALLBACK(PART), (PART & MDH_PF_PSMALL) != 0>(sc, x, index);
   return closest_primitive_info<(PART & MDH_PF_CUSTOM) != 0>(sc, x, index);
}

// ------------------------------------------------------------------------- raymarching
// glsl/raymarching.glsl:25-37
template <int PART> MDH_DEV bool raycast(const KScene &sc, f3 from, f3 dir, int &index, f3 &coll, float &t_out, int &steps)
{
   int n = 0;
   MDH_WORK(0);
   for (float total = 0.0f; total < sc.max_dist;) {
      MDH_WORK(1);
      float dist = sdf_info<PART>(sc, from + dir * total, index);
      ++n;
      if (dist < MDH_EPS) {
         coll = from + dir * total;
         t_out = total;
         steps = n;
         return true;
      }
      total += dist;
   }
   steps = n;
   return false;
}
// glsl/raymarching.glsl:39-56: raycast_visibility = 1 - float(hit)
template <int PART> MDH_DEV float raycast_visibility(const KScene &sc, f3 from, f3 dir, float max_dist)
{
   MDH_WORK(0);
   for (float total = 0.0f; total < max_dist;) {
      MDH_WORK(1);
      float dist = sdf<PART>(sc, from + dir * total);
      if (dist < MDH_EPS) return 0.0f;
      total += dist;
   }
   return 1.0f;
}

// ------------------------------------------------------------------------------ lights
// sample_<Light> (scenes.adb:497-549) dispatched by cumulative RUNTIME counts (scenes.adb:731-764)
template <bool CUSTOM> MDH_DEV f3 sample_light(const KScene &sc, int index, f3 pos, f3 normal, f3 &dir, float &dist)
{
   const int nl = hdr(H_NL);
#pragma unroll 1
   for (int k = 0; k < nl; ++k) {
      const int n = hdr(H_LCOUNT + k);
      if (index < n) {
         if (CUSTOM && hdr(H_LTYPE + k) == LK_CUSTOM) { // the generated sample_<Light> (scenes.adb:497-549) around the kind's programs
            dir = xlight_position(k, index) - pos;
            dist = length(dir);
            dir = dir / dist;
            return xlight_sample(k, index, pos, normal, dir, dist);
         }
         if (hdr(H_LTYPE + k) == LK_POINT) { // madarch-lights-point_lights.ads:20-22
            const int s = hdr(H_LSLOT + k) + 2 * index;
            dir = xyz(s_tab[s]) - pos;
            dist = length(dir);
            dir = dir / dist;
            return sdiv3(xyz(s_tab[s + 1]), (dist * dist) * 0.03f);
         }
         const int s = hdr(H_LSLOT + k) + 3 * index; // madarch-lights-spot_lights.adb:5-24
         float4 a = s_tab[s];
         dir = xyz(a) - pos;
         dist = length(dir);
         dir = dir / dist;
         float attenuation = sdiv(1.0f, (dist * dist) * 0.03f);
         float theta = acos_(max_(dot(-dir, xyz(s_tab[s + 1])), 0.0f));
         float ratio = clamp_(sdiv(theta, a.w), 0.0f, 1.0f);
         float visible = 1.0f - pow8_(ratio);
         return (xyz(s_tab[s + 2]) * min_(attenuation, 1.5f)) * visible;
      }
      index -= n;
   }
   dir = F3(0.0f, 0.0f, 0.0f);
   dist = 0.0f;
   return F3(0.0f, 0.0f, 0.0f);
}

// -------------------------------------------------------------------------------- BRDF
struct Material { f3 albedo; float metallic, roughness; };
MDH_DEV Material get_material(const KScene &sc, int id) // glsl/materials.glsl:1-10
{
   float4 a = s_tab[sc.mat_slot + 2 * id], b = s_tab[sc.mat_slot + 2 * id + 1];
   Material m;
   m.albedo = xyz(a); m.metallic = a.w; m.roughness = b.x;
   return m;
}
// glsl/cook_torrance_brdf.glsl:1-52
MDH_DEV void cook_torrance(f3 N, f3 V, f3 L, float NdotL, f3 albedo, float metallic, float roughness, f3 &kD, f3 &kS)
{
   MDH_SHADING_FP
   f3 H = snormalize(V + L);
   float NdotV = max_(dot(N, V), 0.0f);
   f3 F0 = F3(mix_(0.04f, albedo.x, metallic), mix_(0.04f, albedo.y, metallic), mix_(0.04f, albedo.z, metallic));
   float a = roughness * roughness;
   float a2 = a * a;
   float NdotH = max_(dot(N, H), 0.0f);
   float NdotH2 = NdotH * NdotH;
   float denom = NdotH2 * (a2 - 1.0f) + 1.0f;
   denom = MDH_PI * denom * denom;
   float NDF = sdiv(a2, denom);
   float rr = roughness + 1.0f;
   float kk = (rr * rr) / 8.0f; // (a division by a power of two: exact either way)
   float ggx2 = sdiv(NdotV, NdotV * (1.0f - kk) + kk);
   float ggx1 = sdiv(NdotL, NdotL * (1.0f - kk) + kk);
   float G = ggx1 * ggx2;
   float p5 = pow5_(1.001f - max_(dot(H, V), 0.0f));
   f3 F = F3(F0.x + (1.0f - F0.x) * p5, F0.y + (1.0f - F0.y) * p5, F0.z + (1.0f - F0.z) * p5);
   f3 numerator = F * (NDF * G);
   float denominator = 4.0f * NdotV * NdotL;
   float dm = max_(denominator, 0.001f);
   kD = F3(1.0f - F.x, 1.0f - F.y, 1.0f - F.z) * (1.0f - metallic);
   kS = min3s(sdiv3(numerator, dm), 1.0f);
}
// glsl/lighting.glsl:42-49
MDH_DEV f3 compute_indirect_lighting(f3 irradiance, f3 radiance, f3 V, f3 N, f3 L, f3 albedo, float metallic, float roughness)
{
   f3 kD, kS;
   float NdotL = max_(dot(N, L), 0.0f);
   cook_torrance(N, V, L, NdotL, albedo, metallic, roughness, kD, kS);
   return sdiv3(kD * irradiance, MDH_PI) + (kS * radiance) * NdotL;
}
// ------------------------------------------------------------------------- probe utils
// glsl/probe_utils.glsl:19-56
MDH_DEV i3 probe_id_to_grid(const KProbes &pr, int id)
{
   int xy = pr.gx * pr.gy;
   i3 g;
   g.z = id / xy;
   g.y = (id - g.z * xy) / pr.gx;
   g.x = id - g.z * xy - g.y * pr.gx;
   return g;
}
MDH_DEV f3 grid_to_world(const KProbes &pr, i3 g) { return F3((float)g.x * pr.sx, (float)g.y * pr.sy, (float)g.z * pr.sz); }
MDH_DEV i3 world_to_grid(const KProbes &pr, f3 p)
{
   f3 f = floor3(p / F3(pr.sx, pr.sy, pr.sz));
   i3 g;
   g.x = (int)f.x; g.y = (int)f.y; g.z = (int)f.z;
   return g;
}
// (P2: the kernel variants for power-of-two atlases run on fewer than 65 536 probes -- run_pass -- where the products fit
// the 24-bit multiplier, which issues at the full rate)
template <bool P2 = false> MDH_DEV int grid_to_probe_id(const KProbes &pr, i3 g)
{
   if (P2) return __mul24(g.z, pr.gx * pr.gy) + __mul24(g.y, pr.gx) + g.x;
   return g.z * pr.gx * pr.gy + g.y * pr.gx + g.x;
}
// x / probe_count: a multiplication when the count is a power of two (exact: the same real value, rounded once)
template <bool P2 = false> MDH_DEV float div_pcx(const KProbes &pr, float x) { return (P2 || pr.inv_pcx != 0.0f) ? x * pr.inv_pcx : x / pr.fpcx; }
template <bool P2 = false> MDH_DEV float div_pcy(const KProbes &pr, float y) { return (P2 || pr.inv_pcy != 0.0f) ? y * pr.inv_pcy : y / pr.fpcy; }
template <bool P2 = false> MDH_DEV f2 probe_id_to_coord(const KProbes &pr, int id)
{
   int y = (P2 || pr.pcx_shift >= 0) ? (id >> pr.pcx_shift) : div_magic(id, pr.pcx, pr.m_pcx), x = P2 ? (id & (pr.pcx - 1)) : id - y * pr.pcx;
   return F2(div_pcx<P2>(pr, (float)x), div_pcy<P2>(pr, (float)y));
}
// glsl/probe_utils.glsl:58-92
MDH_DEV float sign_not_zero(float v) { return v >= 0.0f ? 1.0f : -1.0f; }
MDH_DEV f2 float32x3_to_oct(f3 v)
{
   float s = sdiv(1.0f, (__builtin_fabsf(v.x) + __builtin_fabsf(v.y)) + __builtin_fabsf(v.z));
   f2 p = F2(v.x * s, v.y * s);
   if (v.z <= 0.0f) return F2((1.0f - __builtin_fabsf(p.y)) * sign_not_zero(p.x), (1.0f - __builtin_fabsf(p.x)) * sign_not_zero(p.y));
   return p;
}
MDH_DEV f3 oct_to_float32x3(f2 e)
{
   f3 v = F3(e.x, e.y, (1.0f - __builtin_fabsf(e.x)) - __builtin_fabsf(e.y));
   if (v.z < 0.0f) {
      float nx = (1.0f - __builtin_fabsf(v.y)) * sign_not_zero(v.x);
      float ny = (1.0f - __builtin_fabsf(v.x)) * sign_not_zero(v.y);
      v.x = nx;
      v.y = ny;
   }
   return normalize(v);
}
MDH_DEV f3 ray_id_to_ray_dir(f2 id) { return oct_to_float32x3(F2(id.x * 2.0f - 1.0f, id.y * 2.0f - 1.0f)); }
MDH_DEV f2 ray_dir_to_ray_id(f3 d)
{
   f2 raw = float32x3_to_oct(d);
   return F2((raw.x + 1.0f) * 0.5f, (raw.y + 1.0f) * 0.5f);
}

// -------------------------------------------------------------------------- textures
// GL_MIRRORED_REPEAT (support/render_passes.adb:111-112).  P2: n is a power of two (the modulo is a mask; with a
// runtime divisor the compiler hoists the reciprocal set-up of every distinct divisor into VGPRs that stay live through
// the whole kernel -- seven of them in the screen pass, four of which ended up in scratch)
template <bool P2 = false> MDH_DEV int mirror(int i, int n)
{
   if (P2) { // (three instructions, inside the image or not: the period is a mask, the reflection a minimum)
      const int m = i & (2 * n - 1);
      return min(m, 2 * n - 1 - m);
   }
   if ((unsigned)i < (unsigned)n) return i; // inside the image: the usual case
   // one reflection covers [-n, 2n): every tap of this library (coordinates clamped inside a tile, or a texel
   // beside the image) -- the same value as the modulo form below
   if ((unsigned)(i + n) < (unsigned)(3 * n)) return i < 0 ? -1 - i : 2 * n - 1 - i;
   // farther out: the general form.  The divisor goes through an opaque per-lane register HERE so that the set-up of
   // the division stays in this cold path (with a uniform divisor the compiler hoists a reciprocal per distinct
   // image size into VGPRs that live -- or spill -- through the whole pixel program).
   int n2 = 2 * n;
   asm volatile("" : "+v"(n2));
   int m = i % n2;
   if (m < 0) m += n2;
   return m >= n ? 2 * n - 1 - m : m;
}
MDH_DEV float unorm8(float x) { return (x != x) ? 0.0f : __builtin_rintf(clamp_(x, 0.0f, 1.0f) * 255.0f); }

// Probe atlases are stored probe-major -- [probe][y][x], RGBA8 or float4 texels -- so a
// rank's probe slice is one contiguous range (DESIGN.md "HBM layout").  (X, Y) are texel
// coordinates of the reference's 2-D atlas image, X = tile_x * res + x.
// `shift` = log2(res) when res is a power of two (wave-uniform fast path), else -1
// `magic`: div_magic's number for res (KProbes::m_rres / m_ires)
template <bool P2 = false> MDH_DEV unsigned atlas_index(int pcx, int res, int shift, int X, int Y, unsigned magic = 0u)
{
   int tx, ty;
   if (P2 || shift >= 0) { tx = X >> shift; ty = Y >> shift; }
   else { tx = div_magic(X, res, magic); ty = div_magic(Y, res, magic); }
   return ((unsigned)(ty * pcx + tx) * res + (Y - ty * res)) * res + (X - tx * res);
}
// The same index as the sum of a term of the row and a term of the column -- ((ty pcx + tx) res + ry) res + rx =
// (ty pcx res^2 + ry res) + (tx res^2 + rx), exact in the ring of 32-bit integers -- so that the four texels of a
// bilinear tap share two of each; with powers of two (P2) every product is a shift of masked bits.  (32-bit integer
// products issue at a quarter of the rate of the other integer operations: the tap's addressing was the largest single
// item of the screen pass's probe code.)
template <bool P2> MDH_DEV unsigned atlas_col(int res, int shift, int X, unsigned magic)
{
   if (P2) return ((unsigned)(X & ~(res - 1)) << shift) + (unsigned)(X & (res - 1));
   const int tx = shift >= 0 ? X >> shift : div_magic(X, res, magic);
   return (unsigned)(tx * res) * (unsigned)res + (unsigned)(X - tx * res);
}
template <bool P2> MDH_DEV unsigned atlas_row(int pcx, int res, int shift, int Y, unsigned magic)
{
   if (P2) return ((unsigned)(Y & ~(res - 1)) << (shift + (31 - __builtin_clz(pcx)))) + ((unsigned)(Y & (res - 1)) << shift);
   const int ty = shift >= 0 ? Y >> shift : div_magic(Y, res, magic);
   return ((unsigned)(ty * pcx) * (unsigned)res + (unsigned)(Y - ty * res)) * (unsigned)res;
}
// Atlas reads are GLOBAL loads: through the generic pointer of the argument block they are flat loads, which count as
// LDS operations too -- every wait for an LDS read (each step of a march) then also waits for the texels in flight.
// P2: the byte offset of an RGBA8 texel fits 32 bits (run_pass starts these variants on atlases below 4 GiB only), the
// load takes the base from scalar registers and needs no 64-bit address arithmetic.
typedef const unsigned __attribute__((address_space(1))) *GlobalU32;
typedef const char __attribute__((address_space(1))) *GlobalBytes;
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const f4v __attribute__((address_space(1))) *GlobalF4;
template <bool P2 = false> MDH_DEV unsigned atlas_rgba8(const void *base, unsigned idx)
{
   if (P2) return *(GlobalU32)((GlobalBytes)base + (idx << 2));
   return ((GlobalU32)base)[idx];
}
// k / 255 for a byte k, correctly rounded, without a division or a table: 1/255 = c_hi + c_lo with c_hi its nearest
// float, and fma (k, c_hi, fl (k c_lo)) is the correctly rounded quotient for every k in 0 .. 255 (checked in exact
// arithmetic for all 256 values: tests/test_oracle_pins.py).  Three instructions (v_cvt_f32_ubyteN takes the byte out
// of the texel itself); the table costs a bit-field extract, an address and an LDS read per component.
#ifndef MDH_U8_FMA
#define MDH_U8_FMA 1
#endif
MDH_DEV float u8_unorm(float k) { return __builtin_fmaf(k, 0x1.010102p-8f, k * -0x1.fdfdfep-33f); }
// u8_tab = float index of the k / 255 table in LDS (KScene::u8_slot * 4), or < 0: divide
template <bool P2 = false> MDH_DEV f3 atlas_texel(const void *base, int fmt, unsigned idx, int u8_tab)
{
   if (fmt == 0) {
      const unsigned t = atlas_rgba8<P2>(base, idx);
      const unsigned x = t & 255u, y = (t >> 8) & 255u, z = (t >> 16) & 255u;
      if (MDH_U8_FMA) return F3(u8_unorm((float)x), u8_unorm((float)y), u8_unorm((float)z));
      if (u8_tab >= 0) return F3(tab_float(u8_tab + x), tab_float(u8_tab + y), tab_float(u8_tab + z));
      return F3((float)x / 255.0f, (float)y / 255.0f, (float)z / 255.0f);
   }
   const f4v t = ((GlobalF4)base)[idx];
   return F3(t.x, t.y, t.z);
}
MDH_DEV void atlas_store(void *base, int fmt, unsigned idx, f3 v)
{
   if (fmt == 0) {
      uchar4 t;
      t.x = (unsigned char)unorm8(v.x); t.y = (unsigned char)unorm8(v.y); t.z = (unsigned char)unorm8(v.z); t.w = 255;
      ((uchar4 *)base)[idx] = t;
   } else {
      float4 t;
      t.x = (v.x != v.x) ? 0.0f : v.x; t.y = (v.y != v.y) ? 0.0f : v.y; t.z = (v.z != v.z) ? 0.0f : v.z; t.w = 1.0f;
      ((float4 *)base)[idx] = t;
   }
}
// GL_LINEAR on the atlas image of pcx*res x pcy*res texels (render_passes.adb:113-114); Wf, Hf = (float)(pcx * res), (float)(pcy * res)
template <bool P2 = false>
MDH_DEV f3 atlas_sample(const void *base, int fmt, int pcx, int pcy, int res, int shift, float Wf, float Hf, float cx, float cy, int u8_tab, unsigned magic = 0u)
{
   const int W = pcx * res, H = pcy * res;
   float px = cx * Wf - 0.5f, py = cy * Hf - 0.5f;
   float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py);
   float fx = px - fx0, fy = py - fy0;
   int x0 = mirror<P2>((int)fx0, W), x1 = mirror<P2>((int)fx0 + 1, W), y0 = mirror<P2>((int)fy0, H), y1 = mirror<P2>((int)fy0 + 1, H);
   float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
   const unsigned c0 = atlas_col<P2>(res, shift, x0, magic), c1 = atlas_col<P2>(res, shift, x1, magic);
   const unsigned r0 = atlas_row<P2>(pcx, res, shift, y0, magic), r1 = atlas_row<P2>(pcx, res, shift, y1, magic);
   f3 a = atlas_texel<P2>(base, fmt, r0 + c0, u8_tab), b = atlas_texel<P2>(base, fmt, r0 + c1, u8_tab);
   f3 c = atlas_texel<P2>(base, fmt, r1 + c0, u8_tab), d = atlas_texel<P2>(base, fmt, r1 + c1, u8_tab);
   return ((a * w00 + b * w10) + c * w01) + d * w11;
}

// MDH_OPT_RADIANCE_MIPS: textureLod (radiance_data, coord, lod) as GL_LINEAR_MIPMAP_LINEAR reads a mip chain -- lod clamped to
// the chain, the two nearest levels sampled bilinearly and mixed by the fraction (the upper one only when the fraction is
// not 0).  Level l >= 1 has the atlas's probe-major layout with tiles of res >> l texels; the levels follow one another
// in pq.rad_mips (k_radiance_mips).  Only the screen kernel's variant for the optional paths calls this.
MDH_DEV f3 radiance_level_sample(const KProbes &pq, int l, float cx, float cy, int u8_tab)
{
   if (l == 0) return atlas_sample<false>(pq.rad, pq.fmt, pq.pcx, pq.pcy, pq.rres, pq.rshift, pq.rad_w, pq.rad_h, cx, cy, u8_tab, pq.m_rres);
   unsigned off = 0; // texels of levels 1 .. l - 1
   for (int k = 1; k < l; ++k) { const int rk = pq.rres >> k; off += (unsigned)(pq.pcx * pq.pcy) * (unsigned)(rk * rk); }
   const int res = pq.rres >> l;
   const char *base = (const char *)pq.rad_mips + (size_t)off * (pq.fmt == 0 ? 4u : 16u);
   return atlas_sample<false>(base, pq.fmt, pq.pcx, pq.pcy, res, pq.rshift - l, (float)(pq.pcx * res), (float)(pq.pcy * res), cx, cy, u8_tab, 0u);
}
MDH_DEV f3 radiance_lod_sample(const KProbes &pq, float lod, float cx, float cy, int u8_tab)
{
   const float d = clamp_(lod, 0.0f, (float)pq.rad_lods);
   const int l0 = (int)d;
   const float f = d - (float)l0;
   const f3 lo = radiance_level_sample(pq, l0, cx, cy, u8_tab);
   if (!(f > 0.0f)) return lo;
   const f3 hi = radiance_level_sample(pq, l0 + 1, cx, cy, u8_tab);
   return lo * (1.0f - f) + hi * f; // mix ()
}

// The same tap in two halves, so that the four texel loads of an RGBA8 atlas can be issued long
// before their values are needed (L2 latency hidden behind a march loop): atlas_tap_issue does
// the addressing and the loads, atlas_tap_resolve the u8 -> float conversion and the blend.
// float4 atlases load at resolve time.
struct AtlasTap {
   unsigned t00, t10, t01, t11; // RGBA8 texels; float4 atlases: the four texel indices
   float fx, fy;
};
template <bool P2 = false>
MDH_DEV AtlasTap atlas_tap_issue(const void *base, int fmt, int pcx, int pcy, int res, int shift, float Wf, float Hf, float cx, float cy, unsigned magic = 0u)
{
   const int W = pcx * res, H = pcy * res;
   float px = cx * Wf - 0.5f, py = cy * Hf - 0.5f;
   float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py);
   AtlasTap t;
   t.fx = px - fx0; t.fy = py - fy0;
   int x0 = mirror<P2>((int)fx0, W), x1 = mirror<P2>((int)fx0 + 1, W), y0 = mirror<P2>((int)fy0, H), y1 = mirror<P2>((int)fy0 + 1, H);
   const unsigned c0 = atlas_col<P2>(res, shift, x0, magic), c1 = atlas_col<P2>(res, shift, x1, magic);
   const unsigned r0 = atlas_row<P2>(pcx, res, shift, y0, magic), r1 = atlas_row<P2>(pcx, res, shift, y1, magic);
   t.t00 = r0 + c0; t.t10 = r0 + c1; t.t01 = r1 + c0; t.t11 = r1 + c1;
   if (fmt == 0) { t.t00 = atlas_rgba8<P2>(base, t.t00); t.t10 = atlas_rgba8<P2>(base, t.t10); t.t01 = atlas_rgba8<P2>(base, t.t01); t.t11 = atlas_rgba8<P2>(base, t.t11); }
   return t;
}
MDH_DEV f3 u8_texel(unsigned t, int u8_tab)
{
#if MDH_U8_FMA
   return F3(u8_unorm((float)(t & 255u)), u8_unorm((float)((t >> 8) & 255u)), u8_unorm((float)((t >> 16) & 255u)));
#else
   return F3(tab_float(u8_tab + (t & 255u)), tab_float(u8_tab + ((t >> 8) & 255u)), tab_float(u8_tab + ((t >> 16) & 255u)));
#endif
}
MDH_DEV f3 atlas_tap_resolve(const void *base, int fmt, const AtlasTap &t, int u8_tab)
{
   const float fx = t.fx, fy = t.fy;
   float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
   f3 a, b, c, d;
   if (fmt == 0) { a = u8_texel(t.t00, u8_tab); b = u8_texel(t.t10, u8_tab); c = u8_texel(t.t01, u8_tab); d = u8_texel(t.t11, u8_tab); }
   else { a = atlas_texel(base, fmt, t.t00, u8_tab); b = atlas_texel(base, fmt, t.t10, u8_tab); c = atlas_texel(base, fmt, t.t01, u8_tab); d = atlas_texel(base, fmt, t.t11, u8_tab); }
   return ((a * w00 + b * w10) + c * w01) + d * w11;
}

// ------------------------------------------------------------------------ volumetrics
#define MDH_TAU 0.1f // glsl/volumetrics.glsl:12
// glsl/volumetrics.glsl:21-30
MDH_DEV float henvey_greenstein_phase(f3 in_dir, f3 out_dir)
{
   float cos_angle = dot(in_dir, out_dir);
   float t2 = MDH_TAU * MDH_TAU;
   float result = 1.0f - t2;
   const float base = 1.0f + t2 - 2.0f * MDH_TAU * cos_angle;
   result = sdiv(result, 4.0f * MDH_PI * (MDH_HYBRID_NUMERICS ? base * ssqrt(base) : pow1_5_(base)));
   return result;
}
// bilinear tap of a plain row-major texture with C floats per texel
template <int C> MDH_DEV void tex_sample(const float *data, int W, int H, float cx, float cy, float *out)
{
   float px = cx * (float)W - 0.5f, py = cy * (float)H - 0.5f;
   float fx0 = __builtin_floorf(px), fy0 = __builtin_floorf(py);
   float fx = px - fx0, fy = py - fy0;
   int x0 = mirror<false>((int)fx0, W), x1 = mirror<false>((int)fx0 + 1, W), y0 = mirror<false>((int)fy0, H), y1 = mirror<false>((int)fy0 + 1, H);
   float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
   typedef const float __attribute__((address_space(1))) *GlobalF32; // (global loads, not flat ones: see atlas_rgba8)
   const GlobalF32 g = (GlobalF32)data;
   const unsigned r0 = (unsigned)y0 * (unsigned)W, r1 = (unsigned)y1 * (unsigned)W; // (two products for the four texels; fewer than 2^32 texels: mdh_create)
   const GlobalF32 a = g + (size_t)(r0 + (unsigned)x0) * C, b = g + (size_t)(r0 + (unsigned)x1) * C;
   const GlobalF32 c = g + (size_t)(r1 + (unsigned)x0) * C, d = g + (size_t)(r1 + (unsigned)x1) * C;
#pragma unroll
   for (int k = 0; k < C; ++k) out[k] = ((a[k] * w00 + b[k] * w10) + c[k] * w01) + d[k] * w11;
}
// glsl/volumetrics.glsl:34-54; on a miss len = max_dist
MDH_DEV f3 render_volumetrics(const KScene &sc, const KVolumetrics &vol, f3 L, f3 from, f3 to, bool hit, f2 frag_pos)
{
   f2 tc = F2((frag_pos.x + 1.0f) * 0.5f, (frag_pos.y + 1.0f) * 0.5f);
   float len = hit ? length(to - from) : sc.max_dist;
   float closest = sc.max_dist;
   f3 fog = F3(0.0f, 0.0f, 0.0f);
   const float sx = 1.0f / (float)vol.sw, sy = 1.0f / (float)vol.sh;
   for (int x = -1; x <= 1; ++x)
      for (int y = -1; y <= 1; ++y) {
         float d[4];
         tex_sample<4>((const float *)vol.scat, vol.sw, vol.sh, tc.x + (float)x * sx, tc.y + (float)y * sy, d);
         float dist = __builtin_fabsf(d[3] - len);
         if (dist < closest) { closest = dist; fog = F3(d[0], d[1], d[2]); }
      }
   return L * sexp(-len * MDH_TAU) + fog;
}

struct PrimaryHit { int index; float t; int steps; };

// camera ray (glsl/draw_screen.glsl:21-24)
MDH_DEV f3 mat_mul(const float *m, f3 v)
{
   return F3((m[0] * v.x + m[3] * v.y) + m[6] * v.z, (m[1] * v.x + m[4] * v.y) + m[7] * v.z, (m[2] * v.x + m[5] * v.y) + m[8] * v.z);
}
MDH_DEV void camera_ray(const KCamera &cam, float u, float v, f3 &origin, f3 &dir)
{
   f3 frag = F3(u, v, 0.0f);
   f3 d = normalize(frag - F3(0.0f, 0.0f, -1.5f));
   dir = mat_mul(cam.m, d);
   origin = mat_mul(cam.m, frag) + F3(cam.px, cam.py, cam.pz);
}
// texel / pixel centre in [-1, 1]
MDH_DEV float centre(int i, int n) { return (float)(2 * i + 1) / (float)n - 1.0f; }
