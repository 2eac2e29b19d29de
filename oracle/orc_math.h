/* orc_math.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * fp32 scalar/vec3 arithmetic with the semantics the oracle fixes for the GLSL
 * built-ins the reference's shaders use and for Math_Utils
 * (/root/reference/madarch/support/math_utils.ads:12-93).  PARITY UNPINNED: the
 * reference holds no numeric fixture for any of these;
 * where GLSL leaves precision or operation order to the driver the choice made
 * here is written next to the function and is the one the HIP kernels follow.
 *
 * Build with -ffp-contract=off and without -ffast-math: every operation below
 * is one correctly rounded IEEE binary32 operation, in the order written.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>
#include <stdint.h>

typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;
typedef struct { int x, y, z; } iv3;

#define ORC_INLINE static inline __attribute__((always_inline))

/* maths.glsl:1-3 */
#define ORC_PI 3.14159265358f
#define ORC_EPSILON 0.001f
/* raymarching.glsl:1 */
#define ORC_MIN_STEP 0.05f

ORC_INLINE v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
ORC_INLINE v3 v3s(float s) { return V3(s, s, s); }
ORC_INLINE v2 V2(float x, float y) { v2 r = {x, y}; return r; }

/* GLSL min/max (GLSL 4.30 8.3: y < x ? y : x, undefined for NaN).  The oracle
 * fixes the NaN case as IEEE 754-2008 minNum/maxNum -- a NaN operand is
 * ignored -- which is what the gfx950 v_min_f32/v_max_f32 the kernels use do. */
ORC_INLINE float fmin_(float a, float b) { return (a != a) ? b : (b != b) ? a : (b < a) ? b : a; }
ORC_INLINE float fmax_(float a, float b) { return (a != a) ? b : (b != b) ? a : (a < b) ? b : a; }
ORC_INLINE float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
ORC_INLINE int imin_(int a, int b) { return b < a ? b : a; }
ORC_INLINE int imax_(int a, int b) { return a < b ? b : a; }
ORC_INLINE int iclamp_(int x, int lo, int hi) { return imin_(imax_(x, lo), hi); }
/* sign(): -1, 0, +1 (GLSL and Math_Utils.Sign, math_utils.ads:12-16) */
ORC_INLINE float sign_(float x) { return x < 0.0f ? -1.0f : (x > 0.0f ? 1.0f : 0.0f); }
ORC_INLINE float fract_(float x) { return x - floorf(x); }
/* mix(x,y,a) = x*(1-a) + y*a (GLSL 4.30 8.3) */
ORC_INLINE float mix_(float x, float y, float a) { return x * (1.0f - a) + y * a; }

ORC_INLINE v3 add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
ORC_INLINE v3 sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
ORC_INLINE v3 mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
ORC_INLINE v3 vdiv(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
ORC_INLINE v3 scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
ORC_INLINE v3 divs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
ORC_INLINE v3 adds(v3 a, float s) { return V3(a.x + s, a.y + s, a.z + s); }
ORC_INLINE v3 neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
ORC_INLINE v3 vabs(v3 a) { return V3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
ORC_INLINE v3 vmaxs(v3 a, float s) { return V3(fmax_(a.x, s), fmax_(a.y, s), fmax_(a.z, s)); }
ORC_INLINE v3 vmins(v3 a, float s) { return V3(fmin_(a.x, s), fmin_(a.y, s), fmin_(a.z, s)); }
ORC_INLINE v3 vfloor(v3 a) { return V3(floorf(a.x), floorf(a.y), floorf(a.z)); }
ORC_INLINE v3 vsqrt(v3 a) { return V3(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)); }

/* dot: left-to-right sum of the three products, as Singles.Dot_Product is
 * taken to be (OpenGLAda is not vendored; math_utils.ads:55-56) */
ORC_INLINE float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
ORC_INLINE float dot2(v3 a) { return dot(a, a); } /* maths.glsl:5-7 */
/* length = sqrt(dot2) (math_utils.ads:77-79) */
ORC_INLINE float length(v3 a) { return sqrtf(dot2(a)); }
/* normalize = v / length(v), a true division per component (math_utils.ads:81-83) */
ORC_INLINE v3 normalize(v3 a) { return divs(a, length(a)); }
ORC_INLINE v3 cross(v3 a, v3 b)
{
   return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* reflect(I,N) = I - 2 dot(N,I) N (GLSL 4.30 8.5) */
ORC_INLINE v3 reflect(v3 i, v3 n) { return sub(i, scale(n, 2.0f * dot(n, i))); }

ORC_INLINE float f_of_bits(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }
ORC_INLINE uint32_t bits_of_f(float f) { union { uint32_t u; float f; } c; c.f = f; return c.u; }

/* Transcendentals.  GLSL gives acos / exp / pow implementation-defined precision and the
 * hardware approximations of the GPU (v_exp_f32, v_log_f32) have no bit-identical CPU
 * counterpart, so the oracle FIXES them as the explicit fp32 algorithms below (plain mul/add
 * in the order written, no fma); the HIP kernels run the same operation sequences.  Accuracy
 * against libm (tests/test_oracle_pins.py): acos abs error < 5e-7, exp and x^0.4545 within 1.2e-6
 * relative -- two orders below the 1e-4 parity tolerance.  Small literal integer
 * powers are repeated multiplication. */

/* acos(x) = sqrt(1 - |x|) * P7(|x|), Abramowitz & Stegun 4.4.46 (|error| <= 2e-8), mirrored
 * for x < 0 */
ORC_INLINE float acos_(float x)
{
   float ax = fabsf(x);
   float p = -0.0012624911f;
   p = p * ax + 0.0066700901f;
   p = p * ax + -0.0170881256f;
   p = p * ax + 0.0308918810f;
   p = p * ax + -0.0501743046f;
   p = p * ax + 0.0889789874f;
   p = p * ax + -0.2145988016f;
   p = p * ax + 1.5707963050f;
   float r = sqrtf(1.0f - ax) * p;
   return x < 0.0f ? 3.14159265358979f - r : r;
}
/* asin = pi/2 - acos (same polynomial; absolute error < 2e-7) */
ORC_INLINE float asin_(float x) { return 1.5707963267948966f - acos_(x); }
/* sin / cos: k = rint (x 2/pi), r = ((x - k C1) - k C2) - k C3 with C1 + C2 + C3 = pi/2 split so that the
 * first two products are exact for |k| < 2^13 (Cody-Waite), then the degree-7 / degree-8 polynomials of
 * the Cephes single-precision library on [-pi/4, pi/4] and the quadrant symmetries.  |x| up to ~1e4
 * keeps the error within a few 1e-7; beyond that the reduction loses accuracy (the result stays bounded). */
ORC_INLINE void sincos_core_(float x, float *s, float *c, int *q)
{
   float kf = rintf(x * 0.636619772367581f);
   if (!(fabsf(kf) < 1.0e9f)) kf = 0.0f; /* huge or NaN arguments: no reduction (NaN propagates below) */
   float r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
   float z = r * r;
   float ps = -1.9515295891e-4f;
   ps = ps * z + 8.3321608736e-3f;
   ps = ps * z + -1.6666654611e-1f;
   *s = (ps * z) * r + r;
   float pc = 2.443315711809948e-5f;
   pc = pc * z + -1.388731625493765e-3f;
   pc = pc * z + 4.166664568298827e-2f;
   *c = ((pc * z) * z - 0.5f * z) + 1.0f;
   *q = (int)kf & 3;
}
ORC_INLINE float sin_(float x)
{
   float s, c; int q;
   sincos_core_(x, &s, &c, &q);
   float r = (q & 1) ? c : s;
   return (q & 2) ? -r : r;
}
ORC_INLINE float cos_(float x)
{
   float s, c; int q;
   sincos_core_(x, &s, &c, &q);
   float r = (q & 1) ? s : c;
   return ((q + 1) & 2) ? -r : r;
}
ORC_INLINE float tan_(float x) { return sin_(x) / cos_(x); }
/* atan: Cephes atanf -- reduce to |t| <= tan (pi/8) by atan x = pi/2 - atan (1/x) and atan x = pi/4 +
 * atan ((x - 1) / (x + 1)), then an odd degree-9 polynomial */
ORC_INLINE float atan_(float x)
{
   float ax = fabsf(x), y = 0.0f, t = ax;
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
/* 2^z: n = rint(z), 2^(z - n) = exp(u) with u = (z - n) ln2 in [-0.347, 0.347] by its Taylor
 * polynomial of degree 7 (truncation < 6e-9), scaled by 2^n through the exponent field */
ORC_INLINE float exp2_(float z)
{
   if (z != z) return z;
   if (z > 128.0f) return INFINITY;
   if (z < -126.0f) return 0.0f; /* results below the normal range are flushed */
   float n = rintf(z);
   float u = (z - n) * 0.693147182464599609375f;
   float p = 1.0f / 5040.0f;
   p = p * u + 1.0f / 720.0f;
   p = p * u + 1.0f / 120.0f;
   p = p * u + 1.0f / 24.0f;
   p = p * u + 1.0f / 6.0f;
   p = p * u + 0.5f;
   p = p * u + 1.0f;
   p = p * u + 1.0f;
   return p * f_of_bits((uint32_t)((int32_t)n + 127) << 23);
}
/* log2(x), x > 0: x = m 2^e with m in [sqrt(1/2), sqrt(2)); ln m = 2 atanh(s), s = (m-1)/(m+1),
 * by the odd series up to s^9 (|s| <= 0.172, truncation < 2e-9 relative) */
ORC_INLINE float log2_(float x)
{
   int32_t e = 0;
   if (x < 1.17549435e-38f) { x = x * 16777216.0f; e = -24; }
   uint32_t b = bits_of_f(x);
   e += (int32_t)((b >> 23) & 255u) - 127;
   float m = f_of_bits((b & 0x007fffffu) | 0x3f800000u);
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
ORC_INLINE float exp_(float x) { return exp2_(x * 1.44269502162933349609375f); }
/* pow(x, y) for the tonemap (y = 0.4545, draw_screen.glsl:29): 2^(y log2 x); x < 0 is NaN as in
 * GLSL, 0 gives 0 */
ORC_INLINE float pow_(float x, float y)
{
   if (x != x || x < 0.0f) return NAN;
   if (x == 0.0f) return 0.0f;
   if (x > 3.40282347e+38f) return x;
   return exp2_(y * log2_(x));
}
/* pow(x, 5.0) of fresnel_schlick (cook_torrance_brdf.glsl:2) */
ORC_INLINE float pow5_(float x) { float x2 = x * x; return (x2 * x2) * x; }
/* pow(x, 8.0) of the spot light (madarch-lights-spot_lights.adb:18) */
ORC_INLINE float pow8_(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x4; }
/* pow(x, 1.5) of henvey_greenstein_phase (volumetrics.glsl:25-28) */
ORC_INLINE float pow1_5_(float x) { return x * sqrtf(x); }

#endif
