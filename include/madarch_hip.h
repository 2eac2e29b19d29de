/* madarch_hip.h -- C ABI of libmadarch_hip.so, the MI355X (gfx950) back end
 * that replaces the OpenGL/GLSL dispatch of Madarch.Renderers.
 *
 * Every entry point below stands for one public operation of the reference
 * package `Madarch.Renderers` (madarch/madarch-renderers.ads:21-97) or for an
 * output of `Madarch.Scenes.Compile` (madarch/madarch-scenes.ads:47-76); the
 * reference file:line each one replaces is cited next to it.  An Ada body of
 * Madarch.Renderers binds these with `pragma Import (C, ...)` (see
 * INTEGRATION.md and ada/).
 *
 * Conventions (mirroring the reference):
 *  - plain pointers and sizes only; the library copies everything on call and
 *    keeps no caller pointer;
 *  - every function returns an int32 status, 0 = ok; the text of the last
 *    error of the calling thread is returned by mdh_last_error();
 *  - a handle is NOT thread safe (the reference is single threaded and calls
 *    Make_Current on every entry, madarch-renderers.adb:170,304);
 *  - primitives and lights are 1-based (Ada `Positive`), materials 0-based
 *    (`Materials.Id`, madarch-renderers.adb:349-367);
 *  - GL.Types.Single = float, GL.Types.Int = int32_t, Singles.Vector3 = 3
 *    contiguous floats, Singles.Matrix3 = 9 floats passed COLUMN-major;
 *  - entity blobs use the std140 element layout the reference computes in
 *    GPU_Types (support/gpu_types-structs.adb:23-38): vec3 aligned to 16,
 *    float/int aligned to 4, fields in declared component order.
 */
#ifndef MADARCH_HIP_H
#define MADARCH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdh_renderer mdh_renderer; /* opaque; = access Renderer_Internal (renderers.ads:109-145) */

/* status codes; the Ada body raises Program_Error / Constraint_Error from them */
enum {
   MDH_OK = 0,
   MDH_E_INVALID = 1,          /* bad argument (null pointer, negative size, ...)            */
   MDH_E_PROBE_MISMATCH = 2,   /* Program_Error "Probe_Count should match grid dimensions."
                                  (madarch-renderers.adb:63-65)                               */
   MDH_E_UNSUPPORTED_KIND = 3, /* primitive/light kind that is not a built-in one             */
   MDH_E_INDEX = 4,            /* Constraint_Error: index out of the declared range           */
   MDH_E_DEVICE = 5,           /* a HIP call failed (text in mdh_last_error)                  */
   MDH_E_NO_DEVICE = 6,        /* no gfx950 device visible: the library never falls back to
                                  the CPU                                                     */
   MDH_E_STATE = 7,            /* call not valid in this state (e.g. partitioning disabled)   */
   MDH_E_COMM = 8              /* librccl missing, or an RCCL call failed (text in mdh_last_error) */
};

/* = Madarch.Values.Value_Kind order (madarch-values.ads:8) */
typedef enum { MDH_VEC3 = 0, MDH_FLOAT = 1, MDH_INT = 2 } mdh_kind;

/* = Madarch.Components.Component (name, kind)  (madarch-components.ads) */
typedef struct {
   const char *name;
   int32_t kind; /* mdh_kind */
} mdh_component;

/* One primitive or light kind with its declared maximum:
 * Scenes.Primitive_Count / Light_Count (madarch-scenes.ads:15-23).  Built-in
 * names: "Sphere", "Plane", "Box", "Triangle" (madarch-primitives-*.ads),
 * "PointLight", "SpotLight" (madarch-lights-*.ads).
 *
 * Any other primitive name is a USER-DEFINED kind (Primitives.Create,
 * madarch-primitives.ads:24-30): its Distance, Normal and Material expressions --
 * the Exprs trees the reference turns into GLSL text (Exprs.To_GLSL,
 * madarch-exprs.adb:325-711) -- arrive as three MDH_X programs (below) that the
 * kernels interpret.  Built-in kinds leave the program fields NULL / 0.
 *
 * Any other LIGHT name is a user-defined light kind (Lights.Create,
 * madarch-lights.ads:20-24) with two programs in the same fields:
 *   dist_code   = Sample (l, pos, normal, dir, dist) -> vec3 radiance in R0, R1, R2
 *   normal_code = Position (l)                        -> vec3 in R0, R1, R2
 * wrapped as the generated sample_<Light> is (madarch-scenes.adb:497-549):
 * dir = Position (l) - pos; dist = length (dir); dir /= dist; return Sample (...). */
typedef struct {
   const char *name;
   int32_t max_count;
   int32_t n_components;
   const mdh_component *components;
   const int32_t *dist_code;     /* Distance (prim, x) -> float in R0            */
   int32_t dist_len;
   const int32_t *normal_code;   /* Normal (prim, x)   -> vec3 in R0, R1, R2     */
   int32_t normal_len;
   const int32_t *material_code; /* Material (prim)    -> int (raw bits) in R0   */
   int32_t material_len;
} mdh_kind_decl;

/* ---- MDH_X: the register program a user-defined kind's expressions compile to.
 *
 * 64 registers R0..R63 of 32 raw bits per ray; a program is a straight line of
 * int32 words, one instruction per word (two for MDH_X_LIT and MDH_X_SEL):
 *     word = op | dst << 8 | a << 16 | b << 24
 * Vector expressions are lowered by the front end to scalar instructions in the
 * order the GLSL contract of DESIGN.md section 5 fixes: a vec3 lives in three
 * registers, dot = (x x' + y y') + z z', dot2 (v) = dot (v, v), length = sqrt (dot2),
 * normalize = v / length component by component, cross = (ay bz - az by, az bx - ax bz,
 * ax by - ay bx), clamp (x, l, u) = min (max (x, l), u), Min (A, B, C) = min (min (A, B), C)
 * (madarch-exprs.adb:154-155); If_Then_Else evaluates both sides and selects (the
 * expressions have no side effects); Let_In binds registers.  Float literals are
 * rounded through Single'Image like every literal of the generated GLSL
 * (madarch-exprs.adb:330-336); E ** n with a literal integer n = 2..16 is repeated
 * multiplication, squaring from the most significant bit of n ((x^2)^2 x for n = 5).  Arithmetic is IEEE fp32, one operation per
 * instruction, division and square root correctly rounded. */
enum {
   MDH_X_LIT = 0,    /* R[dst] = the next word (raw bits)                          */
   MDH_X_MOV = 1,    /* R[dst] = R[a]                                              */
   MDH_X_COMP = 2,   /* R[dst] = float a of the instance: components in declaration
                        order, a vec3 takes 3 floats, float / int 1 (raw bits)     */
   MDH_X_POINT = 3,  /* R[dst] = argument float a.  Primitive programs: a = 0..2, the point.
                        Light Sample: 0..2 pos, 3..5 normal, 6..8 dir, 9 dist          */
   MDH_X_ADD = 4, MDH_X_SUB = 5, MDH_X_MUL = 6,
   MDH_X_DIV = 7,    /* true division: components of vector "/"                    */
   MDH_X_DIVF = 8,   /* Float "/" Float: true division in the render passes; L + R in
                        Eval_Distance_To when MDH_OPT_ADA_EVAL_DIV is set, as
                        Madarch.Values."/" computes it (madarch-values.adb:112)     */
   MDH_X_NEG = 9, MDH_X_ABS = 10, MDH_X_FLOOR = 11,
   MDH_X_SIGN = 12,  /* 1, -1 or 0 (GLSL sign)                                     */
   MDH_X_MIN = 13, MDH_X_MAX = 14, MDH_X_SQRT = 15,
   MDH_X_POW = 16,   /* the fp32 pow of the shading path (oracle/orc_math.h)       */
   MDH_X_LT = 17, MDH_X_GT = 18, MDH_X_LE = 19, MDH_X_GE = 20, /* 1.0f or 0.0f    */
   MDH_X_SEL = 21,   /* R[dst] = R[a] != 0 ? R[b] : R[c], c = low byte of the next word */
   MDH_X_ITOF = 22,  /* To_Float: R[dst] = (float) (int) R[a]                      */
   MDH_X_ACOS = 23,  /* the fp32 acos of the shading path                           */
   MDH_X_SIN = 24, MDH_X_COS = 25, MDH_X_TAN = 26, MDH_X_ASIN = 27, MDH_X_ATAN = 28,
                     /* explicit fp32 algorithms (oracle/orc_math.h), a few 1e-7 absolute */
   MDH_X_OPS = 29
};
#define MDH_X_REGS 64
#define MDH_X_MAX_WORDS 4096 /* per program */

/* = Scenes.Partitioning_Settings (madarch-scenes.ads:30-41) */
typedef struct {
   int32_t enable;
   int32_t index_count;
   int32_t border_behavior; /* 0 Clamp, 1 Fallback (scenes.ads:28) */
   int32_t grid_dimensions[3];
   float grid_spacing[3];
   float grid_offset[3];
} mdh_partitioning;

/* = Renderers.Probe_Settings (madarch-renderers.ads:23-29) */
typedef struct {
   int32_t radiance_resolution;
   int32_t irradiance_resolution;
   int32_t probe_count[2];
   int32_t grid_dimensions[3];
   float grid_spacing[3];
} mdh_probe_settings;

/* = Renderers.Volumetrics_Settings (madarch-renderers.ads:33-41) */
typedef struct {
   int32_t enabled;
   int32_t visibility_resolution[3];
   float visibility_step_size;
   int32_t scattering_resolution[2];
   float scattering_step_size;
} mdh_volumetrics;

/* = the arguments of Scenes.Compile (madarch-scenes.ads:47-53) */
typedef struct {
   int32_t n_prim_kinds;
   const mdh_kind_decl *prim_kinds;
   int32_t n_light_kinds;
   const mdh_kind_decl *light_kinds;
   mdh_partitioning partitioning;
   float max_dist;
   int32_t loop_strategy; /* 0 Split, 1 Unify (scenes.ads:45); same results, kept for the record */
} mdh_scene_desc;

/* ---- options: switches the reference fixes at shader-compile time (the M_*
 * macros of madarch-renderers.adb:109-143) or that the headless build adds ---- */
enum {
   /* atlas storage: 0 = RGB8 as the reference (render_passes.adb:94), 1 = fp32 */
   MDH_OPT_ATLAS_FORMAT = 0,
   /* screen pass content: 0 = full pixel_color_probes as draw_screen.glsl;
    * 1 = BASELINE config 1 (primary ray only, colour 0.5 n + 0.5, no tonemap);
    * 2 = BASELINE config 2 (direct PBR + AO, irradiance = 0, no indirect specular) */
   MDH_OPT_SCREEN_MODE = 1,
   /* M_AMBIENT_OCCLUSION_STEPS of the screen pass (default 3, renderers.adb:140) */
   MDH_OPT_AO_STEPS = 2,
   /* 1 = also write the primary-ray geometry buffer (hit index, t, steps) */
   MDH_OPT_GBUFFER = 3,
   /* image/probe sharding for one-process-per-GPU runs: this renderer draws the
    * screen tiles and probe slices of `rank` out of `world` (default 0 / 1).  mdh_comm_init sets
    * both; setting them by hand is for callers that bring their own exchange (mdh_frame_*) */
   MDH_OPT_RANK = 4,
   MDH_OPT_WORLD = 5,
   /* 1 = record HIP events around every pass (read with mdh_pass_time) */
   MDH_OPT_TIMING = 6,
   /* Eval_Distance_To arithmetic: 1 = reproduce Madarch.Values."/" on floats
    * (L + R, madarch-values.adb:112) as the Ada evaluator does; 0 = GLSL "/" */
   MDH_OPT_ADA_EVAL_DIV = 7,
   /* how mdh_render schedules consecutive frames (madarch-renderers.adb:302-321;
    * the results are those of the serial order in every case):
    *   0 = strictly one pass after the other;
    *   1 = the probe passes of a frame overlap the screen pass of the frame before
    *       it (second HIP stream, two atlas sets);
    *   2 (default) = also the screen passes of consecutive frames overlap (third
    *       stream, two framebuffers).
    * Sharded renderers keep frames in flight the same way (mdh_frame_begin / _probe_pass / _end);
    * renderers on a caller-supplied stream run serially. */
   MDH_OPT_FRAME_OVERLAP = 8,
   /* user-defined kinds: 1 (default) = the MDH_X programs are compiled into the
    * kernels with hiprtc the first time a pass runs (about two seconds per kernel,
    * cached per process), the analogue of the reference's runtime shader
    * compilation; 0 = the kernels interpret the programs (no compilation, ~60x
    * slower).  Same results bit for bit.  If hiprtc cannot build a scene the
    * renderer falls back to 0 by itself and mdh_last_error says why. */
   MDH_OPT_JIT = 9,
   /* sharded renderers (MDH_OPT_WORLD > 1): 1 (default) = the irradiance pass updates ALL
    * probes on every rank from the gathered radiance atlas instead of its own slice (the pass
    * is one workgroup per probe and far from filling a GPU: the same time, and the second
    * exchange of the frame is not needed); 0 = own slice only, to be exchanged. */
   MDH_OPT_IRRADIANCE_ALL = 10,
   /* where the window's RGBA8 pixels (mdh_swap_buffers) are made: 1 = every screen pass also stores them
    * straight into pinned host memory, over PCIe, beside its float store, and mdh_swap_buffers has nothing
    * to convert or copy; 0 = mdh_swap_buffers converts and copies the framebuffer itself; 2 (default) = 0
    * until the first mdh_swap_buffers, 1 from then on (a renderer that never swaps pays nothing). */
   MDH_OPT_WINDOW = 11,
   /* M_COMPUTE_INDIRECT_SPECULAR of the screen pass (madarch-renderers.adb:138 fixes it at 2; the macro selects
    * among four bodies at glsl/render_probes.glsl:264-272):
    *   0 = no indirect specular;
    *   1 = sample_radiance_with_specular (render_probes.glsl:71-136): the reflection's hit position lit by the
    *       radiance atlases of the eight cage probes of the SHADED point, weighted by soft shadows and trilinearly;
    *   2 (default) = sample_radiance_no_specular (:138-209): the best-facing visible cage probe of the reflection's
    *       hit, plus that point's direct specular (M_ADD_INDIRECT_SPECULAR = 1);
    *   3 = compute_indirect_specular (:211-244): the reflection's hit shaded in full (direct light + irradiance).
    * textureLod on the single-level atlases is level 0 in every mode. */
   MDH_OPT_INDIRECT_SPECULAR = 12,
   /* Temporal blending of the irradiance atlas -- NOT in the reference, off by default (SURVEY.md section 8f-4 lists
    * it as a deviation): the irradiance pass stores mix (fresh, previous, h) with h = value / 1000 (0 .. 999), `previous`
    * being the texel the atlas held before the pass, as stored (RGB8: its 8-bit levels).  0 = the reference. */
   MDH_OPT_HYSTERESIS_PERMILLE = 13,
   /* Scheduling only, no effect on any texel: 1 (default) = the radiance pass records every probe ray's primary-march
    * length and the next frame's pass takes the rays sorted by it (a wavefront pays the longest march among its 64
    * rays); 0 = rays in probe order. */
   MDH_OPT_RADIANCE_ORDER = 14,
   /* Scheduling only, no effect on any pixel: 1 (default) = a screen pass records how long every 8x8 tile's wavefront
    * took, and later passes start the tiles in that order, slowest first (a pass ends with its slowest wavefront: in
    * image order the passes of the reference's scenes spend 15 - 45 % of their time waiting for a few late, long
    * tiles); re-sorted after camera moves and geometry edits; 0 = tiles in image order. */
   MDH_OPT_SCREEN_ORDER = 15,
   /* READ-ONLY: 0 = this library computes the oracle's bits (DESIGN.md "numerics contract": the only kind that ships);
    * 1 = the labelled experiment build (`make -C madarch_amd/csrc fast`: hardware sqrt / rcp / log / exp, fused
    * multiply-adds, the irradiance fold in four partial sums), which holds BASELINE.json's 1e-4 tolerance at best.
    * Setting it to anything but the build's own value is refused. */
   MDH_OPT_NUMERICS = 16,
   /* 1 = a mip chain for the radiance atlas (default 0 = the reference's behaviour).  The reference asks
      textureLod for level 1 (mode 2, render_probes.glsl:197) and for mix (0, radiance_lods, 2 roughness)
      (mode 1, render_probes.glsl:84-86,131; probe_utils.glsl:17) of an atlas it allocates with ONE level
      (render_passes.adb:113): every tap reads level 0.  With the switch on, each screen pass first builds
      levels 1 .. radiance_lods of the radiance atlas it reads (2x2 box filter of the level below,
      ((a + b) + (c + d)) / 4 per channel in fp32, stored in the atlas's format) and the two taps read
      them as GL_LINEAR_MIPMAP_LINEAR would.  Needs a power-of-two radiance resolution
      (MDH_E_INVALID otherwise).  Levels can be read back: mdh_read_texture (MDH_TEX_RADIANCE_MIP0 + l). */
   MDH_OPT_RADIANCE_MIPS = 17,
   /* Scheduling only, no effect on any pixel: a screen launch whose tiles, as two or four wavefronts each (8x4 or 4x4
    * pixels: 32 or 16 of a wavefront's 64 lanes), stay within `value` wavefronts is launched that way -- a launch that
    * leaves the chip's wavefront slots empty (a small window) lasts as long as its slowest
    * wavefront, which then marches for a quarter of the tile only, and the tile's work runs on four SIMDs (whole frames only: a
    * rank's scattered tiles of a sharded frame measured no faster).  Default 2560
    * (half the slots of an MI355X at five wavefronts per SIMD); 0 = every tile one wavefront.  Launches of 2 048 tiles and
    * more (MDH_OPT_SCREEN_ORDER's) are never split. */
   MDH_OPT_SCREEN_SPLIT = 18
};

/* passes of Renderers.Render (madarch-renderers.adb:302-321) */
enum {
   MDH_PASS_RADIANCE = 0,   /* compute_probe_radiance.glsl        */
   MDH_PASS_IRRADIANCE = 1, /* update_probe_irradiance.glsl       */
   MDH_PASS_VISIBILITY = 2, /* compute_frustrum_visibility.glsl   */
   MDH_PASS_SCATTERING = 3, /* accumulate_scattering.glsl         */
   MDH_PASS_SCREEN = 4,     /* draw_screen.glsl                   */
   MDH_PASS_EXCHANGE = 5,   /* no pass of the reference: the all-gather of a sharded frame's atlas slices
                               (mdh_frame_exchange), timed like a pass (mdh_pass_time)                     */
   MDH_PASS_COUNT = 6
};

/* atlases / textures of the renderer (texture units 0-3, renderers.adb:239-279) */
enum { MDH_TEX_RADIANCE = 0, MDH_TEX_IRRADIANCE = 1, MDH_TEX_VISIBILITY = 2, MDH_TEX_SCATTERING = 3 };
#define MDH_TEX_RADIANCE_MIP0 16 /* mdh_read_texture only: + l = level l >= 1 of the radiance atlas (MDH_OPT_RADIANCE_MIPS) */

/* Renderers.Create (madarch-renderers.adb:91-300) together with Scenes.Compile
 * (madarch-scenes.adb:1378-1421).  `width`/`height` replace the window size
 * (Windows.Open; the build is headless).  `device` is the HIP device ordinal
 * (one process per GPU: pass LOCAL_RANK).  SURVEY.md section 8(b) sketches this argument as
 * `n_devices`; with one process per GPU -- the execution model of this build -- a renderer owns ONE
 * device, and the N-device frame is formed by the renderers of N processes joining a communicator
 * (mdh_comm_init below): `n_devices` is that call's `world`. */
int32_t mdh_create(int32_t width, int32_t height, const mdh_scene_desc *scene,
                   const mdh_probe_settings *probes, const mdh_volumetrics *volumetrics,
                   int32_t device, mdh_renderer **out);

/* the reference never frees a Renderer; explicit release is new and harmless */
int32_t mdh_destroy(mdh_renderer *r);

int32_t mdh_set_option(mdh_renderer *r, int32_t option, int32_t value);
int32_t mdh_get_option(mdh_renderer *r, int32_t option, int32_t *value);

/* Renderers.Set_Material (madarch-renderers.adb:349-367); id is 0-based */
int32_t mdh_set_material(mdh_renderer *r, int32_t id0, const float albedo[3], float metallic,
                         float roughness);
/* Renderers.Add_Material (madarch-renderers.adb:369-377) */
int32_t mdh_add_material(mdh_renderer *r, const float albedo[3], float metallic, float roughness,
                         int32_t *out_id0);

/* Renderers.Set_Primitive (madarch-renderers.adb:379-398); index1 is 1-based */
int32_t mdh_set_primitive(mdh_renderer *r, int32_t kind_ix, int32_t index1, const void *std140_blob,
                          int32_t nbytes);
/* Renderers.Add_Primitive (madarch-renderers.adb:435-456) */
int32_t mdh_add_primitive(mdh_renderer *r, int32_t kind_ix, const void *std140_blob, int32_t nbytes,
                          int32_t *out_count);
/* Renderers.Set_Light (madarch-renderers.adb:458-483): also sets the kind's
 * count and total_light_count to index1, as the reference does */
int32_t mdh_set_light(mdh_renderer *r, int32_t index1, int32_t light_kind_ix, const void *std140_blob,
                      int32_t nbytes);

/* Renderers.Set_Camera_Position / Set_Camera_Orientation (renderers.adb:485-497) */
int32_t mdh_set_camera_position(mdh_renderer *r, const float p[3]);
int32_t mdh_set_camera_orientation(mdh_renderer *r, const float m_colmajor[9]);

/* Renderers.Update_Partitioning (madarch-renderers.adb:757-775);
 * method: 0 CPU_Best, 1 CPU_Fast, 2 GPU_Fast (renderers.ads:93).  All three builders run on
 * the device; the call enqueues the build and returns -- frames in flight keep the table they
 * were launched with, everything launched afterwards uses the new one. */
int32_t mdh_update_partitioning(mdh_renderer *r, int32_t method);

/* Renderers.Render (madarch-renderers.adb:302-321): enqueues all passes of one
 * frame on the renderer's stream.  mdh_finish waits for them. */
int32_t mdh_render(mdh_renderer *r);
/* one pass only (for one-process-per-GPU runs that exchange atlas slices
 * between the passes, see DESIGN.md "Multi-GPU") */
int32_t mdh_render_pass(mdh_renderer *r, int32_t pass);
/* Render in three steps, for callers that put work of their own between the
 * passes: mdh_frame_begin, mdh_frame_probe_pass(MDH_PASS_RADIANCE), [exchange],
 * mdh_frame_probe_pass(MDH_PASS_IRRADIANCE), [exchange], mdh_frame_end (volumetric
 * passes + screen pass).  mdh_render is exactly that sequence without the exchanges,
 * and frames opened this way are kept in flight the same way (MDH_OPT_FRAME_OVERLAP).
 * While a frame is open the atlas reads, writes and device pointers refer to the
 * atlas set that frame is producing, and the caller's device work on it must be
 * ordered on the stream mdh_probe_stream names. */
int32_t mdh_frame_begin(mdh_renderer *r);
int32_t mdh_frame_probe_pass(mdh_renderer *r, int32_t pass);
/* the exchange step of an open frame of a renderer that has a communicator (mdh_comm_init): all-gather of
 * the ranks' slices of atlas `tex` (MDH_TEX_RADIANCE / MDH_TEX_IRRADIANCE), in place on the open frame's atlas
 * set, enqueued on the probe stream.  Without a communicator: nothing (MDH_OK). */
int32_t mdh_frame_exchange(mdh_renderer *r, int32_t tex);
int32_t mdh_frame_end(mdh_renderer *r);
int32_t mdh_finish(mdh_renderer *r);

/* ---- one frame on the N GPUs of a node: one process (and one renderer) per GPU.
 *
 * The reference draws a frame with ONE call on one GPU (Renderers.Render, madarch-renderers.adb:302-321).
 * SURVEY.md section 8(b)/(e) asks for the same call to draw it on N: every process creates its renderer on its
 * own device, the processes join a communicator (RCCL over xGMI; librccl is opened on first use, the
 * library does not link it), and from then on mdh_render of every rank IS one frame of the sharded schedule:
 *    radiance pass for the rank's probe slice -> all-gather of the slices, in place on the probe-major atlas,
 *    on the probe stream -> irradiance pass (all probes on every rank, MDH_OPT_IRRADIANCE_ALL; own slice +
 *    a second all-gather otherwise) -> volumetric passes (replicated) -> screen pass for the 8x8 tiles
 *    t = rank (mod world).
 * Frames stay in flight exactly as on one GPU (MDH_OPT_FRAME_OVERLAP).  No caller-side collective, no other
 * runtime in the process: a host written in Ada or C needs only these calls and a way to hand 128 bytes from
 * rank 0 to the others (a file, a pipe, an environment variable).
 *
 *   mdh_comm_unique_id   rank 0 only: the 128-byte id of a new communicator (ncclGetUniqueId)
 *   mdh_comm_init        every rank, collectively: joins (ncclCommInitRank on the renderer's device) and sets
 *                        MDH_OPT_RANK / MDH_OPT_WORLD, which cannot be set by hand while the communicator lives
 *   mdh_comm_destroy     leaves; the renderer is rank 0 of 1 again
 *   mdh_comm_abort       tears the communicator down WITHOUT waiting for collectives in flight
 *                        (ncclCommAbort): what a host's watchdog calls when a peer never arrives
 *   mdh_comm_barrier     mdh_finish + a collective every rank has to reach + host wait
 *   mdh_comm_max_f64     *value = max over the ranks (timing: the slowest rank's wall time)
 *   mdh_comm_reduce_framebuffer
 *                        the ranks' tiles of the last frame summed into `root`'s framebuffer (every pixel is
 *                        non-zero on one rank only, so the sum is the whole frame bit for bit up to the sign of
 *                        zeros): what mdh_read_framebuffer / the window of rank `root` then shows.  Not part of
 *                        a frame; a host that presents the image calls it, a benchmark does not. */
#define MDH_COMM_ID_BYTES 128
int32_t mdh_comm_unique_id(uint8_t id_out[MDH_COMM_ID_BYTES]);
/* MDH_OK when this process can load librccl (what the ranks other than 0 ask before everybody enters mdh_comm_init) */
int32_t mdh_comm_available(void);
int32_t mdh_comm_init(mdh_renderer *r, const uint8_t id[MDH_COMM_ID_BYTES], int32_t rank, int32_t world);
int32_t mdh_comm_destroy(mdh_renderer *r);
int32_t mdh_comm_abort(mdh_renderer *r);
int32_t mdh_comm_barrier(mdh_renderer *r);
int32_t mdh_comm_max_f64(mdh_renderer *r, double *value);
int32_t mdh_comm_reduce_framebuffer(mdh_renderer *r, int32_t root);

/* ---- the same sharded frame WITHOUT a collective library: the peer exchange ("a hand-rolled P2P
 * fan-out").  Every rank exports the interprocess handles (hipIpcGetMemHandle) of its radiance atlases and of a block of
 * frame numbers in device memory; the host hands every rank's 512 bytes to every rank (a file, a pipe, whatever it used
 * for the communicator id); mdh_peer_init opens the peers' handles.  From then on mdh_render is the sharded frame of
 * mdh_comm_init, its exchange step being COPIES ordered on the device: behind its radiance pass a rank stores the
 * frame's number into its block; for each peer it enqueues one wavefront that polls the peer's number until it has
 * reached this frame's, and behind it the copy of the peer's slice out of the peer's atlas into its own
 * (hipMemcpyAsync, device to device) -- all on the probe stream, no host waits, frames stay in flight.  The irradiance
 * pass runs for all probes on every rank (MDH_OPT_IRRADIANCE_ALL must be on).  It is the fall-back between RCCL and the
 * exchange through host memory, and the one device-resident exchange that several processes can run on ONE GPU (RCCL
 * refuses two ranks of a communicator on one device).  The ranks must be processes of one node (one process per rank)
 * and must all render the same frames; a peer that never arrives ends the wait after a few seconds and the next
 * mdh_render / mdh_finish returns MDH_E_COMM.
 *   mdh_peer_export   this rank's handles (MDH_PEER_BLOB_BYTES bytes); starts a new session at frame 0
 *   mdh_peer_init     every rank, with all ranks' blobs in rank order; sets MDH_OPT_RANK / MDH_OPT_WORLD
 *   mdh_comm_destroy / mdh_comm_abort   leave (as for a communicator); mdh_comm_barrier, mdh_comm_max_f64 and
 *                     mdh_comm_reduce_framebuffer need a communicator and return MDH_E_STATE here */
#define MDH_PEER_BLOB_BYTES 512
int32_t mdh_peer_export(mdh_renderer *r, uint8_t blob_out[MDH_PEER_BLOB_BYTES]);
int32_t mdh_peer_init(mdh_renderer *r, const uint8_t *blobs /* world x MDH_PEER_BLOB_BYTES */, int32_t rank, int32_t world);

/* replaces Swap_Buffers (renderers.adb:320): linear RGB floats, H*W*3, row 0 = top.
 * In a sharded run only this rank's tiles are written, the rest is 0. */
int32_t mdh_read_framebuffer(mdh_renderer *r, float *rgb_out);
/* Swap_Buffers itself (renderers.adb:320): what the reference's window shows.  The default framebuffer of
 * that window is RGBA8 without sRGB encoding, so the last frame's colours are clamped to [0, 1] and
 * converted to the nearest of 256 levels per channel (OpenGL 4.3 core 2.3.5.1; ties to even, NaN -> 0,
 * alpha = 255), on the device.  mdh_swap_buffers enqueues that conversion and the copy into a pinned host
 * buffer behind the frame and returns without waiting: frames stay in flight (MDH_OPT_FRAME_OVERLAP)
 * and the copy of one frame runs beside the passes of the next.  mdh_front_buffer waits for the most
 * recent swap only and returns its pixels: H*W*4 bytes, R G B A, row 0 = top, owned by the renderer
 * and valid until the second next mdh_swap_buffers (with MDH_OPT_WINDOW also: until three more
 * screen passes have been started); *swap_count (may be NULL) is the number of swaps so far.
 * With MDH_OPT_WINDOW = 1 the screen pass has stored the pixels in host memory already and the swap
 * only marks them.  In a sharded run only this rank's tiles are written, the rest is 0. */
int32_t mdh_swap_buffers(mdh_renderer *r);
int32_t mdh_front_buffer(mdh_renderer *r, const uint8_t **rgba, int64_t *swap_count);
/* primary-ray geometry buffer (needs MDH_OPT_GBUFFER): per pixel the flat
 * primitive index of closest_primitive_info (-1 = miss), the march length t
 * and the number of SDF evaluations of the primary march */
int32_t mdh_read_gbuffer(mdh_renderer *r, int32_t *index_out, float *t_out, int32_t *steps_out);

/* texture contents as the reference's 2-D images (row 0 = normalised y of 0),
 * `channels` floats per texel (3, or 4 for scattering); out may be NULL to
 * query the size only */
int32_t mdh_read_texture(mdh_renderer *r, int32_t tex, float *out, int32_t *width, int32_t *height,
                         int32_t *channels);
/* deterministic warm start / checkpoint of the DDGI state (same layout) */
int32_t mdh_write_texture(mdh_renderer *r, int32_t tex, const float *in, int32_t width, int32_t height,
                          int32_t channels);

/* probe-major atlas slices through host memory, [probe][res][res][3] floats:
 * the exchange step of a sharded run when the ranks have no device collective */
int32_t mdh_read_atlas_slice(mdh_renderer *r, int32_t tex, int32_t probe_begin, int32_t n_probes, float *out);
int32_t mdh_write_atlas_slice(mdh_renderer *r, int32_t tex, int32_t probe_begin, int32_t n_probes,
                              const float *in);

/* device-resident atlas slices for callers that bring an exchange of their own (mdh_comm_init needs none of this):
 * pointer to the probe-major atlas, its total byte size and the byte range
 * [offset, offset+bytes) this rank owns */
int32_t mdh_atlas_device_ptr(mdh_renderer *r, int32_t tex, void **dptr, int64_t *total_bytes,
                             int64_t *own_offset, int64_t *own_bytes);
/* the HIP stream (hipStream_t) every pass is enqueued on */
int32_t mdh_stream(mdh_renderer *r, void **stream);
/* enqueue on the caller's stream instead (e.g. the one an RCCL communicator is
 * ordered with); NULL returns to the renderer's own stream */
int32_t mdh_set_stream(mdh_renderer *r, void *stream);
/* the stream the probe passes of an open frame run on (hipStream_t) */
int32_t mdh_probe_stream(mdh_renderer *r, void **stream);

/* Renderers.Eval_Distance_To (madarch-renderers.adb:499-526), batched: for
 * each of n points the closest distance over the listed kinds (initial
 * closest 1.0e10) and the normal of the arg-min primitive.  n = 1 is the
 * reference call.  Runs beside the frames in flight (a stream of its own) and
 * returns when its own results are on the host. */
int32_t mdh_eval_distance_to(mdh_renderer *r, int32_t n, const float *points_xyz,
                             const int32_t *kind_ixs, int32_t n_kinds, float *normals_xyz_out,
                             float *dist_out);

/* accumulated HIP-event time and launch count of one pass (MDH_OPT_TIMING) */
int32_t mdh_pass_time(mdh_renderer *r, int32_t pass, double *total_ms, int64_t *launches);
int32_t mdh_reset_pass_times(mdh_renderer *r);

/* std140 layout queries = Scenes.Get_Primitives_Location / Get_Lights_Location
 * (madarch-scenes.adb:1435-1462) and Get_GPU_Type(...).Size */
int32_t mdh_scene_layout(mdh_renderer *r, int32_t is_light, int32_t kind_ix, int32_t *count_offset,
                         int32_t *array_offset, int32_t *stride, int32_t *element_size);
int32_t mdh_scene_buffer_size(mdh_renderer *r, int32_t *size, int32_t *total_light_count_offset);
/* raw std140 image of the scene uniform block (binding 1) for inspection */
int32_t mdh_read_scene_buffer(mdh_renderer *r, void *out, int32_t nbytes);
/* partition table as int32 [cell][n_prim_kinds counts + index_count indices] */
int32_t mdh_read_partitioning(mdh_renderer *r, int32_t *out, int32_t n_ints);
/* cells of the last Update_Partitioning whose candidate list overflowed Index_Count
 * (the reference prints "Warning : partition size too small", renderers.adb:593-598);
 * waits for that build */
int32_t mdh_partition_warnings(mdh_renderer *r);

const char *mdh_last_error(void);
const char *mdh_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MADARCH_HIP_H */
