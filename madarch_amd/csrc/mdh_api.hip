// mdh_api.hip -- libmadarch_hip.so: the C ABI of include/madarch_hip.h.
//
// Host side of the MI355X back end: what Madarch.Renderers does with OpenGL
// objects (reference madarch/madarch-renderers.adb:91-497, support/gpu_buffers.adb,
// support/render_passes.adb) is done here with HIP memory, one stream per renderer
// and the kernels of mdh_kernels.h.  There is no CPU rendering path: without a
// gfx950 device mdh_create fails with MDH_E_NO_DEVICE.
#include "../../include/madarch_hip.h"
#include "mdh_kernels.h"
#include "mdh_jit_sources.inc" // the three device headers as string literals (Makefile), for the hiprtc build of user-defined kinds

#include <dlfcn.h>
#include <unistd.h>
#include <hip/hiprtc.h> // types only: the library is opened on first use (no link-time dependency)
#include <rccl/rccl.h>  // types only, likewise: librccl is opened by the first mdh_comm_* call

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

static thread_local char g_err[512];
static int seterr(int code, const char *msg)
{
   snprintf(g_err, sizeof g_err, "%s", msg);
   return code;
}
#define HIP_TRY(expr)                                                                               \
   do {                                                                                             \
      hipError_t e_ = (expr);                                                                       \
      if (e_ != hipSuccess) {                                                                       \
         snprintf(g_err, sizeof g_err, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
         return MDH_E_DEVICE;                                                                       \
      }                                                                                             \
   } while (0)

extern "C" const char *mdh_last_error(void) { return g_err; }
#if MDH_FAST_NUMERICS
extern "C" const char *mdh_version(void) { return "madarch-hip 0.3 (gfx950) FAST-NUMERICS EXPERIMENT BUILD: tolerance only, not the oracle's bits"; }
#elif MDH_HYBRID_NUMERICS
extern "C" const char *mdh_version(void) { return "madarch-hip 0.4 (gfx950) HYBRID-NUMERICS EXPERIMENT BUILD: exact march loops and primary ray, hardware rcp / sqrt / exp / log in shading"; }
#else
extern "C" const char *mdh_version(void) { return "madarch-hip 0.4 (gfx950)"; }
#endif

// ------------------------------------------------------------------ std140 layout
// = GPU_Types (support/gpu_types-base.ads:21-37, gpu_types-structs.adb:11-38,
// gpu_types-fixed_arrays.adb:17-39)
static int pad_to(int x, int a) { while (x % a) ++x; return x; }
static int base_align(int kind) { return kind == MDH_VEC3 ? 16 : 4; }
static int base_size(int kind) { return kind == MDH_VEC3 ? 12 : 4; }

struct Kind {
   int type = -1, max_count = 0, ncomp = 0;
   std::string comp_name[8];
   int comp_kind[8], comp_off[8];
   int elem_size = 0, stride = 0, count_off = 0, array_off = 0;
   int f_a = -1, f_b = -1, f_c = -1, f_d = -1; // resolved field offsets
   // user-defined kinds (PK_CUSTOM): the three MDH_X programs and the packed instance size
   std::vector<int32_t> x_dist, x_nrm, x_mat;
   int inst_floats = 0;
   int offset_of(const char *name, int kind) const
   {
      for (int i = 0; i < ncomp; ++i)
         if (comp_name[i] == name && comp_kind[i] == kind) return comp_off[i];
      return -1;
   }
};

// An MDH_X program is accepted only if every word is a known instruction with in-range
// operands (nothing the caller hands over is trusted to index the register file or the instance).
static bool valid_program(const int32_t *code, int n, int inst_floats, int n_args)
{
   if (!code || n < 1 || n > MDH_X_MAX_WORDS) return false;
   for (int pc = 0; pc < n; ++pc) {
      const uint32_t w = (uint32_t)code[pc];
      const int op = w & 255, d = (w >> 8) & 255, a = (w >> 16) & 255, b = (w >> 24) & 255;
      if (op >= MDH_X_OPS || d >= MDH_X_REGS) return false;
      switch (op) {
      case MDH_X_LIT: if (++pc >= n) return false; break;
      case MDH_X_COMP: if (a >= inst_floats) return false; break;
      case MDH_X_POINT: if (a >= n_args) return false; break;
      case MDH_X_SEL:
         if (a >= MDH_X_REGS || b >= MDH_X_REGS || ++pc >= n || (uint32_t)code[pc] >= MDH_X_REGS) return false;
         break;
      default: if (a >= MDH_X_REGS || b >= MDH_X_REGS) return false; break;
      }
   }
   return true;
}

static bool resolve_kind(Kind &k, const mdh_kind_decl &d, bool is_light)
{
   static const char *P[4] = {"Sphere", "Plane", "Box", "Triangle"};
   static const char *L[2] = {"PointLight", "SpotLight"};
   if (!d.name || !d.components) return false;
   // A kind's behaviour is its expressions (madarch-primitives.ads:24-30, madarch-lights.ads:20-24), never its name: one
   // that brings programs runs them, whatever it is called; the hand-written device functions are taken only by a kind
   // that brings none and carries one of the library's own names (the reference's own six kinds).
   k.type = -1;
   const bool has_programs = d.dist_code || d.normal_code || d.material_code;
   if (!has_programs)
      for (int t = 0; t < (is_light ? 2 : 4); ++t)
         if (strcmp(d.name, is_light ? L[t] : P[t]) == 0) k.type = t;
   const bool custom = has_programs && d.dist_code && d.normal_code && (is_light || d.material_code);
   if (custom) k.type = is_light ? (int)LK_CUSTOM : (int)PK_CUSTOM;
   if (k.type < 0 || d.n_components < 1 || d.n_components > 8 || d.max_count < 0) return false;
   k.max_count = d.max_count;
   k.ncomp = d.n_components;
   int off = 0;
   for (int i = 0; i < k.ncomp; ++i) {
      k.comp_name[i] = d.components[i].name ? d.components[i].name : "";
      k.comp_kind[i] = d.components[i].kind;
      off = pad_to(off, base_align(k.comp_kind[i]));
      k.comp_off[i] = off;
      off += base_size(k.comp_kind[i]);
   }
   k.elem_size = off;
   k.stride = pad_to(off, 16);
   if (custom) {
      for (int i = 0; i < k.ncomp; ++i) {
         if (k.comp_kind[i] != MDH_VEC3 && k.comp_kind[i] != MDH_FLOAT && k.comp_kind[i] != MDH_INT) return false;
         k.inst_floats += k.comp_kind[i] == MDH_VEC3 ? 3 : 1;
      }
      if (is_light) { // Sample (pos, normal, dir, dist: 10 argument floats) and Position (none)
         if (!valid_program(d.dist_code, d.dist_len, k.inst_floats, 10) || !valid_program(d.normal_code, d.normal_len, k.inst_floats, 0)) return false;
      } else if (!valid_program(d.dist_code, d.dist_len, k.inst_floats, 3) || !valid_program(d.normal_code, d.normal_len, k.inst_floats, 3) ||
                 !valid_program(d.material_code, d.material_len, k.inst_floats, 0))
         return false;
      k.x_dist.assign(d.dist_code, d.dist_code + d.dist_len);
      k.x_nrm.assign(d.normal_code, d.normal_code + d.normal_len);
      if (!is_light) k.x_mat.assign(d.material_code, d.material_code + d.material_len);
      return true;
   }
   if (!is_light) {
      k.f_d = k.offset_of("material_id", MDH_INT);
      switch (k.type) {
      case PK_SPHERE: k.f_a = k.offset_of("center", MDH_VEC3); k.f_b = k.offset_of("radius", MDH_FLOAT); k.f_c = 0; break;
      case PK_PLANE: k.f_a = k.offset_of("normal", MDH_VEC3); k.f_b = k.offset_of("offset", MDH_FLOAT); k.f_c = 0; break;
      case PK_BOX: k.f_a = k.offset_of("center", MDH_VEC3); k.f_b = k.offset_of("side", MDH_VEC3); k.f_c = 0; break;
      default: k.f_a = k.offset_of("v1", MDH_VEC3); k.f_b = k.offset_of("v2", MDH_VEC3); k.f_c = k.offset_of("v3", MDH_VEC3); break;
      }
   } else {
      k.f_a = k.offset_of("position", MDH_VEC3);
      if (k.type == LK_POINT) { k.f_b = k.offset_of("color", MDH_VEC3); k.f_c = 0; k.f_d = 0; }
      else { k.f_b = k.offset_of("direction", MDH_VEC3); k.f_c = k.offset_of("aperture", MDH_FLOAT); k.f_d = k.offset_of("color", MDH_VEC3); }
   }
   return k.f_a >= 0 && k.f_b >= 0 && k.f_c >= 0 && k.f_d >= 0;
}

// Single'Image keeps 6 significant digits: literals that reach the shaders through
// the generated GLSL text are rounded that way (scenes.adb:21-24,1200-1201;
// renderers.adb:119-134)
static float image_roundtrip(float x)
{
   char buf[64];
   snprintf(buf, sizeof buf, "%.5E", (double)x);
   return strtof(buf, nullptr);
}

// -------------------------------------------------------------------- the renderer
#define MAX_MATERIALS 20 // glsl/materials.glsl:9
#ifndef MDH_ATLAS_SETS
#define MDH_ATLAS_SETS 3 // sets of probe atlases (and volumetric textures) that pipelined frames rotate through
#endif
// how often the probe rays (MDH_OPT_RADIANCE_ORDER) and the screen tiles (MDH_OPT_SCREEN_ORDER) are sorted again: every
// MDH_RAD_RESORT passes, and after MDH_RAD_RESORT_MOVING passes when the geometry (or, for the tiles, the camera) changed
#ifndef MDH_RAD_RESORT
#define MDH_RAD_RESORT 64
#endif
#ifndef MDH_RAD_RESORT_MOVING
#define MDH_RAD_RESORT_MOVING 8
#endif


struct mdh_renderer {
   int W = 0, H = 0, device = 0;
   hipStream_t stream = nullptr;
   int npk = 0, nlk = 0;
   Kind pk[MDH_MAX_KINDS], lk[MDH_MAX_LIGHT_KINDS];
   int prim_base[MDH_MAX_KINDS] = {0};
   int host_count[MDH_MAX_KINDS] = {0}; // All_Primitives lengths (renderers.ads:128)
   float max_dist = 20.0f;
   mdh_partitioning part{};
   float pg_spacing[3], pg_offset[3], part_gpu_diag = 0.0f;
   int part_cells = 0, part_warnings = 0;
   mdh_probe_settings probes{};
   mdh_volumetrics vol{};
   float vstep = 0.1f, sstep = 0.1f;
   std::vector<uint8_t> scene_ubo; // std140 image of uniform block 1 (scenes.adb:551-600)
   int total_light_off = 0;
   uint8_t materials_ubo[16 + 32 * MAX_MATERIALS] = {0}; // renderers.adb:77-89
   int last_material_index = 0;
   float cam_pos[3] = {0, 0, 0}, cam_m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; // renderers.adb:225-226
   // options
   int opt_atlas = 0, opt_mode = 0, opt_ao = 3, opt_gbuffer = 0, opt_rank = 0, opt_world = 1, opt_timing = 0, opt_ada_div = 1, opt_spec = 2, opt_hyst = 0;
   // device state
   std::vector<float4> table_host;
   // The scene table lives in a ring of buffers: an edit (Set_Light every frame in the reference's examples) is
   // packed into the NEXT buffer and uploaded asynchronously from pinned memory while the frames in flight keep
   // reading theirs -- a scene edit does not drain the frame pipeline.  A buffer is rewritten only after the last
   // kernels that were launched with it have finished (tab_done, per stream).
   static const int TAB_RING = 4;
   float4 *d_table_ring[TAB_RING] = {nullptr, nullptr, nullptr, nullptr};
   float4 *h_table_ring[TAB_RING] = {nullptr, nullptr, nullptr, nullptr}; // pinned
   int tab_slot = 0;
   static const int NSTREAMS = 5; // main, probe, alternate, query, volumetric (stream_index)
   hipEvent_t tab_done[TAB_RING][NSTREAMS] = {{nullptr}};
   bool tab_used[TAB_RING][NSTREAMS] = {{false}};
   hipEvent_t ev_table = nullptr;       // recorded after the last upload, on table_stream
   hipStream_t table_stream = nullptr;
   unsigned long long table_version = 0, tab_seen[NSTREAMS] = {0, 0, 0, 0, 0}; // per stream: has it waited for ev_table
   size_t table_cap = 0;
   bool table_dirty = true;
   // The space-partition table, in a ring like the scene table: Update_Partitioning builds into the next buffer
   // on the stream that uses it first and returns; frames in flight keep the buffer they were launched with
   // (KScene::part_table travels by value).  The builders' warning count comes back through pinned memory
   // and is waited for only when somebody asks (mdh_partition_warnings).
   static const int PART_RING = 4;
   int *d_part_ring[PART_RING] = {nullptr, nullptr, nullptr, nullptr};
   int part_slot = 0;
   hipEvent_t part_done[PART_RING][NSTREAMS] = {{nullptr}};
   bool part_used[PART_RING][NSTREAMS] = {{false}};
   hipEvent_t ev_part = nullptr, ev_warn = nullptr; // the last build, the last read-back of its warning count
   hipStream_t part_stream = nullptr;
   unsigned long long part_version = 0, part_seen[NSTREAMS] = {0, 0, 0, 0, 0};
   int *d_warn = nullptr, *h_warn = nullptr;
   bool warn_pending = false;
   hipStream_t query_stream = nullptr; // Eval_Distance_To: beside the frames in flight, not behind them
   hipStream_t vol_stream = nullptr;   // the camera-only volumetric passes of pipelined frames, beside their probe passes
   float *d_query = nullptr; // Eval_Distance_To: points, normals, distances of the largest batch so far
   size_t query_cap = 0;
   // Two sets of probe atlases.  `last` is the set the most recent frame wrote: every read, write and
   // single pass works on it in place.  A pipelined mdh_render (frame overlap, see mdh_render) writes the
   // other set while the previous frame's screen pass still reads this one, then flips.
   // (MDH_ATLAS_SETS of them, three since round 3: with two, the probe passes of frame N + 1 write the set the screen pass of
   //  frame N - 1 still reads and wait for it -- the chain screen (N - 1) -> probes (N + 1) -> screen (N + 1) bound the frame
   //  rate of pipelined frames; with three they wait for the screen pass of frame N - 2, long gone)
   static const int NSETS = MDH_ATLAS_SETS;
   void *d_rad2[NSETS] = {nullptr}, *d_irr2[NSETS] = {nullptr};
   void *d_rad_mips[NSETS] = {nullptr}; // MDH_OPT_RADIANCE_MIPS: levels 1 .. radiance_lods of each set's radiance atlas, one behind the other
   int opt_mips = 0;
   int n_cus = 0;                          // compute units of the device
   // RadOrder (mdh_kernels.h): every ray's primary-march steps, the rays sorted by them, the sort's histograms
   float *d_irr_taps = nullptr;  // k_irradiance's scratch: the taps of the pass's probes (mdh_kernels.h)
   size_t irr_taps_cap = 0;      // in floats
   unsigned char *d_rad_steps = nullptr;
   unsigned *d_rad_order = nullptr, *d_rad_hist = nullptr;
   long rad_rays_cap = 0;                  // rays the three buffers are sized for
   long rad_order_rays = 0;                // rays the stored order is of (0: none)
   int rad_order_begin = -1;               // ... and the first probe of their slice
   int rad_order_age = 0;                  // radiance passes since the rays were sorted
   unsigned long rad_order_scene = 0, geometry_edits = 0; // primitives set or added when the rays were sorted / so far
   int opt_rad_order = MDH_RAD_ORDER_DEFAULT;
   // MDH_OPT_SCREEN_ORDER (ScreenArgs, mdh_kernels.h): the screen pass's tiles in the order of their wavefronts' durations
   unsigned char *d_scr_cost = nullptr;          // [tiles] sort keys, written by the pass that is followed by a sort
   unsigned *d_scr_order[2] = {nullptr, nullptr}; // [tiles] two buffers: passes in flight keep reading the one they were launched with
   unsigned *d_scr_hist = nullptr;
   int scr_order_cur = -1;                       // the buffer that holds the order (-1: none)
   int scr_order_n = 0, scr_order_rank = -1, scr_order_world = -1; // what that order is of
   int scr_order_age = 0;                        // screen passes since the tiles were sorted
   unsigned long scr_order_geom = 0;             // geometry_edits when they were
   float scr_order_cam[12] = {0};                // the camera then
   hipEvent_t ev_scr_sort = nullptr, ev_scr_other = nullptr;
   unsigned long long scr_sort_version = 0, scr_sort_seen[NSTREAMS] = {0, 0, 0, 0, 0};
   int opt_scr_order = 1;
   int opt_scr_split = MDH_SCREEN_SPLIT_DEFAULT; // MDH_OPT_SCREEN_SPLIT: the wavefronts a split screen launch may have (0: never split)
   std::map<std::pair<const void *, size_t>, int> resident; // workgroups per CU of (kernel, LDS bytes): rad_first_round
   int last = 0;
   int opt_overlap = 2;
   int opt_irr_all = 1; // MDH_OPT_IRRADIANCE_ALL
   int opt_jit = 1; // user-defined kinds: 1 = compile the MDH_X programs with hiprtc, 0 = interpret them (MDH_OPT_JIT)
   std::string jit_kinds; // mdh_jit_kinds.h of this scene (generated once)
   hipStream_t probe_stream = nullptr;   // radiance + irradiance passes of pipelined frames
   hipStream_t alt_stream = nullptr;     // screen pass of every other pipelined frame
   hipEvent_t ev_screen[NSETS] = {nullptr}, ev_probe[NSETS] = {nullptr}, ev_join = nullptr, ev_join_alt = nullptr;
   hipEvent_t ev_vol[NSETS] = {nullptr}; // the camera-only volumetric passes of a pipelined frame (on vol_stream, beside its probe passes)
   bool ev_screen_valid[NSETS] = {false};
   bool alt_pending = false; // work on alt_stream that `stream` has not been ordered after yet
   // an open frame (mdh_frame_begin .. mdh_frame_end)
   bool in_frame = false, frame_pipelined = false;
   bool in_frame_passes = false; // run_pass is called for the passes that end an open frame (frame_end_passes)
   void *d_rad_rec = nullptr; size_t rad_rec_bytes = 0; // MADARCH_HIP_RAD_SPLIT: the per-ray records between the radiance pass's two kernels
   bool fuse_scat_march = false; // the frame's visibility launch also marches the scattering texels' camera rays (k_visibility)
   int frame_cur = 0;
   int scr_parity = 0; // which screen stream / framebuffer the last pipelined frame drew on
   bool main_dirty = true; // work went to `stream` outside a pipelined frame since the probe stream last joined it
   // froxel and scattering textures, one per atlas set: the volumetric passes of a pipelined frame run on the
   // probe stream into the set that frame produces
   float *d_vis2[NSETS] = {nullptr};
   float4 *d_scat2[NSETS] = {nullptr};
   float4 *d_fb2[2] = {nullptr, nullptr}; // two framebuffers: consecutive pipelined frames draw on two streams
   int fb_last = 0;                       // the one the most recent frame drew
   // mdh_swap_buffers: a ring of RGBA8 copies of a framebuffer, each a device buffer (written on the stream that
   // drew the frame) and a pinned host buffer (filled on a stream of its own, so no screen pass queues behind a copy)
   static constexpr int FRONT_RING = 3;
   unsigned *d_front[FRONT_RING] = {nullptr, nullptr, nullptr};
   unsigned char *h_front[FRONT_RING] = {nullptr, nullptr, nullptr};
   hipEvent_t ev_front[FRONT_RING] = {nullptr, nullptr, nullptr}; // slot's host copy done
   hipEvent_t ev_packed = nullptr;
   hipStream_t copy_stream = nullptr;
   // MDH_OPT_WINDOW: the screen pass itself stores the RGBA8 pixels into a ring of pinned host buffers (over PCIe,
   // no copy and no extra kernel); a swap then only marks the last one with an event
   static constexpr int WIN_RING = 4;
   int opt_window = 2; // MDH_OPT_WINDOW: 0 never, 1 always, 2 from the first mdh_swap_buffers on
   unsigned *h_win[WIN_RING] = {nullptr, nullptr, nullptr, nullptr};
   hipEvent_t ev_win[WIN_RING] = {nullptr, nullptr, nullptr, nullptr};
   long long win_passes = 0; // screen passes that wrote a window slot; the last one wrote slot (win_passes - 1) % WIN_RING
   bool win_valid = false;   // the last screen pass wrote a window slot
   int win_owner[2] = {0, 1};
   // what mdh_front_buffer returns: set by mdh_swap_buffers
   const unsigned char *front_ptr = nullptr;
   hipEvent_t front_ev = nullptr;
   long long swaps = 0; // mdh_swap_buffers calls so far; the last one went to slot (swaps - 1) % FRONT_RING
   // (rank, world) whose tiles are the only non-zero pixels of a framebuffer; {0, 1}: every pixel may be set
   int fb_owner[2][2] = {{-1, -1}, {-1, -1}};
   // geometry buffer: [which framebuffer] x {index, t, steps} (int32 / float / int32 per pixel)
   void *d_gb2[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
   KScene ks{};
   // timing: event pairs recorded around every pass, resolved lazily (no host sync per pass)
   struct Pending { int pass; hipEvent_t e0, e1; };
   std::vector<Pending> pending;
   std::vector<hipEvent_t> free_events;
   double pass_ms[MDH_PASS_COUNT] = {0};
   long long pass_n[MDH_PASS_COUNT] = {0};
   hipStream_t own_stream = nullptr;
#ifdef MDH_DIAG
   unsigned long long work[MDH_PASS_COUNT][4] = {{0}}; // g_work of the last run of each pass (mdh_diag_work)
#endif
   // the communicator of a sharded run (mdh_comm_init): one rank per process and GPU.  With it mdh_render runs the
   // exchange of the atlas slices itself, on the probe stream, between the probe passes.
   ncclComm_t comm = nullptr;
   // mdh_comm_abort may come from a watchdog thread while the owning thread is inside a collective: the
   // watchdog then only aborts (ncclCommAbort, once) and raises the flag; the owning thread forgets the handle when its
   // call returns.  With nobody inside, the aborting thread owns the state and drops everything itself.
   std::atomic<bool> comm_aborted{false};
   std::atomic<int> comm_busy{0};
   double *d_comm_scratch = nullptr; // barrier / max reductions
   struct PeerState *peer = nullptr; // the peer exchange (mdh_peer_init): the same sharded frame, its exchange as copies
   bool irr_lds_granted = false; // k_irradiance may use up to 160 KiB of dynamic LDS on this renderer's device
};

// Order `stream` after everything pipelined frames put on the alternate screen stream (which
// itself waited for the probe stream): called before any operation outside a pipelined frame.
static int join_main(mdh_renderer *r)
{
   if (!r->alt_pending) return MDH_OK;
   HIP_TRY(hipEventRecord(r->ev_join_alt, r->alt_stream));
   HIP_TRY(hipStreamWaitEvent(r->stream, r->ev_join_alt, 0));
   r->alt_pending = false;
   return MDH_OK;
}
// host wait for everything this renderer has enqueued anywhere (its own streams only: other renderers of the process run on)
static int drain_streams(mdh_renderer *r)
{
   for (hipStream_t st : {r->probe_stream, r->alt_stream, r->query_stream, r->vol_stream, r->own_stream})
      if (st) HIP_TRY(hipStreamSynchronize(st));
   if (r->stream && r->stream != r->own_stream) HIP_TRY(hipStreamSynchronize(r->stream));
   return MDH_OK;
}
static hipEvent_t get_event(mdh_renderer *r)
{
   if (!r->free_events.empty()) { hipEvent_t e = r->free_events.back(); r->free_events.pop_back(); return e; }
   hipEvent_t e = nullptr;
   if (hipEventCreate(&e) != hipSuccess) return nullptr;
   return e;
}
// Fold event pairs into the per-pass totals.  wait = true: after every stream a pass can run on has drained, all of
// them; wait = false (the bound on the pending list, at a frame boundary): only the pairs whose end event has already
// completed, the rest stay pending.  Entries leave the list as they are folded, so an error half way leaves no pair behind
// whose events were already handed back.
static int resolve_timing(mdh_renderer *r, bool wait = true)
{
   if (r->pending.empty()) return MDH_OK;
   if (wait) {
      { int jr = join_main(r); if (jr != MDH_OK) return jr; }
      if (r->probe_stream) HIP_TRY(hipStreamSynchronize(r->probe_stream));
      if (r->alt_stream) HIP_TRY(hipStreamSynchronize(r->alt_stream));
      if (r->vol_stream) HIP_TRY(hipStreamSynchronize(r->vol_stream)); // (timed volumetric passes of pipelined frames)
      if (r->own_stream && r->own_stream != r->stream) HIP_TRY(hipStreamSynchronize(r->own_stream));
      HIP_TRY(hipStreamSynchronize(r->stream));
   }
   int rc = MDH_OK;
   size_t keep = 0;
   for (size_t i = 0; i < r->pending.size(); ++i) {
      const mdh_renderer::Pending p = r->pending[i];
      float ms = 0.0f;
      hipError_t e = rc != MDH_OK ? hipErrorNotReady : ((wait || hipEventQuery(p.e1) == hipSuccess) ? hipEventElapsedTime(&ms, p.e0, p.e1) : hipErrorNotReady);
      if (e == hipSuccess) {
         r->pass_ms[p.pass] += ms;
         r->pass_n[p.pass] += 1;
         r->free_events.push_back(p.e0);
         r->free_events.push_back(p.e1);
         continue;
      }
      if (e != hipErrorNotReady || wait) {
         if (rc == MDH_OK) { snprintf(g_err, sizeof g_err, "hipEventElapsedTime failed: %s", hipGetErrorString(e)); rc = MDH_E_DEVICE; }
      }
      r->pending[keep++] = p;
   }
   r->pending.resize(keep);
   if (!wait) (void)hipGetLastError(); // (a hipErrorNotReady of the queries is no error of the next launch)
   return rc;
}
// the bound on the pending list, applied where no frame is open
static int bound_timing(mdh_renderer *r) { return r->pending.size() >= 4096 ? resolve_timing(r, false) : MDH_OK; }

static int probe_total(const mdh_renderer *r) { return r->probes.probe_count[0] * r->probes.probe_count[1]; }
static size_t texel_bytes(const mdh_renderer *r) { return r->opt_atlas == 0 ? 4 : 16; }
static size_t atlas_bytes(const mdh_renderer *r, int tex)
{
   int res = tex == MDH_TEX_RADIANCE ? r->probes.radiance_resolution : r->probes.irradiance_resolution;
   return (size_t)probe_total(r) * res * res * texel_bytes(r);
}
static void own_probes(const mdh_renderer *r, int *b, int *e)
{
   long long P = probe_total(r);
   const long long rank = r->opt_rank < r->opt_world ? r->opt_rank : r->opt_world - 1; // (rank >= world is refused where a slice is used)
   *b = (int)(P * rank / r->opt_world);
   *e = (int)(P * (rank + 1) / r->opt_world);
}

// the partition table of a renderer: [cell][kinds + Index_Count] ints (what the reference's SSBO holds, mdh_read_partitioning),
// then the same candidate sets as bits, [cell][words] (partitioning_closest_bits, mdh_device.h)
// (the bits start on a 16-byte boundary and are a whole number of float4: workgroups stage them into LDS 16 bytes at a time)
static size_t part_table_ints(const mdh_renderer *r) { return ((size_t)r->part_cells * (r->npk + r->part.index_count) + 3) / 4 * 4; }
static int part_mask_words(const mdh_renderer *r)
{
   int declared = 0;
   for (int k = 0; k < r->npk; ++k) declared += r->pk[k].max_count;
   return declared > 32 ? (declared + 31) / 32 : 2; // (at least two: scenes of up to 64 primitives read a cell's bits as one 8-byte pair)
}
static size_t part_bits_ints(const mdh_renderer *r) { return ((size_t)r->part_cells * part_mask_words(r) + 3) / 4 * 4; }
static size_t part_buffer_ints(const mdh_renderer *r) { return part_table_ints(r) + part_bits_ints(r); }
// The bits of the whole grid are staged into LDS by every workgroup when they are small (the reference's simple_scene:
// 2000 cells x 2 words = 16 KB): a march step then reads no memory at all.  Larger grids read their cells' words from
// memory (L1 / L2).
#ifndef MDH_PART_BITS_LDS_MAX
#define MDH_PART_BITS_LDS_MAX 0 // (measured: off.  Staged, simple_scene's 16 KB leave room for five workgroups per CU instead of seven and
                                // its screen pass takes 0.78 instead of 0.70 ms, although no march step reads memory any more)
#endif
static int part_bits_f4(const mdh_renderer *r) { return r->part.enable && part_bits_ints(r) * 4 <= MDH_PART_BITS_LDS_MAX ? (int)(part_bits_ints(r) / 4) : 0; }
static float rd_f(const mdh_renderer *r, int off) { float f; memcpy(&f, r->scene_ubo.data() + off, 4); return f; }
static int rd_i(const mdh_renderer *r, int off) { int32_t i; memcpy(&i, r->scene_ubo.data() + off, 4); return i; }
static float4 mk4(float x, float y, float z, float w) { float4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
static float4 rd_v3w(const mdh_renderer *r, int off, float w) { return mk4(rd_f(r, off), rd_f(r, off + 4), rd_f(r, off + 8), w); }
static float i_as_f(int i) { float f; memcpy(&f, &i, 4); return f; }

// Repack the std140 images into the float4 table the kernels stage into LDS
// (layout in mdh_device.h) and refresh the SGPR header.
static int stream_index(const mdh_renderer *r, hipStream_t st) { return st == r->probe_stream ? 1 : (st == r->alt_stream ? 2 : (st == r->query_stream ? 3 : (st == r->vol_stream ? 4 : 0))); }
// before a kernel that stages the table is launched on `st`: order `st` after the table's upload
static int table_acquire(mdh_renderer *r, hipStream_t st)
{
   const int si = stream_index(r, st);
   if (r->tab_seen[si] != r->table_version) {
      if (st != r->table_stream) HIP_TRY(hipStreamWaitEvent(st, r->ev_table, 0));
      r->tab_seen[si] = r->table_version;
   }
   if (r->part.enable && r->part_seen[si] != r->part_version) { // ... and after the build of the partition table
      if (st != r->part_stream) HIP_TRY(hipStreamWaitEvent(st, r->ev_part, 0));
      r->part_seen[si] = r->part_version;
   }
   return MDH_OK;
}
// after it: the table buffer is in use on `st` until this point of the stream
static int table_release(mdh_renderer *r, hipStream_t st)
{
   const int si = stream_index(r, st);
   HIP_TRY(hipEventRecord(r->tab_done[r->tab_slot][si], st));
   r->tab_used[r->tab_slot][si] = true;
   if (r->part.enable) {
      HIP_TRY(hipEventRecord(r->part_done[r->part_slot][si], st));
      r->part_used[r->part_slot][si] = true;
   }
   return MDH_OK;
}

static int commit_scene(mdh_renderer *r, hipStream_t up)
{
   KScene &s = r->ks;
   s.max_dist = r->max_dist;
   std::vector<float4> &t = r->table_host;
   t.clear();
   int H[H_INTS] = {0};
   t.resize(H_INTS / 4); // the int header, filled in at the end
   H[H_NK] = r->npk;
   H[H_NL] = r->nlk;
   for (int ty = 0; ty < 4; ++ty) { s.tcount[ty] = 0; s.tslot[ty] = 0; }
   // geometry: kind by kind, the elements below the runtime count
   for (int k = 0; k < r->npk; ++k) {
      const Kind &kd = r->pk[k];
      int n = rd_i(r, kd.count_off);
      if (n < 0) n = 0;
      if (n > kd.max_count) n = kd.max_count;
      H[H_KTYPE + k] = kd.type; H[H_KCOUNT + k] = n; H[H_KBASE + k] = r->prim_base[k]; H[H_KMAX + k] = kd.max_count; H[H_KSLOT + k] = (int)t.size();
      if (kd.type == PK_CUSTOM) { // instances packed as MDH_X_COMP addresses them: components in order, vec3 = 3 floats
         const int stride = (kd.inst_floats + 3) / 4;
         H[H_KSTRIDE + k] = stride;
         for (int i = 0; i < n; ++i) {
            const int b = kd.array_off + kd.stride * i;
            float buf[32] = {0};
            int f = 0;
            for (int c = 0; c < kd.ncomp; ++c) {
               const int nf = kd.comp_kind[c] == MDH_VEC3 ? 3 : 1;
               memcpy(buf + f, r->scene_ubo.data() + b + kd.comp_off[c], 4 * nf); // ints travel as raw bits
               f += nf;
            }
            for (int q = 0; q < stride; ++q) t.push_back(mk4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]));
         }
         continue;
      }
      H[H_KSTRIDE + k] = kd.type == PK_TRIANGLE ? 3 : (kd.type == PK_BOX ? 2 : 1);
      s.tcount[kd.type] = n; s.tslot[kd.type] = (int)t.size();
      for (int i = 0; i < n; ++i) {
         int b = kd.array_off + kd.stride * i;
         switch (kd.type) {
         case PK_SPHERE: t.push_back(rd_v3w(r, b + kd.f_a, rd_f(r, b + kd.f_b))); break;
         case PK_PLANE: t.push_back(rd_v3w(r, b + kd.f_a, rd_f(r, b + kd.f_b))); break;
         case PK_BOX: t.push_back(rd_v3w(r, b + kd.f_a, 0.0f)); t.push_back(rd_v3w(r, b + kd.f_b, 0.0f)); break;
         default: t.push_back(rd_v3w(r, b + kd.f_a, 0.0f)); t.push_back(rd_v3w(r, b + kd.f_b, 0.0f)); t.push_back(rd_v3w(r, b + kd.f_c, 0.0f)); break;
         }
      }
   }
   // axis-aligned planes fold into six offsets; the others are repeated as "general planes"
   // for closest_primitive (the per-kind copy above stays complete for the arg-min / normals)
   {
      const float inf = INFINITY;
      for (int g = 0; g < 6; ++g) { s.axis_off[g] = inf; H[H_AXIS_IDX + g] = -1; }
      int per_dir[6] = {0, 0, 0, 0, 0, 0};
      s.n_axis = 0;
      s.gplane_slot = (int)t.size();
      s.gplane_count = 0;
      for (int k = 0; k < r->npk; ++k) {
         const Kind &kd = r->pk[k];
         if (kd.type != PK_PLANE) continue;
         for (int i = 0; i < H[H_KCOUNT + k]; ++i) {
            int b = kd.array_off + kd.stride * i;
            float n[3] = {rd_f(r, b + kd.f_a), rd_f(r, b + kd.f_a + 4), rd_f(r, b + kd.f_a + 8)};
            float o = rd_f(r, b + kd.f_b);
            int axis = -1, nz = 0;
            for (int c = 0; c < 3; ++c)
               if (n[c] != 0.0f) { ++nz; axis = c; }
            if (nz == 1 && (n[axis] == 1.0f || n[axis] == -1.0f) && o == o) {
               int g = 2 * axis + (n[axis] < 0.0f ? 1 : 0);
               if (o < s.axis_off[g]) { s.axis_off[g] = o; H[H_AXIS_IDX + g] = r->prim_base[k] + i; }
               ++per_dir[g];
               ++s.n_axis;
            } else {
               t.push_back(mk4(n[0], n[1], n[2], o));
               ++s.gplane_count;
            }
         }
      }
      // the typed arg-min of closest_primitive_info needs every plane to be one of the six folded ones
      // (two planes of one direction can tie after rounding, and the tie goes to the lower index)
      H[H_FASTINFO] = s.gplane_count == 0;
      for (int g = 0; g < 6; ++g)
         if (per_dir[g] > 1) H[H_FASTINFO] = 0;
      for (int ty = 0; ty < 4; ++ty) H[H_TBASE + ty] = 0;
      for (int k = 0; k < r->npk; ++k) {
         if (r->pk[k].type == PK_CUSTOM) H[H_FASTINFO] = 0;
         else H[H_TBASE + r->pk[k].type] = r->prim_base[k];
      }
   }
   // material ids (int32), 4 per float4
   // the MDH_X programs of the user-defined kinds (ints, read through hdr ())
   for (int k = 0; k < r->npk; ++k) {
      const Kind &kd = r->pk[k];
      if (kd.type != PK_CUSTOM) continue;
      const std::vector<int32_t> *progs[3] = {&kd.x_dist, &kd.x_nrm, &kd.x_mat};
      const int hs[3] = {H_XDIST, H_XNRM, H_XMAT}, hn[3] = {H_XDISTN, H_XNRMN, H_XMATN};
      for (int q = 0; q < 3; ++q) {
         H[hs[q] + k] = (int)t.size() * 4;
         H[hn[q] + k] = (int)progs[q]->size();
         for (size_t w0 = 0; w0 < progs[q]->size(); w0 += 4) {
            float m[4] = {0, 0, 0, 0};
            for (size_t j = 0; j < 4 && w0 + j < progs[q]->size(); ++j) m[j] = i_as_f((*progs[q])[w0 + j]);
            t.push_back(mk4(m[0], m[1], m[2], m[3]));
         }
      }
   }
   for (int k = 0; k < r->npk; ++k) {
      const Kind &kd = r->pk[k];
      if (kd.type == PK_CUSTOM) continue;
      H[H_KMAT + k] = (int)t.size() * 4;
      for (int i0 = 0; i0 < H[H_KCOUNT + k]; i0 += 4) {
         float m[4] = {0, 0, 0, 0};
         for (int j = 0; j < 4 && i0 + j < H[H_KCOUNT + k]; ++j) m[j] = i_as_f(rd_i(r, kd.array_off + kd.stride * (i0 + j) + kd.f_d));
         t.push_back(mk4(m[0], m[1], m[2], m[3]));
      }
   }
   for (int k = 0; k < r->nlk; ++k) {
      const Kind &kd = r->lk[k];
      int n = rd_i(r, kd.count_off);
      if (n < 0) n = 0;
      if (n > kd.max_count) n = kd.max_count;
      H[H_LTYPE + k] = kd.type; H[H_LCOUNT + k] = n; H[H_LSLOT + k] = (int)t.size();
      if (kd.type == LK_CUSTOM) { // instance floats as MDH_X_COMP addresses them, then the two programs
         const int stride = (kd.inst_floats + 3) / 4;
         H[H_LSTRIDE + k] = stride;
         for (int i = 0; i < n; ++i) {
            const int b = kd.array_off + kd.stride * i;
            float buf[32] = {0};
            int f = 0;
            for (int c = 0; c < kd.ncomp; ++c) {
               const int nf = kd.comp_kind[c] == MDH_VEC3 ? 3 : 1;
               memcpy(buf + f, r->scene_ubo.data() + b + kd.comp_off[c], 4 * nf);
               f += nf;
            }
            for (int q = 0; q < stride; ++q) t.push_back(mk4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]));
         }
         const std::vector<int32_t> *progs[2] = {&kd.x_dist, &kd.x_nrm};
         const int hs[2] = {H_XLSAMPLE, H_XLPOS}, hn[2] = {H_XLSAMPLEN, H_XLPOSN};
         for (int q = 0; q < 2; ++q) {
            H[hs[q] + k] = (int)t.size() * 4;
            H[hn[q] + k] = (int)progs[q]->size();
            for (size_t w0 = 0; w0 < progs[q]->size(); w0 += 4) {
               float m4[4] = {0, 0, 0, 0};
               for (size_t j = 0; j < 4 && w0 + j < progs[q]->size(); ++j) m4[j] = i_as_f((*progs[q])[w0 + j]);
               t.push_back(mk4(m4[0], m4[1], m4[2], m4[3]));
            }
         }
         continue;
      }
      for (int i = 0; i < n; ++i) {
         int b = kd.array_off + kd.stride * i;
         if (kd.type == LK_POINT) { t.push_back(rd_v3w(r, b + kd.f_a, 0.0f)); t.push_back(rd_v3w(r, b + kd.f_b, 0.0f)); }
         else { t.push_back(rd_v3w(r, b + kd.f_a, rd_f(r, b + kd.f_c))); t.push_back(rd_v3w(r, b + kd.f_b, 0.0f)); t.push_back(rd_v3w(r, b + kd.f_d, 0.0f)); }
      }
   }
   s.total_lights = rd_i(r, r->total_light_off);
   s.mat_slot = (int)t.size();
   for (int m = 0; m < MAX_MATERIALS; ++m) {
      float f[5];
      memcpy(f, r->materials_ubo + 16 + 32 * m, 20);
      t.push_back(mk4(f[0], f[1], f[2], f[3]));
      t.push_back(mk4(f[4], 0, 0, 0));
   }
#if MDH_U8_FMA
   s.u8_slot = -1; // (RGB8 texels are decoded in registers: u8_unorm, mdh_device.h)
#else
   // k / 255 for k = 0..255: the RGB8 texel decode, one correctly rounded division each
   s.u8_slot = (int)t.size();
   for (int k = 0; k < 256; k += 4) t.push_back(mk4((float)k / 255.0f, (float)(k + 1) / 255.0f, (float)(k + 2) / 255.0f, (float)(k + 3) / 255.0f));
#endif
   for (int k = 0; k < r->npk; ++k) { H[H_KQUAD + 4 * k] = H[H_KTYPE + k]; H[H_KQUAD + 4 * k + 1] = H[H_KSLOT + k]; H[H_KQUAD + 4 * k + 2] = H[H_KBASE + k]; H[H_KQUAD + 4 * k + 3] = H[H_KMAX + k]; }
   memcpy(t.data(), H, sizeof H);
   s.table_f4 = (int)t.size();
   { // (MDH_SDF_SGPR: the words closest_primitive would read from the table; with a count of 0 they belong to the next kind and are not used)
      const size_t ss = (size_t)s.tslot[PK_SPHERE], sb = (size_t)s.tslot[PK_BOX];
      const float4 z = mk4(0, 0, 0, 0), a = ss < t.size() ? t[ss] : z, b0 = sb < t.size() ? t[sb] : z, b1 = sb + 1 < t.size() ? t[sb + 1] : z;
      const float fs[4] = {a.x, a.y, a.z, a.w}, fb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      memcpy(s.first_sphere, fs, sizeof fs); memcpy(s.first_box, fb, sizeof fb);
   }
   // (the march kernels park MDH_PARK_DWORDS floats per thread behind the table, lds_bytes_march)
   if ((size_t)(s.table_f4 + part_bits_f4(r)) * 16 + (size_t)MDH_SCR_PARK_ROWS * MDH_BLOCK * sizeof(float) > 64 * 1024)
      return seterr(MDH_E_INVALID, "scene tables exceed the 64 KiB LDS budget of a workgroup");
   if (t.size() > r->table_cap) { // grow the whole ring (rare: the table only grows with the primitive counts)
      { int dr = drain_streams(r); if (dr != MDH_OK) return dr; }
      r->table_cap = t.size() + 256;
      for (int q = 0; q < mdh_renderer::TAB_RING; ++q) {
         if (r->d_table_ring[q]) HIP_TRY(hipFree(r->d_table_ring[q]));
         if (r->h_table_ring[q]) HIP_TRY(hipHostFree(r->h_table_ring[q]));
         HIP_TRY(hipMalloc(&r->d_table_ring[q], r->table_cap * sizeof(float4)));
         HIP_TRY(hipHostMalloc((void **)&r->h_table_ring[q], r->table_cap * sizeof(float4), hipHostMallocDefault));
         for (int si = 0; si < mdh_renderer::NSTREAMS; ++si) r->tab_used[q][si] = false;
      }
   }
   const int ns = (r->tab_slot + 1) % mdh_renderer::TAB_RING;
   for (int si = 0; si < mdh_renderer::NSTREAMS; ++si)
      if (r->tab_used[ns][si]) { HIP_TRY(hipEventSynchronize(r->tab_done[ns][si])); r->tab_used[ns][si] = false; }
   memcpy(r->h_table_ring[ns], t.data(), t.size() * sizeof(float4));
   HIP_TRY(hipMemcpyAsync(r->d_table_ring[ns], r->h_table_ring[ns], t.size() * sizeof(float4), hipMemcpyHostToDevice, up));
   HIP_TRY(hipEventRecord(r->ev_table, up));
   r->table_stream = up;
   ++r->table_version;
   r->tab_seen[stream_index(r, up)] = r->table_version;
   r->tab_slot = ns;
   s.table = r->d_table_ring[ns];
   s.part_enable = r->part.enable;
   s.part_border = r->part.border_behavior;
   s.part_index_count = r->part.index_count;
   s.part_cells = r->part_cells;
   for (int a = 0; a < 3; ++a) { s.part_dims[a] = r->part.grid_dimensions[a]; s.part_sp[a] = r->pg_spacing[a]; s.part_off[a] = r->pg_offset[a]; }
   s.part_sp_pow2 = 1;
   for (int a = 0; a < 3; ++a) {
      int e = 0;
      const float mant = frexpf(r->pg_spacing[a], &e);
      const bool pow2 = mant == 0.5f && e > -100 && e < 100; // 2^(e-1), well inside the normal range
      s.part_inv_sp[a] = pow2 ? 1.0f / r->pg_spacing[a] : 0.0f;
      if (!pow2) s.part_sp_pow2 = 0;
   }
   // the grid's dimensions as the fp32 values the cell index is computed with (uniform conversions the kernels would repeat at every march step)
   for (int a = 0; a < 3; ++a) s.part_fdims[a] = (float)r->part.grid_dimensions[a];
   s.part_fyz = (float)(r->part.grid_dimensions[1] * r->part.grid_dimensions[2]);
   s.part_table = r->d_part_ring[r->part_slot];
   s.part_mask_off = (int)part_table_ints(r);
   s.part_mask_words = part_mask_words(r);
   s.part_bits_f4 = part_bits_f4(r);
   { // KScene::part_small: one 8-byte load per lookup, a shift and a mask per built-in type
      int declared = 0;
      bool small = r->part.enable != 0;
      for (int ty = 0; ty < 4; ++ty) { s.part_tbit[ty] = 0u; s.part_tmask[ty] = 0u; }
      for (int k = 0; k < r->npk; ++k) {
         const Kind &kd = r->pk[k];
         if (kd.type == PK_CUSTOM || kd.max_count > 32) small = false;
         else if (kd.max_count > 0 && s.part_tmask[kd.type]) small = false; // (two kinds of one built-in type: one shift and mask cannot name both)
         else if (kd.max_count > 0) { s.part_tbit[kd.type] = (unsigned)r->prim_base[k]; s.part_tmask[kd.type] = kd.max_count == 32 ? 0xffffffffu : ((1u << kd.max_count) - 1u); }
         declared += kd.max_count;
      }
      s.part_small = small && declared <= 64 && part_mask_words(r) == 2 ? 1 : 0;
   }
   r->table_dirty = false;
   return MDH_OK;
}

static KProbes make_probes(const mdh_renderer *r)
{
   KProbes p;
   p.pcx = r->probes.probe_count[0]; p.pcy = r->probes.probe_count[1];
   p.gx = r->probes.grid_dimensions[0]; p.gy = r->probes.grid_dimensions[1]; p.gz = r->probes.grid_dimensions[2];
   p.sx = r->probes.grid_spacing[0]; p.sy = r->probes.grid_spacing[1]; p.sz = r->probes.grid_spacing[2];
   p.rres = r->probes.radiance_resolution; p.ires = r->probes.irradiance_resolution;
   p.fmt = r->opt_atlas;
   auto log2_or_neg = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
   p.rshift = log2_or_neg(p.rres); p.ishift = log2_or_neg(p.ires); p.pcx_shift = log2_or_neg(p.pcx);
   p.inv_pcx = log2_or_neg(p.pcx) >= 0 ? 1.0f / (float)p.pcx : 0.0f;
   p.inv_pcy = log2_or_neg(p.pcy) >= 0 ? 1.0f / (float)p.pcy : 0.0f;
   p.rad = r->d_rad2[r->last]; p.irr = r->d_irr2[r->last];
   own_probes(r, &p.probe_begin, &p.probe_end);
   // the uniform fp32 values of the atlas taps: the kernels' own expressions, evaluated here once (IEEE, one rounding each)
   p.irr_lo = 0.5f / (float)p.ires; p.irr_hi = 1.0f - p.irr_lo;
   p.rad_lo = 0.5f / (float)p.rres; p.rad_hi = 1.0f - p.rad_lo;
   p.irr_w = (float)(p.pcx * p.ires); p.irr_h = (float)(p.pcy * p.ires);
   p.rad_w = (float)(p.pcx * p.rres); p.rad_h = (float)(p.pcy * p.rres);
   p.fpcx = (float)p.pcx; p.fpcy = (float)p.pcy;
   p.rad_lods = 0;
   while ((2 << p.rad_lods) <= p.rres) ++p.rad_lods;
   // div_magic: exact for numerators below 65536 -- texel coordinates of an atlas image, probe ids
   auto magic = [](int d) { return (unsigned)((1ull << 32) / (unsigned long long)d + 1ull); };
   const bool small = (long long)p.pcx * p.rres < 65536 && (long long)p.pcy * p.rres < 65536 && (long long)p.pcx * p.ires < 65536 &&
                      (long long)p.pcy * p.ires < 65536 && (long long)p.pcx * p.pcy < 65536;
   p.m_rres = small && p.rres > 1 ? magic(p.rres) : 0u;
   p.m_ires = small && p.ires > 1 ? magic(p.ires) : 0u;
   p.m_pcx = small && p.pcx > 1 ? magic(p.pcx) : 0u;
   p.rad_mips = nullptr; // (the screen pass sets it: run_pass)
   return p;
}
static KCamera make_camera(const mdh_renderer *r)
{
   KCamera c;
   c.px = r->cam_pos[0]; c.py = r->cam_pos[1]; c.pz = r->cam_pos[2];
   memcpy(c.m, r->cam_m, sizeof c.m);
   return c;
}
static KVolumetrics make_vol(const mdh_renderer *r, bool enabled, int set)
{
   KVolumetrics v;
   v.enabled = enabled ? 1 : 0;
   v.vw = r->vol.visibility_resolution[0]; v.vh = r->vol.visibility_resolution[1]; v.vz = r->vol.visibility_resolution[2];
   v.sw = r->vol.scattering_resolution[0]; v.sh = r->vol.scattering_resolution[1];
   v.vstep = r->vstep; v.sstep = r->sstep;
   v.vis = r->d_vis2[set]; v.scat = r->d_scat2[set];
   return v;
}

static int alloc_atlases(mdh_renderer *r)
{
   for (int s = 0; s < mdh_renderer::NSETS; ++s) {
      if (r->d_rad2[s]) HIP_TRY(hipFree(r->d_rad2[s]));
      if (r->d_irr2[s]) HIP_TRY(hipFree(r->d_irr2[s]));
      r->d_rad2[s] = r->d_irr2[s] = nullptr;
      HIP_TRY(hipMalloc(&r->d_rad2[s], atlas_bytes(r, MDH_TEX_RADIANCE)));
      HIP_TRY(hipMalloc(&r->d_irr2[s], atlas_bytes(r, MDH_TEX_IRRADIANCE)));
      // Load_Empty_Texture (render_passes.adb:115-116): contents start as zeros here
      HIP_TRY(hipMemsetAsync(r->d_rad2[s], 0, atlas_bytes(r, MDH_TEX_RADIANCE), r->stream));
      HIP_TRY(hipMemsetAsync(r->d_irr2[s], 0, atlas_bytes(r, MDH_TEX_IRRADIANCE), r->stream));
   }
   r->main_dirty = true;
   return MDH_OK;
}

// MDH_OPT_RADIANCE_MIPS: texels of levels 1 .. radiance_lods together (a third of level 0 at most)
static size_t rad_mips_texels(const mdh_renderer *r)
{
   size_t n = 0;
   for (int res = r->probes.radiance_resolution >> 1; res >= 1; res >>= 1) n += (size_t)probe_total(r) * res * res;
   return n;
}
static int alloc_rad_mips(mdh_renderer *r)
{
   for (int s = 0; s < mdh_renderer::NSETS; ++s) {
      if (r->d_rad_mips[s]) { void *q = r->d_rad_mips[s]; r->d_rad_mips[s] = nullptr; HIP_TRY(hipFree(q)); }
      if (r->opt_mips && rad_mips_texels(r) > 0) HIP_TRY(hipMalloc(&r->d_rad_mips[s], rad_mips_texels(r) * texel_bytes(r)));
   }
   return MDH_OK;
}
// the levels of set `set`'s radiance atlas on stream st (before the screen pass that reads them)
static int build_rad_mips(mdh_renderer *r, int set, hipStream_t st)
{
   const char *src = (const char *)r->d_rad2[set];
   char *dst = (char *)r->d_rad_mips[set];
   for (int res = r->probes.radiance_resolution >> 1; res >= 1; res >>= 1) {
      const long n = (long)probe_total(r) * res * res;
      hipLaunchKernelGGL(k_radiance_mips, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const void *)src, (void *)dst, r->opt_atlas, res, (int)n);
      src = dst;
      dst += (size_t)n * texel_bytes(r);
   }
   HIP_TRY(hipGetLastError());
   return MDH_OK;
}

static int rccl_api_destroy(ncclComm_t c); // (mdh_comm_* below)
static void peer_drop(mdh_renderer *r);      // (mdh_peer_* below)
static bool peer_active(const mdh_renderer *r);
static int peer_check(mdh_renderer *r);
extern "C" int32_t mdh_destroy(mdh_renderer *r)
{
   if (!r) return MDH_OK;
   (void)hipSetDevice(r->device);
   if (r->probe_stream) (void)hipStreamSynchronize(r->probe_stream);
   if (r->alt_stream) (void)hipStreamSynchronize(r->alt_stream);
   if (r->stream) (void)hipStreamSynchronize(r->stream);
   if (r->query_stream) (void)hipStreamSynchronize(r->query_stream);
   if (r->vol_stream) (void)hipStreamSynchronize(r->vol_stream);
   if (r->comm) { ncclComm_t c = r->comm; r->comm = nullptr; (void)rccl_api_destroy(c); }
   if (r->d_comm_scratch) (void)hipFree(r->d_comm_scratch);
   if (r->d_rad_rec) (void)hipFree(r->d_rad_rec);
   peer_drop(r);
   void *ptrs[] = {r->d_table_ring[0], r->d_table_ring[1], r->d_table_ring[2], r->d_table_ring[3], r->d_part_ring[0], r->d_part_ring[1], r->d_part_ring[2], r->d_part_ring[3], r->d_warn, r->d_query, r->d_irr_taps, r->d_rad_steps, r->d_rad_order, r->d_rad_hist, r->d_scr_cost, r->d_scr_order[0], r->d_scr_order[1], r->d_scr_hist, r->d_fb2[0], r->d_fb2[1], r->d_gb2[0][0], r->d_gb2[0][1], r->d_gb2[0][2], r->d_gb2[1][0], r->d_gb2[1][1], r->d_gb2[1][2]};
   for (void *p : ptrs)
      if (p) (void)hipFree(p);
   for (int q = 0; q < mdh_renderer::NSETS; ++q)
      for (void *p : {(void *)r->d_rad2[q], (void *)r->d_irr2[q], (void *)r->d_vis2[q], (void *)r->d_scat2[q], (void *)r->d_rad_mips[q]})
         if (p) (void)hipFree(p);
   for (auto &p : r->pending) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
   for (auto e : r->free_events) (void)hipEventDestroy(e);
   for (hipEvent_t e : {r->ev_join, r->ev_join_alt, r->ev_table})
      if (e) (void)hipEventDestroy(e);
   for (int q = 0; q < mdh_renderer::NSETS; ++q)
      for (hipEvent_t e : {r->ev_screen[q], r->ev_probe[q], r->ev_vol[q]})
         if (e) (void)hipEventDestroy(e);
   for (int q = 0; q < mdh_renderer::TAB_RING; ++q) {
      if (r->h_table_ring[q]) (void)hipHostFree(r->h_table_ring[q]);
      for (int si = 0; si < mdh_renderer::NSTREAMS; ++si)
         if (r->tab_done[q][si]) (void)hipEventDestroy(r->tab_done[q][si]);
   }
   for (int q = 0; q < mdh_renderer::PART_RING; ++q)
      for (int si = 0; si < mdh_renderer::NSTREAMS; ++si)
         if (r->part_done[q][si]) (void)hipEventDestroy(r->part_done[q][si]);
   if (r->ev_scr_sort) (void)hipEventDestroy(r->ev_scr_sort);
   if (r->ev_scr_other) (void)hipEventDestroy(r->ev_scr_other);
   if (r->ev_part) (void)hipEventDestroy(r->ev_part);
   if (r->ev_warn) (void)hipEventDestroy(r->ev_warn);
   if (r->h_warn) (void)hipHostFree(r->h_warn);
   if (r->query_stream) (void)hipStreamDestroy(r->query_stream);
   if (r->vol_stream) (void)hipStreamDestroy(r->vol_stream);
   for (int q = 0; q < mdh_renderer::WIN_RING; ++q) {
      if (r->h_win[q]) (void)hipHostFree(r->h_win[q]);
      if (r->ev_win[q]) (void)hipEventDestroy(r->ev_win[q]);
   }
   if (r->copy_stream) { (void)hipStreamSynchronize(r->copy_stream); (void)hipStreamDestroy(r->copy_stream); }
   if (r->ev_packed) (void)hipEventDestroy(r->ev_packed);
   for (int q = 0; q < mdh_renderer::FRONT_RING; ++q) {
      if (r->d_front[q]) (void)hipFree(r->d_front[q]);
      if (r->h_front[q]) (void)hipHostFree(r->h_front[q]);
      if (r->ev_front[q]) (void)hipEventDestroy(r->ev_front[q]);
   }
   if (r->probe_stream) (void)hipStreamDestroy(r->probe_stream);
   if (r->alt_stream) (void)hipStreamDestroy(r->alt_stream);
   if (r->own_stream) (void)hipStreamDestroy(r->own_stream);
   delete r;
   return MDH_OK;
}

// Renderers.Create (renderers.adb:91-300) + Scenes.Compile (scenes.adb:1378-1421)
extern "C" int32_t mdh_create(int32_t width, int32_t height, const mdh_scene_desc *scene, const mdh_probe_settings *probes,
                              const mdh_volumetrics *vol, int32_t device, mdh_renderer **out)
{
   if (!scene || !probes || !vol || !out || width <= 0 || height <= 0) return seterr(MDH_E_INVALID, "bad argument");
   if (scene->n_prim_kinds < 0 || scene->n_prim_kinds > MDH_MAX_KINDS || scene->n_light_kinds < 0 || scene->n_light_kinds > MDH_MAX_LIGHT_KINDS)
      return seterr(MDH_E_INVALID, "too many kinds");
   // Setup_Probe_Layout, renderers.adb:54-65
   if (probes->grid_dimensions[0] * probes->grid_dimensions[1] * probes->grid_dimensions[2] != probes->probe_count[0] * probes->probe_count[1])
      return seterr(MDH_E_PROBE_MISMATCH, "Probe_Count should match grid dimensions.");
   if (probes->radiance_resolution < 1 || probes->irradiance_resolution < 1 || probes->probe_count[0] < 1 || probes->probe_count[1] < 1)
      return seterr(MDH_E_INVALID, "bad probe settings");
   mdh_renderer *r = new mdh_renderer();
   r->W = width; r->H = height; r->device = device;
   r->npk = scene->n_prim_kinds; r->nlk = scene->n_light_kinds;
   // Compute_Scene_GPU_Type (scenes.adb:1268-1345)
   int off = 0, base = 0;
   for (int k = 0; k < r->npk; ++k) {
      if (!resolve_kind(r->pk[k], scene->prim_kinds[k], false)) { delete r; return seterr(MDH_E_UNSUPPORTED_KIND, "primitive kind is neither one of Sphere, Plane, Box, Triangle with their components nor a user-defined kind with three valid MDH_X programs"); }
      off = pad_to(off, 4); r->pk[k].count_off = off; off += 4;
      off = pad_to(off, 16); r->pk[k].array_off = off; off += r->pk[k].stride * r->pk[k].max_count;
      r->prim_base[k] = base; base += r->pk[k].max_count;
      if (r->pk[k].max_count > 4095) { delete r; return seterr(MDH_E_INVALID, "declared primitive count above 4095"); }
   }
   for (int k = 0; k < r->npk; ++k)
      for (int j = 0; j < k; ++j)
         if (r->pk[k].type != PK_CUSTOM && r->pk[k].type == r->pk[j].type) { delete r; return seterr(MDH_E_INVALID, "a built-in primitive kind is declared twice"); }
   for (int k = 0; k < r->nlk; ++k) {
      if (!resolve_kind(r->lk[k], scene->light_kinds[k], true)) { delete r; return seterr(MDH_E_UNSUPPORTED_KIND, "light kind is neither PointLight or SpotLight with their components nor a user-defined kind with valid Sample and Position programs"); }
      off = pad_to(off, 4); r->lk[k].count_off = off; off += 4;
      off = pad_to(off, 16); r->lk[k].array_off = off; off += r->lk[k].stride * r->lk[k].max_count;
   }
   off = pad_to(off, 4); r->total_light_off = off; off += 4;
   r->scene_ubo.assign((size_t)off, 0);
   r->max_dist = image_roundtrip(scene->max_dist);
   r->part = scene->partitioning;
   r->probes = *probes;
   r->vol = *vol;
   r->vstep = image_roundtrip(vol->visibility_step_size);
   r->sstep = image_roundtrip(vol->scattering_step_size);
   if (r->part.enable) {
      if (r->part.index_count < 1 || r->part.grid_dimensions[0] < 1 || r->part.grid_dimensions[1] < 1 || r->part.grid_dimensions[2] < 1) { delete r; return seterr(MDH_E_INVALID, "bad partitioning settings"); }
      for (int a = 0; a < 3; ++a) { r->pg_spacing[a] = image_roundtrip(r->part.grid_spacing[a]); r->pg_offset[a] = image_roundtrip(r->part.grid_offset[a]); }
      r->part_cells = r->part.grid_dimensions[0] * r->part.grid_dimensions[1] * r->part.grid_dimensions[2];
      float sx = r->part.grid_spacing[0], sy = r->part.grid_spacing[1], sz = r->part.grid_spacing[2];
      r->part_gpu_diag = image_roundtrip(sqrtf((sx * sx + sy * sy) + sz * sz));
   }
   // ---- device
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { delete r; return seterr(MDH_E_NO_DEVICE, "no HIP device visible; libmadarch_hip has no CPU path"); }
   if (device < 0 || device >= ndev) { delete r; return seterr(MDH_E_NO_DEVICE, "device ordinal out of range"); }
   hipDeviceProp_t prop;
   if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) { delete r; return seterr(MDH_E_NO_DEVICE, "device is not gfx950 (MI355X); the kernels are built for gfx950 only"); }
   r->n_cus = prop.multiProcessorCount;
   int rc = MDH_OK;
   auto fail = [&](int code) { mdh_destroy(r); return code; };
#define TRY_OR_FAIL(expr)                                                                                          \
   do {                                                                                                            \
      hipError_t e_ = (expr);                                                                                      \
      if (e_ != hipSuccess) {                                                                                      \
         snprintf(g_err, sizeof g_err, "%s failed: %s", #expr, hipGetErrorString(e_));                             \
         return fail(MDH_E_DEVICE);                                                                                \
      }                                                                                                            \
   } while (0)
   TRY_OR_FAIL(hipSetDevice(device));
   TRY_OR_FAIL(hipStreamCreateWithFlags(&r->own_stream, hipStreamNonBlocking));
   r->stream = r->own_stream;
   {
      int lo = 0, hi = 0; // numerically lower = higher priority
      TRY_OR_FAIL(hipDeviceGetStreamPriorityRange(&lo, &hi));
      // the probe passes are the head of each frame's dependency chain: high priority (measured on MI355X:
      // 3340 vs 3300 Mpix/s at BASELINE config 3).  MADARCH_HIP_PROBE_PRIORITY = -1 high, 0 default, 1 low
      const char *pe = getenv("MADARCH_HIP_PROBE_PRIORITY");
      int prio = pe ? atoi(pe) : -1;
      prio = prio < 0 ? hi : (prio > 0 ? lo : 0);
      TRY_OR_FAIL(hipStreamCreateWithPriority(&r->probe_stream, hipStreamNonBlocking, prio));
      TRY_OR_FAIL(hipStreamCreateWithFlags(&r->alt_stream, hipStreamNonBlocking));
      for (hipEvent_t *e : {&r->ev_join, &r->ev_join_alt, &r->ev_table}) TRY_OR_FAIL(hipEventCreateWithFlags(e, hipEventDisableTiming));
      for (int q = 0; q < mdh_renderer::NSETS; ++q)
         for (hipEvent_t *e : {&r->ev_screen[q], &r->ev_probe[q], &r->ev_vol[q]}) TRY_OR_FAIL(hipEventCreateWithFlags(e, hipEventDisableTiming));
      // (HIP maps a process's streams onto a few hardware queues, four by default: a fifth stream shares one with another
      //  and their work serialises -- measured: light_shafts lost a fifth of its frame rate.  So the volumetric stream exists
      //  only with volumetrics, and the query stream from the first Eval_Distance_To on.)
      if (vol->enabled) TRY_OR_FAIL(hipStreamCreateWithFlags(&r->vol_stream, hipStreamNonBlocking));
      for (hipEvent_t *e : {&r->ev_part, &r->ev_warn}) TRY_OR_FAIL(hipEventCreateWithFlags(e, hipEventDisableTiming));
      for (int q = 0; q < mdh_renderer::TAB_RING; ++q)
         for (int si = 0; si < mdh_renderer::NSTREAMS; ++si) TRY_OR_FAIL(hipEventCreateWithFlags(&r->tab_done[q][si], hipEventDisableTiming));
      for (int q = 0; q < mdh_renderer::PART_RING; ++q)
         for (int si = 0; si < mdh_renderer::NSTREAMS; ++si) TRY_OR_FAIL(hipEventCreateWithFlags(&r->part_done[q][si], hipEventDisableTiming));
   }
   if ((rc = alloc_atlases(r)) != MDH_OK) return fail(rc);
   size_t px = (size_t)width * height;
   for (int s = 0; s < 2; ++s) {
      TRY_OR_FAIL(hipMalloc(&r->d_fb2[s], px * sizeof(float4)));
      TRY_OR_FAIL(hipMemsetAsync(r->d_fb2[s], 0, px * sizeof(float4), r->stream));
   }
   for (int s = 0; s < 2; ++s)
      for (int q = 0; q < 3; ++q) TRY_OR_FAIL(hipMalloc(&r->d_gb2[s][q], px * 4));
   TRY_OR_FAIL(hipMalloc(&r->d_warn, 4));
   TRY_OR_FAIL(hipHostMalloc((void **)&r->h_warn, 4, hipHostMallocDefault));
   *r->h_warn = 0;
   size_t vis_n = (size_t)vol->visibility_resolution[0] * vol->visibility_resolution[1] * vol->visibility_resolution[2] * 3;
   size_t scat_n = (size_t)vol->scattering_resolution[0] * vol->scattering_resolution[1];
   if (vis_n / 3 >= (1ull << 32) || scat_n >= (1ull << 32)) { seterr(MDH_E_INVALID, "a volumetrics texture of 2^32 texels or more"); return fail(MDH_E_INVALID); } // (tex_sample: 32-bit texel index)
   for (int s = 0; s < mdh_renderer::NSETS; ++s) {
      TRY_OR_FAIL(hipMalloc(&r->d_vis2[s], (vis_n ? vis_n : 1) * 4));
      TRY_OR_FAIL(hipMemsetAsync(r->d_vis2[s], 0, (vis_n ? vis_n : 1) * 4, r->stream));
      TRY_OR_FAIL(hipMalloc(&r->d_scat2[s], (scat_n ? scat_n : 1) * sizeof(float4)));
      TRY_OR_FAIL(hipMemsetAsync(r->d_scat2[s], 0, (scat_n ? scat_n : 1) * sizeof(float4), r->stream));
   }
   if (r->part.enable) {
      if (part_buffer_ints(r) >= (1ull << 31)) { seterr(MDH_E_INVALID, "a partition table of 2^31 ints or more"); return fail(MDH_E_INVALID); }
      size_t n = part_buffer_ints(r);
      for (int q = 0; q < mdh_renderer::PART_RING; ++q) {
         TRY_OR_FAIL(hipMalloc(&r->d_part_ring[q], n * 4));
         TRY_OR_FAIL(hipMemsetAsync(r->d_part_ring[q], 0, n * 4, r->stream));
      }
   }
   TRY_OR_FAIL(hipStreamSynchronize(r->stream));
#undef TRY_OR_FAIL
   *out = r;
   return MDH_OK;
}

extern "C" int32_t mdh_set_option(mdh_renderer *r, int32_t option, int32_t value)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   // what an open frame has latched (its atlas set, slice, streams and screen mode) cannot change under it
   if (r->in_frame && (option == MDH_OPT_ATLAS_FORMAT || option == MDH_OPT_RANK || option == MDH_OPT_WORLD || option == MDH_OPT_FRAME_OVERLAP ||
                       option == MDH_OPT_SCREEN_MODE))
      return seterr(MDH_E_STATE, "a frame is open");
   if ((r->comm || peer_active(r)) && (option == MDH_OPT_RANK || option == MDH_OPT_WORLD)) return seterr(MDH_E_STATE, "rank and world belong to the communicator (mdh_comm_init / mdh_peer_init)");
   if (peer_active(r) && option == MDH_OPT_IRRADIANCE_ALL && !value) return seterr(MDH_E_STATE, "the peer exchange moves radiance slices only: MDH_OPT_IRRADIANCE_ALL stays on");
   switch (option) {
   case MDH_OPT_ATLAS_FORMAT:
      if (value != 0 && value != 1) return seterr(MDH_E_INVALID, "atlas format is 0 (RGB8) or 1 (fp32)");
      if (value != r->opt_atlas) {
         if (r->peer) return seterr(MDH_E_STATE, "the peers hold handles of this renderer's atlases (mdh_peer_export): leave first (mdh_comm_destroy)");
         HIP_TRY(hipSetDevice(r->device));
         { int jr = join_main(r); if (jr != MDH_OK) return jr; }
         HIP_TRY(hipStreamSynchronize(r->stream));
         r->opt_atlas = value;
         int rc = alloc_atlases(r);
         if (rc != MDH_OK) return rc;
         if ((rc = alloc_rad_mips(r)) != MDH_OK) return rc;
      }
      break;
   case MDH_OPT_SCREEN_MODE: if (value < 0 || value > 2) return seterr(MDH_E_INVALID, "screen mode is 0, 1 or 2"); r->opt_mode = value; break;
   case MDH_OPT_AO_STEPS: r->opt_ao = value; break;
   case MDH_OPT_GBUFFER: r->opt_gbuffer = value ? 1 : 0; break;
   case MDH_OPT_RANK: if (value < 0) return seterr(MDH_E_INVALID, "rank < 0"); r->opt_rank = value; break;
   case MDH_OPT_WORLD: if (value < 1) return seterr(MDH_E_INVALID, "world < 1"); r->opt_world = value; break;
   case MDH_OPT_TIMING: r->opt_timing = value ? 1 : 0; break;
   case MDH_OPT_ADA_EVAL_DIV: r->opt_ada_div = value ? 1 : 0; break;
   case MDH_OPT_JIT: r->opt_jit = value ? 1 : 0; break;
   case MDH_OPT_IRRADIANCE_ALL: r->opt_irr_all = value ? 1 : 0; break;
   case MDH_OPT_WINDOW:
      if (value < 0 || value > 2) return seterr(MDH_E_INVALID, "bad value");
      r->opt_window = value; r->win_valid = false;
      break;
   case MDH_OPT_FRAME_OVERLAP:
      if (value < 0 || value > 2) return seterr(MDH_E_INVALID, "frame overlap is 0, 1 or 2");
      if ((value != 0) != (r->opt_overlap != 0)) r->scr_order_age = MDH_RAD_RESORT; // (how much of the screen pass is sorted follows the schedule: sort again)
      r->opt_overlap = value;
      break;
   case MDH_OPT_INDIRECT_SPECULAR: if (value < 0 || value > 3) return seterr(MDH_E_INVALID, "indirect specular mode is 0 .. 3"); r->opt_spec = value; break;
   case MDH_OPT_HYSTERESIS_PERMILLE: if (value < 0 || value > 999) return seterr(MDH_E_INVALID, "hysteresis is 0 .. 999 per mille"); r->opt_hyst = value; break;
   case MDH_OPT_RADIANCE_ORDER: r->opt_rad_order = value ? 1 : 0; r->rad_order_rays = 0; break;
   case MDH_OPT_SCREEN_ORDER: r->opt_scr_order = value ? 1 : 0; r->scr_order_cur = -1; break;
   case MDH_OPT_SCREEN_SPLIT: if (value < 0) return seterr(MDH_E_INVALID, "MDH_OPT_SCREEN_SPLIT: a number of wavefronts"); r->opt_scr_split = value; break;
   case MDH_OPT_NUMERICS: if (value != (MDH_FAST_NUMERICS ? 1 : (MDH_HYBRID_NUMERICS ? 2 : 0))) return seterr(MDH_E_STATE, "the numerics are a property of the library build (make fast builds the experiment)"); break;
   case MDH_OPT_RADIANCE_MIPS: {
      const int res = r->probes.radiance_resolution;
      if (value && (res & (res - 1)) != 0) return seterr(MDH_E_INVALID, "radiance mips need a power-of-two radiance resolution");
      if ((value != 0) != (r->opt_mips != 0)) {
         HIP_TRY(hipSetDevice(r->device));
         { int dr = drain_streams(r); if (dr != MDH_OK) return dr; }
         r->opt_mips = value ? 1 : 0;
         int rc = alloc_rad_mips(r);
         if (rc != MDH_OK) return rc;
      }
      break;
   }
   default: return seterr(MDH_E_INVALID, "unknown option");
   }
   return MDH_OK;
}
extern "C" int32_t mdh_get_option(mdh_renderer *r, int32_t option, int32_t *value)
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
   case MDH_OPT_FRAME_OVERLAP: *value = r->opt_overlap; break;
   case MDH_OPT_JIT: *value = r->opt_jit; break;
   case MDH_OPT_IRRADIANCE_ALL: *value = r->opt_irr_all; break;
   case MDH_OPT_WINDOW: *value = r->opt_window; break;
   case MDH_OPT_INDIRECT_SPECULAR: *value = r->opt_spec; break;
   case MDH_OPT_HYSTERESIS_PERMILLE: *value = r->opt_hyst; break;
   case MDH_OPT_RADIANCE_ORDER: *value = r->opt_rad_order; break;
   case MDH_OPT_SCREEN_ORDER: *value = r->opt_scr_order; break;
   case MDH_OPT_SCREEN_SPLIT: *value = r->opt_scr_split; break;
   case MDH_OPT_NUMERICS: *value = MDH_FAST_NUMERICS ? 1 : (MDH_HYBRID_NUMERICS ? 2 : 0); break; // 0 exact (shipped), 1 / 2 the labelled experiments
   case MDH_OPT_RADIANCE_MIPS: *value = r->opt_mips; break;
   default: return seterr(MDH_E_INVALID, "unknown option");
   }
   return MDH_OK;
}

// Set_Material (renderers.adb:349-367)
extern "C" int32_t mdh_set_material(mdh_renderer *r, int32_t id0, const float albedo[3], float metallic, float roughness)
{
   if (!r || !albedo) return seterr(MDH_E_INVALID, "bad argument");
   if (id0 < 0 || id0 >= MAX_MATERIALS) return seterr(MDH_E_INDEX, "material index out of range");
   uint8_t *p = r->materials_ubo + 16 + 32 * id0;
   memcpy(p, albedo, 12);
   memcpy(p + 12, &metallic, 4);
   memcpy(p + 16, &roughness, 4);
   if (id0 >= r->last_material_index) r->last_material_index = id0 + 1;
   r->table_dirty = true;
   return MDH_OK;
}
// Add_Material (renderers.adb:369-377)
extern "C" int32_t mdh_add_material(mdh_renderer *r, const float albedo[3], float metallic, float roughness, int32_t *out_id0)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   int id = r->last_material_index;
   int rc = mdh_set_material(r, id, albedo, metallic, roughness);
   if (rc == MDH_OK && out_id0) *out_id0 = id;
   return rc;
}
// Write_Entity (renderers.adb:335-347) of a whole element
static int write_entity(mdh_renderer *r, const Kind &k, int index1, const void *blob, int nbytes)
{
   if (!blob || nbytes != k.elem_size) return seterr(MDH_E_INVALID, "blob size does not match the std140 element size");
   if (index1 < 1 || index1 > k.max_count) return seterr(MDH_E_INDEX, "index out of the declared range");
   memcpy(r->scene_ubo.data() + k.array_off + k.stride * (index1 - 1), blob, (size_t)nbytes);
   r->table_dirty = true;
   return MDH_OK;
}
// Set_Primitive (renderers.adb:379-398)
extern "C" int32_t mdh_set_primitive(mdh_renderer *r, int32_t kind_ix, int32_t index1, const void *blob, int32_t nbytes)
{
   if (!r || kind_ix < 0 || kind_ix >= r->npk) return seterr(MDH_E_INVALID, "bad kind index");
   if (index1 < 1 || index1 > r->host_count[kind_ix]) return seterr(MDH_E_INDEX, "index past the primitives added");
   ++r->geometry_edits;
   return write_entity(r, r->pk[kind_ix], index1, blob, nbytes);
}
// Add_Primitive (renderers.adb:435-456)
extern "C" int32_t mdh_add_primitive(mdh_renderer *r, int32_t kind_ix, const void *blob, int32_t nbytes, int32_t *out_count)
{
   if (!r || kind_ix < 0 || kind_ix >= r->npk) return seterr(MDH_E_INVALID, "bad kind index");
   int count = r->host_count[kind_ix] + 1;
   ++r->geometry_edits;
   int rc = write_entity(r, r->pk[kind_ix], count, blob, nbytes);
   if (rc != MDH_OK) return rc;
   r->host_count[kind_ix] = count;
   int32_t c = count;
   memcpy(r->scene_ubo.data() + r->pk[kind_ix].count_off, &c, 4);
   if (out_count) *out_count = count;
   return MDH_OK;
}
// Set_Light (renderers.adb:458-483)
extern "C" int32_t mdh_set_light(mdh_renderer *r, int32_t index1, int32_t light_kind_ix, const void *blob, int32_t nbytes)
{
   if (!r || light_kind_ix < 0 || light_kind_ix >= r->nlk) return seterr(MDH_E_INVALID, "bad light kind index");
   int rc = write_entity(r, r->lk[light_kind_ix], index1, blob, nbytes);
   if (rc != MDH_OK) return rc;
   int32_t c = index1;
   memcpy(r->scene_ubo.data() + r->lk[light_kind_ix].count_off, &c, 4);
   memcpy(r->scene_ubo.data() + r->total_light_off, &c, 4);
   return MDH_OK;
}
extern "C" int32_t mdh_set_camera_position(mdh_renderer *r, const float p[3])
{
   if (!r || !p) return seterr(MDH_E_INVALID, "bad argument");
   memcpy(r->cam_pos, p, 12);
   return MDH_OK;
}
extern "C" int32_t mdh_set_camera_orientation(mdh_renderer *r, const float m[9])
{
   if (!r || !m) return seterr(MDH_E_INVALID, "bad argument");
   memcpy(r->cam_m, m, 36);
   return MDH_OK;
}

// `up`: the stream the new table is uploaded on (the one that uses it first); nullptr = the main stream
static int ensure_committed(mdh_renderer *r, hipStream_t up = nullptr)
{
   HIP_TRY(hipSetDevice(r->device));
   if (r->table_dirty) return commit_scene(r, up ? up : r->stream);
   return MDH_OK;
}
// for kernels that never look a cell up (the builders, the distance query): no bits staged behind the scene table
static KScene ks_no_bits(const mdh_renderer *r) { KScene k = r->ks; k.part_bits_f4 = 0; return k; }
static size_t lds_bytes(const mdh_renderer *r) { return (size_t)(r->ks.table_f4 + r->ks.part_bits_f4) * sizeof(float4); }
// the march kernels park MDH_PARK_DWORDS floats per thread behind the table (mdh_march.h)
static size_t lds_bytes_march(const mdh_renderer *r) { return lds_bytes(r) + (size_t)MDH_PARK_DWORDS * MDH_BLOCK * sizeof(float); }
// the screen pass runs without the visibility queue, whose entries, first steps and result words are the park
// slots from 15 up: 3 KiB less per workgroup, room for one more workgroup of a neighbouring pass on the CU
static size_t lds_bytes_screen(const mdh_renderer *r)
{
   // (scenes with a space partition: three more rows, the second shaded point's normal across its visibility marches, MDH_PARK_VD_ROW)
   const int rows = r->opt_mode == 2 ? MDH_DIRECT_PARK_ROWS : (r->part.enable && MDH_PART_PARK_VD ? std::max(MDH_SCR_PARK_ROWS, MDH_PARK_VD_ROW + 3) : MDH_SCR_PARK_ROWS);
   return lds_bytes(r) + (size_t)rows * MDH_BLOCK * sizeof(float);
}

// ------------------------------------------------------------------ hiprtc build of user-defined kinds
// MDH_OPT_JIT: instead of interpreting the MDH_X programs, compile them.  Every program becomes a
// straight-line function (one statement per instruction, the same operations in the same order, so
// the results are those of the interpreter bit for bit), mdh_jit_kinds.h dispatches on the kind, and
// the kernels of this library are compiled again around it with hiprtc -- the analogue of the
// reference's runtime glCompileShader of its generated GLSL.  Modules are cached per process by their
// generated source.
// A Distance program whose result is  sqrt (A) - R,  sqrt (A) + S  or  sqrt (A)  itself (a sphere, a torus, a capsule, a rounded
// anything: most distance functions end that way) can take part in the wave-level culling of the built-in spheres
// (mdh_device.h: closest_primitive): inside  closest = min (closest, d)  the root is not taken when, in every lane of the
// wavefront, A already proves d >= closest -- the same rule, the same margins, the same proof (rounding is monotone), and the
// same operations on the same values whenever the root IS taken.  jit_min_form finds the two instructions: the root S whose
// only reader is the final operation F, and F whose value (through moves) is the program's r0.
struct JitMinForm { int pc_sqrt = -1, pc_final = -1, kind = 0; }; // kind: 1 = sqrt - r[b], 2 = sqrt + r[b], 3 = r[a] + sqrt, 4 = sqrt
static int jit_op_reads(int op) // registers an instruction reads (SEL: three, see below)
{
   switch (op) {
   case MDH_X_LIT: case MDH_X_COMP: case MDH_X_POINT: return 0;
   case MDH_X_MOV: case MDH_X_NEG: case MDH_X_ABS: case MDH_X_FLOOR: case MDH_X_SIGN: case MDH_X_SQRT: case MDH_X_ITOF:
   case MDH_X_ACOS: case MDH_X_SIN: case MDH_X_COS: case MDH_X_TAN: case MDH_X_ASIN: case MDH_X_ATAN: return 1;
   case MDH_X_SEL: return 3;
   default: return 2;
   }
}
static JitMinForm jit_min_form(const std::vector<int32_t> &code)
{
   JitMinForm none, f;
   struct Val { int pc, op, a, b; int readers; };
   std::vector<Val> vals; // value 0: "never written" (registers start as 0.0f)
   vals.push_back({-1, -1, 0, 0, 0});
   int id[MDH_X_REGS];
   for (int i = 0; i < MDH_X_REGS; ++i) id[i] = 0;
   for (size_t pc = 0; pc < code.size(); ++pc) {
      const uint32_t w = (uint32_t)code[pc];
      const int op = w & 255, d = (w >> 8) & 63, a = (w >> 16) & 255, b = (w >> 24) & 63, ra = a & 63;
      const int reads = jit_op_reads(op);
      int c = 0;
      if (op == MDH_X_LIT) ++pc;
      if (op == MDH_X_SEL) c = code[++pc] & 63;
      if (op == MDH_X_MOV) { id[d] = id[ra]; continue; } // (a move hands the value on)
      if (reads >= 1) ++vals[id[ra]].readers;
      if (reads >= 2) ++vals[id[b]].readers;
      if (reads >= 3) ++vals[id[c]].readers;
      vals.push_back({(int)pc - (op == MDH_X_LIT || op == MDH_X_SEL ? 1 : 0), op, reads >= 1 ? id[ra] : 0, reads >= 2 ? id[b] : 0, 0});
      id[d] = (int)vals.size() - 1;
   }
   const Val &F = vals[id[0]];
   if (F.pc < 0) return none;
   auto is_root = [&](int v) { return v > 0 && vals[v].op == MDH_X_SQRT && vals[v].readers == 1; };
   if (F.op == MDH_X_SQRT && F.readers == 0) { f.pc_sqrt = f.pc_final = F.pc; f.kind = 4; return f; }
   if (F.op == MDH_X_SUB && is_root(F.a) && F.b != F.a) { f.pc_sqrt = vals[F.a].pc; f.pc_final = F.pc; f.kind = 1; return f; }
   if (F.op == MDH_X_ADD && is_root(F.a) && F.b != F.a) { f.pc_sqrt = vals[F.a].pc; f.pc_final = F.pc; f.kind = 2; return f; }
   if (F.op == MDH_X_ADD && is_root(F.b) && F.b != F.a) { f.pc_sqrt = vals[F.b].pc; f.pc_final = F.pc; f.kind = 3; return f; }
   return none;
}
// min_form: emit  float NAME (int ent, f3 x, float closest)  =  min (closest, the program's distance)  with the root culled
static void jit_emit_program(std::string &out, const char *name, const std::vector<int32_t> &code, const JitMinForm *min_form = nullptr)
{
   char buf[512];
   bool used[MDH_X_REGS] = {false};
   std::string body;
   static const char *ARG[10] = {"x.x", "x.y", "x.z", "nrm.x", "nrm.y", "nrm.z", "dir.x", "dir.y", "dir.z", "dist"};
   for (size_t pc = 0; pc < code.size(); ++pc) {
      const uint32_t w = (uint32_t)code[pc];
      const int op = w & 255, d = (w >> 8) & 63, a = (w >> 16) & 255, b = (w >> 24) & 63, ra = a & 63;
      if (min_form && (int)pc == min_form->pc_final) { // the final operation: cull, or take the root and finish as the program does
         used[ra] = used[b] = true;
         char rr[32], fin[64];
         switch (min_form->kind) {
         case 1: snprintf(rr, sizeof rr, "r%d", b); snprintf(fin, sizeof fin, "sqrt_(d2_) - r%d", b); break;
         case 2: snprintf(rr, sizeof rr, "-r%d", b); snprintf(fin, sizeof fin, "sqrt_(d2_) + r%d", b); break;
         case 3: snprintf(rr, sizeof rr, "-r%d", ra); snprintf(fin, sizeof fin, "r%d + sqrt_(d2_)", ra); break;
         default: snprintf(rr, sizeof rr, "0.0f"); snprintf(fin, sizeof fin, "sqrt_(d2_)"); break;
         }
         if (min_form->kind == 4) { snprintf(buf, sizeof buf, "   const float d2_ = r%d;\n", ra); body += buf; }
         snprintf(buf, sizeof buf,
                  "   const float tsum_ = closest + (%s);\n"
                  "   const bool need_ = !(tsum_ < 0.0f) && !(d2_ > (tsum_ * tsum_) * 1.000001f);\n"
                  "   if (__ballot(need_) != 0ull) closest = min_raw(closest, %s);\n"
                  "   return closest;\n", rr, fin);
         body += buf;
         break;
      }
      if (min_form && (int)pc == min_form->pc_sqrt) { // the root waits (its register is read by the final operation only)
         used[ra] = true;
         snprintf(buf, sizeof buf, "   const float d2_ = r%d;\n", ra); body += buf;
         continue;
      }
      used[d] = true;
      const char *fmt = nullptr;
      switch (op) {
      case MDH_X_LIT: snprintf(buf, sizeof buf, "   r%d = __builtin_bit_cast(float, (int)0x%08xu);\n", d, (unsigned)code[++pc]); body += buf; continue;
      case MDH_X_COMP: snprintf(buf, sizeof buf, "   r%d = tab_float(ent + %d);\n", d, a); body += buf; continue;
      case MDH_X_POINT: snprintf(buf, sizeof buf, "   r%d = %s;\n", d, ARG[a < 10 ? a : 9]); body += buf; continue;
      case MDH_X_SEL: {
         const int c = code[++pc] & 63;
         used[ra] = used[b] = used[c] = true;
         snprintf(buf, sizeof buf, "   r%d = r%d != 0.0f ? r%d : r%d;\n", d, ra, b, c); body += buf; continue;
      }
      case MDH_X_MOV: fmt = "   r%d = r%d;\n"; break;
      case MDH_X_ADD: fmt = "   r%d = r%d + r%d;\n"; break;
      case MDH_X_SUB: fmt = "   r%d = r%d - r%d;\n"; break;
      case MDH_X_MUL: fmt = "   r%d = r%d * r%d;\n"; break;
      case MDH_X_DIV: fmt = "   r%d = r%d / r%d;\n"; break;
      case MDH_X_DIVF: fmt = "   r%d = ADA_DIV ? r%2$d + r%3$d : r%2$d / r%3$d;\n"; break;
      case MDH_X_NEG: fmt = "   r%d = -r%d;\n"; break;
      case MDH_X_ABS: fmt = "   r%d = __builtin_fabsf(r%d);\n"; break;
      case MDH_X_FLOOR: fmt = "   r%d = __builtin_floorf(r%d);\n"; break;
      case MDH_X_SIGN: fmt = "   r%d = sign_(r%d);\n"; break;
      case MDH_X_MIN: fmt = "   r%d = min_(r%d, r%d);\n"; break;
      case MDH_X_MAX: fmt = "   r%d = max_(r%d, r%d);\n"; break;
      case MDH_X_SQRT: fmt = "   r%d = sqrt_(r%d);\n"; break;
      case MDH_X_POW: fmt = "   r%d = pow_(r%d, r%d);\n"; break;
      case MDH_X_LT: fmt = "   r%d = r%d < r%d ? 1.0f : 0.0f;\n"; break;
      case MDH_X_GT: fmt = "   r%d = r%d > r%d ? 1.0f : 0.0f;\n"; break;
      case MDH_X_LE: fmt = "   r%d = r%d <= r%d ? 1.0f : 0.0f;\n"; break;
      case MDH_X_GE: fmt = "   r%d = r%d >= r%d ? 1.0f : 0.0f;\n"; break;
      case MDH_X_ITOF: fmt = "   r%d = (float)__builtin_bit_cast(int, r%d);\n"; break;
      case MDH_X_ACOS: fmt = "   r%d = acos_(r%d);\n"; break;
      case MDH_X_SIN: fmt = "   r%d = sin_(r%d);\n"; break;
      case MDH_X_COS: fmt = "   r%d = cos_(r%d);\n"; break;
      case MDH_X_TAN: fmt = "   r%d = tan_(r%d);\n"; break;
      case MDH_X_ASIN: fmt = "   r%d = asin_(r%d);\n"; break;
      case MDH_X_ATAN: fmt = "   r%d = atan_(r%d);\n"; break;
      default: fmt = "   r%d = 0.0f;\n"; break;
      }
      used[ra] = used[b] = true;
      if (op == MDH_X_DIVF) snprintf(buf, sizeof buf, "   r%d = ADA_DIV ? r%d + r%d : r%d / r%d;\n", d, ra, b, ra, b);
      else snprintf(buf, sizeof buf, fmt, d, ra, b);
      body += buf;
   }
   used[0] = used[1] = used[2] = true;
   if (min_form) {
      out += "MDH_DEV float ";
      out += name;
      out += "(int ent, f3 x, float closest)\n{\n   constexpr bool ADA_DIV = false;\n   float";
   } else {
      out += "template <bool ADA_DIV> MDH_DEV f3 ";
      out += name;
      out += "(int ent, f3 x, f3 nrm, f3 dir, float dist)\n{\n   float";
   }
   bool first = true;
   for (int i = 0; i < MDH_X_REGS; ++i)
      if (used[i]) { snprintf(buf, sizeof buf, "%s r%d = 0.0f", first ? "" : ",", i); out += buf; first = false; }
   out += ";\n" + body + (min_form ? "}\n" : "   return F3(r0, r1, r2);\n}\n");
}
// mdh_jit_kinds.h of a scene: the programs as functions and the two dispatchers mdh_device.h calls
static std::string jit_kinds_header(const mdh_renderer *r)
{
   std::string s = "// generated by libmadarch_hip (mdh_api.hip: jit_kinds_header)\n", prim_cases, light_cases, closest_all;
   char name[64], buf[512];
   for (int k = 0; k < r->npk; ++k) {
      if (r->pk[k].type != PK_CUSTOM) continue;
      const std::vector<int32_t> *progs[3] = {&r->pk[k].x_dist, &r->pk[k].x_nrm, &r->pk[k].x_mat};
      { // closest_primitive's loop over this kind's instances, the kind a constant; Distance in its min form where it has one
         const JitMinForm mf = jit_min_form(r->pk[k].x_dist);
         snprintf(name, sizeof name, "jit_p%d_min", k);
         if (mf.kind) jit_emit_program(s, name, r->pk[k].x_dist, &mf);
         snprintf(buf, sizeof buf,
                  "   {\n      const int n = hdr(H_KCOUNT + %d), stride = hdr(H_KSTRIDE + %d) * 4;\n      int ent = hdr(H_KSLOT + %d) * 4;\n"
                  "#pragma unroll 1\n      for (int i = 0; i < n; ++i, ent += stride) closest = %s;\n   }\n", k, k, k,
                  mf.kind ? (std::string(name) + "(ent, x, closest)").c_str()
                          : ("min_raw(closest, jit_p" + std::to_string(k) + "_0<false>(ent, x, F3(0.0f, 0.0f, 0.0f), F3(0.0f, 0.0f, 0.0f), 0.0f).x)").c_str());
         closest_all += buf;
      }
      for (int q = 0; q < 3; ++q) {
         snprintf(name, sizeof name, "jit_p%d_%d", k, q);
         jit_emit_program(s, name, *progs[q]);
         snprintf(buf, sizeof buf, "   case %d: return %s<ADA_DIV>(ent, x, F3(0.0f, 0.0f, 0.0f), F3(0.0f, 0.0f, 0.0f), 0.0f);\n", q * 8 + k, name);
         prim_cases += buf;
      }
   }
   for (int k = 0; k < r->nlk; ++k) {
      if (r->lk[k].type != LK_CUSTOM) continue;
      const std::vector<int32_t> *progs[2] = {&r->lk[k].x_dist, &r->lk[k].x_nrm};
      for (int q = 0; q < 2; ++q) {
         snprintf(name, sizeof name, "jit_l%d_%d", k, q);
         jit_emit_program(s, name, *progs[q]);
         snprintf(buf, sizeof buf, "   case %d: return %s<false>(ent, pos, nrm, dir, dist);\n", q * 4 + k, name);
         light_cases += buf;
      }
   }
   s += "// closest_primitive over every instance of every user-defined kind (scenes.adb:602-629)\n"
        "#define MDH_JIT_CLOSEST_ALL 1\nMDH_DEV float jit_closest_all(f3 x, float closest)\n{\n" + closest_all + "   return closest;\n}\n";
   s += "// which: 0 Distance, 1 Normal, 2 Material; k: the kind (wave-uniform)\n"
        "template <bool ADA_DIV> MDH_DEV f3 jit_prim(int which, int k, int ent, f3 x)\n{\n   switch (which * 8 + k) {\n" + prim_cases +
        "   default: break;\n   }\n   return F3(0.0f, 0.0f, 0.0f);\n}\n"
        "// which: 0 Sample, 1 Position\n"
        "MDH_DEV f3 jit_light(int which, int k, int ent, f3 pos, f3 nrm, f3 dir, float dist)\n{\n   switch (which * 4 + k) {\n" + light_cases +
        "   default: break;\n   }\n   return F3(0.0f, 0.0f, 0.0f);\n}\n";
   return s;
}

// hiprtc is loaded when the first user-defined kind is compiled; a box without it falls back to the interpreter
struct HiprtcApi {
   decltype(&hiprtcCreateProgram) CreateProgram = nullptr;
   decltype(&hiprtcDestroyProgram) DestroyProgram = nullptr;
   decltype(&hiprtcAddNameExpression) AddNameExpression = nullptr;
   decltype(&hiprtcCompileProgram) CompileProgram = nullptr;
   decltype(&hiprtcGetCodeSize) GetCodeSize = nullptr;
   decltype(&hiprtcGetCode) GetCode = nullptr;
   decltype(&hiprtcGetLoweredName) GetLoweredName = nullptr;
   decltype(&hiprtcGetProgramLogSize) GetProgramLogSize = nullptr;
   decltype(&hiprtcGetProgramLog) GetProgramLog = nullptr;
   decltype(&hiprtcGetErrorString) GetErrorString = nullptr;
   bool ok = false;
};
static const HiprtcApi &hiprtc_api()
{
   static HiprtcApi api;
   static bool tried = false;
   if (tried) return api;
   tried = true;
   void *h = nullptr;
   for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"})
      if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
   if (!h) return api;
#define MDH_RTC_SYM(field, sym) api.field = (decltype(api.field))dlsym(h, #sym)
   MDH_RTC_SYM(CreateProgram, hiprtcCreateProgram); MDH_RTC_SYM(DestroyProgram, hiprtcDestroyProgram);
   MDH_RTC_SYM(AddNameExpression, hiprtcAddNameExpression); MDH_RTC_SYM(CompileProgram, hiprtcCompileProgram);
   MDH_RTC_SYM(GetCodeSize, hiprtcGetCodeSize); MDH_RTC_SYM(GetCode, hiprtcGetCode); MDH_RTC_SYM(GetLoweredName, hiprtcGetLoweredName);
   MDH_RTC_SYM(GetProgramLogSize, hiprtcGetProgramLogSize); MDH_RTC_SYM(GetProgramLog, hiprtcGetProgramLog);
   MDH_RTC_SYM(GetErrorString, hiprtcGetErrorString);
#undef MDH_RTC_SYM
   api.ok = api.CreateProgram && api.DestroyProgram && api.AddNameExpression && api.CompileProgram && api.GetCodeSize && api.GetCode &&
            api.GetLoweredName && api.GetProgramLogSize && api.GetProgramLog && api.GetErrorString;
   return api;
}

struct JitModule {
   hipModule_t mod = nullptr;
   std::map<std::string, hipFunction_t> fn;
};
static std::mutex g_jit_mutex;
static std::map<std::string, JitModule *> g_jit_cache;

// Compile the kernels named by `exprs` (template-ids) around this scene's mdh_jit_kinds.h.
static JitModule *jit_module(mdh_renderer *r, const std::vector<std::string> &exprs)
{
   const HiprtcApi &rtc = hiprtc_api();
   if (!rtc.ok) { seterr(MDH_E_DEVICE, "libhiprtc.so cannot be loaded: user-defined kinds are interpreted"); return nullptr; }
   if (r->jit_kinds.empty()) {
      r->jit_kinds = jit_kinds_header(r);
      if (const char *dump = getenv("MADARCH_HIP_JIT_DUMP")) // (diagnostic: the generated mdh_jit_kinds.h of the scene)
         if (FILE *f = fopen(dump, "w")) { fputs(r->jit_kinds.c_str(), f); fclose(f); }
   }
   std::string key = "device " + std::to_string(r->device) + "\n" + r->jit_kinds; // (a module and its functions belong to the device they were loaded on)
   for (auto &e : exprs) key += "|" + e;
   std::lock_guard<std::mutex> lock(g_jit_mutex);
   auto it = g_jit_cache.find(key);
   if (it != g_jit_cache.end()) return it->second;
   const char *headers[4] = {MDH_SRC_DEVICE, MDH_SRC_MARCH, MDH_SRC_KERNELS, r->jit_kinds.c_str()};
   const char *names[4] = {"mdh_device.h", "mdh_march.h", "mdh_kernels.h", "mdh_jit_kinds.h"};
   hiprtcProgram prog = nullptr;
   auto fail = [&](const char *what, hiprtcResult e) -> JitModule * {
      std::string log;
      size_t n = 0;
      if (prog && rtc.GetProgramLogSize(prog, &n) == HIPRTC_SUCCESS && n > 1) { log.resize(n); rtc.GetProgramLog(prog, &log[0]); }
      snprintf(g_err, sizeof g_err, "hiprtc %s failed (%s): %.380s", what, rtc.GetErrorString(e), log.c_str());
      if (prog) rtc.DestroyProgram(&prog);
      return nullptr;
   };
   hiprtcResult e = rtc.CreateProgram(&prog, "#define MDH_JIT 1\n#include \"mdh_kernels.h\"\n", "mdh_jit.hip", 4, headers, names);
   if (e != HIPRTC_SUCCESS) return fail("create", e);
   for (auto &x : exprs)
      if ((e = rtc.AddNameExpression(prog, x.c_str())) != HIPRTC_SUCCESS) return fail("name expression", e);
   // the flags of the Makefile: one IEEE operation per source operation, no vectorizers
   const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-fno-vectorize", "-fno-slp-vectorize"};
   if ((e = rtc.CompileProgram(prog, (int)(sizeof opts / sizeof opts[0]), opts)) != HIPRTC_SUCCESS) return fail("compile", e);
   size_t size = 0;
   if ((e = rtc.GetCodeSize(prog, &size)) != HIPRTC_SUCCESS) return fail("code size", e);
   std::vector<char> code(size);
   if ((e = rtc.GetCode(prog, code.data())) != HIPRTC_SUCCESS) return fail("code", e);
   JitModule *m = new JitModule();
   if (hipModuleLoadData(&m->mod, code.data()) != hipSuccess) { delete m; return fail("module load", HIPRTC_ERROR_INTERNAL_ERROR); }
   for (auto &x : exprs) {
      const char *lowered = nullptr;
      if ((e = rtc.GetLoweredName(prog, x.c_str(), &lowered)) != HIPRTC_SUCCESS) { delete m; return fail("lowered name", e); }
      hipFunction_t f = nullptr;
      if (hipModuleGetFunction(&f, m->mod, lowered) != hipSuccess) { delete m; return fail("module function", HIPRTC_ERROR_INTERNAL_ERROR); }
      m->fn[x] = f;
   }
   rtc.DestroyProgram(&prog);
   g_jit_cache[key] = m;
   return m;
}
// Does the radiance pass of this renderer's slice leave most wavefront slots empty (k_radiance's SMALL variant)?
static bool rad_small_launch(const mdh_renderer *r)
{
   const KProbes p = make_probes(r);
   const long waves = ((long)(p.probe_end - p.probe_begin) * p.rres * p.rres + 63) / 64;
   return waves <= (long)r->n_cus * 4 * MDH_RAD_SMALL_WAVES_PER_SIMD;
}
// The workgroups of the radiance pass the chip holds at once, when the launch is that and a remainder smaller than
// it (k_radiance, mdh_kernels.h: the remainder runs at a raised issue priority); 0 otherwise.
#ifndef MDH_RAD_TAIL_PRIO
#define MDH_RAD_TAIL_PRIO 1
#endif
static int rad_first_round(mdh_renderer *r, const void *kernel, hipFunction_t fn, size_t lds, int blocks)
{
   const void *key = kernel ? kernel : (const void *)fn;
   auto it = r->resident.find({key, lds});
   if (it == r->resident.end()) {
      int per_cu = 0;
      const hipError_t e = kernel ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, MDH_BLOCK, lds)
                                  : hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, MDH_BLOCK, lds);
      if (e != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
      it = r->resident.insert({{key, lds}, per_cu}).first;
   }
   const long first = (long)it->second * r->n_cus;
   return MDH_RAD_TAIL_PRIO && first > 0 && blocks > first && blocks < 2 * first ? (int)first : 0;
}
// launch a function of a JIT module: the arguments are the kernel's by-value structs, laid out as the
// kernarg segment lays them out (each at its natural alignment = a struct of them)
template <typename Args> static int jit_launch(hipFunction_t f, int blocks, int block, size_t lds, hipStream_t st, Args &args)
{
   size_t size = sizeof(Args);
   void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
   HIP_TRY(hipModuleLaunchKernel(f, blocks, 1, 1, block, 1, 1, (unsigned)lds, st, nullptr, extra));
   return MDH_OK;
}

// Update_Partitioning (renderers.adb:757-775): all three methods build the table on the device
extern "C" int32_t mdh_update_partitioning(mdh_renderer *r, int32_t method)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (!r->part.enable) return MDH_OK; // renderers.adb:763-765
   if (method < 0 || method > 2) return seterr(MDH_E_INVALID, "bad method");
   if (r->in_frame) return seterr(MDH_E_STATE, "a frame is open");
   // The build goes where the table is used first -- the probe stream when frames are kept in flight -- into the
   // next buffer of the ring, and nothing waits on the host: frames in flight keep their buffer, later launches
   // on any stream are ordered after ev_part (table_acquire).
   const bool piped = r->opt_overlap && r->stream == r->own_stream && r->opt_mode == 0;
   hipStream_t up = piped ? r->probe_stream : r->stream;
   int rc = ensure_committed(r, up);
   if (rc != MDH_OK) return rc;
   if (!piped) {
      if ((rc = join_main(r)) != MDH_OK) return rc;
      r->main_dirty = true;
   }
   const int sup = stream_index(r, up);
   const int ns = (r->part_slot + 1) % mdh_renderer::PART_RING;
   for (int si = 0; si < mdh_renderer::NSTREAMS; ++si)
      if (r->part_used[ns][si]) { // the last kernels that read that buffer (four builds ago)
         if (si != sup) HIP_TRY(hipStreamWaitEvent(up, r->part_done[ns][si], 0));
         r->part_used[ns][si] = false;
      }
   // the warning counter is shared by consecutive builds: behind the previous build and its read-back
   if (r->part_version && r->part_stream != up) HIP_TRY(hipStreamWaitEvent(up, r->ev_warn, 0));
   PartBuildArgs a;
   a.method = method;
   const int *d = r->part.grid_dimensions;
   if (method == 2) { a.gx = 2 * (d[0] / 2); a.gy = 2 * (d[1] / 2); a.gz = 2 * (d[2] / 2); }
   else { a.gx = d[0]; a.gy = d[1]; a.gz = d[2]; }
   for (int i = 0; i < 3; ++i) {
      // the CPU builders read the settings record, the compute shader the GLSL text
      a.sp[i] = method == 2 ? r->pg_spacing[i] : r->part.grid_spacing[i];
      a.off[i] = method == 2 ? r->pg_offset[i] : r->part.grid_offset[i];
   }
   a.gpu_diag = r->part_gpu_diag;
   a.table = r->d_part_ring[ns];
   a.warnings = r->d_warn;
   // The builders write a cell's counts and the candidates it found, nothing else: entries behind them and cells
   // outside the builder's grid (GPU_Fast on odd dimensions) keep what the table held before, in the
   // reference's single buffer.  So the next buffer starts as a copy of the current one (tens of KiB).
   const size_t total = part_buffer_ints(r); // (the lists and their bits)
   const int cells = a.gx * a.gy * a.gz;
   if ((rc = table_acquire(r, up)) != MDH_OK) return rc; // (also orders `up` after the previous build)
   HIP_TRY(hipMemcpyAsync(r->d_part_ring[ns], r->d_part_ring[r->part_slot], total * 4, hipMemcpyDeviceToDevice, up));
   HIP_TRY(hipMemsetAsync(r->d_warn, 0, 4, up));
   if (cells > 0) {
      hipLaunchKernelGGL(k_partition_build, dim3(cells), dim3(64), lds_bytes(r), up, ks_no_bits(r), a); // one wavefront per cell
      HIP_TRY(hipGetLastError());
   }
   // the lists once more as bits, for every cell (cells the builder left alone keep their lists, and so their bits)
   hipLaunchKernelGGL(k_partition_bits, dim3((r->part_cells + 63) / 64), dim3(64), lds_bytes(r), up, ks_no_bits(r), r->d_part_ring[ns]);
   HIP_TRY(hipGetLastError());
   if ((rc = table_release(r, up)) != MDH_OK) return rc; // (the copy read the current buffer)
   HIP_TRY(hipMemcpyAsync(r->h_warn, r->d_warn, 4, hipMemcpyDeviceToHost, up));
   HIP_TRY(hipEventRecord(r->ev_warn, up));
   r->warn_pending = true;
   HIP_TRY(hipEventRecord(r->ev_part, up));
   r->part_stream = up;
   ++r->part_version;
   r->part_seen[sup] = r->part_version;
   r->part_slot = ns;
   r->ks.part_table = r->d_part_ring[ns];
   return MDH_OK;
}

template <int PART, int MODE> static void launch_screen_g(mdh_renderer *r, hipStream_t st, const KProbes &pr, const KVolumetrics &vol, const KCamera &cam, const ScreenArgs &a, int blocks)
{
   if (r->opt_gbuffer) hipLaunchKernelGGL((k_screen<PART, MODE, true>), dim3(blocks), dim3(MDH_BLOCK), lds_bytes_screen(r), st, r->ks, pr, vol, cam, a);
   else hipLaunchKernelGGL((k_screen<PART, MODE, false>), dim3(blocks), dim3(MDH_BLOCK), lds_bytes_screen(r), st, r->ks, pr, vol, cam, a);
}
// (pow2: mode 0 with probe counts and tile resolutions that are all powers of two runs the MDH_PF_POW2 variant;
//  built for scenes without user-defined kinds only -- those compile their own variant at run time, or interpret)
template <int PART_> static void launch_screen_m(mdh_renderer *r, hipStream_t st, const KProbes &pr, const KVolumetrics &vol, const KCamera &cam, const ScreenArgs &a, int blocks, bool pow2)
{
   // (the census variants are built for the renderer's own pixel program only: the optional specular modes and screen modes 1 and 2
   //  run the general scan)
   constexpr int CENSUS = PART_ & (MDH_PF_ROOM | MDH_PF_PSMALL), PART = PART_ & ~CENSUS;
   if (CENSUS && r->opt_mode == 0 && !(a.spec_mode == 1 || a.spec_mode == 3 || (a.spec_mode == 2 && pr.rad_mips))) {
      if (pow2) launch_screen_g<PART_ | MDH_PF_POW2, 0>(r, st, pr, vol, cam, a, blocks);
      else launch_screen_g<PART_, 0>(r, st, pr, vol, cam, a, blocks);
      return;
   }
   if ((CENSUS & MDH_PF_PSMALL) && r->opt_mode == 2) { launch_screen_g<PART_, 2>(r, st, pr, vol, cam, a, blocks); return; } // (simple_scene's direct light + occlusion: BASELINE config 2)
   if (r->opt_mode == 0) {
      if (a.spec_mode == 1 || a.spec_mode == 3 || (a.spec_mode == 2 && pr.rad_mips)) { // the other two bodies of render_probes.glsl:264-272 (and mode 2 over a mip chain): a variant of their own
         if (r->opt_gbuffer) hipLaunchKernelGGL((k_screen<PART, 0, true, true>), dim3(blocks), dim3(MDH_BLOCK), lds_bytes_screen(r), st, r->ks, pr, vol, cam, a);
         else hipLaunchKernelGGL((k_screen<PART, 0, false, true>), dim3(blocks), dim3(MDH_BLOCK), lds_bytes_screen(r), st, r->ks, pr, vol, cam, a);
      }
      else if (pow2 && !(PART & MDH_PF_CUSTOM)) launch_screen_g<(PART & MDH_PF_CUSTOM) ? PART : (PART | MDH_PF_POW2), 0>(r, st, pr, vol, cam, a, blocks);
      else launch_screen_g<PART, 0>(r, st, pr, vol, cam, a, blocks);
   }
   else if (r->opt_mode == 1) launch_screen_g<PART, 1>(r, st, pr, vol, cam, a, blocks);
   else launch_screen_g<PART, 2>(r, st, pr, vol, cam, a, blocks);
}

// MDH_OPT_WINDOW: the pinned host buffer the next screen pass stores the window's pixels in.  A ring: the pixels
// of a frame stay untouched until WIN_RING - 1 further screen passes have been enqueued.
static int window_slot(mdh_renderer *r, int rank, int world, unsigned **out)
{
   const size_t bytes = (size_t)r->W * r->H * 4;
   const bool owner_changed = r->win_owner[0] != rank || r->win_owner[1] != world;
   if (owner_changed) { // other ranks' tiles read 0, as in the framebuffer: clear every slot once nothing writes them any more
      if (r->probe_stream) HIP_TRY(hipStreamSynchronize(r->probe_stream));
      if (r->alt_stream) HIP_TRY(hipStreamSynchronize(r->alt_stream));
      HIP_TRY(hipStreamSynchronize(r->stream));
      for (int q = 0; q < mdh_renderer::WIN_RING; ++q)
         if (r->h_win[q]) memset(r->h_win[q], 0, bytes);
      r->win_owner[0] = rank; r->win_owner[1] = world;
   }
   const int slot = (int)(r->win_passes % mdh_renderer::WIN_RING);
   if (!r->h_win[slot]) {
      HIP_TRY(hipHostMalloc((void **)&r->h_win[slot], bytes, hipHostMallocDefault));
      memset(r->h_win[slot], 0, bytes);
   }
   r->win_passes += 1;
   r->win_valid = true;
   *out = r->h_win[slot];
   return MDH_OK;
}

// One pass on stream `st`.  The radiance pass reads the irradiance atlas of set `src` and writes the
// radiance atlas of set `dst`; every other pass works on set `dst` (in place: src == dst == r->last).
static int run_pass(mdh_renderer *r, int pass, hipStream_t st, int src, int dst, int fbix = -1)
{
   if (fbix < 0) fbix = r->fb_last;
   // kernel variant: bit 0 = space partition, bit 1 = user-defined kinds (mdh_device.h, MDH_PF_*)
   bool has_custom = false;
   for (int k = 0; k < r->npk; ++k) has_custom = has_custom || r->pk[k].type == PK_CUSTOM;
   for (int k = 0; k < r->nlk; ++k) has_custom = has_custom || r->lk[k].type == LK_CUSTOM;
   // ... bit 3 = the space partition's border falls back to the full scan (built-in kinds: a variant of its own; with
   // user-defined kinds the kernels test the setting at run time)
   const int pf = (r->part.enable != 0 ? MDH_PF_PART : 0) | (has_custom ? MDH_PF_CUSTOM : 0) |
                  (r->part.enable != 0 && r->part.border_behavior != 0 && !has_custom ? MDH_PF_FALLBACK : 0);
   // ... bit 4 = the census of the reference's rooms (MDH_PF_ROOM, mdh_device.h: closest_primitive): every plane folded into
   // the axis offsets, one sphere, one box, nothing else, no partition -- the scan's loops as straight-line code
   const bool room = pf == 0 && r->ks.n_axis > 0 && r->ks.gplane_count == 0 && r->ks.tcount[PK_SPHERE] == 1 && r->ks.tcount[PK_BOX] == 1 &&
                     r->ks.tcount[PK_TRIANGLE] == 0 && MDH_ROOM_VARIANTS;
   // ... bit 5 = the partition's small form with its census (MDH_PF_PSMALL, mdh_device.h: partitioning_closest_bits)
   const bool psmall = pf == MDH_PF_PART && r->ks.part_small && r->ks.part_tmask[PK_TRIANGLE] == 0 && r->ks.part_sp_pow2 && r->ks.part_cells < (1 << 24) && MDH_ROOM_VARIANTS;
   const int pfk = room ? MDH_PF_ROOM : (psmall ? (MDH_PF_PART | MDH_PF_PSMALL) : pf); // (the kernel variant by the scene's census)
   // the probe-sampling kernels (radiance, mode-0 screen) have a variant for atlases whose every dimension is a power of two
   auto is_pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
   // (... and small enough for what those variants assume besides: probe ids within 24-bit products, RGBA8 byte offsets
   //  within 32 bits -- mdh_device.h: grid_to_probe_id, atlas_rgba8)
   const long long pcount = (long long)r->probes.probe_count[0] * r->probes.probe_count[1];
   const long long rres = r->probes.radiance_resolution, ires = r->probes.irradiance_resolution;
   const bool pow2 = is_pow2(r->probes.probe_count[0]) && is_pow2(r->probes.probe_count[1]) && is_pow2(r->probes.radiance_resolution) &&
                     is_pow2(r->probes.irradiance_resolution) && pcount < 65536 && pcount * rres * rres <= (1ll << 30) &&
                     pcount * ires * ires <= (1ll << 30);
#define MDH_LAUNCH_PF(KERNEL, GRID, BLOCK, LDS, ...)                                                      \
   do {                                                                                                   \
      switch (pfk) {                                                                                      \
      case 0: hipLaunchKernelGGL(KERNEL<0>, GRID, BLOCK, LDS, st, __VA_ARGS__); break;                     \
      case MDH_PF_ROOM: hipLaunchKernelGGL(KERNEL<MDH_PF_ROOM>, GRID, BLOCK, LDS, st, __VA_ARGS__); break; \
      case MDH_PF_PART | MDH_PF_PSMALL: hipLaunchKernelGGL(KERNEL<MDH_PF_PART | MDH_PF_PSMALL>, GRID, BLOCK, LDS, st, __VA_ARGS__); break; \
      case 1: hipLaunchKernelGGL(KERNEL<1>, GRID, BLOCK, LDS, st, __VA_ARGS__); break;                     \
      case 2: hipLaunchKernelGGL(KERNEL<2>, GRID, BLOCK, LDS, st, __VA_ARGS__); break;                     \
      case 9: hipLaunchKernelGGL(KERNEL<9>, GRID, BLOCK, LDS, st, __VA_ARGS__); break;                     \
      default: hipLaunchKernelGGL(KERNEL<3>, GRID, BLOCK, LDS, st, __VA_ARGS__); break;                    \
      }                                                                                                   \
   } while (0)
   KProbes pr = make_probes(r);
   pr.rad = r->d_rad2[dst];
   pr.irr = r->d_irr2[pass == MDH_PASS_RADIANCE ? src : dst];
   KCamera cam = make_camera(r);
   // user-defined kinds compiled into the kernels (MDH_OPT_JIT); a scene hiprtc cannot build falls back to the
   // interpreter for good (the reason stays in mdh_last_error, MDH_OPT_JIT reads 0 afterwards)
   if (pass != MDH_PASS_IRRADIANCE) { // (the irradiance pass does not stage the scene table)
      int trc = table_acquire(r, st);
      if (trc != MDH_OK) return trc;
   }
   bool jit = has_custom && r->opt_jit;
   char kname[64] = "";
   JitModule *jm = nullptr;
   if (jit) {
      switch (pass) {
      case MDH_PASS_RADIANCE: snprintf(kname, sizeof kname, "k_radiance<%d, %s>", pf | (pow2 ? MDH_PF_POW2 : 0), rad_small_launch(r) ? "true" : "false"); break;
      case MDH_PASS_VISIBILITY: snprintf(kname, sizeof kname, "k_visibility<%d>", pf); break;
      case MDH_PASS_SCATTERING: snprintf(kname, sizeof kname, MDH_SCAT_SPLIT ? "k_scat_march<%d>" : "k_scattering<%d>", pf); break;
      case MDH_PASS_SCREEN: {
         // (the predicate of launch_screen_m below: mode 2 runs the variant of the optional paths only when a mip chain really exists)
         const bool alt = r->opt_mode == 0 && (r->opt_spec == 1 || r->opt_spec == 3 || (r->opt_spec == 2 && r->opt_mips && r->d_rad_mips[dst]));
         snprintf(kname, sizeof kname, "k_screen<%d, %d, %s, %s>", pf | (pow2 && r->opt_mode == 0 && !alt ? MDH_PF_POW2 : 0), r->opt_mode, r->opt_gbuffer ? "true" : "false", alt ? "true" : "false");
         break;
      }
      default: jit = false; break;
      }
      if (jit && !(jm = jit_module(r, {kname}))) { r->opt_jit = 0; jit = false; }
   }
   hipEvent_t e0 = nullptr, e1 = nullptr;
   if (r->opt_timing) {
      e0 = get_event(r);
      e1 = get_event(r);
      if (!e0 || !e1) return seterr(MDH_E_DEVICE, "hipEventCreate failed");
      HIP_TRY(hipEventRecord(e0, st));
   }
   switch (pass) {
   case MDH_PASS_RADIANCE: {
      long n = (long)(pr.probe_end - pr.probe_begin) * pr.rres * pr.rres;
      {
         const int G = MDH_RAD_PROBES_PER_WAVE, T = (G == 1) ? 8 : (G == 4) ? 4 : (G == 16) ? 2 : 1;
         if (pr.rres % T == 0) n = (long)((pr.probe_end - pr.probe_begin + G - 1) / G) * G * pr.rres * pr.rres;
      }
      if (n > 0) {
         // the rays in the order of the previous pass's primary-march lengths (RadOrder, mdh_kernels.h)
         const long rays = (long)(pr.probe_end - pr.probe_begin) * pr.rres * pr.rres;
         RadOrder ro = {nullptr, nullptr, (int)rays, 64, nullptr};
         // (chunks of whole workgroup strides, at most MDH_RO_MAX_CHUNKS of them)
         const long ro_chunk = std::max(2048l, ((rays + MDH_RO_MAX_CHUNKS - 1) / MDH_RO_MAX_CHUNKS + 1023) / 1024 * 1024);
         const int ro_chunks = (int)((rays + ro_chunk - 1) / ro_chunk);
         if (r->opt_rad_order && rays >= 8192 && rays < (1l << 31)) {
            if (rays > r->rad_rays_cap) {
               // (the buffers are only ever used by radiance passes, on the probe stream or the main stream)
               if (r->probe_stream) HIP_TRY(hipStreamSynchronize(r->probe_stream));
               HIP_TRY(hipStreamSynchronize(r->stream));
               r->rad_rays_cap = 0;
               r->rad_order_rays = 0;
               { void *q = r->d_rad_steps; r->d_rad_steps = nullptr; if (q) HIP_TRY(hipFree(q)); }
               { void *q = r->d_rad_order; r->d_rad_order = nullptr; if (q) HIP_TRY(hipFree(q)); }
               { void *q = r->d_rad_hist; r->d_rad_hist = nullptr; if (q) HIP_TRY(hipFree(q)); }
               HIP_TRY(hipMalloc(&r->d_rad_steps, rays));
               HIP_TRY(hipMalloc(&r->d_rad_order, rays * sizeof(unsigned)));
               HIP_TRY(hipMalloc(&r->d_rad_hist, 256 * MDH_RO_MAX_CHUNKS * sizeof(unsigned)));
               r->rad_rays_cap = rays;
               r->rad_order_rays = 0;
            }
            const bool have = r->rad_order_rays == rays && r->rad_order_begin == pr.probe_begin;
            if (have) { ro.order = r->d_rad_order; n = rays; }
            // A probe ray's march changes with the scene's geometry, not with time, lights or materials: the rays are sorted
            // again when primitives were set or added since (at most every MDH_RAD_RESORT_MOVING passes: a stale order costs
            // speed only, and three more launches per frame cost the host of a 0.4 ms frame 10 %) and every MDH_RAD_RESORT
            // passes besides -- the sort kernels (20 us) are out of almost every frame.
            ++r->rad_order_age;
            if (!have || r->rad_order_age >= MDH_RAD_RESORT || (r->rad_order_scene != r->geometry_edits && r->rad_order_age >= MDH_RAD_RESORT_MOVING)) ro.steps = r->d_rad_steps;
         }
         // RadOrder::fill: quarter or half filled wavefronts for launches that leave most slots of the chip empty.  Measured
         // (rank 0's radiance slice of an 8-way sharded frame, 1 024 full wavefronts on 1 024 SIMDs): 0.094 ms full, 0.096 ms
         // half filled, 0.125 ms quarter filled (profiles/r03_x_radiance_partial_fill.log) -- the rays are sorted by their
         // march lengths already, so a wavefront's lanes end together whatever their number, and four wavefronts of 16 issue
         // four times the instructions.  Dropped: full wavefronts unless MADARCH_HIP_RAD_FILL says otherwise.
         {
            static const int fill_env = [] { const char *e = getenv("MADARCH_HIP_RAD_FILL"); return e ? atoi(e) : 0; }();
            ro.fill = fill_env == 16 || fill_env == 32 ? fill_env : 64;
            n = (n + ro.fill - 1) / ro.fill * 64;
         }
         int blocks = (int)((n + MDH_BLOCK - 1) / MDH_BLOCK);
         const size_t lds = lds_bytes_march(r);
         if (jit) {
            struct { KScene sc; KProbes pr; int first_round; RadOrder ro; } args = {r->ks, pr, rad_first_round(r, nullptr, jm->fn[kname], lds, blocks), ro};
            int rc = jit_launch(jm->fn[kname], blocks, MDH_BLOCK, lds, st, args);
            if (rc != MDH_OK) return rc;
         } else {
#define MDH_LAUNCH_RAD_(...) hipLaunchKernelGGL((__VA_ARGS__), dim3(blocks), dim3(MDH_BLOCK), lds, st, r->ks, pr, rad_first_round(r, (const void *)(__VA_ARGS__), nullptr, lds, blocks), ro)
#define MDH_LAUNCH_RAD(P) do { if (rad_small_launch(r)) MDH_LAUNCH_RAD_(k_radiance<P, true>); else MDH_LAUNCH_RAD_(k_radiance<P, false>); } while (0)
            // MADARCH_HIP_RAD_SPLIT=1 (an experiment; the brute-force power-of-two variant only): the pass as two
            // kernels -- marches and direct light into per-ray records, then the probe code from the records
            static const bool rad_split = [] { const char *e = getenv("MADARCH_HIP_RAD_SPLIT"); return e && atoi(e) == 1; }();
            if (rad_split && pow2 && !has_custom && !(pf & MDH_PF_PART)) {
               const size_t need = (size_t)blocks * MDH_BLOCK * sizeof(RadRecord);
               if (need > r->rad_rec_bytes) {
                  if (r->probe_stream) HIP_TRY(hipStreamSynchronize(r->probe_stream));
                  HIP_TRY(hipStreamSynchronize(r->stream));
                  if (r->d_rad_rec) HIP_TRY(hipFree(r->d_rad_rec));
                  r->d_rad_rec = nullptr; r->rad_rec_bytes = 0;
                  HIP_TRY(hipMalloc(&r->d_rad_rec, need));
                  r->rad_rec_bytes = need;
               }
               ro.rec = (RadRecord *)r->d_rad_rec;
               if (rad_small_launch(r)) {
                  hipLaunchKernelGGL((k_radiance<MDH_PF_POW2, true, 1>), dim3(blocks), dim3(MDH_BLOCK), lds, st, r->ks, pr, 0, ro);
                  hipLaunchKernelGGL((k_radiance<MDH_PF_POW2, true, 2>), dim3(blocks), dim3(MDH_BLOCK), lds, st, r->ks, pr, 0, ro);
               } else {
                  hipLaunchKernelGGL((k_radiance<MDH_PF_POW2, false, 1>), dim3(blocks), dim3(MDH_BLOCK), lds, st, r->ks, pr, rad_first_round(r, (const void *)(k_radiance<MDH_PF_POW2, false, 1>), nullptr, lds, blocks), ro);
                  hipLaunchKernelGGL((k_radiance<MDH_PF_POW2, false, 2>), dim3(blocks), dim3(MDH_BLOCK), lds, st, r->ks, pr, rad_first_round(r, (const void *)(k_radiance<MDH_PF_POW2, false, 2>), nullptr, lds, blocks), ro);
               }
            } else
            if (pow2 && !has_custom) {
               if (pf & MDH_PF_FALLBACK) MDH_LAUNCH_RAD(MDH_PF_PART | MDH_PF_POW2 | MDH_PF_FALLBACK);
               else if (psmall) MDH_LAUNCH_RAD(MDH_PF_PART | MDH_PF_POW2 | MDH_PF_PSMALL);
               else if (pf & MDH_PF_PART) MDH_LAUNCH_RAD(MDH_PF_PART | MDH_PF_POW2);
               else if (room) MDH_LAUNCH_RAD(MDH_PF_POW2 | MDH_PF_ROOM);
               else MDH_LAUNCH_RAD(MDH_PF_POW2);
            } else
               switch (pfk) {
               case MDH_PF_ROOM: MDH_LAUNCH_RAD(MDH_PF_ROOM); break;
               case MDH_PF_PART | MDH_PF_PSMALL: MDH_LAUNCH_RAD(MDH_PF_PART | MDH_PF_PSMALL); break;
               case 0: MDH_LAUNCH_RAD(0); break;
               case 1: MDH_LAUNCH_RAD(1); break;
               case 2: MDH_LAUNCH_RAD(2); break;
               case 9: MDH_LAUNCH_RAD(9); break;
               default: MDH_LAUNCH_RAD(3); break;
               }
#undef MDH_LAUNCH_RAD_
#undef MDH_LAUNCH_RAD
         }
         if (ro.steps) { // the next pass's order from this pass's step counts
            hipLaunchKernelGGL(k_rad_hist, dim3(ro_chunks), dim3(256), 0, st, (const unsigned char *)r->d_rad_steps, (int)rays, (int)ro_chunk, r->d_rad_hist);
            hipLaunchKernelGGL(k_rad_scan, dim3(1), dim3(256), 0, st, r->d_rad_hist, ro_chunks);
            hipLaunchKernelGGL(k_rad_scatter, dim3(ro_chunks), dim3(256), 0, st, (const unsigned char *)r->d_rad_steps, (int)rays, (int)ro_chunk, (const unsigned *)r->d_rad_hist, r->d_rad_order);
            r->rad_order_rays = rays;
            r->rad_order_begin = pr.probe_begin;
            r->rad_order_age = 0;
            r->rad_order_scene = r->geometry_edits;
         } else if (!ro.order)
            r->rad_order_rays = 0;
      }
      break;
   }
   case MDH_PASS_IRRADIANCE: {
      if (r->opt_world > 1 && r->opt_irr_all) { pr.probe_begin = 0; pr.probe_end = probe_total(r); } // every rank, every probe
      int n = pr.probe_end - pr.probe_begin; // one workgroup per probe, its taps staged in LDS
      size_t lds = (size_t)2 * pr.rres * pr.rres * sizeof(float4);
      float *tap_planes = nullptr;
#if MDH_FAST_NUMERICS
      if (pr.ires * pr.ires <= 64) lds += (size_t)4 * 64 * sizeof(float4); // (the experiment's fold: all taps staged, four partial sums per texel)
      else
#endif
      if (MDH_IRR_CHANNELS && pr.ires * pr.ires <= 64 && MDH_IRR_BLOCK == 256 && n > 0) { // (k_irradiance: a channel per wavefront, the taps through device memory)
         const size_t need = (size_t)n * 6 * MDH_IRR_CHANNELS_PLANE(pr.rres * pr.rres);
         if (need > r->irr_taps_cap) { // (the scratch of irradiance passes only, which follow one another on their stream)
            if (r->probe_stream) HIP_TRY(hipStreamSynchronize(r->probe_stream));
            HIP_TRY(hipStreamSynchronize(r->stream));
            r->irr_taps_cap = 0;
            { void *q = r->d_irr_taps; r->d_irr_taps = nullptr; if (q) HIP_TRY(hipFree(q)); }
            HIP_TRY(hipMalloc(&r->d_irr_taps, need * sizeof(float)));
            r->irr_taps_cap = need;
         }
         tap_planes = r->d_irr_taps;
         lds = MDH_IRR_CHANNELS_LDS;
      } else
      if (MDH_IRR_WPRE && pr.ires * pr.ires <= 64 && MDH_IRR_BLOCK == 256) lds = MDH_IRR_WPRE_LDS; // (k_irradiance: the weights' pipeline)
      else
      if (MDH_IRR_CHUNK && pr.ires * pr.ires <= 64 && lds > (size_t)4 * MDH_IRR_CHUNK * sizeof(float4)) lds = (size_t)4 * MDH_IRR_CHUNK * sizeof(float4); // two chunk buffers
#ifdef MDH_IRR_LDS_PAD
      lds += MDH_IRR_LDS_PAD; // (experiment: what the pass's LDS footprint costs it beside the march kernels of frames in flight)
#endif
      if (lds > 64 * 1024) { // radiance tiles beyond 45 x 45 texels: up to the whole 160 KiB of a CU (70 x 70)
         if (lds > 160 * 1024) return seterr(MDH_E_INVALID, "radiance resolution too large for the irradiance pass (160 KiB of LDS: at most 70)");
         if (!r->irr_lds_granted) { // (the attribute belongs to the function on THIS device: kept per renderer, not per process)
            HIP_TRY(hipFuncSetAttribute((const void *)k_irradiance, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            r->irr_lds_granted = true;
         }
      }
      // (hysteresis: the previous frame's irradiance is set `src` -- the same set when the pass runs in place)
      if (n > 0) hipLaunchKernelGGL(k_irradiance, dim3(n), dim3(MDH_IRR_BLOCK), lds, st, pr, (const void *)r->d_irr2[src], (float)r->opt_hyst / 1000.0f, tap_planes);
      break;
   }
   case MDH_PASS_VISIBILITY: {
      KVolumetrics vol = make_vol(r, true, dst);
      long n = (long)vol.vw * vol.vh * vol.vz;
      if (n > 0) {
#if MDH_VIS_QUEUE
         const long per_block = (long)MDH_BLOCK * MDH_VIS_ROUNDS; // every wavefront owns MDH_VIS_ROUNDS x 64 froxels (k_visibility)
         int blocks = (int)((n + per_block - 1) / per_block);
#else
         const long per_block = (long)MDH_BLOCK * MDH_VIS_LOOP;
         int blocks = (int)((n + per_block - 1) / per_block);
#endif
         // inside a frame the launch also marches the scattering texels' camera rays (k_visibility's second part, mdh_kernels.h)
         const int vis_blocks = blocks;
         if (MDH_SCAT_SPLIT && r->fuse_scat_march) blocks += (int)(((long)vol.sw * vol.sh + MDH_BLOCK - 1) / MDH_BLOCK);
         const size_t vis_lds = lds_bytes(r) + (MDH_VIS_QUEUE == 2 ? (size_t)(MDH_BLOCK / 64) * MDH_VIS_Q_FLOATS * sizeof(float) : 0); // (the second form of the ray queue keeps its rays in LDS)
         if (jit) {
            struct { KScene sc; KVolumetrics vol; KCamera cam; int vis_blocks; } args = {r->ks, vol, cam, vis_blocks};
            int rc = jit_launch(jm->fn[kname], blocks, MDH_BLOCK, vis_lds, st, args);
            if (rc != MDH_OK) return rc;
         } else
            MDH_LAUNCH_PF(k_visibility, dim3(blocks), dim3(MDH_BLOCK), vis_lds, r->ks, vol, cam, vis_blocks);
      }
      break;
   }
   case MDH_PASS_SCATTERING: {
      KVolumetrics vol = make_vol(r, true, dst);
      long n = (long)vol.sw * vol.sh;
      if (n > 0) {
         int blocks = (int)((n + MDH_BLOCK - 1) / MDH_BLOCK);
         if (jit) {
            if (!(MDH_SCAT_SPLIT && r->fuse_scat_march)) {
               struct { KScene sc; KVolumetrics vol; KCamera cam; } args = {r->ks, vol, cam};
               int rc = jit_launch(jm->fn[kname], blocks, MDH_BLOCK, lds_bytes(r), st, args);
               if (rc != MDH_OK) return rc;
            }
         } else
#if MDH_SCAT_SPLIT
            { if (!r->fuse_scat_march) MDH_LAUNCH_PF(k_scat_march, dim3(blocks), dim3(MDH_BLOCK), lds_bytes(r), r->ks, vol, cam); } // (inside a frame the visibility pass's launch has marched them)
#else
            MDH_LAUNCH_PF(k_scattering, dim3(blocks), dim3(MDH_BLOCK), lds_bytes(r), r->ks, vol, cam);
#endif
#if MDH_SCAT_SPLIT
         // the texels' lengths are in place: their steps, spread over lanes and folded in step order (k_scat_fold, mdh_kernels.h)
         HIP_TRY(hipGetLastError());
         hipLaunchKernelGGL(k_scat_fold, dim3((unsigned)((n + MDH_SCAT_TEXELS - 1) / MDH_SCAT_TEXELS)), dim3(MDH_SCAT_BLOCK), 0, st, vol);
#endif
      }
      break;
   }
   case MDH_PASS_SCREEN: {
      KVolumetrics vol = make_vol(r, r->vol.enabled != 0 && r->opt_mode == 0, dst);
      ScreenArgs a;
      a.W = r->W; a.H = r->H;
      a.tiles_x = (r->W + 7) / 8;
      a.n_tiles = a.tiles_x * ((r->H + 7) / 8);
      a.rank = r->opt_rank; a.world = r->opt_world;
      a.ao_steps = r->opt_ao;
      a.spec_mode = r->opt_spec;
      if (r->opt_mips && r->opt_mode == 0 && r->opt_spec != 0 && r->d_rad_mips[dst]) { // MDH_OPT_RADIANCE_MIPS: the levels of the atlas this pass reads
         int mrc = build_rad_mips(r, dst, st);
         if (mrc != MDH_OK) return mrc;
         pr.rad_mips = r->d_rad_mips[dst];
      }
      a.fb = r->d_fb2[fbix]; a.gb_index = (int *)r->d_gb2[fbix][0]; a.gb_t = (float *)r->d_gb2[fbix][1]; a.gb_steps = (int *)r->d_gb2[fbix][2];
      a.window = nullptr;
      if (r->opt_window == 1 || (r->opt_window == 2 && r->swaps > 0)) { // the window's pixels straight into pinned host memory (mdh_swap_buffers)
         int wrc = window_slot(r, a.rank, a.world, &a.window);
         if (wrc != MDH_OK) return wrc;
      }
      // other ranks' tiles read 0: cleared when the buffer last held another rank's (or a whole) frame, not every frame
      if (a.world > 1 && (r->fb_owner[fbix][0] != a.rank || r->fb_owner[fbix][1] != a.world))
         HIP_TRY(hipMemsetAsync(r->d_fb2[fbix], 0, (size_t)r->W * r->H * sizeof(float4), st));
      r->fb_owner[fbix][0] = a.rank; r->fb_owner[fbix][1] = a.world;
      int own_tiles = (a.n_tiles - a.rank + a.world - 1) / a.world;
      a.n_own = own_tiles;
      a.order = nullptr;
      a.cost = nullptr;
      // MDH_OPT_SCREEN_ORDER: the tiles in the order of an earlier pass's wavefront durations, slowest first
      bool sort_after = false;
      const int si_st = stream_index(r, st);
      if (r->opt_scr_order && own_tiles >= 2048) {
         if (!r->d_scr_cost) {
            HIP_TRY(hipMalloc(&r->d_scr_cost, a.n_tiles));
            for (int q = 0; q < 2; ++q) HIP_TRY(hipMalloc(&r->d_scr_order[q], (size_t)a.n_tiles * sizeof(unsigned)));
            HIP_TRY(hipMalloc(&r->d_scr_hist, 256 * MDH_RO_MAX_CHUNKS * sizeof(unsigned)));
            HIP_TRY(hipEventCreateWithFlags(&r->ev_scr_sort, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&r->ev_scr_other, hipEventDisableTiming));
         }
         float cam_now[12];
         memcpy(cam_now, r->cam_pos, 12); memcpy(cam_now + 3, r->cam_m, 36);
         const bool have = r->scr_order_cur >= 0 && r->scr_order_n == own_tiles && r->scr_order_rank == a.rank && r->scr_order_world == a.world;
         if (have) {
            a.order = r->d_scr_order[r->scr_order_cur];
            if (r->scr_sort_seen[si_st] != r->scr_sort_version) { // the sort ran on another stream
               HIP_TRY(hipStreamWaitEvent(st, r->ev_scr_sort, 0));
               r->scr_sort_seen[si_st] = r->scr_sort_version;
            }
         }
         // which tiles are slow follows the camera and the geometry: sorted again when either changed since (at most every
         // MDH_RAD_RESORT_MOVING passes: a stale order costs speed only) and every MDH_RAD_RESORT passes besides
         ++r->scr_order_age;
         const bool moved = memcmp(cam_now, r->scr_order_cam, sizeof cam_now) != 0 || r->scr_order_geom != r->geometry_edits;
         if (!have || r->scr_order_age >= MDH_RAD_RESORT || (moved && r->scr_order_age >= MDH_RAD_RESORT_MOVING)) {
            sort_after = true;
            a.cost = r->d_scr_cost;
            memcpy(r->scr_order_cam, cam_now, sizeof cam_now);
         }
      } else
         r->scr_order_cur = -1;
      // MDH_OPT_SCREEN_SPLIT (ScreenArgs::split): launches that leave most of the chip's wavefront slots empty give a tile to
      // two or four wavefronts; never with a tile order (launches of 2 048 tiles and more)
      a.split = 0;
      a.split_first = 0;
      if (!a.order && !a.cost && r->opt_scr_split > 0 && a.world == 1) { // (a rank's scattered tiles of a sharded frame: measured 1 - 3 % slower split, profiles/r04_x_split_tiles.log)
         if ((long)own_tiles * 4 <= r->opt_scr_split) a.split = 2;
         else if ((long)own_tiles * 2 <= r->opt_scr_split) a.split = 1;
      }
      // ... and of an ordered launch the slowest tiles (MADARCH_HIP_SPLIT_FIRST, thousandths of the launch's tiles; a pass that
      // records the tiles' durations draws every tile with one wavefront: the key is the tile's)
      static const int split_first_env = [] { const char *e = getenv("MADARCH_HIP_SPLIT_FIRST"); return e ? atoi(e) : MDH_SCREEN_SPLIT_FIRST_PERMILLE; }();
      if (a.order && !a.cost && r->opt_scr_split > 0 && split_first_env > 0) a.split_first = (int)((long)own_tiles * split_first_env / 1000);
      if (own_tiles > 0) {
         const long waves = ((long)own_tiles << a.split) + 3l * a.split_first;
         int blocks = (int)((waves + (MDH_BLOCK / 64) - 1) / (MDH_BLOCK / 64));
         if (jit) {
            struct { KScene sc; KProbes pr; KVolumetrics vol; KCamera cam; ScreenArgs a; } args = {r->ks, pr, vol, cam, a};
            int rc = jit_launch(jm->fn[kname], blocks, MDH_BLOCK, lds_bytes_screen(r), st, args);
            if (rc != MDH_OK) return rc;
         } else
         switch (pfk) {
         case MDH_PF_ROOM: launch_screen_m<MDH_PF_ROOM>(r, st, pr, vol, cam, a, blocks, pow2); break;
         case MDH_PF_PART | MDH_PF_PSMALL: launch_screen_m<MDH_PF_PART | MDH_PF_PSMALL>(r, st, pr, vol, cam, a, blocks, pow2); break;
         case 0: launch_screen_m<0>(r, st, pr, vol, cam, a, blocks, pow2); break;
         case 1: launch_screen_m<1>(r, st, pr, vol, cam, a, blocks, pow2); break;
         case 2: launch_screen_m<2>(r, st, pr, vol, cam, a, blocks, pow2); break;
         case 9: launch_screen_m<9>(r, st, pr, vol, cam, a, blocks, pow2); break;
         default: launch_screen_m<3>(r, st, pr, vol, cam, a, blocks, pow2); break;
         }
      }
      if (sort_after) { // later passes' order from this pass's durations, into the buffer no pass in flight reads
         HIP_TRY(hipGetLastError());
         if (r->opt_timing && e1) { // (the sort is no part of the pass's time: ADVICE r03)
            HIP_TRY(hipEventRecord(e1, st));
            r->pending.push_back({pass, e0, e1});
            e0 = e1 = nullptr;
         }
         const int nb = r->scr_order_cur == 0 ? 1 : 0;
         // (passes that read that buffer were launched before the previous sort; on the other screen stream nothing
         //  orders them against this stream when no probe passes run: wait for what that stream holds)
         hipStream_t other = st == r->alt_stream ? r->stream : r->alt_stream;
         if (other && other != st && r->scr_order_cur >= 0) {
            HIP_TRY(hipEventRecord(r->ev_scr_other, other));
            HIP_TRY(hipStreamWaitEvent(st, r->ev_scr_other, 0));
         }
         const long chunk = std::max(2048l, (((long)own_tiles + MDH_RO_MAX_CHUNKS - 1) / MDH_RO_MAX_CHUNKS + 1023) / 1024 * 1024);
         const int chunks = (int)((own_tiles + chunk - 1) / chunk);
         // Sorted as a whole every screen pass measured is faster on its own (BASELINE config 3: +4.5 %, config 2: +59 %,
         // config 5's frame on one GPU: +7 %) and, with frames in flight, every frame but one kind: where the probe passes
         // are a large share of the frame (config 3 at 1080p: 524 288 probe rays for 2 M pixels), a screen pass that
         // holds every wavefront slot to its very end keeps the NEXT frame's probe passes -- the head of that frame's
         // dependency chain -- waiting for slots that the thin tail of an image-order pass hands over early: -3 %.
         // There only the tiles that took twice the median and more go to the front (the ones that make the tail) and
         // all others keep their place: +-0 in flight (profiles/r03_x_screen_tile_order.log).
         // MADARCH_HIP_ORDER_FLOOR (thousandths of the median) overrides, for experiments.
         static const int floor_env = [] { const char *e = getenv("MADARCH_HIP_ORDER_FLOOR"); return e ? atoi(e) : -1; }();
         const long probe_rays = r->opt_mode == 0 ? (long)(pr.probe_end - pr.probe_begin) * pr.rres * pr.rres : 0;
         const bool probe_heavy = probe_rays * 10 >= (long)own_tiles * 64;
         const int floor_permille = floor_env >= 0 ? floor_env : (r->frame_pipelined && r->in_frame_passes && probe_heavy ? 2000 : 0);
         hipLaunchKernelGGL(k_rad_hist, dim3(chunks), dim3(256), 0, st, (const unsigned char *)r->d_scr_cost, own_tiles, (int)chunk, r->d_scr_hist);
         if (floor_permille > 0) {
            hipLaunchKernelGGL(k_order_floor, dim3(chunks), dim3(256), 0, st, r->d_scr_cost, own_tiles, (int)chunk, (const unsigned *)r->d_scr_hist, chunks, floor_permille);
            hipLaunchKernelGGL(k_rad_hist, dim3(chunks), dim3(256), 0, st, (const unsigned char *)r->d_scr_cost, own_tiles, (int)chunk, r->d_scr_hist);
         }
         hipLaunchKernelGGL(k_rad_scan, dim3(1), dim3(256), 0, st, r->d_scr_hist, chunks);
         hipLaunchKernelGGL(k_order_scatter_stable, dim3(chunks), dim3(256), 0, st, (const unsigned char *)r->d_scr_cost, own_tiles, (int)chunk, (const unsigned *)r->d_scr_hist, r->d_scr_order[nb]);
         HIP_TRY(hipGetLastError());
         HIP_TRY(hipEventRecord(r->ev_scr_sort, st));
         ++r->scr_sort_version;
         r->scr_sort_seen[si_st] = r->scr_sort_version;
         r->scr_order_cur = nb;
         r->scr_order_n = own_tiles; r->scr_order_rank = a.rank; r->scr_order_world = a.world;
         r->scr_order_age = 0;
         r->scr_order_geom = r->geometry_edits;
      }
      break;
   }
   default: return seterr(MDH_E_INVALID, "bad pass");
   }
   HIP_TRY(hipGetLastError());
   if (pass != MDH_PASS_IRRADIANCE) {
      int trc = table_release(r, st);
      if (trc != MDH_OK) return trc;
   }
   if (r->opt_timing && e1) {
      HIP_TRY(hipEventRecord(e1, st));
      r->pending.push_back({pass, e0, e1}); // (folded at frame boundaries: bound_timing)
   }
#ifdef MDH_DIAG
   { // the diagnostic build waits for every pass and takes its work counters
      HIP_TRY(hipStreamSynchronize(st));
      HIP_TRY(hipMemcpyFromSymbol(r->work[pass], HIP_SYMBOL(g_work), sizeof r->work[pass]));
      unsigned long long z[4] = {0, 0, 0, 0};
      HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_work), z, sizeof z));
   }
#endif
   return MDH_OK;
}

extern "C" int32_t mdh_render_pass(mdh_renderer *r, int32_t pass)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (r->in_frame) return seterr(MDH_E_STATE, "a frame is open: use mdh_frame_probe_pass / mdh_frame_end");
   if (r->opt_rank >= r->opt_world) return seterr(MDH_E_STATE, "MDH_OPT_RANK is not below MDH_OPT_WORLD");
   int rc = ensure_committed(r);
   if (rc != MDH_OK) return rc;
   if ((rc = join_main(r)) != MDH_OK) return rc;
   r->main_dirty = true;
   if ((rc = run_pass(r, pass, r->stream, r->last, r->last)) != MDH_OK) return rc;
   return bound_timing(r);
}
// Render (renderers.adb:302-321).
//
// Frame overlap (MDH_OPT_FRAME_OVERLAP, on by default, single-GPU renderers on their own stream).
// The probe passes of frame N+1 do not depend on the screen pass of frame N -- they need the
// irradiance atlas frame N produced and nothing else -- and the screen passes of two frames are
// independent of each other.  Every kernel of a frame ends in a tail of a few slow wavefronts
// (the radiance pass spends more than half of its time below 20 % occupancy; the irradiance pass
// fills a fraction of the chip), so serial frames leave the chip idle a good part of the time.
// Pipelined frames use three HIP streams (four with volumetrics), MDH_ATLAS_SETS = 3 atlas sets and two framebuffers;
// frame N works on set c = N mod 3 and reads the irradiance of set `last` = (N - 1) mod 3, its screen pass draws on
// stream and framebuffer p = N mod 2:
//    probe stream: wait screen(N-3) (the last reader of atlas set c)
//                  -> radiance(irr[last] -> rad[c]) -> [exchange] -> irradiance(rad[c] -> irr[c]) -> ev_probe[c]
//    volumetric stream (renderers with volumetrics): wait screen(N-3) -> visibility, scattering of set c -> ev_vol[c]
//    screen stream p (main / alternate): wait ev_probe[c] (and ev_vol[c]) -> screen(atlas set c -> framebuffer p)
// so that screen(N+1) starts while screen(N) drains and the probe passes of N+2 fill in behind.  Anything outside a pipelined frame first orders the main stream after the others
// (join_main), and the next pipelined frame orders the others after the main stream: results are
// those of the serial order bit for bit.
// A frame in three steps, for callers that put work of their own between the passes (the
// one-process-per-GPU runs all-gather the atlas slices after each probe pass, on the stream
// mdh_probe_stream names): begin -> probe passes -> end.  mdh_render is exactly
// begin, radiance, irradiance, end.
static hipStream_t frame_probe_stream(const mdh_renderer *r) { return r->frame_pipelined ? r->probe_stream : r->stream; }
extern "C" int32_t mdh_frame_begin(mdh_renderer *r)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (r->in_frame) return seterr(MDH_E_STATE, "a frame is already open");
   if (r->opt_rank >= r->opt_world) return seterr(MDH_E_STATE, "MDH_OPT_RANK is not below MDH_OPT_WORLD");
   r->frame_pipelined = r->opt_overlap && r->stream == r->own_stream; // (modes 1 and 2 have no probe passes: their screen passes still alternate streams)
   // an edited scene goes up on the stream that uses it first; frames in flight keep the table buffers they were launched with
   int rc = ensure_committed(r, r->frame_pipelined && r->opt_mode == 0 ? r->probe_stream : nullptr);
   if (rc != MDH_OK) return rc;
   if (!r->frame_pipelined) {
      if ((rc = join_main(r)) != MDH_OK) return rc;
      r->main_dirty = true;
      r->frame_cur = r->last;
   } else {
      // modes 1 and 2 run no probe passes: the atlas sets stay as they are (their screen passes still alternate)
      const int cur = r->opt_mode == 0 ? (r->last + 1) % mdh_renderer::NSETS : r->last;
      if (r->main_dirty) { // the other streams have to see everything that went to the main stream meanwhile
         if ((rc = join_main(r)) != MDH_OK) return rc;
         HIP_TRY(hipEventRecord(r->ev_join, r->stream));
         HIP_TRY(hipStreamWaitEvent(r->probe_stream, r->ev_join, 0));
         HIP_TRY(hipStreamWaitEvent(r->alt_stream, r->ev_join, 0));
         if (r->vol_stream) HIP_TRY(hipStreamWaitEvent(r->vol_stream, r->ev_join, 0)); // (the volumetric passes of pipelined frames: frame_end_passes)
         r->main_dirty = false;
      } else if (r->opt_mode == 0 && r->ev_screen_valid[cur]) { // the last screen pass that read atlas set cur
         HIP_TRY(hipStreamWaitEvent(r->probe_stream, r->ev_screen[cur], 0));
         if (r->vol_stream) HIP_TRY(hipStreamWaitEvent(r->vol_stream, r->ev_screen[cur], 0)); // (it read the froxels of set cur as well)
      }
      r->frame_cur = cur;
   }
   r->in_frame = true;
   return MDH_OK;
}
// a pass of an open frame failed: close the frame and make the next one re-join every stream (the work already
// enqueued on the probe stream is ordered against nothing else)
static int abandon_frame(mdh_renderer *r, int rc)
{
   r->in_frame = false;
   r->main_dirty = true;
   if (r->probe_stream) (void)hipStreamSynchronize(r->probe_stream);
   if (r->vol_stream) (void)hipStreamSynchronize(r->vol_stream);
   return rc;
}
extern "C" int32_t mdh_frame_probe_pass(mdh_renderer *r, int32_t pass)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (!r->in_frame) return seterr(MDH_E_STATE, "no open frame");
   if (pass != MDH_PASS_RADIANCE && pass != MDH_PASS_IRRADIANCE) return seterr(MDH_E_INVALID, "not a probe pass");
   if (r->opt_mode != 0) return MDH_OK; // modes 1 and 2 draw without probes (renderers.adb:302-321 runs them anyway; nothing reads them)
   return run_pass(r, pass, frame_probe_stream(r), r->last, r->frame_cur);
}
#ifndef MDH_VOL_OWN_STREAM
#define MDH_VOL_OWN_STREAM 1
#endif
static int frame_end_passes(mdh_renderer *r);
extern "C" int32_t mdh_frame_end(mdh_renderer *r)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (!r->in_frame) return seterr(MDH_E_STATE, "no open frame");
   r->in_frame = false;
   r->in_frame_passes = true;
   const int rc = frame_end_passes(r);
   r->in_frame_passes = false;
   return rc == MDH_OK ? bound_timing(r) : abandon_frame(r, rc);
}
static int frame_end_passes(mdh_renderer *r)
{
   int rc;
   const int cur = r->frame_cur;
   if (!r->frame_pipelined) {
      if (r->opt_mode == 0 && r->vol.enabled) {
         r->fuse_scat_march = true; // (both passes follow each other: the first one's launch marches the second one's camera rays)
         rc = run_pass(r, MDH_PASS_VISIBILITY, r->stream, cur, cur);
         if (rc == MDH_OK) rc = run_pass(r, MDH_PASS_SCATTERING, r->stream, cur, cur);
         r->fuse_scat_march = false;
         if (rc != MDH_OK) return rc;
      }
      return run_pass(r, MDH_PASS_SCREEN, r->stream, cur, cur);
   }
   const bool dual = r->opt_overlap > 1;
   // (level 1: all screen passes on the main stream, after whatever the alternate stream still holds)
   if (!dual && r->alt_pending && (rc = join_main(r)) != MDH_OK) return rc;
   // framebuffer 1 is only ever written from the alternate stream and framebuffer 0 from the main stream, so
   // each buffer's writes are ordered by its stream; frames that cannot alternate draw on (main, 0)
   r->scr_parity = dual ? r->scr_parity ^ 1 : 0;
   hipStream_t screen_stream = r->scr_parity ? r->alt_stream : r->stream;
   const int fbix = r->scr_parity;
   if (r->opt_mode == 0 && r->vol.enabled) {
      // camera-only passes into this frame's set: on a stream of their own, beside this frame's probe
      // passes and the previous screen pass -- they are a few hundred wavefronts each and wait for nothing the probes make
      hipStream_t vs = MDH_VOL_OWN_STREAM && r->vol_stream ? r->vol_stream : r->probe_stream;
      r->fuse_scat_march = true;
      rc = run_pass(r, MDH_PASS_VISIBILITY, vs, cur, cur);
      if (rc == MDH_OK) rc = run_pass(r, MDH_PASS_SCATTERING, vs, cur, cur);
      r->fuse_scat_march = false;
      if (rc != MDH_OK) return rc;
      if (vs != r->probe_stream) {
         HIP_TRY(hipEventRecord(r->ev_vol[cur], vs));
         HIP_TRY(hipStreamWaitEvent(screen_stream, r->ev_vol[cur], 0));
      }
   }
   HIP_TRY(hipEventRecord(r->ev_probe[cur], r->probe_stream));
   HIP_TRY(hipStreamWaitEvent(screen_stream, r->ev_probe[cur], 0));
   if ((rc = run_pass(r, MDH_PASS_SCREEN, screen_stream, cur, cur, fbix)) != MDH_OK) return rc;
   HIP_TRY(hipEventRecord(r->ev_screen[cur], screen_stream));
   r->ev_screen_valid[cur] = true;
   r->last = cur;
   r->fb_last = fbix;
   if (screen_stream == r->alt_stream) r->alt_pending = true;
   return MDH_OK;
}
extern "C" int32_t mdh_render(mdh_renderer *r)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   int rc;
   if ((rc = mdh_frame_begin(r)) != MDH_OK) return rc;
   // (with a communicator: the exchanges of the sharded schedule between the probe passes, mdh_frame_exchange)
   if ((rc = mdh_frame_probe_pass(r, MDH_PASS_RADIANCE)) != MDH_OK || (rc = mdh_frame_exchange(r, MDH_TEX_RADIANCE)) != MDH_OK ||
       (rc = mdh_frame_probe_pass(r, MDH_PASS_IRRADIANCE)) != MDH_OK ||
       (!r->opt_irr_all && (rc = mdh_frame_exchange(r, MDH_TEX_IRRADIANCE)) != MDH_OK))
      return abandon_frame(r, rc);
   return mdh_frame_end(r);
}

// ------------------------------------------------------------------ one frame on N GPUs: the communicator
// One process per GPU; the processes' renderers join an RCCL communicator and mdh_render of every rank is then one
// frame of the sharded schedule (include/madarch_hip.h "one frame on the N GPUs of a node").  librccl is opened on
// first use: a process that already holds one (a PyTorch process: the wheel carries its own, same SONAME) gets that
// one, any other the system's.
struct RcclApi {
   decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
   decltype(&ncclCommInitRank) CommInitRank = nullptr;
   decltype(&ncclCommDestroy) CommDestroy = nullptr;
   decltype(&ncclCommAbort) CommAbort = nullptr;
   decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
   decltype(&ncclAllGather) AllGather = nullptr;
   decltype(&ncclBroadcast) Broadcast = nullptr;
   decltype(&ncclAllReduce) AllReduce = nullptr;
   decltype(&ncclReduce) Reduce = nullptr;
   decltype(&ncclGroupStart) GroupStart = nullptr;
   decltype(&ncclGroupEnd) GroupEnd = nullptr;
   decltype(&ncclGetErrorString) GetErrorString = nullptr;
   bool ok = false;
};
static const RcclApi &rccl_api()
{
   static RcclApi api;
   static std::once_flag once;
   std::call_once(once, [] {
      void *h = nullptr;
      const char *env = getenv("MADARCH_HIP_RCCL_LIBRARY");
      if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
         if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) return;
#define MDH_RCCL_SYM(field, sym) api.field = (decltype(api.field))dlsym(h, #sym)
      MDH_RCCL_SYM(GetUniqueId, ncclGetUniqueId); MDH_RCCL_SYM(CommInitRank, ncclCommInitRank); MDH_RCCL_SYM(CommDestroy, ncclCommDestroy);
      MDH_RCCL_SYM(CommAbort, ncclCommAbort); MDH_RCCL_SYM(CommGetAsyncError, ncclCommGetAsyncError); MDH_RCCL_SYM(AllGather, ncclAllGather);
      MDH_RCCL_SYM(Broadcast, ncclBroadcast); MDH_RCCL_SYM(AllReduce, ncclAllReduce); MDH_RCCL_SYM(Reduce, ncclReduce);
      MDH_RCCL_SYM(GroupStart, ncclGroupStart); MDH_RCCL_SYM(GroupEnd, ncclGroupEnd); MDH_RCCL_SYM(GetErrorString, ncclGetErrorString);
#undef MDH_RCCL_SYM
      api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.CommAbort && api.CommGetAsyncError && api.AllGather && api.Broadcast &&
               api.AllReduce && api.Reduce && api.GroupStart && api.GroupEnd && api.GetErrorString;
   });
   return api;
}
#define RCCL_TRY(expr)                                                                                             \
   do {                                                                                                            \
      ncclResult_t e_ = (expr);                                                                                    \
      if (e_ != ncclSuccess) {                                                                                     \
         snprintf(g_err, sizeof g_err, "%s failed: %s (%s:%d)", #expr, rccl_api().GetErrorString(e_), __FILE__, __LINE__); \
         return MDH_E_COMM;                                                                                        \
      }                                                                                                            \
   } while (0)
static int rccl_api_destroy(ncclComm_t c) { return rccl_api().ok && rccl_api().CommDestroy(c) == ncclSuccess ? MDH_OK : MDH_E_COMM; }
static int rccl_ready()
{
   return rccl_api().ok ? MDH_OK : seterr(MDH_E_COMM, "librccl.so.1 cannot be loaded (or lacks a symbol): no communicator; MADARCH_HIP_RCCL_LIBRARY names another file");
}

// ------------------------------------------------------------------ the peer exchange (include/madarch_hip.h)
// The sharded frame's exchange as device-to-device COPIES between processes of one node, ordered ON THE DEVICE.  What a rank
// shares: its radiance atlases (one hipIpcMemHandle per atlas set) and a block of FRAME NUMBERS in device memory, one per
// set.  Behind its radiance pass a rank stores the frame's number into its own block (k_peer_publish, on the probe stream);
// a rank that needs a peer's slice enqueues k_peer_wait -- one wavefront that polls the peer's number through the mapped
// handle until it has reached this frame's (system-scope loads; bounded: a peer that never arrives ends the wait with an
// error word in host memory instead of a hung queue) -- and behind it the copy of the slice out of the peer's atlas into
// its own (hipMemcpyAsync, device to device).  No host of any rank waits for anything: frames stay in flight, the screen
// pass of the previous frame runs beside the exchange.
// (Round 4 first used interprocess EVENTS for the ordering.  They work -- tests of 4 frames passed -- and fail: after some
// tens of frames hipStreamWaitEvent on a peer's event returned hipErrorInvalidValue, and HIP implements such waits as host
// callbacks.  hipStreamWaitValue32 enqueued before the peer's write stalled the peer's first HIP call in the probe
// (scripts/probes/ipc_probe.cpp, mode 2).  A polling wavefront needs nothing but memory.)
//
// Why reading a peer's slice is safe against the peer's NEXT radiance pass into the same set (three frames on): that
// pass comes behind the peer's irradiance pass of the frame before it, which waited for this rank's slice of that
// frame, which this rank's probe stream produced behind this copy.
struct PeerBlob { // MDH_PEER_BLOB_BYTES on the wire
   uint32_t magic, version;
   int32_t pid, device, nsets;
   uint64_t rad_bytes;
   hipIpcMemHandle_t rad[MDH_ATLAS_SETS];
   hipIpcMemHandle_t flags;
   uint64_t serial; // (of this export: a blob of an earlier session of the same process is refused)
};
static_assert(sizeof(PeerBlob) <= MDH_PEER_BLOB_BYTES, "MDH_PEER_BLOB_BYTES holds a rank's handles");
struct PeerState {
   bool active = false;
   int rank = 0, world = 1;
   bool exported = false;
   uint64_t serial = 0;
   unsigned *d_flags = nullptr;         // [MDH_ATLAS_SETS] the number of the last frame whose slice of set s is complete (device memory, shared)
   unsigned seq[MDH_ATLAS_SETS] = {0};
   unsigned *h_err = nullptr;           // pinned host word: a wait gave up
   struct Peer {
      void *rad[MDH_ATLAS_SETS] = {nullptr};
      unsigned *flags = nullptr;
   };
   std::vector<Peer> peers;
};
#ifndef MDH_PEER_SPIN_LIMIT
#define MDH_PEER_SPIN_LIMIT (1u << 23) // polls of k_peer_wait, about a microsecond each
#endif
__global__ void k_peer_publish(unsigned *flag, unsigned n)
{
   __hip_atomic_store(flag, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_peer_wait(const unsigned *flag, unsigned n, unsigned *err)
{
   if (threadIdx.x != 0) return;
   for (unsigned spin = 0; spin < MDH_PEER_SPIN_LIMIT; ++spin) {
      // (frame numbers only grow; the difference as a signed number survives the counter's wrap)
      if ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - n) >= 0) return;
      __builtin_amdgcn_s_sleep(32);
   }
   __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // the copy behind this wait reads a stale slice: the host is told
}
static bool peer_active(const mdh_renderer *r) { return r->peer && r->peer->active; }
static void peer_close_peers(PeerState *p)
{
   for (auto &q : p->peers) {
      for (int s = 0; s < MDH_ATLAS_SETS; ++s) {
         if (q.rad[s]) (void)hipIpcCloseMemHandle(q.rad[s]);
         q.rad[s] = nullptr;
      }
      if (q.flags) (void)hipIpcCloseMemHandle(q.flags);
      q.flags = nullptr;
   }
   p->peers.clear();
}
// leave the exchange: the peers' handles are closed, this rank is rank 0 of 1 again (its own exports stay valid until mdh_destroy)
static void peer_leave(mdh_renderer *r)
{
   PeerState *p = r->peer;
   if (!p) return;
   peer_close_peers(p);
   if (p->active) {
      p->active = false;
      r->opt_rank = 0;
      r->opt_world = 1;
      r->rad_order_rays = 0;
      r->fb_owner[0][0] = r->fb_owner[1][0] = -1;
   }
}
static void peer_abort(mdh_renderer *r) { (void)r; } // (nothing on the host waits: a wait on the device ends by itself, MDH_PEER_SPIN_LIMIT)
static void peer_drop(mdh_renderer *r)
{
   PeerState *p = r->peer;
   if (!p) return;
   peer_leave(r);
   if (p->d_flags) (void)hipFree(p->d_flags);
   if (p->h_err) (void)hipHostFree(p->h_err);
   delete p;
   r->peer = nullptr;
}
// a wait that gave up: reported once, by the next call that looks
static int peer_check(mdh_renderer *r)
{
   PeerState *p = r->peer;
   if (p && p->h_err && *(volatile unsigned *)p->h_err) {
      *(volatile unsigned *)p->h_err = 0u;
      return seterr(MDH_E_COMM, "a peer's radiance slice did not arrive in time (its frame number never came): the atlases of the frames since are not the whole frame's");
   }
   return MDH_OK;
}
extern "C" int32_t mdh_peer_export(mdh_renderer *r, uint8_t blob_out[MDH_PEER_BLOB_BYTES])
{
   if (!r || !blob_out) return seterr(MDH_E_INVALID, "bad argument");
   if (r->in_frame) return seterr(MDH_E_STATE, "a frame is open");
   if (r->comm || peer_active(r)) return seterr(MDH_E_STATE, "the renderer has a communicator");
   HIP_TRY(hipSetDevice(r->device));
   { int dr = drain_streams(r); if (dr != MDH_OK) return dr; }
   if (!r->peer) r->peer = new PeerState();
   PeerState *p = r->peer;
   if (!p->d_flags) HIP_TRY(hipMalloc((void **)&p->d_flags, 4096));
   if (!p->h_err) { HIP_TRY(hipHostMalloc((void **)&p->h_err, 64, hipHostMallocDefault)); *p->h_err = 0u; }
   // a new session starts at frame 0: nobody must take an earlier session's numbers for this one's
   HIP_TRY(hipMemset(p->d_flags, 0, 4096));
   for (int s = 0; s < MDH_ATLAS_SETS; ++s) p->seq[s] = 0u;
   static uint64_t serial = 0;
   p->serial = ++serial;
   p->exported = true;
   PeerBlob b;
   memset(&b, 0, sizeof b);
   b.magic = 0x5045484du; b.version = 2;
   b.pid = (int32_t)getpid(); b.device = r->device; b.nsets = MDH_ATLAS_SETS;
   b.rad_bytes = atlas_bytes(r, MDH_TEX_RADIANCE);
   b.serial = p->serial;
   for (int s = 0; s < MDH_ATLAS_SETS; ++s) HIP_TRY(hipIpcGetMemHandle(&b.rad[s], r->d_rad2[s]));
   HIP_TRY(hipIpcGetMemHandle(&b.flags, p->d_flags));
   memset(blob_out, 0, MDH_PEER_BLOB_BYTES);
   memcpy(blob_out, &b, sizeof b);
   return MDH_OK;
}
extern "C" int32_t mdh_peer_init(mdh_renderer *r, const uint8_t *blobs, int32_t rank, int32_t world)
{
   if (!r || !blobs) return seterr(MDH_E_INVALID, "bad argument");
   if (world < 1 || rank < 0 || rank >= world) return seterr(MDH_E_INVALID, "rank is not below world");
   if (r->in_frame) return seterr(MDH_E_STATE, "a frame is open");
   if (r->comm || peer_active(r)) return seterr(MDH_E_STATE, "the renderer already has a communicator");
   PeerState *p = r->peer;
   if (!p || !p->exported) return seterr(MDH_E_STATE, "mdh_peer_export comes first");
   if (!r->opt_irr_all) return seterr(MDH_E_STATE, "the peer exchange moves radiance slices only: MDH_OPT_IRRADIANCE_ALL must be on");
   HIP_TRY(hipSetDevice(r->device));
   { int jr = join_main(r); if (jr != MDH_OK) return jr; }
   HIP_TRY(hipStreamSynchronize(r->stream));
   p->peers.assign((size_t)world, PeerState::Peer());
   for (int q = 0; q < world; ++q) {
      PeerBlob b;
      memcpy(&b, blobs + (size_t)q * MDH_PEER_BLOB_BYTES, sizeof b);
      if (b.magic != 0x5045484du || b.version != 2 || b.nsets != MDH_ATLAS_SETS) { peer_close_peers(p); return seterr(MDH_E_INVALID, "not a peer blob of this library"); }
      if (b.rad_bytes != atlas_bytes(r, MDH_TEX_RADIANCE)) { peer_close_peers(p); return seterr(MDH_E_INVALID, "a peer's radiance atlas has another size: the ranks' probe settings and atlas formats must agree"); }
      if (q == rank) {
         if (b.pid != (int32_t)getpid() || b.serial != p->serial) { peer_close_peers(p); return seterr(MDH_E_INVALID, "the blob at this rank's place is not this renderer's latest export"); }
         continue;
      }
      if (b.pid == (int32_t)getpid()) { peer_close_peers(p); return seterr(MDH_E_INVALID, "two ranks of a peer exchange in one process: a process cannot open its own interprocess handles"); }
      PeerState::Peer &pe = p->peers[(size_t)q];
      for (int s = 0; s < MDH_ATLAS_SETS; ++s)
         if (hipIpcOpenMemHandle(&pe.rad[s], b.rad[s], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); peer_close_peers(p); return seterr(MDH_E_COMM, "hipIpcOpenMemHandle failed on a peer's radiance atlas"); }
      if (hipIpcOpenMemHandle((void **)&pe.flags, b.flags, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); peer_close_peers(p); return seterr(MDH_E_COMM, "hipIpcOpenMemHandle failed on a peer's frame numbers"); }
   }
   *p->h_err = 0u;
   p->rank = rank; p->world = world;
   p->active = true;
   r->opt_rank = rank;
   r->opt_world = world;
   r->rad_order_rays = 0;
   return MDH_OK;
}
static int peer_exchange(mdh_renderer *r, int tex)
{
   PeerState *p = r->peer;
   if (tex != MDH_TEX_RADIANCE) return r->opt_irr_all ? MDH_OK : seterr(MDH_E_STATE, "the peer exchange moves radiance slices only: MDH_OPT_IRRADIANCE_ALL must be on");
   { int pc = peer_check(r); if (pc != MDH_OK) return pc; }
   hipStream_t st = frame_probe_stream(r);
   const int s = r->frame_cur;
   const size_t per = (size_t)r->probes.radiance_resolution * r->probes.radiance_resolution * texel_bytes(r);
   const long long P = probe_total(r), world = p->world;
   hipEvent_t e0 = nullptr, e1 = nullptr;
   if (r->opt_timing) {
      e0 = get_event(r);
      e1 = get_event(r);
      if (!e0 || !e1) return seterr(MDH_E_DEVICE, "hipEventCreate failed");
      HIP_TRY(hipEventRecord(e0, st));
   }
   // my slice of set s is complete behind everything the probe stream holds: its frame number says so
   const unsigned n = ++p->seq[s];
   hipLaunchKernelGGL(k_peer_publish, dim3(1), dim3(1), 0, st, p->d_flags + s, n);
   HIP_TRY(hipGetLastError());
   char *mine = (char *)r->d_rad2[s];
   for (long long k = 1; k < world; ++k) { // (peers in a rotated order: the ranks do not all read the same peer at once)
      const long long q = (p->rank + k) % world;
      const long long b = P * q / world, e = P * (q + 1) / world; // own_probes () of rank q
      if (e <= b) continue;
      PeerState::Peer &pe = p->peers[(size_t)q];
      hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, st, (const unsigned *)(pe.flags + s), n, p->h_err);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(mine + per * (size_t)b, (const char *)pe.rad[s] + per * (size_t)b, per * (size_t)(e - b), hipMemcpyDeviceToDevice, st));
   }
   if (r->opt_timing) {
      HIP_TRY(hipEventRecord(e1, st));
      r->pending.push_back({MDH_PASS_EXCHANGE, e0, e1});
   }
   return MDH_OK;
}

extern "C" int32_t mdh_comm_available(void) { return rccl_ready(); }
extern "C" int32_t mdh_comm_unique_id(uint8_t id_out[MDH_COMM_ID_BYTES])
{
   static_assert(sizeof(ncclUniqueId) == MDH_COMM_ID_BYTES, "MDH_COMM_ID_BYTES is the size of ncclUniqueId");
   if (!id_out) return seterr(MDH_E_INVALID, "bad argument");
   int rc = rccl_ready();
   if (rc != MDH_OK) return rc;
   ncclUniqueId id;
   RCCL_TRY(rccl_api().GetUniqueId(&id));
   memcpy(id_out, &id, sizeof id);
   return MDH_OK;
}
extern "C" int32_t mdh_comm_init(mdh_renderer *r, const uint8_t id_in[MDH_COMM_ID_BYTES], int32_t rank, int32_t world)
{
   if (!r || !id_in) return seterr(MDH_E_INVALID, "bad argument");
   if (world < 1 || rank < 0 || rank >= world) return seterr(MDH_E_INVALID, "rank is not below world");
   if (r->in_frame) return seterr(MDH_E_STATE, "a frame is open");
   if (r->comm || peer_active(r)) return seterr(MDH_E_STATE, "the renderer already has a communicator");
   int rc = rccl_ready();
   if (rc != MDH_OK) return rc;
   HIP_TRY(hipSetDevice(r->device)); // (the communicator binds to the current device)
   if ((rc = join_main(r)) != MDH_OK) return rc;
   ncclUniqueId id;
   memcpy(&id, id_in, sizeof id);
   ncclComm_t comm = nullptr;
   RCCL_TRY(rccl_api().CommInitRank(&comm, world, id, rank));
   if (!r->d_comm_scratch && hipMalloc(&r->d_comm_scratch, 2 * sizeof(double)) != hipSuccess) {
      (void)rccl_api().CommAbort(comm);
      return seterr(MDH_E_DEVICE, "hipMalloc failed");
   }
   r->comm = comm;
   r->opt_rank = rank;
   r->opt_world = world;
   r->rad_order_rays = 0; // (the stored ray order is of another slice)
   return MDH_OK;
}
// the renderer is rank 0 of 1 again (the handle itself is the caller's business)
static void comm_forget(mdh_renderer *r)
{
   r->comm = nullptr;
   r->comm_aborted = false;
   r->opt_rank = 0;
   r->opt_world = 1;
   r->rad_order_rays = 0;
   r->fb_owner[0][0] = r->fb_owner[1][0] = -1; // (the framebuffers hold a rank's tiles: cleared before they are drawn whole)
}
static int comm_drop(mdh_renderer *r, bool abort)
{
   if (!r->comm) return MDH_OK;
   ncclComm_t c = r->comm;
   if (r->comm_aborted) { comm_forget(r); return MDH_OK; } // (a watchdog has aborted -- and thereby freed -- it already)
   if (abort) { comm_forget(r); RCCL_TRY(rccl_api().CommAbort(c)); return MDH_OK; }
   RCCL_TRY(rccl_api().CommDestroy(c)); // (a failed destroy keeps the handle: the caller may abort it)
   comm_forget(r);
   return MDH_OK;
}
// the owning thread's view of a watchdog's abort: the handle is gone
static int comm_gone(mdh_renderer *r)
{
   comm_forget(r);
   if (r->in_frame) { r->in_frame = false; r->main_dirty = true; }
   return seterr(MDH_E_COMM, "the communicator was aborted (mdh_comm_abort)");
}
struct CommBusy { // the owning thread is inside RCCL with r->comm
   mdh_renderer *r;
   explicit CommBusy(mdh_renderer *r_) : r(r_) { ++r->comm_busy; }
   ~CommBusy() { --r->comm_busy; }
};
extern "C" int32_t mdh_comm_destroy(mdh_renderer *r)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (r->in_frame) return seterr(MDH_E_STATE, "a frame is open");
   HIP_TRY(hipSetDevice(r->device));
   { int dr = drain_streams(r); if (dr != MDH_OK) return dr; } // collectives in flight end first
   if (r->peer) { peer_leave(r); return MDH_OK; }
   return comm_drop(r, false);
}
// what a watchdog of the host calls (from any thread) when a collective never returns: no wait for anything
extern "C" int32_t mdh_comm_abort(mdh_renderer *r)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   (void)hipSetDevice(r->device);
   if (r->peer && !r->comm) { peer_abort(r); if (r->in_frame) { r->in_frame = false; r->main_dirty = true; } return MDH_OK; }
   ncclComm_t c = r->comm;
   if (!c) return MDH_OK;
   if (r->comm_busy.load() > 0) { // another thread is inside a collective: abort only, that thread forgets the handle when its call returns
      if (!r->comm_aborted.exchange(true)) RCCL_TRY(rccl_api().CommAbort(c));
      return MDH_OK;
   }
   if (r->in_frame) { r->in_frame = false; r->main_dirty = true; }
   return comm_drop(r, true);
}
// an asynchronous failure of the communicator (a peer died, a transport error) surfaces here
static int comm_check(mdh_renderer *r)
{
   if (r->comm_aborted) return comm_gone(r);
   ncclResult_t ae = ncclSuccess;
   RCCL_TRY(rccl_api().CommGetAsyncError(r->comm, &ae));
   if (ae != ncclSuccess && ae != ncclInProgress) {
      snprintf(g_err, sizeof g_err, "the communicator reports an asynchronous error: %s", rccl_api().GetErrorString(ae));
      return MDH_E_COMM;
   }
   return MDH_OK;
}
// The exchange step of an open frame: every rank's slice of the atlas the open frame is producing goes to every other
// rank, in place (the atlases are probe-major: a slice is one byte range, and a rank's input is its slice where it
// lies in the output -- the in-place form of ncclAllGather, no staging copy), on the stream of the frame's probe
// passes: the screen pass of the previous frame keeps running beside it.  Probe counts the world size does not
// divide (the reference's default 36 probes on 8 ranks) go as one group of broadcasts, a slice each, which RCCL
// fuses into one launch as well.
extern "C" int32_t mdh_frame_exchange(mdh_renderer *r, int32_t tex)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (tex != MDH_TEX_RADIANCE && tex != MDH_TEX_IRRADIANCE) return seterr(MDH_E_INVALID, "not an atlas");
   if (!r->in_frame) return seterr(MDH_E_STATE, "no open frame");
   if (peer_active(r)) return r->opt_mode != 0 ? MDH_OK : peer_exchange(r, tex);
   if (!r->comm || r->opt_mode != 0) return MDH_OK;
   if (r->comm_aborted) return comm_gone(r);
   CommBusy busy(r);
   const RcclApi &n = rccl_api();
   hipStream_t st = frame_probe_stream(r);
   char *buf = (char *)(tex == MDH_TEX_RADIANCE ? r->d_rad2[r->frame_cur] : r->d_irr2[r->frame_cur]);
   const int res = tex == MDH_TEX_RADIANCE ? r->probes.radiance_resolution : r->probes.irradiance_resolution;
   const size_t per = (size_t)res * res * texel_bytes(r);
   const long long P = probe_total(r), world = r->opt_world;
   hipEvent_t e0 = nullptr, e1 = nullptr;
   if (r->opt_timing) {
      e0 = get_event(r);
      e1 = get_event(r);
      if (!e0 || !e1) return seterr(MDH_E_DEVICE, "hipEventCreate failed");
      HIP_TRY(hipEventRecord(e0, st));
   }
   static const bool force_bcast = [] { const char *e = getenv("MADARCH_HIP_EXCHANGE"); return e && strcmp(e, "broadcast") == 0; }();
   // (a failed collective hands its timing events back: ADVICE r03)
   auto failed = [&](const char *what, ncclResult_t e) {
      if (e0) r->free_events.push_back(e0);
      if (e1) r->free_events.push_back(e1);
      snprintf(g_err, sizeof g_err, "%s failed: %s", what, n.GetErrorString(e));
      return r->comm_aborted ? comm_gone(r) : (int)MDH_E_COMM;
   };
   if (P % world == 0 && !force_bcast) {
      const size_t count = per * (size_t)(P / world);
      const ncclResult_t ar = n.AllGather(buf + count * (size_t)r->opt_rank, buf, count, ncclChar, r->comm, st);
      if (ar != ncclSuccess) return failed("ncclAllGather", ar);
   } else {
      ncclResult_t gr = n.GroupStart();
      if (gr != ncclSuccess) return failed("ncclGroupStart", gr);
      for (long long q = 0; q < world; ++q) {
         const long long b = P * q / world, e = P * (q + 1) / world; // own_probes () of rank q
         if (e > b) {
            ncclResult_t br = n.Broadcast(buf + per * (size_t)b, buf + per * (size_t)b, per * (size_t)(e - b), ncclChar, (int)q, r->comm, st);
            if (br != ncclSuccess) { (void)n.GroupEnd(); return failed("ncclBroadcast", br); }
         }
      }
      if ((gr = n.GroupEnd()) != ncclSuccess) return failed("ncclGroupEnd", gr);
   }
   if (r->comm_aborted) { if (e0) r->free_events.push_back(e0); if (e1) r->free_events.push_back(e1); return comm_gone(r); }
   if (r->opt_timing) {
      HIP_TRY(hipEventRecord(e1, st));
      r->pending.push_back({MDH_PASS_EXCHANGE, e0, e1});
   }
   return MDH_OK;
}
extern "C" int32_t mdh_comm_barrier(mdh_renderer *r)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (r->in_frame) return seterr(MDH_E_STATE, "a frame is open");
   int rc = mdh_finish(r);
   if (rc == MDH_OK && peer_active(r)) return seterr(MDH_E_STATE, "the peer exchange has no collectives: the host's own channel is the barrier");
   if (rc != MDH_OK || !r->comm) return rc;
   if (r->comm_aborted) return comm_gone(r);
   CommBusy busy(r);
   HIP_TRY(hipMemsetAsync(r->d_comm_scratch, 0, sizeof(double), r->stream));
   RCCL_TRY(rccl_api().AllReduce(r->d_comm_scratch, r->d_comm_scratch, 1, ncclFloat64, ncclSum, r->comm, r->stream));
   HIP_TRY(hipStreamSynchronize(r->stream));
   return comm_check(r);
}
extern "C" int32_t mdh_comm_max_f64(mdh_renderer *r, double *value)
{
   if (!r || !value) return seterr(MDH_E_INVALID, "bad argument");
   if (r->in_frame) return seterr(MDH_E_STATE, "a frame is open");
   if (peer_active(r)) return seterr(MDH_E_STATE, "the peer exchange has no collectives");
   if (!r->comm) return MDH_OK;
   if (r->comm_aborted) return comm_gone(r);
   CommBusy busy(r);
   HIP_TRY(hipSetDevice(r->device));
   { int jr = join_main(r); if (jr != MDH_OK) return jr; }
   HIP_TRY(hipMemcpyAsync(r->d_comm_scratch + 1, value, sizeof(double), hipMemcpyHostToDevice, r->stream));
   RCCL_TRY(rccl_api().AllReduce(r->d_comm_scratch + 1, r->d_comm_scratch + 1, 1, ncclFloat64, ncclMax, r->comm, r->stream));
   HIP_TRY(hipMemcpyAsync(value, r->d_comm_scratch + 1, sizeof(double), hipMemcpyDeviceToHost, r->stream));
   HIP_TRY(hipStreamSynchronize(r->stream));
   return comm_check(r);
}
extern "C" int32_t mdh_comm_reduce_framebuffer(mdh_renderer *r, int32_t root)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (r->in_frame) return seterr(MDH_E_STATE, "a frame is open");
   if (peer_active(r)) return seterr(MDH_E_STATE, "the peer exchange has no collectives: read every rank's framebuffer and add them");
   if (!r->comm) return MDH_OK;
   if (root < 0 || root >= r->opt_world) return seterr(MDH_E_INVALID, "root is not a rank");
   if (r->comm_aborted) return comm_gone(r);
   CommBusy busy(r);
   HIP_TRY(hipSetDevice(r->device));
   { int jr = join_main(r); if (jr != MDH_OK) return jr; }
   r->main_dirty = true;
   float4 *fb = r->d_fb2[r->fb_last];
   RCCL_TRY(rccl_api().Reduce(fb, fb, (size_t)r->W * r->H * 4, ncclFloat32, ncclSum, root, r->comm, r->stream));
   // the root's buffer now holds every rank's tiles: its next sharded screen pass clears it first
   if (r->opt_rank == root) r->fb_owner[r->fb_last][0] = -1;
   HIP_TRY(hipStreamSynchronize(r->stream));
   return comm_check(r);
}

// the stream the probe passes of the open frame run on (what a caller's collectives must be ordered on)
extern "C" int32_t mdh_probe_stream(mdh_renderer *r, void **stream)
{
   if (!r || !stream) return seterr(MDH_E_INVALID, "bad argument");
   if (r->in_frame) *stream = (void *)frame_probe_stream(r);
   else *stream = (void *)((r->opt_overlap && r->stream == r->own_stream) ? r->probe_stream : r->stream);
   return MDH_OK;
}
extern "C" int32_t mdh_finish(mdh_renderer *r)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   HIP_TRY(hipSetDevice(r->device));
   { int jr = join_main(r); if (jr != MDH_OK) return jr; }
   HIP_TRY(hipStreamSynchronize(r->stream));
   { int pc = peer_check(r); if (pc != MDH_OK) return pc; }
   return resolve_timing(r);
}

extern "C" int32_t mdh_read_framebuffer(mdh_renderer *r, float *rgb_out)
{
   if (!r || !rgb_out) return seterr(MDH_E_INVALID, "bad argument");
   HIP_TRY(hipSetDevice(r->device));
   size_t px = (size_t)r->W * r->H;
   std::vector<float4> tmp(px);
   { int jr = join_main(r); if (jr != MDH_OK) return jr; }
   HIP_TRY(hipMemcpyAsync(tmp.data(), r->d_fb2[r->fb_last], px * sizeof(float4), hipMemcpyDeviceToHost, r->stream));
   HIP_TRY(hipStreamSynchronize(r->stream));
   for (size_t i = 0; i < px; ++i) { rgb_out[3 * i] = tmp[i].x; rgb_out[3 * i + 1] = tmp[i].y; rgb_out[3 * i + 2] = tmp[i].z; }
   return MDH_OK;
}
// Swap_Buffers (renderers.adb:320): the last frame as the window's RGBA8 pixels in pinned host memory, without a
// host wait, so that frames stay in flight.
//  - MDH_OPT_WINDOW = 1: the screen pass has already stored them there itself (64 lanes x 4 bytes over PCIe beside
//    its float4 store); the swap only records an event behind it.
//  - otherwise: k_present converts the framebuffer on the stream that drew it and a copy stream takes the result
//    to the host beside the passes of the next frame.
extern "C" int32_t mdh_swap_buffers(mdh_renderer *r)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (r->in_frame) return seterr(MDH_E_STATE, "a frame is open");
   HIP_TRY(hipSetDevice(r->device));
   const size_t px = (size_t)r->W * r->H;
   // framebuffer 1 is written from the alternate stream only; once the main stream has been ordered after it
   // (join_main) the main stream holds its latest contents as well
   const bool on_alt = r->fb_last == 1 && r->alt_pending;
   hipStream_t st = on_alt ? r->alt_stream : r->stream;
   if (r->opt_window && r->win_valid) {
      const int slot = (int)((r->win_passes - 1) % mdh_renderer::WIN_RING);
      if (!r->ev_win[slot]) HIP_TRY(hipEventCreateWithFlags(&r->ev_win[slot], hipEventDisableTiming));
      HIP_TRY(hipEventRecord(r->ev_win[slot], st));
      r->front_ptr = (const unsigned char *)r->h_win[slot];
      r->front_ev = r->ev_win[slot];
      r->swaps += 1;
      return MDH_OK;
   }
   const int slot = (int)(r->swaps % mdh_renderer::FRONT_RING);
   if (!r->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&r->copy_stream, hipStreamNonBlocking));
   if (!r->ev_packed) HIP_TRY(hipEventCreateWithFlags(&r->ev_packed, hipEventDisableTiming));
   if (!r->d_front[slot]) HIP_TRY(hipMalloc(&r->d_front[slot], px * 4));
   if (!r->h_front[slot]) HIP_TRY(hipHostMalloc((void **)&r->h_front[slot], px * 4, hipHostMallocDefault));
   if (!r->ev_front[slot]) HIP_TRY(hipEventCreateWithFlags(&r->ev_front[slot], hipEventDisableTiming));
   else HIP_TRY(hipStreamWaitEvent(st, r->ev_front[slot], 0)); // the copy that read this slot's device buffer three swaps ago
   hipLaunchKernelGGL(k_present, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, st, r->d_fb2[r->fb_last], r->d_front[slot], (int)px);
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipEventRecord(r->ev_packed, st));
   HIP_TRY(hipStreamWaitEvent(r->copy_stream, r->ev_packed, 0));
   HIP_TRY(hipMemcpyAsync(r->h_front[slot], r->d_front[slot], px * 4, hipMemcpyDeviceToHost, r->copy_stream));
   HIP_TRY(hipEventRecord(r->ev_front[slot], r->copy_stream));
   // framebuffer 1 read from the main stream: its next writer (the alternate stream) has to come after this
   if (r->fb_last == 1 && !on_alt) r->main_dirty = true;
   r->front_ptr = r->h_front[slot];
   r->front_ev = r->ev_front[slot];
   r->swaps += 1;
   return MDH_OK;
}
extern "C" int32_t mdh_front_buffer(mdh_renderer *r, const uint8_t **rgba, int64_t *swap_count)
{
   if (!r || !rgba) return seterr(MDH_E_INVALID, "bad argument");
   if (r->swaps == 0) return seterr(MDH_E_STATE, "no mdh_swap_buffers yet");
   HIP_TRY(hipSetDevice(r->device));
   HIP_TRY(hipEventSynchronize(r->front_ev));
   *rgba = r->front_ptr;
   if (swap_count) *swap_count = r->swaps;
   return MDH_OK;
}
extern "C" int32_t mdh_read_gbuffer(mdh_renderer *r, int32_t *index_out, float *t_out, int32_t *steps_out)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   HIP_TRY(hipSetDevice(r->device));
   size_t n = (size_t)r->W * r->H * 4;
   { int jr = join_main(r); if (jr != MDH_OK) return jr; }
   if (index_out) HIP_TRY(hipMemcpyAsync(index_out, r->d_gb2[r->fb_last][0], n, hipMemcpyDeviceToHost, r->stream));
   if (t_out) HIP_TRY(hipMemcpyAsync(t_out, r->d_gb2[r->fb_last][1], n, hipMemcpyDeviceToHost, r->stream));
   if (steps_out) HIP_TRY(hipMemcpyAsync(steps_out, r->d_gb2[r->fb_last][2], n, hipMemcpyDeviceToHost, r->stream));
   HIP_TRY(hipStreamSynchronize(r->stream));
   return MDH_OK;
}

// The atlas set and the stream reads, writes and device pointers refer to: inside an open frame the
// set that frame is producing, on the stream its probe passes run on; otherwise the last frame's
// set on the main stream.
static int atlas_set(const mdh_renderer *r) { return r->in_frame ? r->frame_cur : r->last; }
static hipStream_t atlas_stream(const mdh_renderer *r) { return r->in_frame ? frame_probe_stream(r) : r->stream; }
// host copies of the probe-major atlases as float RGB per texel
// (texels [first, first + n) of the probe-major atlas; n = 0: all of it)
static int atlas_to_host(mdh_renderer *r, int tex, std::vector<float> &rgb, size_t first = 0, size_t n = 0)
{
   int res = tex == MDH_TEX_RADIANCE ? r->probes.radiance_resolution : r->probes.irradiance_resolution;
   if (n == 0 && first == 0) n = (size_t)probe_total(r) * res * res;
   const char *src = (const char *)(tex == MDH_TEX_RADIANCE ? r->d_rad2[atlas_set(r)] : r->d_irr2[atlas_set(r)]) + first * texel_bytes(r);
   if (n == 0) { rgb.clear(); return MDH_OK; }
   hipStream_t st = atlas_stream(r);
   if (!r->in_frame) { int jr = join_main(r); if (jr != MDH_OK) return jr; }
   rgb.resize(n * 3);
   if (r->opt_atlas == 0) {
      std::vector<uchar4> tmp(n);
      HIP_TRY(hipMemcpyAsync(tmp.data(), src, n * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      for (size_t i = 0; i < n; ++i) { rgb[3 * i] = (float)tmp[i].x / 255.0f; rgb[3 * i + 1] = (float)tmp[i].y / 255.0f; rgb[3 * i + 2] = (float)tmp[i].z / 255.0f; }
   } else {
      std::vector<float4> tmp(n);
      HIP_TRY(hipMemcpyAsync(tmp.data(), src, n * 16, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      for (size_t i = 0; i < n; ++i) { rgb[3 * i] = tmp[i].x; rgb[3 * i + 1] = tmp[i].y; rgb[3 * i + 2] = tmp[i].z; }
   }
   return MDH_OK;
}
static float unorm8_host(float x)
{
   if (x != x) return 0.0f;
   float c = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
   return rintf(c * 255.0f);
}
// upload texels [first, first + n) of a probe-major atlas from float RGB
static int atlas_from_host(mdh_renderer *r, int tex, size_t first, size_t n, const float *rgb)
{
   void *dst = tex == MDH_TEX_RADIANCE ? r->d_rad2[atlas_set(r)] : r->d_irr2[atlas_set(r)];
   hipStream_t st = atlas_stream(r);
   if (!r->in_frame) {
      int jr = join_main(r);
      if (jr != MDH_OK) return jr;
      r->main_dirty = true;
   }
   if (r->opt_atlas == 0) {
      std::vector<uchar4> tmp(n);
      for (size_t i = 0; i < n; ++i) {
         tmp[i].x = (unsigned char)unorm8_host(rgb[3 * i]); tmp[i].y = (unsigned char)unorm8_host(rgb[3 * i + 1]);
         tmp[i].z = (unsigned char)unorm8_host(rgb[3 * i + 2]); tmp[i].w = 255;
      }
      HIP_TRY(hipMemcpyAsync((uchar4 *)dst + first, tmp.data(), n * 4, hipMemcpyHostToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));
   } else {
      std::vector<float4> tmp(n);
      for (size_t i = 0; i < n; ++i) {
         float a = rgb[3 * i], b = rgb[3 * i + 1], c = rgb[3 * i + 2];
         tmp[i] = mk4(a != a ? 0.0f : a, b != b ? 0.0f : b, c != c ? 0.0f : c, 1.0f);
      }
      HIP_TRY(hipMemcpyAsync((float4 *)dst + first, tmp.data(), n * 16, hipMemcpyHostToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));
   }
   return MDH_OK;
}

extern "C" int32_t mdh_read_texture(mdh_renderer *r, int32_t tex, float *out, int32_t *w, int32_t *h, int32_t *c)
{
   if (r && tex > MDH_TEX_RADIANCE_MIP0 && tex <= MDH_TEX_RADIANCE_MIP0 + 15) { // level l of the radiance atlas, built now from the current atlas
      const int l = tex - MDH_TEX_RADIANCE_MIP0, res = r->probes.radiance_resolution >> l, pcx = r->probes.probe_count[0];
      if (!r->opt_mips || res < 1) return seterr(MDH_E_INVALID, "no such level (MDH_OPT_RADIANCE_MIPS)");
      HIP_TRY(hipSetDevice(r->device));
      { int jr = join_main(r); if (jr != MDH_OK) return jr; }
      const int set = atlas_set(r), W = res * pcx, H = res * r->probes.probe_count[1];
      if (out) {
         int rc = build_rad_mips(r, set, r->stream);
         if (rc != MDH_OK) return rc;
         size_t off = 0;
         for (int k = 1; k < l; ++k) off += (size_t)probe_total(r) * (r->probes.radiance_resolution >> k) * (r->probes.radiance_resolution >> k);
         const size_t n = (size_t)probe_total(r) * res * res;
         std::vector<float> rgb(n * 3);
         const char *src = (const char *)r->d_rad_mips[set] + off * texel_bytes(r);
         if (r->opt_atlas == 0) {
            std::vector<uchar4> tmp(n);
            HIP_TRY(hipMemcpyAsync(tmp.data(), src, n * 4, hipMemcpyDeviceToHost, r->stream));
            HIP_TRY(hipStreamSynchronize(r->stream));
            for (size_t i = 0; i < n; ++i) { rgb[3 * i] = (float)tmp[i].x / 255.0f; rgb[3 * i + 1] = (float)tmp[i].y / 255.0f; rgb[3 * i + 2] = (float)tmp[i].z / 255.0f; }
         } else {
            std::vector<float4> tmp(n);
            HIP_TRY(hipMemcpyAsync(tmp.data(), src, n * 16, hipMemcpyDeviceToHost, r->stream));
            HIP_TRY(hipStreamSynchronize(r->stream));
            for (size_t i = 0; i < n; ++i) { rgb[3 * i] = tmp[i].x; rgb[3 * i + 1] = tmp[i].y; rgb[3 * i + 2] = tmp[i].z; }
         }
         for (int Y = 0; Y < H; ++Y)
            for (int X = 0; X < W; ++X) {
               const int tx = X / res, ty = Y / res;
               const size_t idx = ((size_t)(ty * pcx + tx) * res + (Y - ty * res)) * res + (X - tx * res);
               memcpy(out + ((size_t)Y * W + X) * 3, &rgb[idx * 3], 12);
            }
      }
      if (w) *w = W;
      if (h) *h = H;
      if (c) *c = 3;
      return MDH_OK;
   }
   if (!r || tex < 0 || tex > 3) return seterr(MDH_E_INVALID, "bad argument");
   HIP_TRY(hipSetDevice(r->device));
   { int jr = join_main(r); if (jr != MDH_OK) return jr; }
   int W, H, C;
   if (tex == MDH_TEX_RADIANCE || tex == MDH_TEX_IRRADIANCE) {
      int res = tex == MDH_TEX_RADIANCE ? r->probes.radiance_resolution : r->probes.irradiance_resolution;
      W = res * r->probes.probe_count[0]; H = res * r->probes.probe_count[1]; C = 3;
      if (out) {
         std::vector<float> rgb;
         int rc = atlas_to_host(r, tex, rgb);
         if (rc != MDH_OK) return rc;
         int pcx = r->probes.probe_count[0];
         for (int Y = 0; Y < H; ++Y)
            for (int X = 0; X < W; ++X) {
               int tx = X / res, ty = Y / res;
               size_t idx = ((size_t)(ty * pcx + tx) * res + (Y - ty * res)) * res + (X - tx * res);
               memcpy(out + ((size_t)Y * W + X) * 3, &rgb[idx * 3], 12);
            }
      }
   } else if (tex == MDH_TEX_VISIBILITY) {
      W = r->vol.visibility_resolution[0]; H = r->vol.visibility_resolution[1] * r->vol.visibility_resolution[2]; C = 3;
      if (out) { HIP_TRY(hipMemcpyAsync(out, r->d_vis2[atlas_set(r)], (size_t)W * H * 12, hipMemcpyDeviceToHost, r->stream)); HIP_TRY(hipStreamSynchronize(r->stream)); }
   } else {
      W = r->vol.scattering_resolution[0]; H = r->vol.scattering_resolution[1]; C = 4;
      if (out) { HIP_TRY(hipMemcpyAsync(out, r->d_scat2[atlas_set(r)], (size_t)W * H * 16, hipMemcpyDeviceToHost, r->stream)); HIP_TRY(hipStreamSynchronize(r->stream)); }
   }
   if (w) *w = W;
   if (h) *h = H;
   if (c) *c = C;
   return MDH_OK;
}
extern "C" int32_t mdh_write_texture(mdh_renderer *r, int32_t tex, const float *in, int32_t w, int32_t h, int32_t c)
{
   if (!r || tex < 0 || tex > 3 || !in) return seterr(MDH_E_INVALID, "bad argument");
   HIP_TRY(hipSetDevice(r->device));
   { int jr = join_main(r); if (jr != MDH_OK) return jr; }
   r->main_dirty = true;
   int W, H, C;
   int rc = mdh_read_texture(r, tex, nullptr, &W, &H, &C);
   if (rc != MDH_OK) return rc;
   if (w != W || h != H || c != C) return seterr(MDH_E_INVALID, "texture shape mismatch");
   if (tex == MDH_TEX_RADIANCE || tex == MDH_TEX_IRRADIANCE) {
      int res = tex == MDH_TEX_RADIANCE ? r->probes.radiance_resolution : r->probes.irradiance_resolution;
      int pcx = r->probes.probe_count[0];
      std::vector<float> rgb((size_t)W * H * 3);
      for (int Y = 0; Y < H; ++Y)
         for (int X = 0; X < W; ++X) {
            int tx = X / res, ty = Y / res;
            size_t idx = ((size_t)(ty * pcx + tx) * res + (Y - ty * res)) * res + (X - tx * res);
            memcpy(&rgb[idx * 3], in + ((size_t)Y * W + X) * 3, 12);
         }
      return atlas_from_host(r, tex, 0, (size_t)W * H, rgb.data());
   }
   void *dst = tex == MDH_TEX_VISIBILITY ? (void *)r->d_vis2[atlas_set(r)] : (void *)r->d_scat2[atlas_set(r)];
   HIP_TRY(hipMemcpyAsync(dst, in, (size_t)W * H * C * 4, hipMemcpyHostToDevice, r->stream));
   HIP_TRY(hipStreamSynchronize(r->stream));
   return MDH_OK;
}
extern "C" int32_t mdh_read_atlas_slice(mdh_renderer *r, int32_t tex, int32_t probe_begin, int32_t n_probes, float *out)
{
   if (!r || (tex != MDH_TEX_RADIANCE && tex != MDH_TEX_IRRADIANCE) || !out) return seterr(MDH_E_INVALID, "bad argument");
   if (probe_begin < 0 || n_probes < 0 || probe_begin + n_probes > probe_total(r)) return seterr(MDH_E_INDEX, "probe range");
   HIP_TRY(hipSetDevice(r->device));
   if (n_probes == 0) return MDH_OK;
   // the slice only: one device-to-host copy of its bytes (a sharded run's host exchange reads its own slice every frame)
   int res = tex == MDH_TEX_RADIANCE ? r->probes.radiance_resolution : r->probes.irradiance_resolution;
   std::vector<float> rgb;
   int rc = atlas_to_host(r, tex, rgb, (size_t)probe_begin * res * res, (size_t)n_probes * res * res);
   if (rc != MDH_OK) return rc;
   memcpy(out, rgb.data(), rgb.size() * sizeof(float));
   return MDH_OK;
}
extern "C" int32_t mdh_write_atlas_slice(mdh_renderer *r, int32_t tex, int32_t probe_begin, int32_t n_probes, const float *in)
{
   if (!r || (tex != MDH_TEX_RADIANCE && tex != MDH_TEX_IRRADIANCE) || !in) return seterr(MDH_E_INVALID, "bad argument");
   if (probe_begin < 0 || n_probes < 0 || probe_begin + n_probes > probe_total(r)) return seterr(MDH_E_INDEX, "probe range");
   HIP_TRY(hipSetDevice(r->device));
   int res = tex == MDH_TEX_RADIANCE ? r->probes.radiance_resolution : r->probes.irradiance_resolution;
   return atlas_from_host(r, tex, (size_t)probe_begin * res * res, (size_t)n_probes * res * res, in);
}
extern "C" int32_t mdh_atlas_device_ptr(mdh_renderer *r, int32_t tex, void **dptr, int64_t *total_bytes, int64_t *own_offset, int64_t *own_bytes)
{
   if (!r || (tex != MDH_TEX_RADIANCE && tex != MDH_TEX_IRRADIANCE)) return seterr(MDH_E_INVALID, "bad argument");
   if (r->opt_rank >= r->opt_world) return seterr(MDH_E_STATE, "MDH_OPT_RANK is not below MDH_OPT_WORLD");
   int res = tex == MDH_TEX_RADIANCE ? r->probes.radiance_resolution : r->probes.irradiance_resolution;
   int b, e;
   own_probes(r, &b, &e);
   int64_t per = (int64_t)res * res * (int64_t)texel_bytes(r);
   if (!r->in_frame) { // (inside a frame the caller works on the probe stream, which the frame orders)
      int jr = join_main(r);
      if (jr != MDH_OK) return jr;
      r->main_dirty = true; // the caller may write through the pointer
   }
   if (dptr) *dptr = tex == MDH_TEX_RADIANCE ? r->d_rad2[atlas_set(r)] : r->d_irr2[atlas_set(r)];
   if (total_bytes) *total_bytes = (int64_t)atlas_bytes(r, tex);
   if (own_offset) *own_offset = per * b;
   if (own_bytes) *own_bytes = per * (e - b);
   return MDH_OK;
}
extern "C" int32_t mdh_stream(mdh_renderer *r, void **stream)
{
   if (!r || !stream) return seterr(MDH_E_INVALID, "bad argument");
   { int jr = join_main(r); if (jr != MDH_OK) return jr; } // the stream handed out is ordered after all work so far
   *stream = (void *)r->stream;
   return MDH_OK;
}
extern "C" int32_t mdh_set_stream(mdh_renderer *r, void *stream)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   HIP_TRY(hipSetDevice(r->device));
   { int jr = join_main(r); if (jr != MDH_OK) return jr; }
   r->main_dirty = true;
   HIP_TRY(hipStreamSynchronize(r->stream));
   int rc = resolve_timing(r);
   if (rc != MDH_OK) return rc;
   if (r->probe_stream) HIP_TRY(hipStreamSynchronize(r->probe_stream));
   if (r->alt_stream) HIP_TRY(hipStreamSynchronize(r->alt_stream));
   if (r->query_stream) HIP_TRY(hipStreamSynchronize(r->query_stream));
   if (r->vol_stream) HIP_TRY(hipStreamSynchronize(r->vol_stream));
   r->stream = stream ? (hipStream_t)stream : r->own_stream;
   // the new main stream has not waited for anything: uploads and earlier work are complete (synchronised above)
   for (int si = 0; si < mdh_renderer::NSTREAMS; ++si) { r->tab_seen[si] = r->table_version; r->part_seen[si] = r->part_version; }
   return MDH_OK;
}

// Eval_Distance_To (renderers.adb:499-526), batched on the device
extern "C" int32_t mdh_eval_distance_to(mdh_renderer *r, int32_t n, const float *pts, const int32_t *kind_ixs, int32_t n_kinds,
                                        float *normals_out, float *dist_out)
{
   if (!r || !pts || !kind_ixs || !dist_out || n < 0 || n_kinds < 0 || n_kinds > MDH_MAX_KINDS) return seterr(MDH_E_INVALID, "bad argument");
   for (int i = 0; i < n_kinds; ++i)
      if (kind_ixs[i] < 0 || kind_ixs[i] >= r->npk) return seterr(MDH_E_INVALID, "bad kind index");
   if (n == 0) return MDH_OK;
   // on a stream of its own: the query waits for the table it reads, not for the frames in flight
   HIP_TRY(hipSetDevice(r->device));
   if (!r->query_stream) HIP_TRY(hipStreamCreateWithFlags(&r->query_stream, hipStreamNonBlocking));
   hipStream_t qs = r->query_stream;
   int rc = ensure_committed(r, qs);
   if (rc != MDH_OK) return rc;
   if ((size_t)n > r->query_cap) { // 7 floats per query: point, normal, distance
      if (r->d_query) { HIP_TRY(hipStreamSynchronize(qs)); HIP_TRY(hipFree(r->d_query)); r->d_query = nullptr; }
      r->query_cap = (size_t)n < 256 ? 256 : (size_t)n;
      HIP_TRY(hipMalloc(&r->d_query, r->query_cap * 7 * sizeof(float)));
   }
   float *d_pts = r->d_query, *d_n = d_pts + 3 * r->query_cap, *d_d = d_n + 3 * r->query_cap;
   HIP_TRY(hipMemcpyAsync(d_pts, pts, (size_t)n * 12, hipMemcpyHostToDevice, qs));
   EvalArgs a;
   a.n = n; a.n_kinds = n_kinds;
   for (int i = 0; i < MDH_MAX_KINDS; ++i) { a.kinds[i] = i < n_kinds ? kind_ixs[i] : 0; a.host_count[i] = r->host_count[i]; }
   a.pts = d_pts; a.normals = d_n; a.dist = d_d;
   if ((rc = table_acquire(r, qs)) != MDH_OK) return rc;
   if (r->opt_ada_div) hipLaunchKernelGGL(k_eval_distance<true>, dim3((n + 63) / 64), dim3(64), lds_bytes(r), qs, ks_no_bits(r), a);
   else hipLaunchKernelGGL(k_eval_distance<false>, dim3((n + 63) / 64), dim3(64), lds_bytes(r), qs, ks_no_bits(r), a);
   HIP_TRY(hipGetLastError());
   if ((rc = table_release(r, qs)) != MDH_OK) return rc;
   HIP_TRY(hipMemcpyAsync(dist_out, d_d, (size_t)n * 4, hipMemcpyDeviceToHost, qs));
   if (normals_out) HIP_TRY(hipMemcpyAsync(normals_out, d_n, (size_t)n * 12, hipMemcpyDeviceToHost, qs));
   HIP_TRY(hipStreamSynchronize(qs));
   return MDH_OK;
}

extern "C" int32_t mdh_pass_time(mdh_renderer *r, int32_t pass, double *ms, int64_t *launches)
{
   if (!r || pass < 0 || pass >= MDH_PASS_COUNT) return seterr(MDH_E_INVALID, "bad argument");
   HIP_TRY(hipSetDevice(r->device));
   int rc = resolve_timing(r);
   if (rc != MDH_OK) return rc;
   if (ms) *ms = r->pass_ms[pass];
   if (launches) *launches = r->pass_n[pass];
   return MDH_OK;
}
extern "C" int32_t mdh_reset_pass_times(mdh_renderer *r)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   HIP_TRY(hipSetDevice(r->device));
   int rc = resolve_timing(r);
   if (rc != MDH_OK) return rc;
   for (int i = 0; i < MDH_PASS_COUNT; ++i) { r->pass_ms[i] = 0; r->pass_n[i] = 0; }
   return MDH_OK;
}

// Scenes.Get_Primitives_Location / Get_Lights_Location (scenes.adb:1435-1462)
extern "C" int32_t mdh_scene_layout(mdh_renderer *r, int32_t is_light, int32_t kind_ix, int32_t *count_off, int32_t *array_off, int32_t *stride, int32_t *elem_size)
{
   if (!r || kind_ix < 0 || kind_ix >= (is_light ? r->nlk : r->npk)) return seterr(MDH_E_INVALID, "bad kind index");
   const Kind &k = is_light ? r->lk[kind_ix] : r->pk[kind_ix];
   if (count_off) *count_off = k.count_off;
   if (array_off) *array_off = k.array_off;
   if (stride) *stride = k.stride;
   if (elem_size) *elem_size = k.elem_size;
   return MDH_OK;
}
extern "C" int32_t mdh_scene_buffer_size(mdh_renderer *r, int32_t *size, int32_t *total_light_off)
{
   if (!r) return seterr(MDH_E_INVALID, "null renderer");
   if (size) *size = (int32_t)r->scene_ubo.size();
   if (total_light_off) *total_light_off = r->total_light_off;
   return MDH_OK;
}
extern "C" int32_t mdh_read_scene_buffer(mdh_renderer *r, void *out, int32_t nbytes)
{
   if (!r || !out || nbytes < 0 || (size_t)nbytes > r->scene_ubo.size()) return seterr(MDH_E_INVALID, "bad argument");
   memcpy(out, r->scene_ubo.data(), (size_t)nbytes);
   return MDH_OK;
}
extern "C" int32_t mdh_read_partitioning(mdh_renderer *r, int32_t *out, int32_t n_ints)
{
   if (!r || !out) return seterr(MDH_E_INVALID, "bad argument");
   if (!r->part.enable) return seterr(MDH_E_STATE, "partitioning disabled");
   size_t total = (size_t)r->part_cells * (r->npk + r->part.index_count);
   if ((size_t)n_ints != total) return seterr(MDH_E_INVALID, "size mismatch");
   HIP_TRY(hipSetDevice(r->device));
   if (r->part_version) HIP_TRY(hipEventSynchronize(r->ev_part)); // the last build
   HIP_TRY(hipMemcpy(out, r->d_part_ring[r->part_slot], total * 4, hipMemcpyDeviceToHost));
   return MDH_OK;
}
// the warnings of the last Update_Partitioning (scenes.adb:861-867 prints them): waits for that build only
extern "C" int32_t mdh_partition_warnings(mdh_renderer *r)
{
   if (!r) return 0;
   if (r->warn_pending) {
      (void)hipSetDevice(r->device);
      if (hipEventSynchronize(r->ev_warn) != hipSuccess) return -1;
      r->part_warnings = *r->h_warn;
      r->warn_pending = false;
   }
   return r->part_warnings;
}

#ifdef MDH_DIAG
// rays started, SDF evaluations inside march loops, SDF evaluations in all, arg-min evaluations at hit points (lanes) of the
// last run of `pass`: what the oracle reports as orc_work_counters
extern "C" int32_t mdh_diag_work(mdh_renderer *r, int32_t pass, unsigned long long *out4)
{
   if (!r || !out4 || pass < 0 || pass >= MDH_PASS_COUNT) return MDH_E_INVALID;
   memcpy(out4, r->work[pass], sizeof r->work[pass]);
   return MDH_OK;
}
// debug: read and reset the lane-utilisation counters of mdh_march.h
extern "C" int32_t mdh_diag_read(unsigned long long *out16)
{
   if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_diag), sizeof(unsigned long long) * 16) != hipSuccess) return MDH_E_DEVICE;
   unsigned long long z[16] = {0};
   if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag), z, sizeof z) != hipSuccess) return MDH_E_DEVICE;
   return MDH_OK;
}
#endif
#ifdef MDH_PHASES
extern "C" int32_t mdh_diag_phases(unsigned long long *out16)
{
   if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 16) != hipSuccess) return MDH_E_DEVICE;
   unsigned long long z[16] = {0};
   if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof z) != hipSuccess) return MDH_E_DEVICE;
   return MDH_OK;
}
#endif
#ifdef MDH_TIMELINE
// debug: the wave timeline of the last k_radiance launch (3 words per wave)
extern "C" int32_t mdh_diag_waves(unsigned long long *out, int32_t n_waves)
{
   if (n_waves > MDH_DIAG_WAVES) n_waves = MDH_DIAG_WAVES;
   if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_t), sizeof(unsigned long long) * 3 * n_waves) != hipSuccess) return MDH_E_DEVICE;
   return MDH_OK;
}
#endif
