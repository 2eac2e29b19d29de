// mdh_kernels.h -- the gfx950 kernels, one per pass of Renderers.Render
// (reference madarch/madarch-renderers.adb:302-321) plus the partition-table build
// and the batched distance query.  See mdh_device.h for the device functions and
// DESIGN.md for the launch shapes and the HBM layout.
#pragma once

#include "mdh_device.h"

#ifndef MDH_BLOCK
#define MDH_BLOCK 256 // 4 wavefronts; every wavefront owns one 8x8 tile
#endif

#include "mdh_march.h"
#define MDH_SHADE shade_structured
#ifdef MDH_PHASES
#define PH_KERNEL_BEGIN() float *pk = park_base(sc); if ((threadIdx.x & 63) < 32) ph_acc_(pk)[threadIdx.x & 63] = 0ull; PH_T0(pk_t)
#define PH_KERNEL_END() do { PH_ADD(pk_t, 11); if ((threadIdx.x & 63) < 16) atomicAdd(&g_phase[threadIdx.x & 63], ph_acc_(pk)[threadIdx.x & 63]); } while (0)
#else
#define PH_KERNEL_BEGIN() do { } while (0)
#define PH_KERNEL_END() do { } while (0)
#endif
#ifndef MDH_RAD_PROBES_PER_WAVE
#define MDH_RAD_PROBES_PER_WAVE 64 // 1, 4, 16 or 64 (measured on MI355X: see DESIGN.md)
#endif
// kernels of scenes with user-defined kinds carry the 64-register file of the MDH_X interpreter
// (not in a hiprtc build, where the programs are plain code)
// Mode 2 through the space partition (simple_scene: direct light + occlusion) waits for the partition's L2 round trips and
// has no probe code: more wavefronts per SIMD hide more of them (measured at 1080p with frames in flight: 5 -> 4 416,
// 6 -> 4 871, 7 -> 5 110 Mpixels/s; with the probe code of mode 0 the same scene LOSES at 6 and 7: ball_game 1 407 / 1 338 /
// 1 272).
#ifndef MDH_DIRECT_WAVES_PER_SIMD
#define MDH_DIRECT_WAVES_PER_SIMD 7
#endif
// Mode 0 through the space partition (the reference's simple_scene and ball_game as they run): its march steps wait for a
// cell's bits and three rounds of LDS gathers -- six wavefronts per SIMD measured +5 % / +3 % on those frames (four: -12 %).
#ifndef MDH_PART_WAVES_PER_SIMD
#define MDH_PART_WAVES_PER_SIMD 6
#endif
// The variants for the census of the reference's rooms (MDH_PF_ROOM) need 80 registers where the general scan needs 96: six
// wavefronts per SIMD without a further spill (measured at config 3: five 5 681 - 5 726, six 5 818 - 5 863 Mpixels/s in flight).
#ifndef MDH_ROOM_VARIANTS
#define MDH_ROOM_VARIANTS 1 // (0: no scene is given a kernel variant built for its census, MDH_PF_ROOM / MDH_PF_PSMALL -- A/B runs)
#endif
#ifndef MDH_ROOM_WAVES_PER_SIMD
#define MDH_ROOM_WAVES_PER_SIMD 6
#endif
// (mode 2 through the partition's census variant: 60 registers, eight wavefronts per SIMD without a further spill -- simple_scene
//  direct 8 743 -> 8 982 Mpixels/s in flight)
#ifndef MDH_PSMALL_DIRECT_WAVES_PER_SIMD
#define MDH_PSMALL_DIRECT_WAVES_PER_SIMD 8
#endif
#define MDH_OCC_BUILTIN(PART, MODE) (((PART) & MDH_PF_PART) ? ((MODE) == 2 ? (((PART) & MDH_PF_PSMALL) ? MDH_PSMALL_DIRECT_WAVES_PER_SIMD : MDH_DIRECT_WAVES_PER_SIMD) : MDH_PART_WAVES_PER_SIMD) : (((PART) & MDH_PF_ROOM) && (MODE) == 0 ? MDH_ROOM_WAVES_PER_SIMD : MDH_WAVES_PER_SIMD))
#ifdef MDH_JIT
#define MDH_OCC(PART, MODE) MDH_OCC_BUILTIN(PART, MODE)
#else
#define MDH_OCC(PART, MODE) (((PART) & MDH_PF_CUSTOM) ? 2 : MDH_OCC_BUILTIN(PART, MODE))
#endif
// The radiance kernel is built for more wavefronts per SIMD than the screen kernel: beside two screen passes it then
// takes less of the register file (measured with frames in flight: 5 -> 3775, 6 -> 3800, 7 -> 3830, 8 -> 3845 Mpix/s;
// at 8 a wavefront alone on its SIMD, as in an 8-way sharded frame, runs 5 % slower on its spills: 7).
#ifndef MDH_RAD_WAVES_PER_SIMD
#define MDH_RAD_WAVES_PER_SIMD 7
#endif
// A launch that leaves most wavefront slots empty anyway (the reference's default 36 probes; a rank's slice of a sharded
// frame) gains nothing from a small register budget: its kernel variant (SMALL) is built for five wavefronts per SIMD --
// 86 VGPRs, 16 instead of 96 bytes of scratch (the radiance pass of a rank's slice -2 %, light_shafts +0.6 to +1.3 %:
// profiles/r02_x_radiance_small_launch.log).
#ifndef MDH_RAD_SMALL_WAVES_PER_SIMD
#define MDH_RAD_SMALL_WAVES_PER_SIMD 5
#endif
#define MDH_OCC_RAD_BUILTIN(SMALL) ((SMALL) ? MDH_RAD_SMALL_WAVES_PER_SIMD : MDH_RAD_WAVES_PER_SIMD)
#ifdef MDH_JIT
#define MDH_OCC_RAD(PART, SMALL) MDH_OCC_RAD_BUILTIN(SMALL)
#else
#define MDH_OCC_RAD(PART, SMALL) (((PART) & MDH_PF_CUSTOM) ? 2 : MDH_OCC_RAD_BUILTIN(SMALL))
#endif
#ifndef MDH_RAD_REDERIVE
#define MDH_RAD_REDERIVE 1 // k_radiance derives its texel again behind the pixel program instead of keeping it (see there)
#endif
#ifndef MDH_RAD_QVIS
#define MDH_RAD_QVIS 1 // probe-visibility rays through the wave's ray queue (mdh_march.h: queued_visibility)
#endif
#ifndef MDH_SCR_QVIS
#define MDH_SCR_QVIS 0
#endif
#ifndef MDH_SCAT_BATCH
#define MDH_SCAT_BATCH 8 // scattering steps whose froxel taps are in flight together
#endif
#ifndef MDH_RELOAD_ARGS
#define MDH_RELOAD_ARGS 1
#endif
#ifndef MDH_WAVES_PER_SIMD
#define MDH_WAVES_PER_SIMD 5 // register budget of the march kernels (measured, pipelined frames: 5 > 6 > 4 waves/SIMD)
#endif

// ------------------------------------------------------------------------ screen pass
// a colour as the RGBA8 default framebuffer of the reference's window holds it (see k_present below;
// unorm8 is the atlas store's conversion, mdh_device.h)
__device__ __forceinline__ unsigned pack_rgba8(float4 c) { return (unsigned)unorm8(c.x) | ((unsigned)unorm8(c.y) << 8) | ((unsigned)unorm8(c.z) << 16) | ((unsigned)unorm8(c.w) << 24); }
struct ScreenArgs {
   int W, H;
   int tiles_x, n_tiles; // 8x8 tiles of the whole image
   int rank, world;      // this launch draws tiles rank, rank + world, ...
   int ao_steps;
   int spec_mode; // M_COMPUTE_INDIRECT_SPECULAR (MDH_OPT_INDIRECT_SPECULAR)
   float4 *fb; // W*H, row 0 = top
   int *gb_index;
   float *gb_t;
   int *gb_steps;
   unsigned *window; // MDH_OPT_WINDOW: the window's RGBA8 pixels (pinned host memory, written over PCIe), or null
   // MDH_OPT_SCREEN_ORDER: the launch's tiles in the order of an earlier pass's wavefront durations, slowest first.  A
   // pass ends with its slowest wavefront; in image order the long tiles (rays that graze surfaces for hundreds of
   // steps) start anywhere, and the pass of simple_scene spent 45 % of its time with a few of them on an empty chip.
   int n_own;             // tiles this launch draws: own index 0 .. n_own - 1, tile = rank + own * world
   const unsigned *order; // [n_own] the own index at every place of the launch, or null: image order
   unsigned char *cost;   // [n_own] written by the pass when not null: the tile's wavefront duration as a sort key
   // A launch too small to fill the chip (a rank's share of a sharded frame, a small window) is the latency of its slowest
   // wavefront, and a wavefront alone on its SIMD issues one instruction every four cycles whatever its lanes do: `split` = 1, 2
   // gives a tile to two wavefronts of 8x4 pixels or four of 4x4 (32 / 16 lanes each, the others idle from the start) -- every
   // wavefront marches only as long as ITS pixels need and the tile's work runs on several SIMDs.  A pixel's operations do not
   // depend on which lanes march beside it: the same bits (tests/test_gpu_parity.py::test_split_tiles_change_no_pixel).
   int split;
   // ... and in an ORDERED launch (slowest tiles first) the first `split_first` places -- the tiles whose wavefronts make the
   // pass's tail, a sum of marches each as long as the longest among 64 pixels -- are drawn by four wavefronts of 4x4 pixels
   // each (places 0 .. 4 split_first - 1), the others by one
   int split_first;
};
// place of a launch -> its index into the launch's tiles, the part of the tile, the split factor (ScreenArgs::split, split_first)
MDH_DEV void screen_place(const ScreenArgs &a, int place, int &idx, int &sub, int &split)
{
   if (a.split_first > 0) { // (wave-uniform)
      if (place < 4 * a.split_first) { idx = place >> 2; sub = place & 3; split = 2; }
      else { idx = place - 3 * a.split_first; sub = 0; split = 0; }
   } else { idx = place >> a.split; sub = place & ((1 << a.split) - 1); split = a.split; }
}
#ifndef MDH_SCREEN_SPLIT_FIRST_PERMILLE
#define MDH_SCREEN_SPLIT_FIRST_PERMILLE 0 // of an ordered launch's tiles, the slowest so many thousandths as four wavefronts each (ScreenArgs::split_first)
#endif
#ifndef MDH_SCREEN_SPLIT_DEFAULT
#define MDH_SCREEN_SPLIT_DEFAULT 2560 // MDH_OPT_SCREEN_SPLIT's initial value
#endif
// a wavefront's duration in ticks of s_memtime (which need not be the shader clock) as a key of the counting sort (k_rad_hist ...)
#ifndef MDH_TILE_COST_SHIFT
#define MDH_TILE_COST_SHIFT 12
#endif
MDH_DEV unsigned char tile_cost_key(unsigned cycles) { return (unsigned char)min(cycles >> MDH_TILE_COST_SHIFT, 255u); }

// draw_screen.glsl:20-30.  One wavefront per 8x8 pixel tile: lane = (y & 7) * 8 + (x & 7);
// the tile's 64 pixels walk the structured pixel program of mdh_march.h together.
//
// Nothing of the prologue stays live across the pixel program: the pixel's coordinates, its camera ray and the
// arguments only the epilogue needs (volumetrics, framebuffer, camera) are derived AGAIN after it, from the kernel
// argument segment read through a pointer the compiler cannot trace back and from a lane index taken from the hardware
// (tile_pixel).  Kept live they cost ~10 VGPRs and ~20 SGPRs which the compiler spilled to scratch (80 bytes per
// lane written and read back through HBM: 172 MB per launch at 1080p against a 33 MB framebuffer).
MDH_DEV void tile_pixel(const ScreenArgs &a, int tile, int sub, int split, int lane, int &i, int &j, float &u, float &v)
{
   if (split == 0) { // (wave-uniform)
      i = (tile % a.tiles_x) * 8 + (lane & 7);
      j = (tile / a.tiles_x) * 8 + (lane >> 3);
   } else if (split == 1) { // the tile's upper or lower 8x4 pixels
      i = (tile % a.tiles_x) * 8 + (lane & 7);
      j = (tile / a.tiles_x) * 8 + sub * 4 + ((lane >> 3) & 3);
      if (lane >= 32) i = a.W; // (no pixel: not valid)
   } else { // one of its four 4x4 quadrants
      i = (tile % a.tiles_x) * 8 + (sub & 1) * 4 + (lane & 3);
      j = (tile / a.tiles_x) * 8 + (sub >> 1) * 4 + ((lane >> 2) & 3);
      if (lane >= 16) i = a.W;
   }
   u = centre(i, a.W);
   v = -centre(j, a.H); // row 0 = top
}
// ALT: the variant whose second shaded point runs render_probes.glsl's other two indirect-specular bodies (modes 1 and 3)
template <int PART, int MODE, bool GBUF, bool ALT = false>
__global__ __launch_bounds__(MDH_BLOCK, MDH_OCC(PART, MODE)) void k_screen(KScene sc, KProbes pr, KVolumetrics vol, KCamera cam, ScreenArgs a)
{
   stage_table(sc);
   const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
   const int place = blockIdx.x * (MDH_BLOCK / 64) + wave;
   int pidx, psub, psplit;
   screen_place(a, place, pidx, psub, psplit);
   if (pidx >= a.n_own) return; // wave-uniform
   const unsigned t_begin = (unsigned)__builtin_amdgcn_s_memtime();
   const int own = a.order ? (int)a.order[pidx] : pidx;
   const int tile = a.rank + own * a.world;
   PH_KERNEL_BEGIN();
   MDH_DIAG_WAVE(own);
   f3 c;
   PrimaryHit ph;
   bool hit;
   f3 pos;
   {
      int i, j;
      float u, v;
      tile_pixel(a, tile, psub, psplit, lane, i, j, u, v);
      const bool valid = i < a.W && j < a.H;
      f3 origin, dir;
      camera_ray(cam, u, v, origin, dir);
      MachineCfg cfg; // renderers.adb:136-143
      cfg.direct_specular = true;
      cfg.spec_mode = a.spec_mode;
      cfg.ao_steps = a.ao_steps;
      c = MDH_SHADE<PART, MODE, ALT ? 2 : 1, MDH_SCR_QVIS != 0>(sc, pr, cfg, valid, origin, dir, ph, hit, pos);
   }
#if MDH_RELOAD_ARGS
   struct KArgs { KScene sc; KProbes pr; KVolumetrics vol; KCamera cam; ScreenArgs a; };
   // (the kernel argument segment lays the by-value arguments out like this struct: each at its natural alignment, all of
   //  them 8-byte aligned aggregates of 4- and 8-byte members -- checked here rather than assumed)
   static_assert(alignof(KScene) == 8 && alignof(KProbes) == 8 && alignof(KVolumetrics) == 8 && alignof(ScreenArgs) == 8 && alignof(KCamera) == 4, "kernel argument alignments");
   static_assert(__builtin_offsetof(KArgs, pr) == sizeof(KScene) && __builtin_offsetof(KArgs, vol) == sizeof(KScene) + sizeof(KProbes) &&
                 __builtin_offsetof(KArgs, cam) == sizeof(KScene) + sizeof(KProbes) + sizeof(KVolumetrics) && __builtin_offsetof(KArgs, a) % 8 == 0, "KArgs is the kernel argument segment of k_screen");
   typedef const KArgs __attribute__((address_space(4))) *KArgsPtr; // constant address space: scalar loads
   KArgsPtr ka = (KArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
   asm volatile("" : "+s"(ka));
   KVolumetrics vol2;
   ScreenArgs a2;
   KCamera cam2;
   {
      typedef const int __attribute__((address_space(4))) *IntPtr;
      IntPtr pv = (IntPtr)&ka->vol, pa = (IntPtr)&ka->a, pc = (IntPtr)&ka->cam;
      int *dv = (int *)&vol2, *da = (int *)&a2, *dc = (int *)&cam2;
#pragma unroll
      for (int q = 0; q < (int)(sizeof(KVolumetrics) / 4); ++q) dv[q] = pv[q];
#pragma unroll
      for (int q = 0; q < (int)(sizeof(ScreenArgs) / 4); ++q) da[q] = pa[q];
#pragma unroll
      for (int q = 0; q < (int)(sizeof(KCamera) / 4); ++q) dc[q] = pc[q];
   }
#define vol vol2
#define a a2
#define cam cam2
#endif
   // the pixel again (the same integer and fp32 operations on the same inputs: the same values)
   int i, j;
   float u, v;
   {
      int wave2 = wave;
      asm volatile("" : "+s"(wave2));
      const int place2 = (int)blockIdx.x * (MDH_BLOCK / 64) + wave2;
      int pidx2, psub2, psplit2;
      screen_place(a, place2, pidx2, psub2, psplit2);
      const int own2 = a.order ? (int)a.order[pidx2] : pidx2;
      const int tile2 = a.rank + own2 * a.world;
      tile_pixel(a, tile2, psub2, psplit2, lane_index_fresh(), i, j, u, v);
      // (everything but the tone map and the stores is behind the wavefront: what it took decides its place in later passes)
      if (a.cost && lane_index_fresh() == 0) a.cost[own2] = tile_cost_key((unsigned)__builtin_amdgcn_s_memtime() - t_begin);
   }
   const bool valid = i < a.W && j < a.H;
#ifdef MDH_PHASES
   if (!valid) { PH_KERNEL_END(); return; }
#endif
   if (!valid) return;
   if (MODE == 0 && vol.enabled) {
      f3 origin, dir;
      camera_ray(cam, u, v, origin, dir);
      c = render_volumetrics(sc, vol, c, origin, pos, hit, F2(u, v));
   }
   if (MODE != 1) // draw_screen.glsl:29
      c = F3(spow(sdiv(c.x, c.x + 1.0f), 0.4545f), spow(sdiv(c.y, c.y + 1.0f), 0.4545f), spow(sdiv(c.z, c.z + 1.0f), 0.4545f));
   const size_t px = (size_t)j * a.W + i;
   a.fb[px] = make_float4(c.x, c.y, c.z, 1.0f);
   if (a.window) a.window[px] = pack_rgba8(make_float4(c.x, c.y, c.z, 1.0f));
   if (GBUF) {
      a.gb_index[px] = ph.index;
      a.gb_t[px] = ph.t;
      a.gb_steps[px] = ph.steps;
   }
#if MDH_RELOAD_ARGS
#undef vol
#undef a
#undef cam
#endif
   PH_KERNEL_END();
}

// ---------------------------------------------------------------------- radiance pass
// compute_probe_radiance.glsl:16-27.  One lane per radiance texel (one probe ray); a wavefront = the same
// octahedral texel of MDH_RAD_PROBES_PER_WAVE = 64 consecutive probes (64 parallel rays from 64 origins).
#ifndef MDH_RAD_ORDER_DEFAULT
#define MDH_RAD_ORDER_DEFAULT 1 // MDH_OPT_RADIANCE_ORDER's initial value
#endif
// a step count on a scale of 16 levels: 0 .. 7 as they are, then 8-11, 12-15, 16-23, 24-31, 32-47, 48-63, 64-95, 96 and more
// (rays of one level differ by a third at most; the two levels of a ray are one byte, one pass of the counting sort)
MDH_DEV int rad_level(int steps)
{
   if (steps < 8) return steps;
   const int l = 31 - __builtin_clz((unsigned)steps); // 3 for 8-15, 4 for 16-31 ...
   return min(15, 8 + 2 * (l - 3) + ((steps >> (l - 1)) & 1));
}
struct RadOrder {
   const unsigned *order; // [n_rays] ray at every place, or nullptr: rays in probe order
   unsigned char *steps;  // [n_rays] every ray's sort key (rad_level of its primary-march and shadow steps), written by the pass; or nullptr
   int n_rays;
   // Lanes of every wavefront that carry a ray: 64, or 32 / 16 for launches that would leave most wavefront slots of
   // the chip empty (a rank's slice of a sharded frame: 1 024 full wavefronts on 1 024 SIMDs, each alone on its SIMD and
   // paying every latency of its dependency chain in full).  Spread over four times as many wavefronts the same rays
   // hide each other's latencies, and a wavefront's marches end with the longest of 16 rays instead of 64.
   int fill;
   RadRecord *rec; // [launch lanes] MADARCH_HIP_RAD_SPLIT: what the first kernel leaves for the second (null: one kernel)
};
// `first_round`: the workgroups the chip holds at once (0: not told).  The pass is ONE round of wavefronts and a
// remainder -- 8 192 wavefronts on 7 168 slots at the headline size -- and ends with its slowest wavefront.  A SIMD
// issues from its OLDEST wavefront first (measured: within the first round a wavefront's duration follows its launch
// position, 36 us for the first eighth to 88 us for the seventh, whichever probes it holds), so the remainder, which
// starts 35-65 us late in slots between six older wavefronts, would finish last by far: it raises its issue priority
// instead and runs at the speed of a wavefront alone (radiance pass 0.165 -> 0.147 ms).
// `ro`: the rays of the pass in the order of the PREVIOUS pass's primary-march lengths (RadOrder below), or no order.
// The texel (x, y) of probe probe_raw that lane `lin` of the launch computes (probe_raw = pr.probe_end: none)
MDH_DEV void radiance_texel(const KProbes &pr, const RadOrder &ro, long lin, int &x, int &y, int &probe_raw)
{
   const int per_probe = pr.rres * pr.rres;
   if (ro.fill < 64) { // the launch's lanes that carry rays, renumbered: wavefront w holds rays w * fill .. w * fill + fill - 1
      const int l = (int)(lin & 63);
      if (l >= ro.fill) { x = y = 0; probe_raw = pr.probe_end; return; }
      lin = (lin >> 6) * ro.fill + l;
   }
   // a wavefront = a TxT texel tile (T*T = 64 / G) of the octahedral maps of G consecutive probes:
   // G = 1 is one 8x8 tile of one probe, G = 64 the same ray direction from 64 probes
   constexpr int G = MDH_RAD_PROBES_PER_WAVE, T = (G == 1) ? 8 : (G == 4) ? 4 : (G == 16) ? 2 : 1;
   const int lane = (int)(lin & 63);
   const long wave_global = lin >> 6;
   if (ro.order) { // lane = the ray at this place of the order (a ray: probe of the slice * texels + texel)
      const unsigned ray = lin < ro.n_rays ? ro.order[lin] : 0xffffffffu;
      if (pr.rshift >= 0) {
         probe_raw = ray == 0xffffffffu ? pr.probe_end : pr.probe_begin + (int)(ray >> (2 * pr.rshift));
         y = (int)(ray >> pr.rshift) & (pr.rres - 1);
         x = (int)ray & (pr.rres - 1);
      } else {
         probe_raw = ray == 0xffffffffu ? pr.probe_end : pr.probe_begin + (int)(ray / (unsigned)per_probe);
         const int rem = (int)(ray % (unsigned)per_probe);
         y = rem / pr.rres;
         x = rem - y * pr.rres;
      }
   } else
   if ((pr.rres % T) == 0) {
      const int tpr = pr.rres / T, tiles = tpr * tpr;      // texel tiles per probe
      const int group = (int)(wave_global / tiles), tile = (int)(wave_global % tiles);
      probe_raw = pr.probe_begin + group * G + lane / (T * T);
      const int l = lane % (T * T);
      x = (tile % tpr) * T + (l % T);
      y = (tile / tpr) * T + (l / T);
   } else { // resolutions that are no multiple of the tile edge: row-major, one probe per run
      probe_raw = pr.probe_begin + (int)(lin / per_probe);
      const int rem = (int)(lin % per_probe);
      y = rem / pr.rres;
      x = rem - y * pr.rres;
   }
}
// PHASE 1 / 2: the pass as two kernels (an experiment, MADARCH_HIP_RAD_SPLIT): 1 = every ray's primary and
// shadow marches, its hit point and direct light into ro.rec[launch lane] (no probe code: eight wavefronts per SIMD); 2 = the
// probe visibility queue, the irradiance taps and the store, from the records.  The same arithmetic per ray: the same texels.
#ifndef MDH_RAD_A_WAVES_PER_SIMD
#define MDH_RAD_A_WAVES_PER_SIMD 8
#endif
template <int PART, bool SMALL = false, int PHASE = 0> __global__ __launch_bounds__(MDH_BLOCK, PHASE == 1 ? MDH_RAD_A_WAVES_PER_SIMD : MDH_OCC_RAD(PART, SMALL)) void k_radiance(KScene sc, KProbes pr, int first_round, RadOrder ro)
{
   if (first_round > 0 && (int)blockIdx.x >= first_round) __builtin_amdgcn_s_setprio(3);
#if MDH_QVIS_SHARED && MDH_RAD_QVIS
   qvis_shared_init(sc); // (made visible to the workgroup by the barrier of stage_table)
#endif
   stage_table(sc);
   f3 c;
   PrimaryHit ph;
   PH_KERNEL_BEGIN();
   {
      const long lin = (long)blockIdx.x * MDH_BLOCK + threadIdx.x;
      int x, y, probe_raw;
      radiance_texel(pr, ro, lin, x, y, probe_raw);
      const bool valid = probe_raw < pr.probe_end;
      MDH_DIAG_WAVE(lin >> 6);
      const int probe = valid ? probe_raw : pr.probe_begin;
      const int ty = probe / pr.pcx, tx = probe - ty * pr.pcx;
      const int i = tx * pr.rres + x, j = ty * pr.rres + y; // texel of the reference's 2-D atlas image
      const f2 nc = F2((centre(i, pr.pcx * pr.rres) + 1.0f) * 0.5f, (centre(j, pr.pcy * pr.rres) + 1.0f) * 0.5f);
      // coord_to_probe_id / probe position (probe_utils.glsl:19-40)
      const int probe_id = (int)(nc.y * (float)pr.pcy) * pr.pcx + (int)(nc.x * (float)pr.pcx);
      const f3 world = grid_to_world(pr, probe_id_to_grid(pr, probe_id));
      const f3 ray_dir = ray_id_to_ray_dir(F2(fract_(nc.x * (float)pr.pcx), fract_(nc.y * (float)pr.pcy)));
      MachineCfg cfg; // renderers.adb:115-117: no specular; AO and volumetrics macros undefined
      cfg.direct_specular = false;
      cfg.spec_mode = 0;
      cfg.ao_steps = 0;
      bool hit;
      f3 pos;
      c = MDH_SHADE<PART, 0, 0, MDH_RAD_QVIS != 0, PHASE>(sc, pr, cfg, valid, world, ray_dir, ph, hit, pos, PHASE ? ro.rec + lin : nullptr);
#if !MDH_RAD_REDERIVE
      if (valid) atlas_store(pr.rad, pr.fmt, atlas_index(pr.pcx, pr.rres, pr.rshift, i, j), c);
      if (valid && ro.steps) // the ray's sort key: the levels of its primary-march and soft-shadow step counts
         ro.steps[(size_t)(probe_raw - pr.probe_begin) * (pr.rres * pr.rres) + y * pr.rres + x] = (unsigned char)((rad_level(ph.steps & 0xffff) << 4) | rad_level(ph.steps >> 16));
#endif
   }
#if MDH_RAD_REDERIVE
   // The texel again: nothing of the prologue stays live across the pixel program (its coordinates, the order's entry and
   // the arguments the store needs were eleven dwords that the seven-wavefront build kept in scratch memory).  The lane
   // index comes from the hardware, the arguments from the kernel argument segment through a pointer the compiler
   // cannot trace: the same integer operations on the same inputs.
   {
      struct KArgs { KScene sc; KProbes pr; int first_round; RadOrder ro; };
      static_assert(alignof(RadOrder) == 8 && sizeof(RadOrder) % 8 == 0 && __builtin_offsetof(KArgs, pr) == sizeof(KScene) && __builtin_offsetof(KArgs, first_round) == sizeof(KScene) + sizeof(KProbes) &&
                    __builtin_offsetof(KArgs, ro) == sizeof(KScene) + sizeof(KProbes) + 8, "KArgs is the kernel argument segment of k_radiance");
      typedef const KArgs __attribute__((address_space(4))) *KArgsPtr;
      KArgsPtr ka = (KArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ka));
      typedef const int __attribute__((address_space(4))) *IntPtr;
      KProbes pr2;
      RadOrder ro2;
      {
         IntPtr pp = (IntPtr)&ka->pr, po = (IntPtr)&ka->ro;
         int *dp = (int *)&pr2, *dq = (int *)&ro2;
#pragma unroll
         for (int q = 0; q < (int)(sizeof(KProbes) / 4); ++q) dp[q] = pp[q];
#pragma unroll
         for (int q = 0; q < (int)(sizeof(RadOrder) / 4); ++q) dq[q] = po[q];
      }
      int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
      asm volatile("" : "+s"(wave));
      const long lin = (long)blockIdx.x * MDH_BLOCK + wave * 64 + lane_index_fresh();
      int x, y, probe_raw;
      radiance_texel(pr2, ro2, lin, x, y, probe_raw);
      if (probe_raw < pr2.probe_end) {
         const int ty = probe_raw / pr2.pcx, tx = probe_raw - ty * pr2.pcx;
         const int i = tx * pr2.rres + x, j = ty * pr2.rres + y;
         if (PHASE != 1) atlas_store(pr2.rad, pr2.fmt, atlas_index(pr2.pcx, pr2.rres, pr2.rshift, i, j), c);
         if (ro2.steps && PHASE != 2) // the ray's sort key: the levels of its primary-march and soft-shadow step counts
            ro2.steps[(size_t)(probe_raw - pr2.probe_begin) * (pr2.rres * pr2.rres) + y * pr2.rres + x] = (unsigned char)((rad_level(ph.steps & 0xffff) << 4) | rad_level(ph.steps >> 16));
      }
   }
#endif
   PH_KERNEL_END();
}

// MDH_OPT_RADIANCE_MIPS: one level of the radiance atlas's mip chain from the level below -- the 2x2 box filter over the
// atlas image, ((a + b) + (c + d)) * 0.25 per channel in fp32 on the texels as stored, stored in the atlas's format.  Both
// levels are probe-major ([probe][y][x]); the resolution is a power of two, so no box crosses a probe's tile.
// thread = texel of the destination level (n = probes * res * res), res = its tile resolution
__global__ __launch_bounds__(256) void k_radiance_mips(const void *src, void *dst, int fmt, int res, int n)
{
   const int idx = blockIdx.x * 256 + threadIdx.x;
   if (idx >= n) return;
   const int per = res * res, p = idx / per, rem = idx - p * per, y = rem / res, x = rem - y * res;
   const int rs = 2 * res;
   const unsigned s0 = ((unsigned)(p * rs + 2 * y) * (unsigned)rs) + (unsigned)(2 * x), s1 = s0 + (unsigned)rs;
   const f3 a = atlas_texel<false>(src, fmt, s0, -1), b = atlas_texel<false>(src, fmt, s0 + 1, -1);
   const f3 c = atlas_texel<false>(src, fmt, s1, -1), d = atlas_texel<false>(src, fmt, s1 + 1, -1);
   atlas_store(dst, fmt, (unsigned)idx, ((a + b) + (c + d)) * 0.25f);
}

// ---- Rays in the order of their primary-march lengths.  In lock step a wavefront pays the LONGEST primary march
// among its 64 rays: 20 of 64 lanes are alive in an average step of that march, which is 60 % of the pass's SDF
// evaluations.  A probe ray's march hardly changes from frame to frame (the probes do not move): every pass leaves each
// ray's step count behind, three small kernels sort the rays by it -- a counting sort, longest first -- and the next pass
// takes its rays in that order, so that the rays of a wavefront end their marches together.  Which lane computes a
// texel changes, never what it holds.
//   k_rad_hist:    per chunk of rays (at most MDH_RO_MAX_CHUNKS chunks) a histogram of the 256 keys -> hist[chunk][key]
//   k_rad_scan:    exclusive prefix over keys (descending) and chunks -> the first place of every (key, chunk)
//   k_rad_scatter: the chunk's rays to their places
#define MDH_RO_MAX_CHUNKS 128 // (the scan keeps a key's counts of all chunks in registers)
// hist is [chunk][key]: every access below is 256 consecutive words
__global__ __launch_bounds__(256) void k_rad_hist(const unsigned char *steps, int n, int chunk, unsigned *hist)
{
   __shared__ unsigned h[256];
   h[threadIdx.x] = 0u;
   __syncthreads();
   const int base = blockIdx.x * chunk; // (chunk is a multiple of 1 024: four rays per thread and turn)
   for (int i = 4 * threadIdx.x; i < chunk; i += 1024) {
      if (base + i + 3 < n) {
         const unsigned w = *(const unsigned *)(steps + base + i);
         atomicAdd(&h[w & 255u], 1u); atomicAdd(&h[(w >> 8) & 255u], 1u); atomicAdd(&h[(w >> 16) & 255u], 1u); atomicAdd(&h[w >> 24], 1u);
      } else
         for (int q = 0; q < 4; ++q)
            if (base + i + q < n) atomicAdd(&h[steps[base + i + q]], 1u);
   }
   __syncthreads();
   hist[blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_rad_scan(unsigned *hist, int chunks) // one workgroup; thread = key
{
   __shared__ unsigned after[256];
   unsigned v[MDH_RO_MAX_CHUNKS];
#pragma unroll
   for (int c = 0; c < MDH_RO_MAX_CHUNKS; ++c) v[c] = c < chunks ? hist[c * 256 + threadIdx.x] : 0u; // (all loads in flight together)
   unsigned sum = 0u;
#pragma unroll
   for (int c = 0; c < MDH_RO_MAX_CHUNKS; ++c) { const unsigned x = v[c]; v[c] = sum; sum += x; }
   // rays with a larger key come first: the sum of the totals of the keys above this one (a suffix scan in LDS)
   after[threadIdx.x] = sum;
   __syncthreads();
   for (int d = 1; d < 256; d <<= 1) {
      const unsigned add = (int)threadIdx.x + d < 256 ? after[threadIdx.x + d] : 0u;
      __syncthreads();
      after[threadIdx.x] += add;
      __syncthreads();
   }
   const unsigned before = after[threadIdx.x] - sum;
#pragma unroll
   for (int c = 0; c < MDH_RO_MAX_CHUNKS; ++c)
      if (c < chunks) hist[c * 256 + threadIdx.x] = v[c] + before;
}
__global__ __launch_bounds__(256) void k_rad_scatter(const unsigned char *steps, int n, int chunk, const unsigned *hist, unsigned *order)
{
   __shared__ unsigned place[256];
   place[threadIdx.x] = hist[blockIdx.x * 256 + threadIdx.x];
   __syncthreads();
   const int base = blockIdx.x * chunk;
   for (int i = 4 * threadIdx.x; i < chunk; i += 1024) {
      if (base + i + 3 < n) {
         const unsigned w = *(const unsigned *)(steps + base + i);
         order[atomicAdd(&place[w & 255u], 1u)] = (unsigned)(base + i);
         order[atomicAdd(&place[(w >> 8) & 255u], 1u)] = (unsigned)(base + i + 1);
         order[atomicAdd(&place[(w >> 16) & 255u], 1u)] = (unsigned)(base + i + 2);
         order[atomicAdd(&place[w >> 24], 1u)] = (unsigned)(base + i + 3);
      } else
         for (int q = 0; q < 4; ++q)
            if (base + i + q < n) order[atomicAdd(&place[steps[base + i + q]], 1u)] = (unsigned)(base + i + q);
   }
}

// The same scatter, STABLE: elements with equal keys keep their order.  The screen pass's tiles are sorted with it: tiles
// that took about as long stay in image order, where neighbours tap the same probes' texels (the unstable scatter above
// shuffles the tiles of a chunk: 3 % slower screen passes at equal keys).  Rounds of 256 elements; an element's place =
// the key's next place + the equal keys in front of it in its round.
__global__ __launch_bounds__(256) void k_order_scatter_stable(const unsigned char *keys, int n, int chunk, const unsigned *hist, unsigned *order)
{
   __shared__ unsigned place[256], cnt[256];
   __shared__ short kk[256];
   const int tid = threadIdx.x;
   place[tid] = hist[blockIdx.x * 256 + tid];
   const int base = blockIdx.x * chunk;
   for (int r0 = 0; r0 < chunk; r0 += 256) {
      const int i = base + r0 + tid;
      const bool in = r0 + tid < chunk && i < n;
      const int k = in ? (int)keys[i] : -1;
      kk[tid] = (short)k;
      cnt[tid] = 0u;
      __syncthreads();
      unsigned before = 0u;
      if (in) {
         for (int j = 0; j < tid; ++j) before += kk[j] == k ? 1u : 0u;
         atomicAdd(&cnt[k], 1u);
      }
      __syncthreads();
      if (in) order[place[k] + before] = (unsigned)i;
      __syncthreads();
      place[tid] += cnt[tid];
      __syncthreads();
   }
}

// Keys below `permille` thousandths of the MEDIAN key become 0 (behind k_rad_hist over the raw keys; the histogram is taken
// again afterwards): only the tiles that took clearly longer than most are moved to the front of the launch, all others
// stay in image order.  Every workgroup finds the median from the histogram by itself (256 keys x at most 128 chunks).
__global__ __launch_bounds__(256) void k_order_floor(unsigned char *keys, int n, int chunk, const unsigned *hist, int chunks, int permille)
{
   __shared__ unsigned total[256];
   __shared__ int floor_key;
   const int tid = threadIdx.x;
   unsigned sum = 0u;
   for (int c = 0; c < chunks; ++c) sum += hist[c * 256 + tid];
   total[tid] = sum;
   __syncthreads();
   if (tid == 0) {
      unsigned acc = 0u;
      int m = 0;
      for (; m < 255; ++m) { acc += total[m]; if (2u * acc >= (unsigned)n) break; }
      floor_key = (int)(((long)m * permille + 999) / 1000);
   }
   __syncthreads();
   const int f = floor_key, base = blockIdx.x * chunk;
   for (int i = tid; i < chunk; i += 256)
      if (base + i < n && (int)keys[base + i] < f) keys[base + i] = 0;
}

// -------------------------------------------------------------------- irradiance pass
// update_probe_irradiance.glsl:8-43: every irradiance texel sums its probe's rres x rres
// radiance taps (bilinear, at texel CORNERS, so the first row/column bleed in from the
// neighbouring tiles) in the reference's order.
//
// One workgroup per probe.  A tap's radiance and its ray direction depend on the probe and
// the tap only, not on the output texel: the workgroup evaluates each of the rres^2 taps
// once (bilinear fetch + octahedral decode, spread over all lanes) into LDS, then every
// lane owns one output texel and folds the taps in the reference's y, x order with LDS
// broadcast reads -- ~13 VALU per tap and lane instead of ~370.
#ifndef MDH_IRR_BLOCK
#define MDH_IRR_BLOCK 256 // all of them evaluate taps; the first ires^2 (one wavefront at 8x8) fold
#endif
#ifndef MDH_IRR_PACKED
#define MDH_IRR_PACKED 1
#endif
#ifndef MDH_IRR_PREFETCH
#define MDH_IRR_PREFETCH 1
#endif
#ifndef MDH_IRR_FOLD_WAVES
// wavefronts that fold a probe's taps: 1 = one carries all four channels; 2 = (x, y) on one and (z, weight) on another -- the
// same bits with 7 instead of 13 instructions per tap and wavefront, measured SLOWER (0.040 against 0.038 ms for 512 probes,
// with two, four or six staging wavefronts beside them: profiles/r03_x_irradiance_two_folders.log): the fold does not wait
// for its instructions but for its taps' LDS reads, and two wavefronts read every tap twice.  Kept as a switch; default 1.
#define MDH_IRR_FOLD_WAVES 1
#endif
#ifndef MDH_IRR_CHUNK
#define MDH_IRR_CHUNK 256 // taps per LDS buffer when one wavefront folds (0 = all taps staged at once)
#endif
#ifndef MDH_IRR_WPRE
// the taps' weights computed by the wavefronts beside the folding one (k_irradiance, round 4: VERDICT r03 item 3).  Built, bit-exact,
// and NOT faster: 0.0368 - 0.0417 ms against 0.0370 (profiles/r04_x_irradiance_weights_pipeline.log) -- the folding
// wavefront needs 74 cycles per tap for two LDS reads and seven plain fp32 operations just as it needed them for two reads and
// thirteen operations; a lone workgroup takes 30 us whatever its instruction count.  Off.
#define MDH_IRR_WPRE 0
#endif
#ifndef MDH_IRR_CHANNELS
#define MDH_IRR_CHANNELS 1 // one channel of the fold per wavefront, the weights shared through LDS (k_irradiance)
#endif
#ifndef MDH_IRR_PRIO
#define MDH_IRR_PRIO 3 // s_setprio of that form's wavefronts (0 = none)
#endif
#ifndef MDH_IRR_CCHUNK
#define MDH_IRR_CCHUNK 32 // taps per weight buffer of that form (32: 18 KiB of LDS, 64: 34 KiB -- the same speed alone, 32 starts sooner beside the march kernels of frames in flight)
#endif
// LDS of that form: two weight buffers of 64 rows; its scratch in device memory: six planes of the taps (rounded up to whole chunks) per probe
#define MDH_IRR_CHANNELS_LDS ((size_t)2 * 64 * (MDH_IRR_CCHUNK + 4) * sizeof(float))
#define MDH_IRR_CHANNELS_PLANE(n) ((size_t)(((n) + 63) / 64 * 64))
#ifndef MDH_IRR_ABL
#define MDH_IRR_ABL 0 // (timing experiments only: 1 no fold, 2 no tap evaluation, 3 no weights -- wrong results)
#endif
#ifndef MDH_IRR_WCHUNK
#define MDH_IRR_WCHUNK 64 // taps per pipeline stage of that form (at most 64: one wavefront evaluates a chunk's taps)
#endif
// LDS of that form: three radiance and two direction buffers of a chunk's taps, two buffers of its weights for 64 texels
#define MDH_IRR_WPRE_LDS ((size_t)5 * MDH_IRR_WCHUNK * sizeof(float4) + (size_t)2 * MDH_IRR_WCHUNK * 64 * sizeof(float))
typedef float pk2 __attribute__((ext_vector_type(2))); // two fp32 per VALU instruction (v_pk_mul_f32 / v_pk_add_f32: IEEE per component)
// `prev`, `hyst`: MDH_OPT_HYSTERESIS_PERMILLE (not in the reference; 0 = off): the texel is stored as
// mix (fresh, what the previous frame's atlas holds, hyst)
MDH_DEV f3 irradiance_blend(const KProbes &pr, const void *prev, float hyst, unsigned idx, f3 fresh)
{
   if (hyst == 0.0f) return fresh;
   const f3 old = atlas_texel(prev, pr.fmt, idx, -1);
   return F3(mix_(fresh.x, old.x, hyst), mix_(fresh.y, old.y, hyst), mix_(fresh.z, old.z, hyst));
}
// The pieces of k_irradiance's channel form (one channel of a probe's fold per wavefront, below): a wavefront's quarter of a
// chunk's weights (irr_produce) and a chunk folded into the wavefront's channel (irr_fold; RAD: the channel has a radiance
// plane, the weight's own chain adds w).
//   A tap's radiance and direction are the same for all 64 texels.  Read as LDS broadcasts they cost an LDS instruction and
//   four LDS cycles per four taps each, and with two probes on a CU the LDS array, not instruction issue, bounded the pass.
//   Here sixteen taps sit in ONE register, lane l holding tap l mod 16 (one 4-byte read per sixteen taps and plane), and
//   every multiplication takes its tap from that register through the DPP row broadcast (v_mul_f32_dpp row_newbcast:n -- lane
//   n of each row of sixteen to the whole row: the operand's route changes, the IEEE operation does not).
//   The LDS reads run ahead of their use -- a wavefront alone on its SIMD has nothing else to hide their latency behind:
//   everything the chunk reads is asked for before anything is computed (reads return in order: counted waits).
template <int N> MDH_DEV float irr_row_bcast(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + N, 0xf, 0xf, true)); }
template <int K> MDH_DEV float irr_comp(const float4 &v) { return K == 0 ? v.x : K == 1 ? v.y : K == 2 ? v.z : v.w; }
#define MDH_IRR_X16(X_) X_(0) X_(1) X_(2) X_(3) X_(4) X_(5) X_(6) X_(7) X_(8) X_(9) X_(10) X_(11) X_(12) X_(13) X_(14) X_(15)
// PT taps' weights for this lane's texel into the texel's row: the taps' directions in dxv, dyv, dzv (lane l holds tap l mod 16 of
// the register's sixteen), taps OFF .. OFF + PT - 1 of them
template <int OFF, int PT> MDH_DEV void irr_produce(float dxv, float dyv, float dzv, float *wrow, f3 irr_dir)
{
   float wn[16];
#define MDH_IRR_W1(n_) if ((n_) < PT) wn[n_] = max_((irr_dir.x * irr_row_bcast<(OFF + (n_)) & 15>(dxv) + irr_dir.y * irr_row_bcast<(OFF + (n_)) & 15>(dyv)) + irr_dir.z * irr_row_bcast<(OFF + (n_)) & 15>(dzv), 0.0f);
   MDH_IRR_X16(MDH_IRR_W1)
#undef MDH_IRR_W1
#pragma unroll
   for (int g = 0; g < PT / 4; ++g) *(float4 *)(wrow + 4 * g) = make_float4(wn[4 * g], wn[4 * g + 1], wn[4 * g + 2], wn[4 * g + 3]);
}
// 16 * NB taps folded into one channel: their weights w (this lane's texel), their radiance rv (lane l holds tap l mod 16 of each sixteen)
template <bool RAD, int NB> MDH_DEV float irr_fold(const float *rv, const float4 *w, float acc)
{
   // (four products, then their four additions: a DPP instruction that follows the one before it at a distance of two reuses
   //  its destination and waits a slot for it)
#define MDH_IRR_F4(q_)                                                                                                    \
   {                                                                                                                     \
      const float4 wq = w[4 * b + (q_)];                                                                                 \
      if (RAD) {                                                                                                         \
         const float p0 = irr_row_bcast<4 * (q_)>(rvb) * wq.x, p1 = irr_row_bcast<4 * (q_) + 1>(rvb) * wq.y, p2 = irr_row_bcast<4 * (q_) + 2>(rvb) * wq.z, p3 = irr_row_bcast<4 * (q_) + 3>(rvb) * wq.w; \
         acc = acc + p0; acc = acc + p1; acc = acc + p2; acc = acc + p3;                                                 \
      } else { acc = acc + wq.x; acc = acc + wq.y; acc = acc + wq.z; acc = acc + wq.w; }                                 \
   }
#pragma unroll
   for (int b = 0; b < NB; ++b) {
      const float rvb = rv[b];
      MDH_IRR_F4(0) MDH_IRR_F4(1) MDH_IRR_F4(2) MDH_IRR_F4(3)
   }
#undef MDH_IRR_F4
   return acc;
}
__global__ __launch_bounds__(MDH_IRR_BLOCK) void k_irradiance(KProbes pr, const void *prev, float hyst, float *tap_planes)
{
   extern __shared__ float4 s_taps[]; // [2 * rres * rres]: {rad.xyz, 1} {dir.xyz, -}
   const int probe = pr.probe_begin + blockIdx.x;
   if (probe >= pr.probe_end) return;
#if MDH_IRR_ABL == 9
   return; // (timing experiments only: what the pass's launch and its two events cost by themselves)
#endif
   const int ntaps = pr.rres * pr.rres;
   const int ty = probe / pr.pcx, tx = probe - ty * pr.pcx;
   const float pcx = (float)pr.pcx, pcy = (float)pr.pcy;
   const f2 step = F2(1.0f / pcx / (float)pr.rres, 1.0f / pcy / (float)pr.rres);
   // probe_id_to_coord of the probe the output texels belong to (probe_utils.glsl:52-56)
   const f2 rad_coord = F2((float)tx / pcx, (float)ty / pcy);
   // one tap: the bilinear fetch of the radiance atlas and the ray direction of its texel
#define MDH_IRR_STAGE(tap_, slot_)                                                                                      \
   do {                                                                                                                 \
      const int yy = (tap_) / pr.rres, xx = (tap_) - yy * pr.rres;                                                       \
      f2 c = F2(clamp_(rad_coord.x + (float)xx * step.x, step.x, 1.0f - step.x), clamp_(rad_coord.y + (float)yy * step.y, step.y, 1.0f - step.y)); \
      f3 rad = atlas_sample(pr.rad, pr.fmt, pr.pcx, pr.pcy, pr.rres, pr.rshift, pr.rad_w, pr.rad_h, c.x, c.y, -1);      \
      f3 rad_dir = ray_id_to_ray_dir(F2(fract_(c.x * pcx), fract_(c.y * pcy)));                                          \
      s_taps[2 * (slot_)] = make_float4(rad.x, rad.y, rad.z, 1.0f);                                                      \
      s_taps[2 * (slot_) + 1] = make_float4(rad_dir.x, rad_dir.y, rad_dir.z, 0.0f);                                      \
   } while (0)
#if MDH_FAST_NUMERICS
   if (pr.ires * pr.ires <= 64 && MDH_IRR_BLOCK == 256) {
      // The experiment's fold: every wavefront folds a QUARTER of the taps for all texels, the four partial sums are added
      // in a fixed order, (0 + 1) + (2 + 3) -- not the reference's order of additions (a tolerance-only result), but a
      // quarter of the pass's critical path.
      for (int t = threadIdx.x; t < ntaps; t += MDH_IRR_BLOCK) MDH_IRR_STAGE(t, t);
      __syncthreads();
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      const int x = lane % pr.ires, y = lane / pr.ires;
      const int i = tx * pr.ires + x, j = ty * pr.ires + y;
      const f2 nc = F2((centre(i, pr.pcx * pr.ires) + 1.0f) * 0.5f, (centre(j, pr.pcy * pr.ires) + 1.0f) * 0.5f);
      const f3 irr_dir = ray_id_to_ray_dir(F2(fract_(nc.x * pcx), fract_(nc.y * pcy)));
      float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      const int t0 = wv * (ntaps / 4), t1 = wv == 3 ? ntaps : t0 + ntaps / 4;
      for (int t = t0; t < t1; ++t) {
         const float4 r = s_taps[2 * t], d = s_taps[2 * t + 1];
         const float w = max_(dot(irr_dir, xyz(d)), 0.0f);
         acc.x += r.x * w; acc.y += r.y * w; acc.z += r.z * w; acc.w += w;
      }
      float4 *part = s_taps + 2 * ntaps;
      part[wv * 64 + lane] = acc;
      __syncthreads();
      if (wv == 0 && lane < pr.ires * pr.ires) {
         const float4 a = part[lane], b = part[64 + lane], c = part[128 + lane], d = part[192 + lane];
         const f3 sum = F3((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z));
         const f3 irradiance = sum / ((a.w + b.w) + (c.w + d.w));
         const unsigned idx = atlas_index(pr.pcx, pr.ires, pr.ishift, i, j);
         atlas_store(pr.irr, pr.fmt, idx, irradiance_blend(pr, prev, hyst, idx, irradiance));
      }
      return;
   }
#endif
#if MDH_IRR_CHANNELS
   if (pr.ires * pr.ires <= 64 && MDH_IRR_BLOCK == 256 && tap_planes) { // (the host's test: mdh_api.hip, MDH_PASS_IRRADIANCE)
      // Round 4 (second session): ONE CHANNEL PER WAVEFRONT.  A wavefront that is alone on its SIMD issues one instruction every
      // four cycles whatever the instruction (vector, LDS or scalar), so the lone folding wavefront of the forms below pays
      // ~17 issue slots per tap -- 74 cycles, 30 us for 1 024 taps -- while three wavefronts beside it idle.  The four sums of a
      // texel (x, y, z, weight) are four independent chains of additions: wavefront k carries channel k of all 64 texels, two
      // instructions per tap (one for the weight's chain), with a quarter of the taps' weights w = max (dot (irr_dir, rad_dir), 0)
      // computed by each wavefront one chunk ahead and shared through LDS ([texel][tap]: 16-byte reads and writes, rows 4 * odd
      // dwords apart).  The same multiplications and additions in the reference's order on every chain; the issue slots of a
      // probe's fold spread over four SIMDs instead of one.
#if MDH_IRR_PRIO
      // all four wavefronts are the pass's critical path and meet at a barrier every chunk; beside the march kernels of the
      // neighbouring frames (five or seven wavefronts per SIMD, all of them older) the youngest wavefront of a SIMD gets the
      // issue slots the others leave: raised priority for the pass that heads the next frame's dependency chain
      __builtin_amdgcn_s_setprio(MDH_IRR_PRIO);
#endif
      constexpr int CH = MDH_IRR_CCHUNK, S = CH + 4, CHQ = CH / 4;
      static_assert((CH == 64 || CH == 32) && ((S / 4) & 1) == 1, "a wavefront's quarter of a chunk is a register of sixteen taps or half of one; weight rows 16-byte aligned, an odd number of quads apart");
      const int ntp = (ntaps + 63) / 64 * 64, nchunks = (ntaps + CH - 1) / CH;
      // The taps themselves (radiance and direction: the same for all 64 texels) go through a scratch buffer in device memory,
      // six planes of ntp floats per probe of the pass -- written by this workgroup, read back by it behind a barrier, sixteen
      // taps to a register, a turn ahead of their use: the LDS holds the weights only (18 KiB for chunks of 32 taps; with the taps
      // beside chunks of 64 the pass needed 59 KiB and, with frames in flight, waited for a CU that had them free: -8 % frame
      // rate, measured).
      float *g_dpl = tap_planes + (size_t)blockIdx.x * 6 * ntp; // [3][ntp] the taps' directions, a plane per component
      float *g_rch = g_dpl + 3 * ntp;                            // [3][ntp] their radiance, a plane per channel
      float *s_w = (float *)s_taps;                              // [2][64][S] weights of two chunks, a row per texel
      const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l16 = lane & 15;
      const int ntex = pr.ires * pr.ires;
      const int x = lane % pr.ires, y = lane / pr.ires;
      const int i = tx * pr.ires + x, j = ty * pr.ires + y;
      const f2 nc = F2((centre(i, pr.pcx * pr.ires) + 1.0f) * 0.5f, (centre(j, pr.pcy * pr.ires) + 1.0f) * 0.5f);
      const f3 irr_dir = ray_id_to_ray_dir(F2(fract_(nc.x * pcx), fract_(nc.y * pcy)));
#ifdef MDH_PHASES
      const unsigned long long ph_k0 = __builtin_amdgcn_s_memtime();
#endif
      // a thread's taps four at a time: the texels of all four asked for before the first direction is decoded (one round trip
      // to the atlas instead of four)
      for (int t0 = threadIdx.x; t0 < ntaps; t0 += 4 * MDH_IRR_BLOCK) {
         AtlasTap tap[4];
         f2 cc[4];
#pragma unroll
         for (int k = 0; k < 4; ++k) {
            const int t = min(t0 + k * MDH_IRR_BLOCK, ntaps - 1); // (a tap beyond the last: the last one again, not stored)
            const int yy = t / pr.rres, xx = t - yy * pr.rres;
            cc[k] = F2(clamp_(rad_coord.x + (float)xx * step.x, step.x, 1.0f - step.x), clamp_(rad_coord.y + (float)yy * step.y, step.y, 1.0f - step.y));
            tap[k] = atlas_tap_issue(pr.rad, pr.fmt, pr.pcx, pr.pcy, pr.rres, pr.rshift, pr.rad_w, pr.rad_h, cc[k].x, cc[k].y);
         }
#pragma unroll
         for (int k = 0; k < 4; ++k) {
            const int t = t0 + k * MDH_IRR_BLOCK;
            const f3 rad_dir = ray_id_to_ray_dir(F2(fract_(cc[k].x * pcx), fract_(cc[k].y * pcy)));
            const f3 rad = atlas_tap_resolve(pr.rad, pr.fmt, tap[k], -1);
            if (t < ntaps) {
               g_dpl[t] = rad_dir.x; g_dpl[ntp + t] = rad_dir.y; g_dpl[2 * ntp + t] = rad_dir.z;
               g_rch[t] = rad.x; g_rch[ntp + t] = rad.y; g_rch[2 * ntp + t] = rad.z;
            }
         }
      }
      __syncthreads(); // (the workgroup's stores are in L2 behind it; none of these lines was read before: nothing stale in this CU's L1)
#ifdef MDH_PHASES
      const unsigned long long ph_k1 = __builtin_amdgcn_s_memtime();
      unsigned long long ph_work = 0ull, ph_wait = 0ull;
      if (lane == 0) atomicAdd(&g_phase[8 + wv], ph_k1 - ph_k0); // (staging, its barrier included)
#endif
      float acc = 0.0f;
      const float *g_rc = g_rch + (wv < 3 ? wv : 0) * ntp; // this wavefront's channel
      if (ntaps % 64 == 0) {
         // Turns of 64 taps in 64 / CH chunks.  Chunk c: the weights' rows of the chunk asked for, this wavefront's quarter of
         // chunk c + 1's weights computed and stored, the chunk folded, barrier.  The registers of taps of the NEXT turn (radiance
         // of its 64 taps; directions of the sixteen taps whose weights this wavefront computes during it) are asked for at the
         // top of this one.  A direction register of turn s holds, for chunk 2 s + 1 + h (CH = 32; chunk s + 1 for CH = 64),
         // the wavefront's taps in lanes 8 h .. 8 h + 7.
         constexpr int HALVES = 64 / CH, PT = CH / 4, NB = CH / 16, NW = CH / 4;
         const int h16 = HALVES == 2 ? l16 >> 3 : 0, n16 = HALVES == 2 ? l16 & 7 : l16;
         auto dir_tap = [&](int turn) { return min(((turn * HALVES + 1 + h16) * CH + wv * PT + n16), ntp - 1); }; // (beyond the last chunk: never used)
         float dx, dy, dz, rv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
         { const int t = wv * PT + n16; dx = g_dpl[t]; dy = g_dpl[ntp + t]; dz = g_dpl[2 * ntp + t]; }
         irr_produce<0, PT>(dx, dy, dz, s_w + lane * S + wv * PT, irr_dir);
         { const int t = dir_tap(0); dx = g_dpl[t]; dy = g_dpl[ntp + t]; dz = g_dpl[2 * ntp + t]; }
         if (wv < 3) {
#pragma unroll
            for (int b = 0; b < 4; ++b) rv[b] = g_rc[16 * b + l16];
         }
         // (vmcnt (0): with loads pending on entry the compiler's wait-count pass makes every turn of the loop wait for the loads
         //  that turn has just issued)
         __builtin_amdgcn_s_waitcnt(0x0f70);
         asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
         const int nturns = ntaps / 64;
#pragma unroll 1
         for (int turn = 0; turn < nturns; ++turn) {
#ifdef MDH_PHASES
            const unsigned long long ph_t0 = __builtin_amdgcn_s_memtime();
            unsigned long long ph_b = 0ull;
#endif
            float dxn = 0.0f, dyn = 0.0f, dzn = 0.0f, rvn[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (turn + 1 < nturns) {
               const int t = dir_tap(turn + 1);
               dxn = g_dpl[t]; dyn = g_dpl[ntp + t]; dzn = g_dpl[2 * ntp + t];
               if (wv < 3) {
#pragma unroll
                  for (int b = 0; b < 4; ++b) rvn[b] = g_rc[(turn + 1) * 64 + 16 * b + l16];
               }
            }
#pragma unroll
            for (int hf = 0; hf < HALVES; ++hf) {
               const int c = turn * HALVES + hf;
               const float *wr = s_w + (c & 1) * 64 * S + lane * S;
               float4 w[NW];
#pragma unroll
               for (int g = 0; g < NW; ++g) w[g] = *(const float4 *)(wr + 4 * g);
               if (c + 1 < nchunks) {
                  float *wn = s_w + ((c + 1) & 1) * 64 * S + lane * S + wv * PT;
                  if (hf == 0) irr_produce<0, PT>(dx, dy, dz, wn, irr_dir);
                  else irr_produce<8, PT>(dx, dy, dz, wn, irr_dir);
               }
               acc = wv < 3 ? irr_fold<true, NB>(rv + hf * NB, w, acc) : irr_fold<false, NB>(rv + hf * NB, w, acc);
#ifdef MDH_PHASES
               const unsigned long long ph_t1 = __builtin_amdgcn_s_memtime();
#endif
               asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef MDH_PHASES
               ph_b += __builtin_amdgcn_s_memtime() - ph_t1;
#endif
            }
#ifdef MDH_PHASES
            ph_wait += ph_b; ph_work += __builtin_amdgcn_s_memtime() - ph_t0 - ph_b;
#endif
            dx = dxn; dy = dyn; dz = dzn;
#pragma unroll
            for (int b = 0; b < 4; ++b) rv[b] = rvn[b];
         }
      } else {
         // radiance tiles whose texels are no multiple of 64 (not the reference's: 32 x 32): the same four chains tap by tap
         for (int c = 0; c <= nchunks; ++c) { // (turn c: the weights of chunk c, the fold of chunk c - 1)
            if (c < nchunks) {
               const int t0 = c * CH + wv * CHQ, t1 = min(t0 + CHQ, ntaps);
               float *wb = s_w + (c & 1) * 64 * S + lane * S - c * CH;
               for (int t = t0; t < t1; ++t) wb[t] = max_((irr_dir.x * g_dpl[t] + irr_dir.y * g_dpl[ntp + t]) + irr_dir.z * g_dpl[2 * ntp + t], 0.0f);
            }
            if (c > 0) {
               const int nh = min(CH, ntaps - (c - 1) * CH);
               const float *wr = s_w + ((c - 1) & 1) * 64 * S + lane * S, *rc = g_rc + (c - 1) * CH;
               for (int t = 0; t < nh; ++t) acc = wv < 3 ? acc + rc[t] * wr[t] : acc + wr[t];
            }
            __syncthreads();
         }
      }
#ifdef MDH_PHASES
      if (lane == 0) { atomicAdd(&g_phase[2 * wv], ph_work); atomicAdd(&g_phase[2 * wv + 1], ph_wait); }
#endif
      // the four sums of a texel back to one lane (through the weight rows: the loop's last barrier is behind every read of them)
      if (wv > 0) s_w[(wv - 1) * 64 + lane] = acc;
      __syncthreads();
      if (wv == 0 && lane < ntex) {
         const f3 irradiance = F3(acc, s_w[lane], s_w[64 + lane]) / s_w[128 + lane];
         const unsigned idx = atlas_index(pr.pcx, pr.ires, pr.ishift, i, j);
         atlas_store(pr.irr, pr.fmt, idx, irradiance_blend(pr, prev, hyst, idx, irradiance));
      }
      return;
   }
#endif
#if MDH_IRR_WPRE
   if (pr.ires * pr.ires <= 64 && MDH_IRR_BLOCK == 256) {
      // Round 4: the weights off the fold's critical path.  A tap's weight for a texel, w = max (dot (irr_dir, rad_dir), 0), is
      // five instructions of the fold's thirteen and depends on nothing the fold produces: the wavefronts that used to wait for
      // the folding one compute it, a lane per texel, into LDS.  Chunks of 64 taps run through a three-stage pipeline, one
      // barrier per chunk:      wavefront 1     evaluates the taps of chunk c + 2 (bilinear fetch, octahedral decode: one lane per tap)
      //                         wavefronts 2, 3 the weights of chunk c + 1 for all 64 texels (tap by tap, LDS broadcast reads)
      //                         wavefront 0     folds chunk c: per tap one per-lane weight, one broadcast radiance, two v_pk_mul
      //                                         and two v_pk_add -- the reference's additions in the reference's order
      // -- about 380 issue slots per chunk on every wavefront instead of 1 000 on the folding one.
      constexpr int CH = MDH_IRR_WCHUNK;
      float4 *s_rad = s_taps;                      // [3][CH] {rad.xyz, 1}
      float4 *s_dir = s_taps + 3 * CH;             // [2][CH] {dir.xyz, -}
      float *s_w = (float *)(s_taps + 5 * CH);     // [2][CH][64]
      const int nchunks = (ntaps + CH - 1) / CH;
      const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
      const int ntex = pr.ires * pr.ires;
      const int x = lane % pr.ires, y = lane / pr.ires; // (a lane's texel, in the weight and fold wavefronts)
      const int i = tx * pr.ires + x, j = ty * pr.ires + y;
      const f2 nc = F2((centre(i, pr.pcx * pr.ires) + 1.0f) * 0.5f, (centre(j, pr.pcy * pr.ires) + 1.0f) * 0.5f);
      const f3 irr_dir = ray_id_to_ray_dir(F2(fract_(nc.x * pcx), fract_(nc.y * pcy)));
      float acc_x = 0.0f, acc_y = 0.0f, acc_z = 0.0f, acc_w = 0.0f;
#define MDH_IRR_TAP_COORD(tap_, cc_)                                                                                     \
      const int yy_ = (tap_) / pr.rres, xx_ = (tap_) - yy_ * pr.rres;                                                    \
      const f2 cc_ = F2(clamp_(rad_coord.x + (float)xx_ * step.x, step.x, 1.0f - step.x), clamp_(rad_coord.y + (float)yy_ * step.y, step.y, 1.0f - step.y))
#define MDH_IRR_TAP_STORE(rad_, cc_, c_, slot_)                                                                          \
      do {                                                                                                               \
         const f3 rad_dir_ = ray_id_to_ray_dir(F2(fract_((cc_).x * pcx), fract_((cc_).y * pcy)));                         \
         s_rad[((c_) % 3) * CH + (slot_)] = make_float4((rad_).x, (rad_).y, (rad_).z, 1.0f);                              \
         s_dir[((c_) & 1) * CH + (slot_)] = make_float4(rad_dir_.x, rad_dir_.y, rad_dir_.z, 0.0f);                        \
      } while (0)
      // the weights of chunk c_: taps t0, t0 + stride, ... of it, eight directions on their way from LDS at a time
#define MDH_IRR_WEIGHTS(c_, t0_, stride_)                                                                                 \
      do {                                                                                                               \
         const int nh_ = min(CH, ntaps - (c_) * CH);                                                                      \
         const float4 *db_ = s_dir + ((c_) & 1) * CH;                                                                     \
         float *wb_ = s_w + ((c_) & 1) * CH * 64 + lane;                                                                  \
         if (lane < ntex) {                                                                                              \
            int t = (t0_);                                                                                               \
            _Pragma("unroll 1") for (; t + 7 * (stride_) < nh_; t += 8 * (stride_)) {                                      \
               float4 d8[8];                                                                                             \
               _Pragma("unroll") for (int q = 0; q < 8; ++q) d8[q] = db_[t + q * (stride_)];                              \
               _Pragma("unroll") for (int q = 0; q < 8; ++q)                                                              \
                  wb_[(t + q * (stride_)) * 64] = max_((irr_dir.x * d8[q].x + irr_dir.y * d8[q].y) + irr_dir.z * d8[q].z, 0.0f); \
            }                                                                                                            \
            for (; t < nh_; t += (stride_)) {                                                                             \
               const float4 d = db_[t];                                                                                  \
               wb_[t * 64] = max_((irr_dir.x * d.x + irr_dir.y * d.y) + irr_dir.z * d.z, 0.0f);                           \
            }                                                                                                            \
         }                                                                                                               \
      } while (0)
      // (LDS only: a wavefront's loads from the atlas stay in flight across it -- __syncthreads would wait for them)
#define MDH_IRR_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
      // prologue: the taps of chunks 0 and 1 (all wavefronts); wavefront 1 asks for the texels of chunk 2; the weights of chunk 0
      for (int t = threadIdx.x; t < min(2 * CH, ntaps); t += MDH_IRR_BLOCK) {
         MDH_IRR_TAP_COORD(t, cc);
         const f3 rad = atlas_sample(pr.rad, pr.fmt, pr.pcx, pr.pcy, pr.rres, pr.rshift, pr.rad_w, pr.rad_h, cc.x, cc.y, -1);
         MDH_IRR_TAP_STORE(rad, cc, t / CH, t % CH);
      }
      __syncthreads();
      AtlasTap tap_next;
      tap_next.t00 = tap_next.t10 = tap_next.t01 = tap_next.t11 = 0u; tap_next.fx = tap_next.fy = 0.0f;
      if (wv == 1 && 2 * CH + lane < ntaps && lane < CH) {
         MDH_IRR_TAP_COORD(2 * CH + lane, cc);
         tap_next = atlas_tap_issue(pr.rad, pr.fmt, pr.pcx, pr.pcy, pr.rres, pr.rshift, pr.rad_w, pr.rad_h, cc.x, cc.y);
      }
      MDH_IRR_WEIGHTS(0, wv, 4);
      MDH_IRR_BARRIER();
#ifdef MDH_PHASES
      unsigned long long ph_work = 0ull, ph_wait = 0ull; // (diagnostic: cycles of this wavefront in its stage and at the barrier)
#endif
      for (int c = 0; c < nchunks; ++c) {
#ifdef MDH_PHASES
         const unsigned long long ph_t0 = __builtin_amdgcn_s_memtime();
#endif
         if (wv == 0) {
#if MDH_IRR_ABL != 1
            if (lane < ntex) {
               const int nh = min(CH, ntaps - c * CH);
               const float4 *rb = s_rad + (c % 3) * CH;
               const float *wb = s_w + (c & 1) * CH * 64 + lane;
               int t = 0;
               // groups of four taps, two groups in flight: while one is added the next one is on its way from LDS (registers
               // A and B alternate: no copies)
#define MDH_IRR_LOAD4(R_, W_, t_) _Pragma("unroll") for (int q = 0; q < 4; ++q) { R_[q] = rb[(t_) + q]; W_[q] = wb[((t_) + q) * 64]; }
#define MDH_IRR_ADD4(R_, W_) _Pragma("unroll") for (int q = 0; q < 4; ++q) { acc_x = acc_x + R_[q].x * W_[q]; acc_y = acc_y + R_[q].y * W_[q]; acc_z = acc_z + R_[q].z * W_[q]; acc_w = acc_w + R_[q].w * W_[q]; }
               // (plain fp32 operations, and the tap's fourth component is 1: 1 * w = w)
               if (nh >= 8) {
                  float4 ra[4], rb4[4];
                  float wa[4], wb4[4];
                  MDH_IRR_LOAD4(ra, wa, 0)
#pragma unroll 1
                  for (; t + 8 <= nh; t += 8) {
                     MDH_IRR_LOAD4(rb4, wb4, t + 4)
                     MDH_IRR_ADD4(ra, wa)
                     if (t + 12 <= nh) { MDH_IRR_LOAD4(ra, wa, t + 8) }
                     MDH_IRR_ADD4(rb4, wb4)
                  }
                  if (t + 4 <= nh) { MDH_IRR_ADD4(ra, wa) t += 4; } // (a last group of four already loaded)
               }
#undef MDH_IRR_LOAD4
#undef MDH_IRR_ADD4
               for (; t < nh; ++t) {
                  const float4 r = rb[t];
                  const float w = wb[t * 64];
                  acc_x = acc_x + r.x * w; acc_y = acc_y + r.y * w; acc_z = acc_z + r.z * w;
                  acc_w = acc_w + w;
               }
            }
#endif
         } else if (wv == 1) { // the taps of chunk c + 2 from the texels asked for one chunk ago; then the texels of chunk c + 3
            if (lane < CH && MDH_IRR_ABL != 2) {
               const int tap = (c + 2) * CH + lane;
               if (c + 2 < nchunks && tap < ntaps) {
                  MDH_IRR_TAP_COORD(tap, cc);
                  const f3 rad = atlas_tap_resolve(pr.rad, pr.fmt, tap_next, -1);
                  MDH_IRR_TAP_STORE(rad, cc, c + 2, lane);
               }
               const int tap3 = (c + 3) * CH + lane;
               if (c + 3 < nchunks && tap3 < ntaps) {
                  MDH_IRR_TAP_COORD(tap3, cc);
                  tap_next = atlas_tap_issue(pr.rad, pr.fmt, pr.pcx, pr.pcy, pr.rres, pr.rshift, pr.rad_w, pr.rad_h, cc.x, cc.y);
               }
            }
         } else if (c + 1 < nchunks && MDH_IRR_ABL != 3)
            MDH_IRR_WEIGHTS(c + 1, wv - 2, 2);
#ifdef MDH_PHASES
         const unsigned long long ph_t1 = __builtin_amdgcn_s_memtime();
#endif
         MDH_IRR_BARRIER();
#ifdef MDH_PHASES
         ph_work += ph_t1 - ph_t0; ph_wait += __builtin_amdgcn_s_memtime() - ph_t1;
#endif
      }
#ifdef MDH_PHASES
      if (lane == 0) { atomicAdd(&g_phase[2 * wv], ph_work); atomicAdd(&g_phase[2 * wv + 1], ph_wait); }
#endif
#undef MDH_IRR_TAP_COORD
#undef MDH_IRR_TAP_STORE
#undef MDH_IRR_BARRIER
#undef MDH_IRR_WEIGHTS
      if (wv == 0 && lane < ntex) {
         const f3 irradiance = F3(acc_x, acc_y, acc_z) / acc_w;
         const unsigned idx = atlas_index(pr.pcx, pr.ires, pr.ishift, i, j);
         atlas_store(pr.irr, pr.fmt, idx, irradiance_blend(pr, prev, hyst, idx, irradiance));
      }
      return;
   }
#endif
#if MDH_IRR_CHUNK
   if (pr.ires * pr.ires <= 64) {
      // One wavefront folds (a lane per texel) -- or two (MDH_IRR_FOLD_WAVES: the sums of a texel's four channels are four
      // independent chains of additions, so one wavefront can carry (x, y) and another (z, weight), each with the
      // reference's taps in the reference's order: the same bits).  The taps go through LDS in chunks of MDH_IRR_CHUNK,
      // two buffers: while the folding wavefronts work on chunk c, the others stage chunk c + 1.  The same taps in the
      // same order; a quarter of the LDS (room for the march kernels' workgroups beside it) and the staging hidden.
      constexpr int CH = MDH_IRR_CHUNK;
      constexpr int FOLDERS = MDH_IRR_FOLD_WAVES; // 1 or 2
      const int nchunks = (ntaps + CH - 1) / CH;
      const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
      const int x = lane % pr.ires, y = lane / pr.ires; // (used by the folding lanes only)
      const int i = tx * pr.ires + x, j = ty * pr.ires + y;
      const f2 nc = F2((centre(i, pr.pcx * pr.ires) + 1.0f) * 0.5f, (centre(j, pr.pcy * pr.ires) + 1.0f) * 0.5f);
      const f3 irr_dir = ray_id_to_ray_dir(F2(fract_(nc.x * pcx), fract_(nc.y * pcy)));
      pk2 acc_xy = {0.0f, 0.0f}, acc_zw = {0.0f, 0.0f};
      const pk2 dir_xy = {irr_dir.x, irr_dir.y};
      for (int t = threadIdx.x; t < min(CH, ntaps); t += MDH_IRR_BLOCK) MDH_IRR_STAGE(t, t);
      __syncthreads();
      for (int c = 0; c < nchunks; ++c) {
         const int base = c * CH, n_here = min(CH, ntaps - base);
         const float4 *buf = s_taps + (size_t)(c & 1) * 2 * CH;
         if (wv < FOLDERS) {
            if (lane < pr.ires * pr.ires) {
               // WHICH: 0 = all four channels (one folding wavefront), 1 = (x, y), 2 = (z, weight)
#define MDH_IRR_FOLD(WHICH, r_, d_)                                                           \
               do {                                                                           \
                  const float4 r = (r_), d = (d_);                                            \
                  const pk2 d_xy = {d.x, d.y}, r_xy = {r.x, r.y}, r_z1 = {r.z, r.w};          \
                  const pk2 p = dir_xy * d_xy;                                                \
                  const float w = max_((p.x + p.y) + irr_dir.z * d.z, 0.0f);                  \
                  const pk2 ww = {w, w};                                                      \
                  if (WHICH != 2) acc_xy = acc_xy + r_xy * ww;                                \
                  if (WHICH != 1) acc_zw = acc_zw + r_z1 * ww;                                \
               } while (0)
               // six taps per turn in two groups of three, each group on its way from LDS while the other is folded (the
               // folding wavefronts are the pass's critical path: 1 024 taps in the reference's order, each waiting for its two
               // LDS reads otherwise).  Three: two groups in flight are 12 LDS reads, and the counter a wavefront waits on holds 15.
#define MDH_IRR_FOLD_CHUNK(WHICH)                                                             \
               do {                                                                           \
                  constexpr int GR = 3;                                                       \
                  const int nt = n_here - n_here % (2 * GR);                                  \
                  int t = 0;                                                                  \
                  float4 ra[GR], da[GR], rb[GR], db[GR];                                      \
                  if (nt > 0) {                                                               \
                     _Pragma("unroll") for (int k = 0; k < GR; ++k) { ra[k] = buf[2 * k]; da[k] = buf[2 * k + 1]; } \
                  }                                                                           \
                  _Pragma("unroll 1") for (; t < nt; t += 2 * GR) {                           \
                     _Pragma("unroll") for (int k = 0; k < GR; ++k) { rb[k] = buf[2 * (t + GR + k)]; db[k] = buf[2 * (t + GR + k) + 1]; } \
                     _Pragma("unroll") for (int k = 0; k < GR; ++k) MDH_IRR_FOLD(WHICH, ra[k], da[k]); \
                     const int tn = t + 2 * GR < nt ? t + 2 * GR : t; /* (the last turn reads its own taps again) */ \
                     _Pragma("unroll") for (int k = 0; k < GR; ++k) { ra[k] = buf[2 * (tn + k)]; da[k] = buf[2 * (tn + k) + 1]; } \
                     _Pragma("unroll") for (int k = 0; k < GR; ++k) MDH_IRR_FOLD(WHICH, rb[k], db[k]); \
                  }                                                                           \
                  for (; t < n_here; ++t) MDH_IRR_FOLD(WHICH, buf[2 * t], buf[2 * t + 1]);    \
               } while (0)
               if (FOLDERS == 1) MDH_IRR_FOLD_CHUNK(0);
               else if (wv == 0) MDH_IRR_FOLD_CHUNK(1);
               else MDH_IRR_FOLD_CHUNK(2);
#undef MDH_IRR_FOLD_CHUNK
#undef MDH_IRR_FOLD
            }
         } else if (c + 1 < nchunks) {
            const int nbase = base + CH, n_next = min(CH, ntaps - nbase), off = ((c + 1) & 1) * CH;
            for (int t = (int)threadIdx.x - 64 * FOLDERS; t < n_next; t += MDH_IRR_BLOCK - 64 * FOLDERS) MDH_IRR_STAGE(nbase + t, off + t);
         }
         __syncthreads();
      }
      if (FOLDERS == 2) { // (z, weight) of every texel to the first wavefront, through the tap buffers (no longer read)
         pk2 *hand = (pk2 *)s_taps;
         if (wv == 1) hand[lane] = acc_zw;
         __syncthreads();
         if (wv == 0) acc_zw = hand[lane];
      }
      if (wv == 0 && lane < pr.ires * pr.ires) {
         const f3 irradiance = F3(acc_xy.x, acc_xy.y, acc_zw.x) / acc_zw.y;
         const unsigned idx = atlas_index(pr.pcx, pr.ires, pr.ishift, i, j);
         atlas_store(pr.irr, pr.fmt, idx, irradiance_blend(pr, prev, hyst, idx, irradiance));
      }
      return;
   }
#endif
   for (int tap = threadIdx.x; tap < ntaps; tap += MDH_IRR_BLOCK) MDH_IRR_STAGE(tap, tap);
#undef MDH_IRR_STAGE
   __syncthreads();
   for (int rem = threadIdx.x; rem < pr.ires * pr.ires; rem += MDH_IRR_BLOCK) {
      const int y = rem / pr.ires, x = rem - y * pr.ires;
      const int i = tx * pr.ires + x, j = ty * pr.ires + y;
      const f2 nc = F2((centre(i, pr.pcx * pr.ires) + 1.0f) * 0.5f, (centre(j, pr.pcy * pr.ires) + 1.0f) * 0.5f);
      const f3 irr_dir = ray_id_to_ray_dir(F2(fract_(nc.x * pcx), fract_(nc.y * pcy)));
      f3 irradiance = F3(0.0f, 0.0f, 0.0f);
      float total_weight = 0.0f;
#if MDH_IRR_PACKED
      // the same operations, two per instruction: (x, y) and (z, weight) pairs.  1.0f * w = w exactly.
      pk2 acc_xy = {0.0f, 0.0f}, acc_zw = {0.0f, 0.0f};
      const pk2 dir_xy = {irr_dir.x, irr_dir.y};
      for (int tap = 0; tap < ntaps; ++tap) {
         const float4 r = s_taps[2 * tap], d = s_taps[2 * tap + 1];
         const pk2 d_xy = {d.x, d.y}, r_xy = {r.x, r.y}, r_z1 = {r.z, r.w};
         const pk2 p = dir_xy * d_xy;
         const float w = max_((p.x + p.y) + irr_dir.z * d.z, 0.0f);
         const pk2 ww = {w, w};
         acc_xy = acc_xy + r_xy * ww;
         acc_zw = acc_zw + r_z1 * ww;
      }
      irradiance = F3(acc_xy.x, acc_xy.y, acc_zw.x);
      total_weight = acc_zw.y;
#else
      for (int tap = 0; tap < ntaps; ++tap) {
         const float4 r = s_taps[2 * tap], d = s_taps[2 * tap + 1];
         float w = max_(dot(irr_dir, xyz(d)), 0.0f);
         irradiance = irradiance + xyz(r) * w;
         total_weight += w;
      }
#endif
      irradiance = irradiance / total_weight;
      const unsigned idx = atlas_index(pr.pcx, pr.ires, pr.ishift, i, j);
      atlas_store(pr.irr, pr.fmt, idx, irradiance_blend(pr, prev, hyst, idx, irradiance));
   }
}

// -------------------------------------------------------------------- visibility pass
// compute_frustrum_visibility.glsl:8-42: one visibility ray per froxel (x, y, depth slice) and light.
//
// Round 4: the rays of a wavefront's froxels as a QUEUE WITH REPLACEMENT.  With one froxel per lane in lock step (the form
// until round 3, kept as MDH_VIS_QUEUE = 0 and in the literal build) a wavefront marched until its longest ray ended:
// 34 of 64 lanes alive per step (r03_c4_*), half of them blocked within a step or two while their neighbours cross the room.
// Here a wavefront owns MDH_VIS_ROUNDS x 64 consecutive froxels (in the same 8x8-tile order) and a lane whose ray has ended
// is SERVED -- its light's term added, the next light started, or the texel stored and the next froxel of the batch taken --
// as soon as MDH_VIS_REFILL lanes wait (or nobody marches).  Every froxel's arithmetic is what it was, operation for
// operation: WHICH lane computes a texel and WHEN changes, nothing else.
#ifndef MDH_VIS_QUEUE
#define MDH_VIS_QUEUE 0 // (measured on MI355X, light_shafts 1080p: the pass 0.067 -> 0.109 ms on its own and -1.5 % in flight --
                        //  serving a lane is ~250 instructions under a sparse mask, as much as its march saves: DESIGN.md, dropped)
#endif
#ifndef MDH_VIS_LOOP
#define MDH_VIS_LOOP 1 // froxels per lane of the lock-step form, one after the other (measured 2 and 4: the pass 0.074 -> 0.100 / 0.153 ms alone, +-0 in flight -- it is the length of a wavefront's chain, not the staging of the table)
#endif
#ifndef MDH_VIS_ROUNDS
#define MDH_VIS_ROUNDS 4
#endif
// LDS of the second form of the queue, per wavefront, in floats: seven per ray, the list (u16) and the reached bytes
#define MDH_VIS_Q_FLOATS (7 * 64 * MDH_VIS_ROUNDS + 64 * MDH_VIS_ROUNDS / 2 + 64 * MDH_VIS_ROUNDS / 4)
#ifndef MDH_VIS_REFILL
#define MDH_VIS_REFILL 16
#endif
// the froxel texel (i, j) at place `lin` of the launch.  Which froxels share a wavefront decides how many of its lanes march
// together: the texture is vw x (vh * vz) texels, row j = slice * vh + y, and in launch order a wavefront of the reference's
// 100^3 froxels was a strip of 64 froxels along x (100 is no multiple of 8: no 8x8 tiles either).  MDH_VIS_BLOCK3: blocks of
// 4 x 4 froxels x 4 depth slices -- neighbours in space, whose rays to a light pass the same geometry.
#ifndef MDH_VIS_BLOCK3
#define MDH_VIS_BLOCK3 1
#endif
MDH_DEV void froxel_texel(const KVolumetrics &vol, int W, int H, long lin, int &i, int &j)
{
   if (MDH_VIS_BLOCK3 && ((vol.vw | vol.vh | vol.vz) & 3) == 0) {
      const long blk = lin >> 6;
      const int l = (int)(lin & 63), bxn = vol.vw >> 2, byn = vol.vh >> 2;
      const int bx = (int)(blk % bxn);
      const long rest = blk / bxn;
      const int by = (int)(rest % byn), bz = (int)(rest / byn);
      i = bx * 4 + (l & 3);
      j = (bz * 4 + (l >> 4)) * vol.vh + by * 4 + ((l >> 2) & 3);
   } else if ((W & 7) == 0 && (H & 7) == 0) {
      const long tile = lin >> 6;
      const int l = (int)(lin & 63), tpr = W >> 3;
      i = (int)(tile % tpr) * 8 + (l & 7);
      j = (int)(tile / tpr) * 8 + (l >> 3);
   } else {
      j = (int)(lin / W);
      i = (int)(lin - (long)j * W);
   }
}
// its sample point and the direction of its camera ray (compute_frustrum_visibility.glsl:22-37)
MDH_DEV void froxel_point(const KVolumetrics &vol, const KCamera &cam, int W, int H, int i, int j, f3 &pos, f3 &dir)
{
   const float px = centre(i, W), py = centre(j, H);
   const float norm_height = (py + 1.0f) * 0.5f;
   const float tex_height = norm_height * (float)vol.vz;
   const float depth = __builtin_floorf(tex_height);
   const float fract_height = tex_height - depth;
   const float frag_height = fract_height * 2.0f - 1.0f;
   f3 origin;
   camera_ray(cam, px, frag_height, origin, dir);
   pos = origin + (dir * depth) * vol.vstep;
}
// the texel's ray length: raycast (raymarching.glsl:25-37) without the arg-min -- only the collision point is used here
template <int PART> MDH_DEV float scattering_length(const KScene &sc, const KVolumetrics &vol, const KCamera &cam, int i, int j)
{
   const float px = centre(i, vol.sw), py = centre(j, vol.sh);
   f3 from, dir;
   camera_ray(cam, px, py, from, dir);
   const float max_depth = vol.vstep * (float)vol.vz; // volumetrics.glsl:3-4
   f3 to = from + dir * max_depth;
   int steps;
   float t;
   if (march_plain<PART>(sc, from, dir, sc.max_dist, t, steps)) to = from + dir * t;
   return min_(length(to - from), max_depth);
}
// `vis_blocks`: the workgroups of the launch that work on froxels; the ones behind them march the camera rays of the
// SCATTERING texels (k_scat_march's work: the texels' lengths, which depend on the camera and the scene only).  That march is
// a thousand wavefronts whose longest ray alone takes 40 us: inside this launch it hides under the froxels' rays (frames
// only: a single visibility pass leaves the scattering texture alone, run_pass).
template <int PART> __global__ __launch_bounds__(MDH_BLOCK) void k_visibility(KScene sc, KVolumetrics vol, KCamera cam, int vis_blocks)
{
   stage_table(sc);
   if ((int)blockIdx.x >= vis_blocks) { // (workgroup-uniform)
      const long lin = (long)((int)blockIdx.x - vis_blocks) * MDH_BLOCK + threadIdx.x;
      if (lin >= (long)vol.sw * vol.sh) return;
      const int j = (int)(lin / vol.sw), i = (int)(lin - (long)j * vol.sw);
      vol.scat[(size_t)j * vol.sw + i].w = scattering_length<PART>(sc, vol, cam, i, j);
      return;
   }
   const int W = vol.vw, H = vol.vh * vol.vz;
   const long n = (long)W * H;
#if MDH_VIS_QUEUE == 2
   // The second form of the queue (round 4): the set-up and the finish of a froxel stay DENSE -- a wavefront owns MDH_VIS_ROUNDS x 64
   // froxels and walks them round by round at full lanes -- and only the MARCH goes through a list with replacement, as
   // queued_visibility does for the probe rays: the rays that need marching are listed (ballot ranks), their origin, direction and
   // length wait in LDS, a lane whose ray has ended takes the next entry (seven LDS reads), a ray that reaches its light sets its byte.
   {
      constexpr int R = MDH_VIS_ROUNDS, NJ = 64 * R;
      const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
      const long wave_global = (long)blockIdx.x * (MDH_BLOCK / 64) + wv;
      const long begin = wave_global * NJ;
      if (begin >= n) return; // (wave-uniform)
      float *wbase = (float *)(s_tab + sc.table_f4 + sc.part_bits_f4) + (size_t)wv * MDH_VIS_Q_FLOATS;
      float *jp = wbase;                                             // [7][NJ]: pos.xyz, L.xyz, L_dist of every froxel's ray
      unsigned short *list = (unsigned short *)(wbase + 7 * NJ);     // [NJ] the rays to march
      unsigned char *reached = (unsigned char *)(list + NJ);         // [NJ] 1: the ray reached its light (or never had to be marched)
      f3 result[R], dirs[R], Ls[R], rads[R];
      float Ld[R];
#pragma unroll
      for (int r = 0; r < R; ++r) result[r] = F3(0.0f, 0.0f, 0.0f);
      for (int l = 0; l < sc.total_lights; ++l) { // sample_lights :8-19, light by light for all the wavefront's froxels
         int njobs = 0;
#pragma unroll
         for (int r = 0; r < R; ++r) { // set-up, dense
            const long lin = begin + r * 64 + lane;
            int i, j;
            froxel_texel(vol, W, H, lin < n ? lin : n - 1, i, j);
            f3 pos;
            froxel_point(vol, cam, W, H, i, j, pos, dirs[r]);
            rads[r] = sample_light<(PART & MDH_PF_CUSTOM) != 0>(sc, l, pos, F3(1.0f, 0.0f, 0.0f), Ls[r], Ld[r]); // compute_frustrum_visibility.glsl:12
            MDH_WORK(0);
            const int e = r * 64 + lane;
            jp[0 * NJ + e] = pos.x; jp[1 * NJ + e] = pos.y; jp[2 * NJ + e] = pos.z;
            jp[3 * NJ + e] = Ls[r].x; jp[4 * NJ + e] = Ls[r].y; jp[5 * NJ + e] = Ls[r].z; jp[6 * NJ + e] = Ld[r];
            const bool need = lin < n && 0.0f < Ld[r]; // raycast_visibility's first test: a loop that is never entered leaves the light visible
            reached[e] = need ? 0 : 1;
            const unsigned long long m = __ballot(need);
            if (need) list[njobs + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (unsigned short)e;
            njobs += (int)__popcll(m);
         }
         __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
         __builtin_amdgcn_wave_barrier();
         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
         { // the march, through the list with replacement (raymarching.glsl:39-56 from its second test on)
            int head = 0, job = -1;
            float total = 0.0f, vmax = 0.0f;
            f3 o = F3(0.0f, 0.0f, 0.0f), d = o;
            for (;;) {
               const unsigned long long idle = __ballot(job < 0);
               const int n_idle = (int)__popcll(idle);
               if (head < njobs && n_idle >= MDH_VIS_REFILL) {
                  if (job < 0) {
                     const int my = head + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
                     if (my < njobs) {
                        const int e = list[my];
                        o = F3(jp[0 * NJ + e], jp[1 * NJ + e], jp[2 * NJ + e]);
                        d = F3(jp[3 * NJ + e], jp[4 * NJ + e], jp[5 * NJ + e]);
                        vmax = jp[6 * NJ + e];
                        total = 0.0f;
                        job = e;
                     }
                  }
                  head += n_idle;
               }
               if (__ballot(job >= 0) == 0ull) break;
               if (job >= 0) {
                  MDH_WORK(1);
                  const float dist = sdf<PART>(sc, o + d * total);
                  if (dist < MDH_EPS) job = -1; // blocked: the byte stays 0
                  else {
                     total += dist;
                     if (!(total < vmax)) { reached[job] = 1; job = -1; }
                  }
               }
            }
         }
         __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
         __builtin_amdgcn_wave_barrier();
         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
         for (int r = 0; r < R; ++r) { // the light's term, dense
            const float visibility = reached[r * 64 + lane] ? 1.0f : 0.0f;
            const f3 L_in = rads[r] * (sexp(-Ld[r] * MDH_TAU) * visibility);
            result[r] = result[r] + (L_in * MDH_TAU) * henvey_greenstein_phase(Ls[r], dirs[r]);
         }
         __builtin_amdgcn_wave_barrier(); // (the next light's set-up overwrites what this finish read)
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
         const long lin = begin + r * 64 + lane;
         if (lin < n) {
            int i, j;
            froxel_texel(vol, W, H, lin, i, j);
            float *o = vol.vis + ((size_t)j * W + i) * 3;
            o[0] = result[r].x; o[1] = result[r].y; o[2] = result[r].z;
         }
      }
   }
#elif MDH_VIS_QUEUE
   const long wave_global = (long)blockIdx.x * (MDH_BLOCK / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   const long begin = wave_global * (64 * MDH_VIS_ROUNDS);
   if (begin >= n) return; // (wave-uniform)
   const long end = min(n, begin + 64 * MDH_VIS_ROUNDS);
   long next = begin; // (wave-uniform) the first froxel of the batch nobody has taken
   // a lane's froxel (lin < 0: none), its light, its ray
   long lin = -1;
   int light = 0;
   bool marching = false;
   float vis = 0.0f, total = 0.0f, L_dist = 0.0f;
   f3 pos = F3(0.0f, 0.0f, 0.0f), dir = pos, L = pos, radiance = pos, result = pos;
   for (;;) {
      const unsigned long long mm = __ballot(marching);
      const bool waiting = !marching && (lin >= 0 || next < end); // a ray that has ended, or an empty lane while froxels are left
      const unsigned long long wm = __ballot(waiting);
      if (mm == 0ull && wm == 0ull) break;
      if (mm == 0ull || __popcll(wm) >= MDH_VIS_REFILL) {
         if (waiting) {
            if (lin >= 0) { // sample_lights :8-19: the ended ray's term, in light order
               const f3 L_in = radiance * (sexp(-L_dist * MDH_TAU) * vis);
               result = result + (L_in * MDH_TAU) * henvey_greenstein_phase(L, dir);
               ++light;
            }
            if (lin >= 0 && light >= sc.total_lights) {
               int i, j;
               froxel_texel(vol, W, H, lin, i, j);
               float *o = vol.vis + ((size_t)j * W + i) * 3;
               o[0] = result.x; o[1] = result.y; o[2] = result.z;
               lin = -1;
            }
         }
         { // empty lanes take the next froxels of the batch, in lane order
            const bool empty = waiting && lin < 0;
            const unsigned long long em = __ballot(empty);
            if (empty) {
               const long mine = next + (long)__builtin_amdgcn_mbcnt_hi((unsigned)(em >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)em, 0u));
               if (mine < end) {
                  lin = mine;
                  int i, j;
                  froxel_texel(vol, W, H, lin, i, j);
                  froxel_point(vol, cam, W, H, i, j, pos, dir);
                  result = F3(0.0f, 0.0f, 0.0f);
                  light = 0;
                  if (sc.total_lights <= 0) { // no light: the texel is 0
                     float *o = vol.vis + ((size_t)j * W + i) * 3;
                     o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f;
                     lin = -1;
                  }
               }
            }
            next = min(end, next + (long)__popcll(em));
         }
         if (waiting && lin >= 0) { // the next light's ray (raycast_visibility, raymarching.glsl:39-56, from its first test on)
            radiance = sample_light<(PART & MDH_PF_CUSTOM) != 0>(sc, light, pos, F3(1.0f, 0.0f, 0.0f), L, L_dist); // compute_frustrum_visibility.glsl:12
            MDH_WORK(0);
            total = 0.0f;
            vis = 1.0f;
            marching = total < L_dist; // (a loop that is never entered: visible)
         }
      }
      if (marching) {
         MDH_WORK(1);
         const float dist = sdf<PART>(sc, pos + L * total);
         if (dist < MDH_EPS) { vis = 0.0f; marching = false; }
         else {
            total += dist;
            if (!(total < L_dist)) marching = false;
         }
      }
   }
#else
   // (MDH_VIS_LOOP froxels per lane, one after the other: a workgroup stages the scene table once for all of them)
#pragma unroll 1
   for (int turn = 0; turn < MDH_VIS_LOOP; ++turn) {
      const long lin = ((long)blockIdx.x * MDH_VIS_LOOP + turn) * MDH_BLOCK + threadIdx.x;
      if (lin >= n) return;
      int i, j;
      froxel_texel(vol, W, H, lin, i, j);
      f3 pos, dir;
      froxel_point(vol, cam, W, H, i, j, pos, dir);
      f3 result = F3(0.0f, 0.0f, 0.0f);
      for (int l = 0; l < sc.total_lights; ++l) { // sample_lights :8-19
         f3 L;
         float L_dist;
         f3 radiance = sample_light<(PART & MDH_PF_CUSTOM) != 0>(sc, l, pos, F3(1.0f, 0.0f, 0.0f), L, L_dist); // compute_frustrum_visibility.glsl:12
         float visibility = raycast_visibility<PART>(sc, pos, L, L_dist);
         f3 L_in = radiance * (sexp(-L_dist * MDH_TAU) * visibility);
         result = result + (L_in * MDH_TAU) * henvey_greenstein_phase(L, dir);
      }
      float *o = vol.vis + ((size_t)j * W + i) * 3;
      o[0] = result.x; o[1] = result.y; o[2] = result.z;
   }
#endif
}

// -------------------------------------------------------------------- scattering pass
// accumulate_scattering.glsl:9-48.  Round 4: two kernels.
//   k_scat_march  one lane per scattering texel: the camera ray's collision point -> len (the texel's fourth component)
//   k_scat_fold   a texel's steps SPREAD OVER LANES.  Until round 3 one lane walked its texel's <= 100 froxel taps by itself
//                 (batches of 8 in flight): 980 wavefronts on 1 024 SIMDs, 142 VGPRs, 10.8 % of the VALU issue peak.  A step's
//                 tap and its exp (-f tau) factor depend on the texel and the step only, and f runs through the same values
//                 for every texel (f_0 = 0, f_k+1 = f_k + step in fp32): a workgroup takes MDH_SCAT_TEXELS texels, computes
//                 f_k, floor (f_k / froxel step) and exp (-f_k tau) once per step into LDS, then every (texel, step) pair's
//                 product tap * factor -- one lane each, all in flight together -- and finally one lane per texel and CHANNEL
//                 adds its products IN STEP ORDER (the same additions in the same order: the same bits).
#ifndef MDH_SCAT_SPLIT
#define MDH_SCAT_SPLIT 1
#endif
#ifndef MDH_SCAT_TEXELS
#define MDH_SCAT_TEXELS 16 // texels per workgroup of k_scat_fold (a power of two, at most 64)
#endif
#ifndef MDH_SCAT_CHUNK
#define MDH_SCAT_CHUNK 128 // steps whose products are in LDS together (16 texels x 128 steps x 12 B = 24 KB)
#endif
#ifndef MDH_SCAT_BLOCK
#define MDH_SCAT_BLOCK 256 // threads of k_scat_fold (measured 1 024: two workgroups per CU, the pass 0.045 -> 0.172 ms)
#endif
template <int PART> __global__ __launch_bounds__(MDH_BLOCK) void k_scat_march(KScene sc, KVolumetrics vol, KCamera cam)
{
   stage_table(sc);
   const int W = vol.sw, H = vol.sh;
   const long lin = (long)blockIdx.x * MDH_BLOCK + threadIdx.x;
   if (lin >= (long)W * H) return;
   const int j = (int)(lin / W), i = (int)(lin - (long)j * W);
   vol.scat[(size_t)j * W + i].w = scattering_length<PART>(sc, vol, cam, i, j);
}
#ifndef MDH_JIT
__global__ __launch_bounds__(MDH_SCAT_BLOCK) void k_scat_fold(KVolumetrics vol)
{
   constexpr int T = MDH_SCAT_TEXELS, CH = MDH_SCAT_CHUNK;
   __shared__ float s_f[CH], s_rel[CH], s_e[CH]; // per step of the chunk: f, floor (f / froxel step), exp (-f tau)
   __shared__ float s_len[T];
   __shared__ int s_n[T];                        // steps of each texel: the k with f_k < len
   __shared__ float s_prod[CH][3][T];
   const int W = vol.sw, H = vol.sh, tid = threadIdx.x;
   const long first = (long)blockIdx.x * T, n_tex = (long)W * H;
   const float max_depth = vol.vstep * (float)vol.vz;
   if (tid < T) {
      const long lin = first + tid;
      s_len[tid] = lin < n_tex ? vol.scat[lin].w : 0.0f; // (k_scat_march)
      s_n[tid] = 0;
   }
   // lanes of the fold: channel c of texel t
   const int fc = tid / T, ft = tid % T;
   const bool folder = tid < 3 * T && first + ft < n_tex;
   float acc = 0.0f;
   for (int c0 = 0;; c0 += CH) {
      // the chunk's steps: f by the loop's own additions (every texel's loop runs through these values)
      if (tid < T) s_n[tid] = 0; // (steps of this chunk: set by the pair that holds the texel's last one)
      if (tid < CH) {
         float f = 0.0f;
         for (int q = 0; q < c0 + tid; ++q) f += vol.sstep;
         s_f[tid] = f;
         s_rel[tid] = __builtin_floorf(f / vol.vstep); // sample_visibility :9-15
         s_e[tid] = sexp(-f * MDH_TAU);
      }
      __syncthreads();
      if (!(s_f[0] < max_depth)) break; // (len <= max_depth: no texel has a step here; uniform)
      // (texel, step) pairs: the tap and its factor, one lane each -- four pairs per lane and turn, their 48 texel loads all in
      // flight together (pairs without a step -- f past the texel's length, a texel past the image -- load nothing: sampled
      // all the same, as the one-kernel form did, they double the pass's loads and it takes 0.066 instead of 0.045 ms)
      for (int p0 = tid; p0 < T * CH; p0 += 4 * MDH_SCAT_BLOCK) {
         float tx[4][3];
         bool live[4];
#pragma unroll
         for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * MDH_SCAT_BLOCK;
            const int t = p % T, k = min(p / T, CH - 1);
            const long lin = min(first + t, n_tex - 1);
            live[u] = p < T * CH && first + t < n_tex && s_f[k] < s_len[t];
            tx[u][0] = tx[u][1] = tx[u][2] = 0.0f;
            if (live[u]) {
               const int j = (int)(lin / W), i = (int)(lin - (long)j * W);
               const f2 norm_pos = F2(0.5f * (centre(i, W) + 1.0f), 0.5f * (centre(j, H) + 1.0f));
               tex_sample<3>(vol.vis, vol.vw, vol.vh * vol.vz, norm_pos.x, (norm_pos.y + s_rel[k]) / (float)vol.vz, tx[u]);
               // (the last of the texel's steps in this chunk says how many there are: f grows, the steps are a prefix)
               if (k + 1 == CH || !(s_f[k + 1] < s_len[t])) s_n[t] = k + 1;
            }
         }
#pragma unroll
         for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * MDH_SCAT_BLOCK;
            const int t = p % T, k = min(p / T, CH - 1);
            if (live[u]) {
               const f3 pr = F3(tx[u][0], tx[u][1], tx[u][2]) * s_e[k];
               s_prod[k][0][t] = pr.x; s_prod[k][1][t] = pr.y; s_prod[k][2][t] = pr.z;
            }
         }
      }
      __syncthreads();
      if (folder) { // L = L + tap * factor, step by step (eight products on their way from LDS at a time)
         const int m = s_n[ft];
         int k = 0;
         for (; k + 8 <= m; k += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = s_prod[k + q][fc][ft];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = acc + v[q];
         }
         for (; k < m; ++k) acc = acc + s_prod[k][fc][ft];
      }
      if (!__syncthreads_or(tid < T && s_n[tid] == CH)) break; // some texel filled the chunk: its loop goes on
   }
   if (folder) ((float *)&vol.scat[first + ft])[fc] = acc * vol.sstep;
}
#endif
// (the one-kernel form of rounds 1 to 3: MDH_SCAT_SPLIT = 0 and the literal build)
template <int PART> __global__ __launch_bounds__(MDH_BLOCK) void k_scattering(KScene sc, KVolumetrics vol, KCamera cam)
{
   stage_table(sc);
   const int W = vol.sw, H = vol.sh;
   const long lin = (long)blockIdx.x * MDH_BLOCK + threadIdx.x;
   if (lin >= (long)W * H) return;
   const int j = (int)(lin / W), i = (int)(lin - (long)j * W);
   const float px = centre(i, W), py = centre(j, H);
   const f2 norm_pos = F2(0.5f * (px + 1.0f), 0.5f * (py + 1.0f));
   const float len = scattering_length<PART>(sc, vol, cam, i, j);
   f3 L = F3(0.0f, 0.0f, 0.0f);
   // The froxel taps of consecutive steps are independent, but each costs a trip to L2/HBM and the
   // pass has one wavefront per SIMD: steps go in batches of MDH_SCAT_BATCH whose loads are all in
   // flight together, then fold into L in step order (a step past `len` loads a valid texel and is
   // not added).  f runs through the same values as the reference's loop.
   for (float f = 0.0f; f < len;) {
      float fs[MDH_SCAT_BATCH], tx[MDH_SCAT_BATCH][3];
#pragma unroll
      for (int k = 0; k < MDH_SCAT_BATCH; ++k) {
         fs[k] = f;
         const float rel = __builtin_floorf(f / vol.vstep); // sample_visibility :9-15
         tex_sample<3>(vol.vis, vol.vw, vol.vh * vol.vz, norm_pos.x, (norm_pos.y + rel) / (float)vol.vz, tx[k]);
         f += vol.sstep;
      }
#pragma unroll
      for (int k = 0; k < MDH_SCAT_BATCH; ++k)
         if (fs[k] < len) L = L + F3(tx[k][0], tx[k][1], tx[k][2]) * exp_(-fs[k] * MDH_TAU);
   }
   L = L * vol.sstep;
   vol.scat[(size_t)j * W + i] = make_float4(L.x, L.y, L.z, len);
}

// ---------------------------------------------------------------- partition-table build
struct PartBuildArgs {
   int method;     // 0 CPU_Best, 1 CPU_Fast, 2 GPU_Fast (renderers.ads:93)
   int gx, gy, gz; // cells visited; GPU_Fast: 2 * (dims / 2), the compute dispatch of renderers.adb:539-549
   float sp[3], off[3];
   float gpu_diag; // Length(spacing) as the generated GLSL text carries it (scenes.adb:1140-1143)
   int *table;     // [cell][nk + index_count]
   int *warnings;
};
#define MDH_PART_MAX_PRE 256

// distance of primitive i of kind k through the table; ADA_DIV for the CPU builders, which
// go through Primitives.Eval_Dist (madarch-primitives.adb:90-108)
template <bool ADA_DIV> MDH_DEV float part_dist(const KScene &sc, int k, int i, f3 x)
{
   // k may differ per lane here (candidate lists): plain per-lane header reads, not hdr()
   const int type = tab_int(H_KTYPE + k), slot = tab_int(H_KSLOT + k) + prim_slots(type) * i;
   if (type == PK_CUSTOM) { // the program must be wave-uniform: one round per user-defined kind, its lanes active
      float r = 0.0f;
      const int nk = hdr(H_NK);
      for (int kk = 0; kk < nk; ++kk)
         if (hdr(H_KTYPE + kk) == PK_CUSTOM && k == kk) r = xdist<ADA_DIV>(kk, i, x);
      return r;
   }
   if (type == PK_TRIANGLE) return sd_triangle<ADA_DIV>(xyz(s_tab[slot]), xyz(s_tab[slot + 1]), xyz(s_tab[slot + 2]), x);
   return prim_dist(type, slot, x);
}

// ONE WAVEFRONT PER GRID CELL (round 4; until round 3 one LANE per cell walked every loop below by itself, its candidate
// arrays in 784 bytes of scratch memory: 0.49 ms for simple_scene's 2 000 cells on 32 wavefronts, every frame of ball_game).
// Update_Partitioning_CPU (renderers.adb:551-755) for methods 0/1, partitioning_compute_grid_cell (scenes.adb:1120-1187)
// for method 2.  The reference's loops are kept as SETS and ORDERS, not as schedules:
//   * the closest primitive at the cell's centre: lanes = the instances of one kind at a time (a uniform type: no
//     divergence), a wave minimum -- min is order-free;
//   * the pre-candidates (closer than closest + the cell's diagonal) in scene order: ballot ranks, kind by kind;
//   * CPU_Best's 27 sample points (Find_Candidates, renderers.adb:669-723): lane s = sample point s in the reference's
//     loop order, the pre-candidates walked uniformly (LDS broadcasts) with the reference's strict `<` (the first of
//     equal distances wins); the points then accept their winners in order, which numbers them as the reference does;
//   * the lists, kind by kind in acceptance order, cut at Index_Count with one warning per overflowing kind
//     (Write_Partitioning_Info, renderers.adb:578-608).
MDH_DEV float wave_min(float v)
{
#pragma unroll
   for (int o = 32; o > 0; o >>= 1) v = min_(v, __shfl_xor(v, o));
   return v;
}
__global__ __launch_bounds__(64) void k_partition_build(KScene sc, PartBuildArgs a)
{
   __shared__ unsigned short s_pre[MDH_PART_MAX_PRE]; // (kind << 12) | index, in scene order
   __shared__ unsigned char s_acc[MDH_PART_MAX_PRE];  // method 0: the candidate's acceptance number (0: not accepted)
   __shared__ unsigned short s_ord[32];               // method 0: the candidate accepted as number s + 1
   stage_table(sc);
   const int lin = blockIdx.x, lane = threadIdx.x;
   if (lin >= a.gx * a.gy * a.gz) return;
   const int X = lin / (a.gy * a.gz), Y = (lin / a.gz) % a.gy, Z = lin % a.gz;
   const int cell = X * a.gy * a.gz + Y * a.gz + Z;
   if (cell >= sc.part_cells) return;
   const f3 sp = F3(a.sp[0], a.sp[1], a.sp[2]), off = F3(a.off[0], a.off[1], a.off[2]);
   const int nk = hdr(H_NK);
   int npre = 0, nacc = 0;
   // every instance of every kind against `thr` at `center`, appended in scene order
#define MDH_PART_COLLECT(ADA_, center_, thr_)                                                                                \
   for (int k = 0; k < nk; ++k) {                                                                                           \
      const int n = hdr(H_KCOUNT + k);                                                                                      \
      for (int c = 0; c < n; c += 64) {                                                                                     \
         const int i = c + lane;                                                                                            \
         const bool in = i < n;                                                                                             \
         const float d = part_dist<ADA_>(sc, k, in ? i : 0, center_);                                                       \
         const bool take = in && d < (thr_);                                                                                \
         const unsigned long long m = __ballot(take);                                                                       \
         const int at = npre + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); \
         if (take && at < MDH_PART_MAX_PRE) { s_pre[at] = (unsigned short)((k << 12) | i); s_acc[at] = 0; }                  \
         npre = min(MDH_PART_MAX_PRE, npre + (int)__popcll(m));                                                             \
      }                                                                                                                     \
   }
   if (a.method == 2) {
      const f3 center = (F3((float)X, (float)Y, (float)Z) + F3s(0.5f)) * sp + off;
      const float thr = closest_primitive<true>(sc, center) + a.gpu_diag;
      MDH_PART_COLLECT(false, center, thr)
   } else {
      const f3 grid_pos = F3((float)X, (float)Y, (float)Z) * sp + off;
      const float cell_diag = length(sp);
      const f3 center = grid_pos + sp * 0.5f;
      float closest = 1.0e10f;
      for (int k = 0; k < nk; ++k) {
         const int n = hdr(H_KCOUNT + k);
         for (int c = 0; c < n; c += 64)
            if (c + lane < n) closest = min_(closest, part_dist<true>(sc, k, c + lane, center));
      }
      closest = wave_min(closest);
      MDH_PART_COLLECT(true, center, closest + cell_diag)
      if (a.method == 0) { // Find_Candidates, renderers.adb:669-723: 3x3x3 sample points, lane = point
         const int sx = lane / 9 + 1, sy = (lane / 3) % 3 + 1, sz = lane % 3 + 1;
         const f3 offv = (F3((float)sx, (float)sy, (float)sz) - F3s(1.0f)) / (F3s(3.0f) - F3s(1.0f));
         const f3 pt = offv * sp + grid_pos;
         float c = 1.0e10f;
         int ci = -1;
         for (int q = 0; q < npre; ++q) {
            const int e = __builtin_amdgcn_readfirstlane((int)s_pre[q]);
            const float d = part_dist<true>(sc, e >> 12, e & 0xfff, pt);
            if (d < c) { c = d; ci = q; }
         }
         // acceptance ORDER defines the order of a kind's indices: the points accept in the reference's loop order
         // (every lane keeps the same books: its own stores are what it reads back)
         for (int s = 0; s < 27; ++s) {
            const int q = __builtin_amdgcn_readlane(ci, s);
            if (q >= 0 && s_acc[q] == 0) { s_ord[nacc] = (unsigned short)q; s_acc[q] = (unsigned char)(++nacc); }
         }
      }
   }
#undef MDH_PART_COLLECT
   // write counts and indices kind by kind (Write_Partitioning_Info, renderers.adb:578-608);
   // the reference only warns when a cell overflows Index_Count, here the list is cut
   int *rec = a.table + (size_t)cell * (nk + sc.part_index_count);
   int written = 0;
   const int nlist = a.method == 0 ? nacc : npre; // the accepted candidates, in the order their kind's entries keep
   for (int k = 0; k < nk; ++k) {
      int n = 0;
      for (int c = 0; c < nlist; c += 64) {
         const int j = c + lane;
         const int e = j < nlist ? (int)s_pre[a.method == 0 ? (int)s_ord[j] : j] : -1;
         const bool mine = e >= 0 && (e >> 12) == k;
         const unsigned long long m = __ballot(mine);
         const int at = written + n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
         if (mine && at < sc.part_index_count) rec[nk + at] = e & 0xfff;
         n += (int)__popcll(m);
      }
      if (written + n > sc.part_index_count) {
         if (lane == 0) atomicAdd(a.warnings, 1);
         n = sc.part_index_count - written;
      }
      if (lane == 0) rec[k] = n;
      written += n;
   }
}

// The candidate lists of every cell once more as one bit per declared primitive (partitioning_closest_bits,
// mdh_device.h), derived from the lists exactly as the lookup walks them: kind by kind, a kind's entries up to its
// count, the whole walk cut at Index_Count.  One lane per cell; runs behind every build on the build's stream.
__global__ __launch_bounds__(64) void k_partition_bits(KScene sc, int *table)
{
   stage_table(sc);
   const int cell = blockIdx.x * 64 + threadIdx.x;
   if (cell >= sc.part_cells) return;
   const int nk = hdr(H_NK);
   const int *rec = table + (size_t)cell * (nk + sc.part_index_count);
   unsigned *words = (unsigned *)(table + sc.part_mask_off) + (size_t)cell * sc.part_mask_words;
   for (int q = 0; q < sc.part_mask_words; ++q) words[q] = 0u;
   int i = 0;
   for (int k = 0; k < nk; ++k) {
      const int size = i + rec[k], base = hdr(H_KBASE + k), kmax = hdr(H_KMAX + k);
      const int stop = min(size, sc.part_index_count);
      for (; i < stop; ++i) {
         const int pi = rec[nk + i];
         if (pi >= 0 && pi < kmax) words[(base + pi) >> 5] |= 1u << ((base + pi) & 31);
      }
      i = size;
   }
}

// ------------------------------------------------------------------- Eval_Distance_To
// renderers.adb:499-526, one query point per lane
struct EvalArgs {
   int n, n_kinds;
   int kinds[MDH_MAX_KINDS];
   int host_count[MDH_MAX_KINDS];
   const float *pts;
   float *normals;
   float *dist;
};
template <bool ADA_DIV> __global__ __launch_bounds__(64) void k_eval_distance(KScene sc, EvalArgs a)
{
   stage_table(sc);
   const int q = blockIdx.x * 64 + threadIdx.x;
   if (q >= a.n) return;
   const f3 p = F3(a.pts[3 * q], a.pts[3 * q + 1], a.pts[3 * q + 2]);
   float closest = 1.0e10f;
   f3 normal = F3(0.0f, 0.0f, 0.0f);
   for (int kk = 0; kk < a.n_kinds; ++kk) {
      const int k = a.kinds[kk], type = hdr(H_KTYPE + k);
      for (int i = 0; i < a.host_count[k]; ++i) {
         float d = part_dist<ADA_DIV>(sc, k, i, p);
         if (d < closest) {
            closest = d;
            const int slot = hdr(H_KSLOT + k) + prim_slots(type) * i;
            float4 A = s_tab[slot];
            switch (type) {
            case PK_CUSTOM: normal = xnormal<ADA_DIV>(k, i, p); break;
            case PK_SPHERE: normal = normalize(p - xyz(A)); break;
            case PK_PLANE: normal = xyz(A); break;
            case PK_BOX: normal = nrm_box(A, s_tab[slot + 1], p); break;
            default: normal = nrm_triangle<ADA_DIV>(xyz(A), xyz(s_tab[slot + 1]), xyz(s_tab[slot + 2]), p); break;
            }
         }
      }
   }
   a.dist[q] = closest;
   if (a.normals) { a.normals[3 * q] = normal.x; a.normals[3 * q + 1] = normal.y; a.normals[3 * q + 2] = normal.z; }
}

// ---------------------------------------------------------------------- the window's pixels
// What Swap_Buffers (renderers.adb:320) puts on screen: the screen pass writes float colours and
// the default framebuffer of the reference's window is RGBA8 without sRGB encoding, so each
// channel is clamped to [0, 1] and converted to the nearest of the 256 levels (OpenGL 4.3 core
// section 2.3.5.1, round to nearest; ties to even; a NaN shows as 0).  16 bytes read and 4 written per pixel.
#ifndef MDH_JIT
__global__ __launch_bounds__(256) void k_present(const float4 *__restrict__ fb, unsigned *__restrict__ out, int n)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) out[q] = pack_rgba8(fb[q]);
}
#endif
