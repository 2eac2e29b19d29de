// mdh_march.h -- pixel_color_probes (reference glsl/render_probes.glsl:246-291) for gfx950.
//
// Two formulations were built and measured on MI355X (BASELINE config 3, 1080p):
//  * a per-lane ray STATE MACHINE -- one shared march loop, lanes whose ray ended wait for a
//    ballot threshold and then run their transition together (the classic persistent-ray
//    design).  Bit-exact, but 3-4x SLOWER than the structured form at every threshold
//    (1.4-2.4 ms radiance, 3.4-4.4 ms screen vs 0.58 / 0.97 ms): the transitions (BRDF,
//    light sampling, atlas taps: ~40 % of the work) then run under sparse exec masks, and the
//    machine keeps every variable of every stage live across the loop.  Removed.
//  * the lock-step STRUCTURED form below, which is what the kernels run.
#pragma once

#include "mdh_device.h"

#ifndef MDH_REUSE_FOLDED
#define MDH_REUSE_FOLDED 1
#endif
#ifndef MDH_SHARE_FIRST_STEP
#define MDH_SHARE_FIRST_STEP 1
#endif
#ifndef MDH_TAP_EARLY
#define MDH_TAP_EARLY 1
#endif
#ifndef MDH_TWIN_PREV_PART
#define MDH_TWIN_PREV_PART 0
#endif
#ifndef MDH_TAP_EARLY_PART
#define MDH_TAP_EARLY_PART 0
#endif
#ifndef MDH_PART_PARK_VD
#define MDH_PART_PARK_VD 1
#endif
#define MDH_PARK_VD_ROW 19 // (rows 19-21: allocated by the host for scenes with a space partition, lds_bytes_screen; six wavefronts per SIMD need 22 rows or fewer)
#ifndef MDH_SKIP_NULL_RAYS
#define MDH_SKIP_NULL_RAYS 1
#endif
#ifndef MDH_TAP_SQRT_CORE
#define MDH_TAP_SQRT_CORE 1
#endif
#ifndef MDH_TWIN_PREV
#define MDH_TWIN_PREV 1
#endif

struct MachineCfg {
   bool direct_specular;   // M_COMPUTE_DIRECT_SPECULAR
   int spec_mode;          // M_COMPUTE_INDIRECT_SPECULAR: 0 none, 1 .. 3 the three bodies of render_probes.glsl:264-272
   int ao_steps;           // M_AMBIENT_OCCLUSION_STEPS
};

// cage probe i of the cell that holds p (render_probes.glsl:13-20)
MDH_DEV i3 cage_probe(const KProbes &pr, i3 gp, int i)
{
   i3 q;
   q.x = iclamp_(gp.x + (i & 1), 0, pr.gx - 1);
   q.y = iclamp_(gp.y + ((i >> 1) & 1), 0, pr.gy - 1);
   q.z = iclamp_(gp.z + ((i >> 2) & 1), 0, pr.gz - 1);
   return q;
}

// =============================================================================================
// shade_structured -- the same pixel program as lock-step structured code.
//
// All lanes of a wave walk the program together; the two shaded points of a pixel (the
// primary hit, and the hit of its reflection ray: `ctx` 0 and 1) go through ONE loop body,
// so every kind of ray has exactly one march loop in the code: hit rays (primary /
// reflection), soft shadow rays, probe visibility rays, AO taps.  Five SDF sites instead of
// the fifteen of a literal transcription -- the kernel stays inside the instruction cache
// and well under 128 VGPRs.
// =============================================================================================

// -DMDH_PHASES: wall cycles (s_memtime) per wave spent in each region of the pixel program, summed over waves
#ifdef MDH_PHASES
__device__ unsigned long long g_phase[16];
#define MDH_PH_SLOT 19 // one extra park slot: 32 u64 accumulators per wave
MDH_DEV unsigned long long *ph_acc_(float *pk) { return (unsigned long long *)(pk + MDH_PH_SLOT * MDH_BLOCK + (threadIdx.x & ~63)); }
MDH_DEV void ph_add_(float *pk, int id, unsigned long long dt)
{
   if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) ph_acc_(pk)[id] += dt;
}
#define PH_T0(v) unsigned long long v = __builtin_amdgcn_s_memtime()
#define PH_ADD(v, id) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph_add_(pk, id, n_ - v); v = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PH_T0(v) do { } while (0)
#define PH_ADD(v, id) do { } while (0)
#endif
#ifdef MDH_TIMELINE
// wave timeline: [wave] = {start, end} in s_memrealtime ticks (100 MHz), {evals, hw id}
#define MDH_DIAG_WAVES 65536
__device__ unsigned long long g_wave_t[MDH_DIAG_WAVES * 3];
struct DiagWaveTimer {
   long wave;
   unsigned long long t0;
   __device__ DiagWaveTimer(long w) : wave(w), t0(__builtin_amdgcn_s_memrealtime()) {}
   __device__ ~DiagWaveTimer()
   {
      if ((threadIdx.x & 63) == 0 && wave < MDH_DIAG_WAVES) {
         g_wave_t[3 * wave] = t0;
         g_wave_t[3 * wave + 1] = __builtin_amdgcn_s_memrealtime();
         unsigned hw;
         asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
         g_wave_t[3 * wave + 2] = hw;
      }
   }
};
#define MDH_DIAG_WAVE(w) DiagWaveTimer diag_wave_timer_(w)
#else
#define MDH_DIAG_WAVE(w) do { } while (0)
#endif

// raymarching.glsl:25-51: plain sphere trace until a hit or tmax; `steps` counts SDF evaluations
template <int PART> MDH_DEV bool march_plain(const KScene &sc, f3 o, f3 d, float tmax, float &t_out, int &steps)
{
   int n = 0;
   bool h = false;
   float total = 0.0f;
   // the first sphere and box stay in registers for the whole march (measured: +0.7 %; the same in the shadow,
   // visibility-queue and occlusion marches ±0, in the per-corner visibility march -1 %: register pressure)
   const SdfRegs regs = sdf_regs(sc);
   MDH_WORK(0);
   while (total < tmax) {
      MDH_DIAG_STEP(0);
      MDH_WORK(1);
      float dist = sdf<PART>(sc, o + d * total, regs);
      ++n;
      if (dist < MDH_EPS) { h = true; break; }
      total += dist;
   }
   t_out = total;
   steps = n;
   return h;
}

// Per-lane parking space in LDS behind the scene table (MDH_PARK_DWORDS floats per thread,
// [slot][thread] so that a wave's access is one conflict-free row).  What the first shaded
// point leaves behind for the final combine -- position, normal, view direction, direct light,
// irradiance -- is cold while the reflection's point is shaded: parking it takes 15 VGPRs
// out of the live set of the inner loops, where hipcc would otherwise spill them to scratch
// (HBM round trips inside the probe loop; measured 47 % of the wave's cycles waiting).
// slots: 0-2 P, 3-5 N, 6-8 view direction, 9-11 direct light, 12-14 irradiance, 15-17 the reflection's colour,
// 18 material id; the visibility queue (radiance pass) uses 12-15 for its entries, 16 and 17 for first step and result
// before the irradiance is parked
#define MDH_PARK_SPEC 15
#define MDH_PARK_MAT 18
// mode 2 (direct light and occlusion only: no probes, no reflection) parks P, N, view direction, direct light and the
// material id: 13 rows, so that eight of its workgroups fit a CU's LDS
#define MDH_PARK_MAT_DIRECT 12
#ifdef MDH_PHASES
#define MDH_DIRECT_PARK_ROWS 20 // (room for the accumulators of the diagnostic, MDH_PH_SLOT)
#else
#define MDH_DIRECT_PARK_ROWS 13
#endif
#ifndef MDH_QVIS_SHARED
#define MDH_QVIS_SHARED 0 // the radiance pass's visibility queue shared by the four wavefronts of a workgroup (queued_visibility_shared)
#endif
#if defined(MDH_PHASES) || MDH_QVIS_SHARED || defined(MDH_PARK_PAD) // (MDH_PARK_PAD: what one more row costs, measured by itself)
#define MDH_PARK_DWORDS 20 // (row 19: the diagnostic's accumulators / the shared queue's second list and counters)
#else
#define MDH_PARK_DWORDS 19
#endif
// screen pass only, during the FIRST point's corner loop: the rows of its irradiance (12-14, parked behind the loop) and
// of the reflection's colour (15-17, cleared behind the loop) are free and hold what the loop needs at every corner but
// must not keep in registers across its marches -- the position inside the cage (alpha) and the clamped octahedral
// texel of N
#define MDH_PARK_ALPHA 12
#define MDH_PARK_RIDN 15
#define MDH_SCR_PARK_ROWS MDH_PARK_DWORDS
MDH_DEV float *park_base(const KScene &sc) { return (float *)(s_tab + sc.table_f4 + sc.part_bits_f4); }
// A thread's own column of the park rows is addressed WITHOUT an address register: ds_write_addtid_b32 /
// ds_read_addtid_b32 take M0[15:0] + offset + 4 * lane (scripts/addtid_probe.hip checks that on the box), so a park
// access costs one LDS instruction per dword and one scalar -- the LDS byte address of the wave's 64 columns in row 0
// -- stays live instead of a per-lane address that the compiler kept in scratch.  M0 is saved and restored (it is
// compiler-reserved); `s_nop 0`: one wait state between a write of M0 and an LDS add-TID instruction.  The reads wait
// for their data inside the statement (the compiler does not count the LDS operations of an asm statement).
#ifndef MDH_PARK_ADDTID
#define MDH_PARK_ADDTID 1
#endif
MDH_DEV int park_wave_base(const float *pk)
{
   const unsigned lds = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)pk;
   return __builtin_amdgcn_readfirstlane((int)(lds + (threadIdx.x & ~63u) * 4u));
}
MDH_DEV int park_col(const float *pk, int wb) // (the generic form: thread index in the park rows)
{
   const unsigned lds = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)pk;
   return (int)((unsigned)wb - lds) / 4 + lane_index_fresh();
}
template <int SLOT> MDH_DEV void park_store3(float *pk, int wb, f3 v)
{
#if MDH_PARK_ADDTID
   unsigned keep;
   asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:%5\n\tds_write_addtid_b32 %2 offset:%6\n\t"
                "ds_write_addtid_b32 %3 offset:%7\n\ts_mov_b32 m0, %0"
                : "=&s"(keep) : "v"(v.x), "v"(v.y), "v"(v.z), "s"(wb), "i"(SLOT * MDH_BLOCK * 4), "i"((SLOT + 1) * MDH_BLOCK * 4), "i"((SLOT + 2) * MDH_BLOCK * 4) : "memory");
#else
   const int t = park_col(pk, wb);
   pk[(SLOT + 0) * MDH_BLOCK + t] = v.x;
   pk[(SLOT + 1) * MDH_BLOCK + t] = v.y;
   pk[(SLOT + 2) * MDH_BLOCK + t] = v.z;
#endif
}
template <int SLOT> MDH_DEV f3 park_load3(const float *pk, int wb)
{
#if MDH_PARK_ADDTID
   unsigned keep;
   f3 v;
   asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tds_read_addtid_b32 %1 offset:%5\n\tds_read_addtid_b32 %2 offset:%6\n\t"
                "ds_read_addtid_b32 %3 offset:%7\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %0"
                : "=&s"(keep), "=&v"(v.x), "=&v"(v.y), "=&v"(v.z) : "s"(wb), "i"(SLOT * MDH_BLOCK * 4), "i"((SLOT + 1) * MDH_BLOCK * 4), "i"((SLOT + 2) * MDH_BLOCK * 4) : "memory");
   return v;
#else
   const int t = park_col(pk, wb);
   return F3(pk[(SLOT + 0) * MDH_BLOCK + t], pk[(SLOT + 1) * MDH_BLOCK + t], pk[(SLOT + 2) * MDH_BLOCK + t]);
#endif
}
template <int SLOT> MDH_DEV f2 park_load2(const float *pk, int wb)
{
#if MDH_PARK_ADDTID
   unsigned keep;
   f2 v;
   asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tds_read_addtid_b32 %1 offset:%4\n\tds_read_addtid_b32 %2 offset:%5\n\t"
                "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %0"
                : "=&s"(keep), "=&v"(v.x), "=&v"(v.y) : "s"(wb), "i"(SLOT * MDH_BLOCK * 4), "i"((SLOT + 1) * MDH_BLOCK * 4) : "memory");
   return v;
#else
   const int t = park_col(pk, wb);
   return F2(pk[(SLOT + 0) * MDH_BLOCK + t], pk[(SLOT + 1) * MDH_BLOCK + t]);
#endif
}
template <int SLOT> MDH_DEV void park_store1(float *pk, int wb, float v)
{
#if MDH_PARK_ADDTID
   unsigned keep;
   asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:%3\n\ts_mov_b32 m0, %0"
                : "=&s"(keep) : "v"(v), "s"(wb), "i"(SLOT * MDH_BLOCK * 4) : "memory");
#else
   pk[SLOT * MDH_BLOCK + park_col(pk, wb)] = v;
#endif
}
template <int SLOT> MDH_DEV float park_load1(const float *pk, int wb)
{
#if MDH_PARK_ADDTID
   unsigned keep;
   float v;
   asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tds_read_addtid_b32 %1 offset:%3\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %0"
                : "=&s"(keep), "=&v"(v) : "s"(wb), "i"(SLOT * MDH_BLOCK * 4) : "memory");
   return v;
#else
   return pk[SLOT * MDH_BLOCK + park_col(pk, wb)];
#endif
}


// ---------------------------------------------------------------------------------------------
// queued_visibility -- the probe-visibility rays of the wave's 64 shaded points as ONE queue.
//
// In lock step a wave pays, for each of the 8 cage corners, the longest visibility ray among
// its lanes, while folded corners and rays that end at once leave lanes empty: the radiance
// pass ran its visibility marches at 15 of 64 lanes.  Here every lane first lists the rays its
// point really needs (corner i of lane l = entry l | i << 6, appended with a ballot prefix sum),
// then the wave marches the list with ray REPLACEMENT: a lane whose ray has ended takes the
// next entry, re-deriving origin, direction and length from the owner lane's parked P, N and
// first-step distance (bit for bit the arithmetic of the lock-step loop), and ORs the
// result into the owner's visibility word.  Results per ray are unchanged; only the schedule is.
//
// LDS: entries are u16 in park slots 12..15 of the wave (free until the irradiance is parked),
// slot 16 = first-step distance, slot 17 = visibility word.
// ---------------------------------------------------------------------------------------------
#ifndef MDH_CORNER_UNROLL
#define MDH_CORNER_UNROLL 8 // the cage-corner loop of the pixel program, unrolled: constant corner bits, no loop branch (measured 1, 2, 4, 8)
#endif
#ifndef MDH_QVIS_FAR_FIRST
#define MDH_QVIS_FAR_FIRST 1
#endif
#ifndef MDH_MATERIAL_PER_LIGHT
#define MDH_MATERIAL_PER_LIGHT 1
#endif
#ifndef MDH_QVIS_REDERIVE
#define MDH_QVIS_REDERIVE 1
#endif
#ifndef MDH_QVIS_REFILL
#define MDH_QVIS_REFILL 16 // idle lanes that trigger a refill
#endif
MDH_DEV unsigned short *qvis_entry(float *pk, int wbase, int j)
{
   return (unsigned short *)(pk + (12 + (j >> 7)) * MDH_BLOCK + wbase) + (j & 127);
}
template <int PART>
MDH_DEV int queued_visibility(const KScene &sc, const KProbes &pr, float *pk, f3 P, f3 N, i3 gp, int folded, float sd0)
{
   const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
   int *words = (int *)(pk + 17 * MDH_BLOCK);
   pk[16 * MDH_BLOCK + threadIdx.x] = sd0;
   words[threadIdx.x] = 0;
   int bits = 0, njobs = 0;
   // The queue is taken from its head: every lane lists its FARTHEST cage corner first (the far side of each axis: the
   // corner j ^ far for j = 0 .. 7 goes from the farthest to the nearest), so that the longest rays of the wave start first
   // and its last rays are short ones -- the queue drains with fewer idle lanes.  The order of the list decides nothing else.
   int far = 0;
#if MDH_QVIS_FAR_FIRST
   {
      const f3 lo = grid_to_world(pr, gp);
      far = ((P.x - lo.x) < (lo.x + pr.sx - P.x) ? 1 : 0) | ((P.y - lo.y) < (lo.y + pr.sy - P.y) ? 2 : 0) | ((P.z - lo.z) < (lo.z + pr.sz - P.z) ? 4 : 0);
   }
#endif
#pragma unroll 1
   for (int j = 0; j < 8; ++j) {
      const int i = j ^ far;
      bool need = false;
      if (!(i & folded)) {
         const f3 hvec = grid_to_world(pr, cage_probe(pr, gp, i)) - P;
         const float vmax = length(hvec) - MDH_MIN_STEP * 5.0f;
         // raycast_visibility (raymarching.glsl:39-56) whose first step is known: sd0 at t = 0
         if (!(0.0f < vmax)) bits |= 1 << i;     // the loop is never entered
         else if (sd0 < MDH_EPS) { }             // blocked at once
         else if (!(sd0 < vmax)) bits |= 1 << i; // the first step already passes the probe
         else need = true;
      }
      const unsigned long long m = __ballot(need);
      if (need) *qvis_entry(pk, wbase, njobs + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))) = (unsigned short)(lane | (i << 6));
      njobs += __popcll(m);
   }
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
   __builtin_amdgcn_wave_barrier();
   __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
   int head = 0, job = -1;
   float total = 0.0f, vmax = 0.0f;
   f3 o = F3(0.0f, 0.0f, 0.0f), d = F3(0.0f, 0.0f, 0.0f);
   // only the lanes whose point was hit are here: a wave with fewer of them than the refill threshold (rays that
   // left an open scene) must still start its queue
   const int refill = min((int)MDH_QVIS_REFILL, (int)__popcll(__ballot(true)));
   for (;;) {
      const unsigned long long idle = __ballot(job < 0);
      const int n_idle = __popcll(idle);
      if (head < njobs && n_idle >= refill) {
         if (job < 0) {
            const int my = head + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
            if (my < njobs) {
               const int e = *qvis_entry(pk, wbase, my);
               const int owner = wbase + (e & 63), corner = e >> 6;
               const f3 oP = F3(pk[0 * MDH_BLOCK + owner], pk[1 * MDH_BLOCK + owner], pk[2 * MDH_BLOCK + owner]);
               const f3 oN = F3(pk[3 * MDH_BLOCK + owner], pk[4 * MDH_BLOCK + owner], pk[5 * MDH_BLOCK + owner]);
               const f3 hvec = grid_to_world(pr, cage_probe(pr, world_to_grid(pr, oP), corner)) - oP;
               const float dist = length(hvec);
               d = sdiv3(hvec, dist);
               vmax = dist - MDH_MIN_STEP * 5.0f;
               o = oP + (oN * MDH_MIN_STEP) * 5.0f;
               total = pk[16 * MDH_BLOCK + owner];
               job = e;
            }
         }
         head += n_idle;
      }
      if (__ballot(job >= 0) == 0ull) break;
      if (job >= 0) {
#ifdef MDH_DIAG_DRAIN
         if (head < njobs) { MDH_DIAG_STEP(3); } else { MDH_DIAG_STEP(4); }
#else
         MDH_DIAG_STEP(3);
#endif
         MDH_WORK(1);
         const float sd = sdf<PART>(sc, o + d * total);
         if (sd < MDH_EPS) job = -1; // blocked: the bit stays 0
         else {
            total += sd;
            if (!(total < vmax)) {
               atomicOr(&words[wbase + (job & 63)], 1 << (job >> 6));
               job = -1;
            }
         }
      }
   }
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
   __builtin_amdgcn_wave_barrier();
   __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
   return bits | words[threadIdx.x];
}


// ---------------------------------------------------------------------------------------------
// queued_visibility_shared (MDH_QVIS_SHARED) -- the same queue with its END shared by the workgroup.
//
// What the per-wavefront queue loses is its end: once its list is handed out a wavefront marches on until its longest
// ray ends, with 12.6 of 64 lanes alive on average (98 k of the radiance pass's 274 k wave-evaluations).  Here every
// wavefront still lists its own jobs in its own rows and draws from them, but the lists' heads are LDS counters, so a
// wavefront whose list is empty draws from its neighbours' (a STOLEN job), and a wavefront down to a few rays while
// another one still marches DONATES them -- resumable jobs (entry + the distance marched so far) in a second, small
// list -- and waits for its results instead of marching at a fraction of its width.  The last marching wavefront
// never donates (a counter of marching wavefronts decides, atomically), so every job is finished by somebody.  Nothing
// here waits at a barrier (a wavefront that hits nothing never comes here), and every wait is bounded: if a bound were
// ever reached the results would be wrong and the parity tests would say so, but no wavefront spins for ever.
//
// The LDS unit serves the DS instructions of a workgroup's wavefronts one after the other, each wavefront's in program
// order: a wavefront that sees a counter move sees everything its writer stored before.  The fences below are therefore
// wavefront-scoped (they order the compiler, not the hardware), and the counters a march step looks at are read one
// step AHEAD (their latency hides behind the distance evaluation; acting on a stale value is harmless: a draw that comes
// too late gets nothing).
//
// LDS of the workgroup (behind the scene table): rows 12-15 = the four wavefronts' lists (u16 entries lane | corner << 6),
// row 16 = first-step distances, row 17 = result words, row 19 = the counters and list 2:
#define QS_ROW 19
#define QS_LIST 0      // [4] head | njobs << 16 of each wavefront's list (a draw adds to the head, whatever is left)
#define QS_Q2 4        // head | tail << 16 of list 2 (draws by compare-and-swap: its tail grows)
#define QS_MARCHING 5  // wavefronts in the march loop
#define QS_PENDING 8   // [4] per wavefront: its jobs in other wavefronts' hands
#define QS_E2 64       // [64] entries of list 2 (-1 = not written yet)
#define QS_T2 128      // [64] distances marched so far
#ifndef QS_DONATE_MAX
#define QS_DONATE_MAX 16
#endif
#ifndef QS_BISECT
#define QS_BISECT 0 // (1: the per-wavefront queue inside the shared build; 2: no stealing)
#endif
static_assert(QS_DONATE_MAX * 4 <= 64, "list 2 holds one donation of every wavefront");
#define QS_SPIN_MAX (1 << 16) // (a few milliseconds: a legitimate wait is some tens of march steps)
#define QS_FOREIGN 0x800      // job bit: counted in QS_PENDING of its owner (stolen or donated)
MDH_DEV void qvis_shared_init(const KScene &sc) // all threads, before the kernel's first barrier
{
   int *ctl = (int *)(park_base(sc) + QS_ROW * MDH_BLOCK);
   ctl[threadIdx.x] = threadIdx.x >= QS_E2 && threadIdx.x < QS_T2 ? -1 : 0;
}
template <int PART>
MDH_DEV int queued_visibility_shared(const KScene &sc, const KProbes &pr, float *pk, f3 P, f3 N, i3 gp, int folded, float sd0)
{
   const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   const bool leader = lane == (int)__ffsll((long long)__ballot(true)) - 1; // (only the lanes whose point was hit are here)
   int *words = (int *)(pk + 17 * MDH_BLOCK);
   int *ctl = (int *)(pk + QS_ROW * MDH_BLOCK);
   // (reads of what other wavefronts write: relaxed atomic loads through an LDS-typed pointer -- a volatile access
   // through the generic pointer is a system-scope FLAT load with a full wait behind it, 40 % of the pass when the
   // counters were read that way)
   typedef __attribute__((address_space(3))) int *LdsInt;
   const LdsInt lctl = (LdsInt)ctl;
#define QS_LD(k) __hip_atomic_load(lctl + (k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define QS_ST(k, v) __hip_atomic_store(lctl + (k), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
   pk[16 * MDH_BLOCK + threadIdx.x] = sd0;
   words[threadIdx.x] = 0;
   int bits = 0, njobs = 0;
   int far = 0;
#if MDH_QVIS_FAR_FIRST
   {
      const f3 lo = grid_to_world(pr, gp);
      far = ((P.x - lo.x) < (lo.x + pr.sx - P.x) ? 1 : 0) | ((P.y - lo.y) < (lo.y + pr.sy - P.y) ? 2 : 0) | ((P.z - lo.z) < (lo.z + pr.sz - P.z) ? 4 : 0);
   }
#endif
#pragma unroll 1
   for (int j = 0; j < 8; ++j) { // (as in queued_visibility)
      const int i = j ^ far;
      bool need = false;
      if (!(i & folded)) {
         const f3 hvec = grid_to_world(pr, cage_probe(pr, gp, i)) - P;
         const float vmax = length(hvec) - MDH_MIN_STEP * 5.0f;
         if (!(0.0f < vmax)) bits |= 1 << i;
         else if (sd0 < MDH_EPS) { }
         else if (!(sd0 < vmax)) bits |= 1 << i;
         else need = true;
      }
      const unsigned long long m = __ballot(need);
      if (need) *qvis_entry(pk, wbase, njobs + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))) = (unsigned short)(lane | (i << 6));
      njobs += __popcll(m);
   }
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
   if (leader) { // the list is open: its length, and one more marching wavefront
      atomicAdd(&ctl[QS_MARCHING], 1);
      atomicAdd(&ctl[QS_LIST + wv], njobs << 16);
   }
   int job = -1;
   float total = 0.0f, vmax = 0.0f;
   f3 o = F3(0.0f, 0.0f, 0.0f), d = F3(0.0f, 0.0f, 0.0f);
   bool donated = false;
   const int n_here = (int)__popcll(__ballot(true));
   const int refill = min((int)MDH_QVIS_REFILL, n_here);
   int own_left = njobs; // (an upper bound: thieves take from it too)
   // the counters as they were one step ago
   int c_list[4] = {0, 0, 0, 0}, c_q2 = 0, c_marching = 0;
#define QS_OPEN(w) ((c_list[w] & 0xffff) < ((unsigned)c_list[w] >> 16))
#define QS_READ_COUNTERS() do { c_list[0] = QS_LD(QS_LIST + 0); c_list[1] = QS_LD(QS_LIST + 1); c_list[2] = QS_LD(QS_LIST + 2); c_list[3] = QS_LD(QS_LIST + 3); \
                                c_q2 = QS_LD(QS_Q2); c_marching = QS_LD(QS_MARCHING); } while (0)
   for (;;) {
      const unsigned long long idle = __ballot(job < 0);
      const int n_idle = __popcll(idle);
      const int n_active = n_here - n_idle;
      if (n_idle >= refill) {
         // where to draw from: the own list, a neighbour's, the donated rays
         int from = -1;
         if (own_left > 0) from = wv;
         else {
            int any = __builtin_amdgcn_readfirstlane((QS_OPEN(0) ? 1 : 0) | (QS_OPEN(1) ? 2 : 0) | (QS_OPEN(2) ? 4 : 0) | (QS_OPEN(3) ? 8 : 0) |
                                                           ((c_q2 & 0xffff) < ((unsigned)c_q2 >> 16) ? 16 : 0));
#if QS_BISECT == 2
            any = 0; // (no stealing)
#endif
            if (any) from = __ffs(any) - 1;
         }
         if (from >= 0 && from < 4) {
            const bool foreign = from != wv;
            int old = 0;
            if (leader) {
               if (foreign) atomicAdd(&ctl[QS_PENDING + from], n_idle); // (before the draw: its owner may be about to look)
               old = atomicAdd(&ctl[QS_LIST + from], n_idle);
            }
            old = __builtin_amdgcn_readfirstlane(old);
            const int h = old & 0xffff, got = max(0, min(n_idle, (int)((unsigned)old >> 16) - h));
            if (foreign && leader && got < n_idle) atomicSub(&ctl[QS_PENDING + from], n_idle - got);
            if (!foreign) own_left = (int)((unsigned)old >> 16) - h - got;
            if (job < 0) {
               const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
               if (rank < got) {
                  const int e = *qvis_entry(pk, from * 64, h + rank);
                  const int owner = from * 64 + (e & 63), corner = e >> 6;
                  const f3 oP = F3(pk[0 * MDH_BLOCK + owner], pk[1 * MDH_BLOCK + owner], pk[2 * MDH_BLOCK + owner]);
                  const f3 oN = F3(pk[3 * MDH_BLOCK + owner], pk[4 * MDH_BLOCK + owner], pk[5 * MDH_BLOCK + owner]);
                  const f3 hvec = grid_to_world(pr, cage_probe(pr, world_to_grid(pr, oP), corner)) - oP;
                  const float dist = length(hvec);
                  d = sdiv3(hvec, dist);
                  vmax = dist - MDH_MIN_STEP * 5.0f;
                  o = oP + (oN * MDH_MIN_STEP) * 5.0f;
                  total = pk[16 * MDH_BLOCK + owner];
                  job = e | (from << 9) | (foreign ? QS_FOREIGN : 0);
               }
            }
            if (foreign || got == 0) { QS_READ_COUNTERS(); }
            continue; // (the lanes have changed: count them again)
         } else if (from == 4) { // the donated rays: their list grows, so the draw is a compare-and-swap
            int h = 0, got = 0;
            if (leader)
               for (int tries = 0; tries < 64; ++tries) {
                  const int q = QS_LD(QS_Q2);
                  const int want = min(n_idle, (int)((unsigned)q >> 16) - (q & 0xffff));
                  if (want <= 0) break;
                  if (atomicCAS(&ctl[QS_Q2], q, q + want) == q) { h = q & 0xffff; got = want; break; }
               }
            h = __builtin_amdgcn_readfirstlane(h); got = __builtin_amdgcn_readfirstlane(got);
            if (job < 0) {
               const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
               if (rank < got) {
                  int e = -1;
                  for (int spin = 0; spin < QS_SPIN_MAX; ++spin) { e = QS_LD(QS_E2 + ((h + rank) & 63)); if (e >= 0) break; __builtin_amdgcn_s_sleep(1); }
                  if (e >= 0) {
                     const int owner = ((e >> 9) & 3) * 64 + (e & 63), corner = (e >> 6) & 7;
                     const f3 oP = F3(pk[0 * MDH_BLOCK + owner], pk[1 * MDH_BLOCK + owner], pk[2 * MDH_BLOCK + owner]);
                     const f3 oN = F3(pk[3 * MDH_BLOCK + owner], pk[4 * MDH_BLOCK + owner], pk[5 * MDH_BLOCK + owner]);
                     const f3 hvec = grid_to_world(pr, cage_probe(pr, world_to_grid(pr, oP), corner)) - oP;
                     const float dist = length(hvec);
                     d = hvec / dist;
                     vmax = dist - MDH_MIN_STEP * 5.0f;
                     o = oP + (oN * MDH_MIN_STEP) * 5.0f;
                     total = __int_as_float(QS_LD(QS_T2 + ((h + rank) & 63)));
                     job = e;
                  }
               }
            }
            QS_READ_COUNTERS();
            continue;
         } else if (n_active == 0) {
            // nothing runs and, one step ago, nothing was left to draw: look now
            QS_READ_COUNTERS();
            int any = __builtin_amdgcn_readfirstlane((QS_OPEN(0) || QS_OPEN(1) || QS_OPEN(2) || QS_OPEN(3) || (c_q2 & 0xffff) < ((unsigned)c_q2 >> 16)) ? 1 : 0);
#if QS_BISECT == 2
            any = 0;
#endif
            if (any) continue;
            // leave the march -- unless this is its last wavefront and a donor has appended meanwhile (donors append before they leave)
            int was = 0;
            if (leader) was = atomicSub(&ctl[QS_MARCHING], 1);
            was = __builtin_amdgcn_readfirstlane(was);
            if (was > 1) break;
            const int q = __builtin_amdgcn_readfirstlane(QS_LD(QS_Q2));
            if (!((q & 0xffff) < (int)((unsigned)q >> 16))) break;
            if (leader) atomicAdd(&ctl[QS_MARCHING], 1);
            QS_READ_COUNTERS();
            continue;
         }
      }
      if (!donated && own_left <= 0 && n_active <= QS_DONATE_MAX && __builtin_amdgcn_readfirstlane(c_marching) > 1) {
         // hand the last rays to the wavefronts that still march
         const unsigned long long act = __ballot(job >= 0);
         int b2 = 0;
         if (leader) b2 = atomicAdd(&ctl[QS_Q2], n_active << 16);
         b2 = (int)((unsigned)__builtin_amdgcn_readfirstlane(b2) >> 16);
         if (job >= 0) {
            const int r2 = (b2 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u))) & 63;
            if (!(job & QS_FOREIGN)) atomicAdd(&ctl[QS_PENDING + ((job >> 9) & 3)], 1);
            QS_ST(QS_T2 + r2, __float_as_int(total));
            QS_ST(QS_E2 + r2, job | QS_FOREIGN);
            job = -1;
         }
         donated = true;
         int was = 0;
         if (leader) was = atomicSub(&ctl[QS_MARCHING], 1);
         was = __builtin_amdgcn_readfirstlane(was);
         if (was > 1) break;                           // somebody marches on: wait for the results below
         if (leader) atomicAdd(&ctl[QS_MARCHING], 1);  // the others left meanwhile: take the rays back from list 2
         QS_READ_COUNTERS();
         continue;
      }
      QS_READ_COUNTERS(); // (for the next step: the reads complete behind the distance evaluation)
      if (job >= 0) {
         MDH_DIAG_STEP(3);
         MDH_WORK(1);
         const float sd = sdf<PART>(sc, o + d * total);
         bool done = false;
         if (sd < MDH_EPS) done = true; // blocked: the bit stays 0
         else {
            total += sd;
            if (!(total < vmax)) { atomicOr(&words[((job >> 9) & 3) * 64 + (job & 63)], 1 << ((job >> 6) & 7)); done = true; }
         }
         if (done) {
            if (job & QS_FOREIGN) atomicSub(&ctl[QS_PENDING + ((job >> 9) & 3)], 1); // (behind the result, in this wavefront's DS order)
            job = -1;
         }
      }
   }
   // this wavefront's own jobs may still be in other hands
   for (int spin = 0; spin < QS_SPIN_MAX; ++spin) {
      if (__builtin_amdgcn_readfirstlane(QS_LD(QS_PENDING + wv)) <= 0) break;
      __builtin_amdgcn_s_sleep(2);
   }
   __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
   return bits | __hip_atomic_load((LdsInt)words + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#undef QS_OPEN
#undef QS_READ_COUNTERS
#undef QS_LD
#undef QS_ST
}

// sample_radiance_with_specular (render_probes.glsl:71-136, M_COMPUTE_INDIRECT_SPECULAR == 1) from the reflection's hit
// position on: the eight cage probes of the FIRST point (parked: slots 0-2, its material id in MDH_PARK_MAT) light
// that position, each weighted by a soft shadow ray from it towards the probe (k = 0.5, raymarching.glsl:4-23 with
// its min_dist) and the trilinear factor of the first point.  textureLod is level 0 (one level): lod
// only narrows the clamp of the tap; a total weight of 0 gives 0.
template <int PART>
MDH_DEV f3 radiance_with_specular(const KScene &sc, const KProbes &pr, const float *pk, int wb, f3 spec_pos, int u8_tab)
{
   const f3 pos = park_load3<0>(pk, wb);
   const float roughness = tab_float((sc.mat_slot + 2 * __float_as_int(park_load1<MDH_PARK_MAT>(pk, wb)) + 1) * 4);
   const f3 pos_to_spec_pos = spec_pos - pos;
   const i3 gp = world_to_grid(pr, pos);
   const f3 alpha = pos / F3(pr.sx, pr.sy, pr.sz) - F3((float)gp.x, (float)gp.y, (float)gp.z);
   const float lod = mix_(0.0f, (float)pr.rad_lods, roughness * 2.0f);
   const int new_res = pr.rres / (int)(lod + 1.0f);
   const float rmin = 0.5f / (float)new_res, rmax = 1.0f - rmin;
   float total_weight = 0.0f;
   f3 radiance = F3(0.0f, 0.0f, 0.0f);
#pragma unroll 1
   for (int i = 0; i < 8; ++i) {
      const i3 q = cage_probe(pr, gp, i);
      f3 pts = (pos - grid_to_world(pr, q)) + pos_to_spec_pos; // probe_to_spec
      const float distance = length(pts);
      pts = pts / distance;
      float res = 1.0f, prev = 1e20f;
      bool blocked = false;
      const float tmax = distance - MDH_MIN_STEP * 5.0f;
      MDH_WORK(0);
      for (float total = MDH_MIN_STEP * 5.0f; total < tmax;) { // softshadows (spec_pos, -probe_to_spec, .., 0.5)
         MDH_WORK(1);
         const float dist = sdf<PART>(sc, spec_pos + (-pts) * total);
         if (dist < MDH_EPS) { blocked = true; break; }
         const float y = dist * dist / (2.0f * prev);
         const float d = sqrt_(dist * dist - y * y);
         res = min_(res, 0.5f * d / max_(0.0f, total - y));
         prev = dist;
         total += dist;
      }
      float weight = max_(blocked ? 0.0f : res, 0.001f);
      const f3 tri = F3(mix_(1.0f - alpha.x, alpha.x, (float)(i & 1)), mix_(1.0f - alpha.y, alpha.y, (float)((i >> 1) & 1)),
                        mix_(1.0f - alpha.z, alpha.z, (float)((i >> 2) & 1)));
      weight *= tri.x * tri.y * tri.z;
      const f2 base = probe_id_to_coord(pr, grid_to_probe_id(pr, q));
      f2 rid = ray_dir_to_ray_id(pts);
      rid = F2(clamp_(rid.x, rmin, rmax), clamp_(rid.y, rmin, rmax));
      const f3 tx = pr.rad_mips ? radiance_lod_sample(pr, lod, base.x + div_pcx(pr, rid.x), base.y + div_pcy(pr, rid.y), u8_tab) // (MDH_OPT_RADIANCE_MIPS)
                                : atlas_sample(pr.rad, pr.fmt, pr.pcx, pr.pcy, pr.rres, pr.rshift, pr.rad_w, pr.rad_h, base.x + div_pcx(pr, rid.x), base.y + div_pcy(pr, rid.y), u8_tab, pr.m_rres);
      radiance = radiance + tx * weight;
      total_weight += weight;
   }
   if (total_weight == 0.0f) return F3(0.0f, 0.0f, 0.0f);
   return radiance / total_weight;
}

// The probe settings again, from the kernel argument segment, at the point of use.  Both march kernels take
// (KScene, KProbes, ...): held in scalar registers from the kernel's entry, the 48 dwords of KProbes stay live across
// every march loop of the pixel program, and what does not fit is copied to and from VGPR lanes with a vector
// instruction per use (the screen kernel had a thousand such copies, a third of them in inner loops).  Re-read where a
// block of probe code begins -- a few scalar loads through a pointer the compiler cannot trace, served by the scalar
// cache -- they are live for that block only.
#ifndef MDH_PROBES_FRESH
#define MDH_PROBES_FRESH 1
#endif
MDH_DEV KProbes probes_fresh(const KProbes &pr)
{
#if MDH_PROBES_FRESH
   struct KHead { KScene sc; KProbes pr; };
   typedef const KHead __attribute__((address_space(4))) *KHeadPtr;
   KHeadPtr ka = (KHeadPtr)__builtin_amdgcn_kernarg_segment_ptr();
   asm volatile("" : "+s"(ka));
   typedef const int __attribute__((address_space(4))) *IntPtr;
   IntPtr src = (IntPtr)&ka->pr;
   KProbes r;
   int *dst = (int *)&r;
#pragma unroll
   for (int q = 0; q < (int)(sizeof(KProbes) / 4); ++q) dst[q] = src[q];
   return r;
#else
   return pr;
#endif
}

// SPEC: 0 = no second point (the radiance pass), 1 = the reflection as the reference's renderer fixes it
// (M_COMPUTE_INDIRECT_SPECULAR = 2, or none), 2 = the kernel variant that holds the two other bodies of
// render_probes.glsl:264-272 (cfg.spec_mode 1 or 3; MDH_OPT_INDIRECT_SPECULAR)
// PHASE (the radiance pass as two kernels, an experiment: MDH_OPT... MADARCH_HIP_RAD_SPLIT): 0 = the whole program; 1 = up to the
// direct light of the hit point, which goes to `rec` (RadRecord) instead of being shaded further; 2 = from such a record on
struct RadRecord { float4 p_pm, n_sd0, lo; }; // {P, material id or -1: no hit}, {N, first-step distance}, {direct light, -}
template <int PART, int MODE, int SPEC, bool QVIS, int PHASE = 0>
MDH_DEV f3 shade_structured(const KScene &sc, const KProbes &pr, const MachineCfg cfg, bool lane_valid, f3 from, f3 dir_in,
                            PrimaryHit &ph, bool &hit, f3 &pos_out, RadRecord *rec = nullptr)
{
   constexpr bool P2 = (PART & MDH_PF_POW2) != 0;
   constexpr bool REFLECT = SPEC != 0 && MODE == 0; // (modes 1 and 2 never shade a second point: no loop, and nothing kept for one)
   // The cage-corner loop unrolled where the registers allow it (the brute-force screen kernel of the reference's fixed mode:
   // 96 VGPRs with and without; the partition variants and the one for the optional specular modes would spill 80 - 200 bytes
   // per lane, the radiance kernel gains nothing): constant corner bits, no loop branch, the x-twin's terms at hand.
   constexpr int CORNER_UNROLL = ((PART & MDH_PF_PART) || SPEC != 1 || QVIS) ? 1 : MDH_CORNER_UNROLL;
   constexpr int PARK_MAT = MODE == 2 ? MDH_PARK_MAT_DIRECT : MDH_PARK_MAT; // (no probe rows in mode 2: MDH_DIRECT_PARK_ROWS)
   // the irradiance tap of a cage corner issued before its visibility march (its loads land during the march) -- not in the
   // space-partition variants, whose march needs the registers (MDH_TAP_EARLY_PART)
   constexpr bool TAP_EARLY = MDH_TAP_EARLY != 0 && (MDH_TAP_EARLY_PART != 0 || !(PART & MDH_PF_PART));
   // the x-twin's probe term kept for the corner behind it (item 6 of "Exact work elimination") -- not in the space-partition
   // variants, where its four registers across the visibility march are scratch memory (MDH_TWIN_PREV_PART)
   constexpr bool TWIN_PREV = MDH_TWIN_PREV != 0 && (MDH_TWIN_PREV_PART != 0 || !(PART & MDH_PF_PART));
   // the space-partition variants of the screen pass: the shaded point and its normal come back from park rows behind every
   // corner's visibility march (the second point's wait in the rows of its colour and in three rows of their own) and the probe's grid position is derived again there:
   // the lookups of that march need the registers -- kept live across it, these values went to scratch memory eight times per
   // shaded point (300 MB of spill stores per 1080p launch at six wavefronts per SIMD)
   constexpr bool PARK_VD = MDH_PART_PARK_VD != 0 && (PART & MDH_PF_PART) != 0 && SPEC == 1 && MODE == 0 && CORNER_UNROLL == 1;
   // the second point goes through the whole of pixel_color_probes' lighting (compute_indirect_specular) ...
   const bool full2 = SPEC == 2 && cfg.spec_mode == 3;
   // ... or is only a position that the cage probes of the FIRST point light (sample_radiance_with_specular)
   const bool cage1 = SPEC == 2 && cfg.spec_mode == 1;
   const int u8_tab = sc.u8_slot * 4;
   float *pk = park_base(sc);
   const int wb = park_wave_base(pk);
   hit = false;
   ph.index = -1; ph.t = 0.0f; ph.steps = 0;
   // (the reflection's colour and the primary hit's material id wait in LDS for the combine, like P, N and the light;
   //  the colour's rows are cleared behind the first point's corner loop, which uses them meanwhile)
   bool shaded = false; // the primary ray hit and the full shading ran
#if MDH_QVIS_SHARED
   f3 irr_keep = F3(0.0f, 0.0f, 0.0f);
#endif
   // the ray that finds the next point to shade
   f3 ro = from, rd = dir_in;
   bool active = lane_valid;
   park_store3<6>(pk, wb, dir_in);
#pragma unroll 1
   for (int ctx = 0; ctx < (REFLECT ? 2 : 1); ++ctx) {
      if (active) {
         float t;
         int steps;
         PH_T0(pt);
         RadRecord rr; // (PHASE 2: what phase 1 left of this ray)
         if (PHASE == 2) { rr = *rec; t = 0.0f; steps = 0; }
         const bool h = PHASE == 2 ? __float_as_int(rr.p_pm.w) >= 0 : march_plain<PART>(sc, ro, rd, sc.max_dist, t, steps);
         PH_ADD(pt, 0);
         if (ctx == 0) { hit = h; ph.steps = steps; }
         active = false; // a miss ends the chain (ctx 1: specular_col stays 0, render_probes.glsl:142-144)
         if (PHASE == 1 && !h) { rec->p_pm = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1)); return F3(0.0f, 0.0f, 0.0f); }
         if (SPEC == 2 && ctx == 1 && full2 && !h) { // ... or is the sky seen along the reflection, render_probes.glsl:216-218
            const float s = rd.y * 0.7f;
            park_store3<MDH_PARK_SPEC>(pk, wb, F3(0.30f - s, 0.36f - s, 0.60f - s));
         }
         if (SPEC == 2 && ctx == 1 && cage1) {
            if (h) park_store3<MDH_PARK_SPEC>(pk, wb, radiance_with_specular<PART>(sc, pr, pk, wb, ro + rd * t, u8_tab));
         } else
         if (h) {
            f3 P = ro + rd * t;
            int index = -1;
            f3 N;
            int pm;
            if (PHASE == 2) { P = xyz(rr.p_pm); N = xyz(rr.n_sd0); pm = __float_as_int(rr.p_pm.w); }
            else {
            MDH_WORK(3);
            (void)sdf_info<PART>(sc, P, index);
            primitive_info<(PART & MDH_PF_CUSTOM) != 0>(sc, index, P, N, pm);
            }
            if (ctx == 0) {
               park_store1<PARK_MAT>(pk, wb, __int_as_float(pm));
               ph.index = index; ph.t = t;
               park_store3<0>(pk, wb, P);
               park_store3<3>(pk, wb, N);
            } else if (PARK_VD) { // (the second point's P in the rows of its colour -- written behind its corner loop -- and its N in rows of their own)
               park_store3<MDH_PARK_SPEC>(pk, wb, P);
               park_store3<MDH_PARK_VD_ROW>(pk, wb, N);
            }
            PH_ADD(pt, 1);
            if (MODE != 1) {
               const f3 from_off = P + (N * MDH_MIN_STEP) * 5.0f;
               // Every shadow and probe-visibility ray of this point starts AT from_off with t = 0, so
               // their first SDF evaluation is at the same position (from_off + dir * 0): it is done
               // once here and each ray replays its first iteration with this value.
               const float sd0 = PHASE == 2 ? rr.n_sd0.w : (MDH_SHARE_FIRST_STEP ? sdf<PART>(sc, from_off) : 0.0f);
               // ---- compute_direct_lighting (lighting.glsl:1-40) at P, seen along rd
               f3 Lo = PHASE == 2 ? xyz(rr.lo) : F3(0.0f, 0.0f, 0.0f);
               if (PHASE != 2) {
#pragma unroll 1
                  for (int li = 0; li < sc.total_lights; ++li) {
                     // (the material is read again for every light, through an id the compiler cannot look through: read
                     //  once in front of the loop, what the BRDF derives from it -- F0, 1 - F0, k -- is hoisted and then
                     //  lives across the shadow march: eight dwords of scratch memory per ray in the seven-wavefront builds)
#if MDH_MATERIAL_PER_LIGHT
                     int pm_l = pm;
                     asm volatile("" : "+v"(pm_l));
                     Material m = get_material(sc, pm_l);
#else
                     Material m = get_material(sc, pm);
#endif
                     if (ctx && !full2) m.albedo = F3(0.0f, 0.0f, 0.0f); // render_probes.glsl:202-205
                     f3 L;
                     float L_dist;
                     f3 radiance = sample_light<(PART & MDH_PF_CUSTOM) != 0>(sc, li, P, N, L, L_dist);
                     float NdotL = max_(dot(N, L), 0.0f);
                     f3 kD, kS;
                     cook_torrance(N, -rd, L, NdotL, m.albedo, m.metallic, m.roughness, kD, kS);
                     if (ctx == 0 && !cfg.direct_specular) kS = F3(0.0f, 0.0f, 0.0f);
                     const f3 contrib = ((sdiv3(kD * m.albedo, MDH_PI) + kS) * radiance) * NdotL;
                     float shadows = 0.0f;
                     PH_ADD(pt, 2);
                     // a light that contributes exactly nothing here (outside a spot's cone, black BRDF)
                     // needs no shadow ray: Lo + (+-0) * shadows = Lo for every shadows in [0, 1]
                     const bool lit = MDH_SKIP_NULL_RAYS ? (contrib.x != 0.0f || contrib.y != 0.0f || contrib.z != 0.0f) : true;
#ifdef MDH_ABL_NO_SHADOW
                     if (false) {
#else
                     if (NdotL > MDH_EPS && lit) { // softshadows, raymarching.glsl:4-23
#endif
                        float res = 1.0f, prev = 1e20f, total = 0.0f;
                        bool blocked = false;
                        bool first = MDH_SHARE_FIRST_STEP != 0;
                        MDH_WORK(0);
                        while (total < L_dist) {
                           MDH_DIAG_STEP(1 + ctx);
                           MDH_WORK(1);
                           if (SPEC == 0) ph.steps += 1 << 16; // (the radiance pass's sort key: shadow steps above the primary ones)
                           float dist = first ? sd0 : sdf<PART>(sc, from_off + L * total);
                           first = false;
                           if (dist < MDH_EPS) { blocked = true; break; }
                           float y = dist * dist / (2.0f * prev);
                           float d = sqrt_(dist * dist - y * y);
                           res = min_raw(res, 64.0f * d / max_(0.0f, total - y));
                           prev = dist;
                           total += dist;
                        }
                        shadows = blocked ? 0.0f : res;
                     }
                     Lo = Lo + contrib * shadows;
                     PH_ADD(pt, 3);
                  }
               }
               if (PHASE == 1) { // the record: the point, its normal and material, its first step and its direct light
                  rec->p_pm = make_float4(P.x, P.y, P.z, __int_as_float(pm));
                  rec->n_sd0 = make_float4(N.x, N.y, N.z, sd0);
                  rec->lo = make_float4(Lo.x, Lo.y, Lo.z, 0.0f);
                  return Lo;
               }
               if (ctx == 0) park_store3<9>(pk, wb, Lo); // = direct
               f3 specular_col = Lo;                      // ctx 1: + the radiance tap below (render_probes.glsl:197-206)
               if (MODE == 2) {
                  shaded = true;
               } else {
                  // ---- the 8 cage probes of P (render_probes.glsl:13-63 and :156-184)
                  const KProbes pg = probes_fresh(pr);
                  i3 gp = world_to_grid(pg, P);
                  // sample_irradiance at this point (the first point; the second one in mode 3), or the best cage probe of the second
                  const bool irrp = ctx == 0 || full2;
                  // irrp: acc = sum sqrt(irradiance) * w, accw = sum w; else: acc = best probe_to_spec, accw = best weight
                  f3 acc = irrp ? F3(0.0f, 0.0f, 0.0f) : F3(0.0f, 0.0f, 1.0f);
                  float accw = irrp ? 0.0f : -2.0f;
                  int best_q = 0; // x | y << 10 | z << 20
                  // Cage corners that the clamp to the grid folds onto an earlier corner (P outside the
                  // probe grid along an axis: all six walls of the example rooms are) name the SAME probe,
                  // hence the same visibility ray: its result is reused instead of marched again.
                  // bit a of `folded`: corners differing only in axis a coincide.
                  int folded = ((gp.x < 0 || gp.x >= pg.gx - 1) ? 1 : 0) | ((gp.y < 0 || gp.y >= pg.gy - 1) ? 2 : 0) |
                               ((gp.z < 0 || gp.z >= pg.gz - 1) ? 4 : 0);
                  int vis_bits = 0; // bit i: visibility of corner i
                  PH_ADD(pt, 2);
                  if (QVIS && ctx == 0) {
#if MDH_QVIS_SHARED && QS_BISECT == 1
                     vis_bits = queued_visibility<PART>(sc, pr, pk, P, N, gp, folded, sd0);
#elif MDH_QVIS_SHARED
                     vis_bits = queued_visibility_shared<PART>(sc, pr, pk, P, N, gp, folded, sd0);
#else
                     vis_bits = queued_visibility<PART>(sc, pr, pk, P, N, gp, folded, sd0);
#endif
#if MDH_QVIS_REDERIVE
                     // Nothing of the point stays in registers across the queue (it is the pass's longest loop, and the
                     // kernel is built for seven wavefronts per SIMD: what stayed live -- P, N, the cage cell, twenty dwords
                     // -- went to scratch and back, 45 MB per launch at BASELINE config 3).  P and N wait in their park rows,
                     // where the queue itself reads them; the cage cell and its folds are derived again: the same operations
                     // on the same values.
                     P = park_load3<0>(pk, wb);
                     N = park_load3<3>(pk, wb);
                     const KProbes pg2 = probes_fresh(pr);
                     gp = world_to_grid(pg2, P);
                     folded = ((gp.x < 0 || gp.x >= pg2.gx - 1) ? 1 : 0) | ((gp.y < 0 || gp.y >= pg2.gy - 1) ? 2 : 0) | ((gp.z < 0 || gp.z >= pg2.gz - 1) ? 4 : 0);
#endif
                  }
                  PH_ADD(pt, 5);
                  // (Reusing the whole irradiance term of a folded corner's twin through a small register ring was
                  // measured: the 16 extra VGPRs cost what the taps saved -- DESIGN.md, dropped experiments.)
                  // what does not depend on the corner, once (the compiler can no longer hoist it by itself: inside the loop the
                  // probe parameters are read afresh per corner): the position inside the cage and the clamped octahedral
                  // texel of N
                  f3 alpha = F3(0.0f, 0.0f, 0.0f);
                  f2 rid_n = F2(0.0f, 0.0f);
                  const bool parked = REFLECT && ctx == 0; // (the rows are free then: see MDH_PARK_ALPHA)
                  if (irrp) {
                     alpha = P / F3(pg.sx, pg.sy, pg.sz) - F3((float)gp.x, (float)gp.y, (float)gp.z);
                     rid_n = ray_dir_to_ray_id(N);
                     rid_n = F2(clamp_(rid_n.x, pg.irr_lo, pg.irr_hi), clamp_(rid_n.y, pg.irr_lo, pg.irr_hi));
                     if (parked) { // five registers less through the visibility marches
                        park_store3<MDH_PARK_ALPHA>(pk, wb, alpha);
                        park_store1<MDH_PARK_RIDN>(pk, wb, rid_n.x);
                        park_store1<MDH_PARK_RIDN + 1>(pk, wb, rid_n.y);
                     }
                  }
                  // A corner folded onto its twin along x follows it directly (i - 1): the twin's probe term -- direction,
                  // visibility, weight before the trilinear factor, irradiance tap -- is this corner's, value for value,
                  // and is still in registers.  (Twins along y and z are two and four corners back: a ring of terms in LDS for
                  // y was measured at +0.4 %, DESIGN.md.)
                  f3 s_keep = F3(0.0f, 0.0f, 0.0f);
                  float w_keep = 0.0f;
#pragma unroll CORNER_UNROLL
                  for (int i = 0; i < 8; ++i) {
                     // best probe: a folded corner has the weight of its twin, which is not strictly larger
                     if (MDH_REUSE_FOLDED && !irrp && (i & folded)) continue;
                     KProbes pq = probes_fresh(pr); // (live up to the visibility march, read again behind it)
                     f3 s_term = F3(0.0f, 0.0f, 0.0f); // sqrt(irradiance tap) of this corner's probe
                     float wpre = 0.0f;                // its weight before the trilinear factor
                     const i3 q = cage_probe(pq, gp, i);
                     const bool twin_prev = TWIN_PREV && irrp && (i & 1) && (folded & 1);
                     if (twin_prev) {
                        s_term = s_keep;
                        wpre = w_keep;
                        vis_bits |= ((vis_bits >> (i - 1)) & 1) << i;
                     } else {
                     const f3 pw = grid_to_world(pq, q);
                     const f3 hvec = irrp ? (pw - P) : (P - pw);
                     const float dist = length(hvec);
                     f3 vd = sdiv3(hvec, dist); // irrp: dir_to_probe, else: probe_to_spec
                     if (!irrp) vd = -vd; // the visibility ray always runs from the point to the probe
                     // the irradiance tap of this corner (render_probes.glsl:44-58) depends on the probe and
                     // N only: its texel loads go out now and land while the visibility ray is marched
                     AtlasTap tap;
                     if (TAP_EARLY && irrp) {
                        const f2 rid = parked ? park_load2<MDH_PARK_RIDN>(pk, wb) : rid_n;
                        const f2 base = probe_id_to_coord<P2>(pq, grid_to_probe_id<P2>(pq, q));
                        tap = atlas_tap_issue<P2>(pq.irr, pq.fmt, pq.pcx, pq.pcy, pq.ires, pq.ishift, pq.irr_w, pq.irr_h, base.x + div_pcx<P2>(pq, rid.x), base.y + div_pcy<P2>(pq, rid.y), pq.m_ires);
                     }
                     // raycast_visibility, raymarching.glsl:39-56
                     float vis = 1.0f, total = 0.0f;
                     float vmax = dist - MDH_MIN_STEP * 5.0f;
                     if (QVIS && ctx == 0) { // every distinct corner was traced by queued_visibility
                        vis = ((vis_bits >> (i & ~folded)) & 1) ? 1.0f : 0.0f;
                        vmax = 0.0f;
                     }
#if MDH_REUSE_FOLDED
                     if (i & folded) { // same probe as corner i & ~folded, already traced
                        vis = ((vis_bits >> (i & ~folded)) & 1) ? 1.0f : 0.0f;
                        vmax = 0.0f;
                     }
#endif
#if MDH_SKIP_NULL_RAYS
                     // ctx 1 keeps the probe with the largest dot(probe_to_spec, -N) * vis (strictly larger
                     // than the best so far).  With vis in {0, 1} the candidate is d or d * 0: once the best
                     // is >= 0, a probe with d <= best cannot win whatever its visibility.
                     if (!irrp && accw >= 0.0f && dot(-vd, -N) <= accw) vmax = 0.0f;
#endif
                     bool first = MDH_SHARE_FIRST_STEP != 0;
                     PH_ADD(pt, 4);
#ifdef MDH_ABL_NO_VIS
                     if (false)
#endif
                     if (!(QVIS && !REFLECT)) // (with the queue and no second point this loop is dead code)
                     MDH_WORK(0);
                     if (!(QVIS && !REFLECT))
                     while (total < vmax) {
                        MDH_DIAG_STEP(3 + ctx);
                        MDH_WORK(1);
                        float sd = first ? sd0 : sdf<PART>(sc, from_off + vd * total);
                        first = false;
                        if (sd < MDH_EPS) { vis = 0.0f; break; }
                        total += sd;
                     }
                     vis_bits |= (vis != 0.0f ? 1 : 0) << i;
                     pq = probes_fresh(pr);
                     i3 q2 = q;
                     if (PARK_VD) { // the point again, from its park rows (nothing in the march reads P or N: they need not live through it)
                        if (ctx == 0) { P = park_load3<0>(pk, wb); N = park_load3<3>(pk, wb); }
                        else { P = park_load3<MDH_PARK_SPEC>(pk, wb); N = park_load3<MDH_PARK_VD_ROW>(pk, wb); }
                        q2 = cage_probe(pq, gp, i); // (the same clamps of the same cell: the same probe)
                     }
                     PH_ADD(pt, 5);
                     if (irrp) { // render_probes.glsl:26-62
                        float angle = (dot(vd, N) + 1.0f) * 0.5f;
                        float weight = angle * angle + 0.2f;
                        weight *= vis;
                        const float crush = 0.2f;
                        if (weight < crush) weight *= weight * weight * (1.0f / (crush * crush));
                        wpre = weight;
#ifdef MDH_ABL_NO_TAPS
                        f3 tx = F3((float)q.x, (float)q.y, N.x);
#else
                        f3 tx;
                        if (TAP_EARLY) tx = atlas_tap_resolve(pq.irr, pq.fmt, tap, u8_tab);
                        else { // (the space-partition variants: the tap's eight registers do not live across the visibility march)
                           const f2 rid = parked ? park_load2<MDH_PARK_RIDN>(pk, wb) : rid_n;
                           const f2 base = probe_id_to_coord<P2>(pq, grid_to_probe_id<P2>(pq, q2));
                           tx = atlas_sample<P2>(pq.irr, pq.fmt, pq.pcx, pq.pcy, pq.ires, pq.ishift, pq.irr_w, pq.irr_h, base.x + div_pcx<P2>(pq, rid.x), base.y + div_pcy<P2>(pq, rid.y), u8_tab, pq.m_ires);
                        }
#endif
#if MDH_TAP_SQRT_CORE
                        // A bilinear tap of an RGBA8 atlas is 0 or at least 2^-56: its texels are k / 255, its weights products of
                        // two factors from {0} and [2^-24, 1] (fx = px - floor (px) with px = cx W - 0.5 a multiple of 2^-24 below 1 --
                        // the subtraction is exact there -- and of ulp (px) above; 1 - fx likewise), its terms are not negative.  No
                        // rescaling can apply: the square root's nine-instruction core, 7 instructions less x 3 channels x 8 corners.
                        if (pq.fmt == 0 && !MDH_HYBRID_NUMERICS) s_term = F3(sqrt_unscaled_(tx.x), sqrt_unscaled_(tx.y), sqrt_unscaled_(tx.z));
                        else
#endif
                        s_term = ssqrt3(tx);
                     } else { // render_probes.glsl:170-183; probe_to_spec = -vd
                        float weight = dot(-vd, -N);
                        weight *= vis;
                        if (weight > accw) { accw = weight; best_q = q2.x | (q2.y << 10) | (q2.z << 20); acc = -vd; }
                     }
                     }
                     if (irrp) { // render_probes.glsl:34-62: the trilinear factor and the sum
                        float weight = wpre;
                        if (parked) alpha = park_load3<MDH_PARK_ALPHA>(pk, wb);
                        f3 tri = F3(mix_(1.0f - alpha.x, alpha.x, (float)(i & 1)), mix_(1.0f - alpha.y, alpha.y, (float)((i >> 1) & 1)),
                                    mix_(1.0f - alpha.z, alpha.z, (float)((i >> 2) & 1)));
                        weight *= tri.x * tri.y * tri.z;
                        acc = acc + s_term * weight;
                        accw += weight;
                        if (TWIN_PREV) { s_keep = s_term; w_keep = wpre; }
                     }
                     PH_ADD(pt, 6);
                  }
                  if (SPEC == 2 && ctx == 1 && full2) { // render_probes.glsl:233-243: indirect (no specular of its own) + direct
                     f3 irr = F3(0.0f, 0.0f, 0.0f);
                     if (accw != 0.0f) { irr = sdiv3(acc, accw); irr = irr * irr; }
                     const Material m = get_material(sc, pm);
                     specular_col = compute_indirect_lighting(irr, F3(0.0f, 0.0f, 0.0f), -rd, N, reflect(rd, N), m.albedo, m.metallic, m.roughness) + specular_col;
                     park_store3<MDH_PARK_SPEC>(pk, wb, specular_col);
                  } else
                  if (ctx == 0) {
                     // render_probes.glsl:65-66 (0/0 fixed as 0)
                     f3 irr = F3(0.0f, 0.0f, 0.0f);
                     if (accw != 0.0f) { irr = sdiv3(acc, accw); irr = irr * irr; }
#if MDH_QVIS_SHARED
                     if (QVIS && !REFLECT) irr_keep = irr; // (rows 12-15 are the workgroup's job list until the kernel ends)
                     else
#endif
                     park_store3<12>(pk, wb, irr);
                     if (REFLECT) park_store3<MDH_PARK_SPEC>(pk, wb, F3(0.0f, 0.0f, 0.0f));
                     shaded = true;
                     // the reflection ray of render_probes.glsl:262-275 finds the next point
                     // (the material id comes back from its park slot: kept in a register across the corner loop it is spilled)
                     active = cfg.spec_mode != 0 && tab_float((sc.mat_slot + 2 * __float_as_int(park_load1<PARK_MAT>(pk, wb)) + 1) * 4) < 0.75f;
#ifdef MDH_ABL_NO_REFLECT
                     active = false;
#endif
                     ro = from_off;
                     rd = reflect(rd, N);
                  } else { // render_probes.glsl:186-208
                     i3 bq;
                     bq.x = best_q & 1023; bq.y = (best_q >> 10) & 1023; bq.z = (best_q >> 20) & 1023;
                     const KProbes pq = probes_fresh(pr);
                     f2 base = probe_id_to_coord<P2>(pq, grid_to_probe_id<P2>(pq, bq));
                     f2 brid = ray_dir_to_ray_id(acc);
                     brid = F2(clamp_(brid.x, pq.rad_lo, pq.rad_hi), clamp_(brid.y, pq.rad_lo, pq.rad_hi));
                     f3 radiance = (SPEC == 2 && pq.rad_mips) // (MDH_OPT_RADIANCE_MIPS, in the kernel variant of the optional paths only: textureLod (.., 1.0))
                        ? radiance_lod_sample(pq, 1.0f, base.x + div_pcx<P2>(pq, brid.x), base.y + div_pcy<P2>(pq, brid.y), u8_tab)
                        : atlas_sample<P2>(pq.rad, pq.fmt, pq.pcx, pq.pcy, pq.rres, pq.rshift, pq.rad_w, pq.rad_h, base.x + div_pcx<P2>(pq, brid.x), base.y + div_pcy<P2>(pq, brid.y), u8_tab, pq.m_rres);
                     specular_col = radiance + specular_col;
                     park_store3<MDH_PARK_SPEC>(pk, wb, specular_col);
                  }
                  PH_ADD(pt, 7);
               }
            }
         }
      }
   }
   // ---- everything parked comes back for the combine
   PH_T0(pc);
   const f3 dir = park_load3<6>(pk, wb);
   f3 result;
   pos_out = F3(0.0f, 0.0f, 0.0f);
   if (!hit) { // render_probes.glsl:287
      float s = dir.y * 0.7f;
      result = F3(0.30f - s, 0.36f - s, 0.60f - s);
      if (!lane_valid) result = F3(0.0f, 0.0f, 0.0f);
   } else {
      const f3 pos = park_load3<0>(pk, wb), normal = park_load3<3>(pk, wb);
      pos_out = pos;
      if (MODE == 1) {
         result = normal * 0.5f + F3s(0.5f);
      } else { // render_probes.glsl:277-285 and lighting.glsl:51-69
         (void)shaded;
         f3 direct = park_load3<9>(pk, wb);
         if (MODE == 0) {
#if MDH_QVIS_SHARED
            const f3 irr = (QVIS && !REFLECT) ? irr_keep : park_load3<12>(pk, wb);
#else
            const f3 irr = park_load3<12>(pk, wb);
#endif
            const f3 specular_col = REFLECT ? park_load3<MDH_PARK_SPEC>(pk, wb) : F3(0.0f, 0.0f, 0.0f);
            Material m = get_material(sc, __float_as_int(park_load1<PARK_MAT>(pk, wb)));
            const f3 specular_dir = reflect(dir, normal);
            direct = direct + compute_indirect_lighting(irr, specular_col, -dir, normal, specular_dir, m.albedo, m.metallic, m.roughness);
         }
         float ao = 1.0f;
         if (cfg.ao_steps > 0) {
            float ao_sum = 0.0f, max_ao_sum = 0.0f, factor = 1.0f;
#pragma unroll 1
            for (int i = 0; i < cfg.ao_steps; ++i) {
               f3 p = pos + (normal * (float)(i + 1)) * 0.1f;
               ao_sum += factor * sdf<PART>(sc, p);
               max_ao_sum += factor * (float)(i + 1) * 0.1f;
               factor = factor * 0.5f;
            }
            ao = 0.6f + sdiv(0.4f * ao_sum, max_ao_sum);
         }
         result = direct * ao;
      }
   }
   PH_ADD(pc, 8);
   return result;
}
