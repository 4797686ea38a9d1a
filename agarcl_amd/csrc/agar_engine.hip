// libagarcl_hip.so -- HIP kernels (gfx950) and the C ABI of include/agarcl_batch.h.
//
// One 64-lane wavefront per arena (grid = num_arenas single-wave workgroups): a launch steps all
// arenas in lock-step; per-arena state is staged HBM -> LDS once per launch, all `ticks` ticks of a
// step() run out of LDS, and the changed state is written back once.  No torch types cross this ABI.
//
// Build (product):   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared ...   (see build.py)
// Build (test-only): g++ -x c++ -DAGAR_CPU_EMU ...   -> tests/_build/libagarcl_emu.so: the same wave-level
//                    source with lanes executed as loops, used by CPU tests to diff kernel logic against
//                    the oracle.  agarcl_amd never loads it.
#ifndef AGAR_CPU_EMU
#include <hip/hip_runtime.h>
#endif
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/agarcl_batch.h"
#include "../../include/agarcl_vec.h"
#include "agar_core.inl"
#include "agar_quiet.inl"
#ifndef AG_PART_NS   // (split build: the observation kernels live in the main unit only)
#include "agar_obs.inl"
#include "agar_screen.inl"
#include "agar_gobigger.inl"
#include "agar_ram.inl"
#endif

// ---- thread-local error string -------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string &m) { g_err = m; return code; }
#ifndef AG_PART_NS   // (split build: the C ABI lives in the main unit only)
extern "C" const char *agarcl_last_error(void) { return g_err.c_str(); }
#endif

// host index of word w of arena a in a tile-transposed array with R words per arena (agar_types.h)
static inline size_t tix(int ag_ts_lg, size_t a, size_t R, size_t w) { return AG_TILE_BASE(a, R) + AG_TW(w); }

// ---- memory / launch abstraction -----------------------------------------------------------------
#ifdef AGAR_CPU_EMU
typedef void *ag_stream_t;
static void *dmalloc(size_t n, ag_stream_t) { return calloc(1, n ? n : 1); }
static void dfree(void *p) { free(p); }
static int h2d(void *d, const void *h, size_t n, ag_stream_t) { memcpy(d, h, n); return 0; }
static int d2h(void *h, const void *d, size_t n, ag_stream_t) { memcpy(h, d, n); return 0; }
static int dsync(ag_stream_t) { return 0; }
#else
typedef hipStream_t ag_stream_t;
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(AGARCL_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
// zero-filled device memory; the fill is enqueued on the env's own stream, so it orders against everything the env
// uploads or launches later (a hipMemset on the NULL stream would not: the env's stream is non-blocking)
static void *dmalloc(size_t n, hipStream_t s) { void *p = nullptr; if (hipMalloc(&p, n ? n : 1) != hipSuccess) return nullptr; if (hipMemsetAsync(p, 0, n ? n : 1, s) != hipSuccess) { (void)hipFree(p); return nullptr; } return p; }
static void dfree(void *p) { if (p) (void)hipFree(p); }
static int h2d(void *d, const void *h, size_t n, ag_stream_t s) { return hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess ? 0 : -1; }
static int d2h(void *h, const void *d, size_t n, ag_stream_t s) { return hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess ? 0 : -1; }
static int dsync(ag_stream_t s) { return hipStreamSynchronize(s) == hipSuccess ? 0 : -1; }
#endif

// k_fused (one launch) vs k_quiet + k_step on C2, us per step, with 16 lanes per arena in k_fused and the best group size
// in k_quiet: 4096: 9.2 / 12.5, 8192: 11.2 / 13.5, 12288: 15.7 / 14.6, 16384: 18.1 / 15.0, 32768: 31.0 / 18.1 -- the single
// launch loses where its batch no longer fits the 2 wavefronts per SIMD it can keep resident.  Its front part therefore gets
// the lane-group size as a parameter too (the general engine behind it is a real call, shared by all of them).
#define AG_FUSED_WAVES 2048L   // k_fused keeps 2 wavefronts per SIMD: the single-launch step is used while the batch fits them
#ifndef AG_KSTEP_SMALL_GRID
#define AG_KSTEP_SMALL_GRID 256
#endif
struct agarcl_env {
  agarcl_config cfg;
  AgDims d; AgParams g; AgState s;   // host copy of the descriptor
  AgState *d_state;                  // HBM-resident copy the kernels read (== &s in the test-only build)
  const float *act_dxdy; const int32_t *act;
  int device;
  ag_stream_t stream; bool own_stream;
  size_t lds_bytes; int ns; bool all_vis;
  int slot;  // ring index of the packed result buffer written by the NEXT step
  std::vector<void *> allocs; size_t alloc_bytes = 0; bool alloc_failed = false;   // (any alloc<>() that returned null: checked once, before the first upload or launch)
  float *d_act_dxdy; int32_t *d_act;  // env-owned action buffers (host-copy path)
  float *lut_r, *lut_ms, *lut_ss, *lut_anti;
  void *timer_ev[2];               // agarcl_timer_mark / agarcl_timer_elapsed_ms
  void *order_ev[2];               // agarcl_stream_wait / agarcl_stream_signal (created on first use)
  int32_t *obs_buf; size_t obs_cap;  // staging for host-side grid observations
  int32_t *undo_list, *undo_count; const int32_t *undo_out; int undo_key;  // incremental clearing of the grid observation (AgObsUndo)
  uint8_t *undo_sig; int undo_sig_g;   // ... and the out-of-bounds channel's row / column signature of the previous call ([frames][2 G] bytes)
  std::vector<uint32_t> seeds;  // last seed of every arena (BaseEnvironment::seed_, written into JSON snapshots)
  bool fused;     // single-launch step (k_fused) instead of k_quiet + k_step: see k_fused
  bool fused_fixed;                // AGARCL_FUSED=0/1 pins the choice
  int fused_wg;                    // threads per workgroup of k_fused (64, 128 or 256)
  int fused_qg; bool fused_ok;     // lanes per arena of k_fused (from the arena count) / the batch is one the single launch may serve
  int quiet_qg;                    // lanes per arena of k_quiet (1, 2, 4, 8 or 16): see agarcl_create
  uint32_t *d_stage; size_t stage_words;  // staging buffer of pull_t / push_t (one arena's largest transposed block)
  int32_t *h_stat; void *stat_ev;  // pinned copy of {qstat, flag watch word} + the event that says it has arrived
  uint8_t *d_mask;                 // [A] reset mask staging (host masks are copied here, stream-ordered)
  uint32_t flags_seen;             // OR of every flag watch sample so far (agarcl_poll_flags)
  long work_step0, work_front0; int32_t work_unf0; int64_t work_pass0;  // agarcl_debug_work baselines
  long next_poll; int poll_gap;          // when the next statistics sample is requested (poll_stats)
  bool stat_pending, stat_stale_flags;   // a sample is in flight / it was requested before the last reset: its flag word is void
  long step_no, front_runs, stat_req_front, stat_last_front; int32_t stat_last_total;
  int parity;     // launch parity of the k_quiet / k_step pair (selects the unfinished-arena counter)
  int sched_parity; // launch parity of k_step (selects its work counter: see k_step)
  int kstep_grid;   // workgroups of a k_step launch over the whole batch (4096; AGARCL_KSTEP_GRID caps it for tests)
  long order_age; bool order_ready, no_order;   // k_order: k_step launches over the whole batch so far / order[] holds a permutation / AGARCL_NO_ORDER=1 (A/B timing)
  bool no_front;  // AGARCL_NO_FRONT=1 in the environment: skip k_quiet (diagnostics / A-B timing only)
  bool few_unfinished; // adaptive: the front part leaves < 64 arenas per step to k_step (see launch_step)
  bool front_off; // adaptive: the front part finishes (almost) no arena-step, so the two-kernel step runs k_step alone
};

// ---- episode bookkeeping of the vector surface (include/agarcl_vec.h) ---------------------------------------------------------------
// What AgarioEnv.step does per env on the host after the engine's step (/root/reference/gym_agario/AgarioEnv.py:105-123): done = the
// engine's done flag or, for an episodic env, number_steps played -- compared BEFORE this step is counted (:111-112) --, the step counter,
// f64 -> f32 rewards, plus the episode statistics; returns whether the arena's episode ended (it is then reset by the caller).
struct AgVecPost {
  const uint8_t *dones; const double *rewards; int n, number_steps, episodic;
  int32_t *steps; float *reward; uint8_t *done, *trunc, *ended; float *ep_return, *final_return; int32_t *final_length;
  const int32_t *ar; int ts_lg, reset_flagged;   // reset_flagged: an arena that carries a capacity flag ends here too (its rows are truncated)
};
AG_DEV bool ag_vec_post_arena(const AgVecPost &v, int a) {
  const int n = v.n, played = v.steps[a];
  const bool timeout = v.episodic != 0 && played >= v.number_steps;
  bool any = timeout;
  // a capacity flag = the arena has left what this engine (or, mostly, the reference itself: README "capacity flags") can represent: with
  // reset_flagged its episode is cut here like any other -- truncated, never terminated -- and the reset clears the flag
  if (v.reset_flagged) { const int ag_ts_lg = v.ts_lg; any = any || v.ar[AG_TILE_BASE(a, AR_WORDS) + AG_TW(AR_FLAGS)] != 0; }
  for (int i = 0; i < n; i++) any = any || v.dones[(size_t)a * n + i] != 0;
  for (int i = 0; i < n; i++) {
    const size_t k = (size_t)a * n + i;
    const bool d = v.dones[k] != 0 || timeout;
    v.done[k] = d ? 1 : 0;
    v.trunc[k] = (any && !d) ? 1 : 0;   // the arena is reset under an agent that was not done: its episode is cut, not finished
    const float r = (float)v.rewards[k];
    v.reward[k] = r;
    const float e = v.ep_return[k] + r;
    if (any) { v.final_return[k] = e; v.ep_return[k] = 0.0f; } else v.ep_return[k] = e;
  }
  if (any) v.final_length[a] = played + 1;
  v.steps[a] = any ? 0 : played + 1;
  v.ended[a] = any ? 1 : 0;
  return any;
}

// ---- kernels ----------------------------------------------------------------------------------------
// NS = pellet register slots per lane (64 pellets per slot): 4 / 8 / 16 / 32 <=> up to 256 / 512 / 1024 / 2048 pellets
template <int NS, bool AV> AG_DEV void ag_ctx_init(AgCtx<NS, AV> &c, const AgState *gs, int arena, unsigned char *lds, const float *act_dxdy, const int32_t *act, int slot = 0) {
  c.slot = slot;
  c.gs = gs; c.arena = arena; c.lds = lds; c.act_dxdy = (const AG_GLOBAL float *)act_dxdy; c.act = (const AG_GLOBAL int32_t *)act;
  c.P = gs->d.P; c.PC = gs->d.PC; c.ts_lg = gs->d.ts_lg;
  c.VC = gs->d.VC; c.FC = gs->d.FC;
  ag_lds_layout(c.P, c.VC, c.FC, gs->d.EC, gs->d.KC, &c.cells_off, &c.vir_off, &c.food_off);
  c.food_dirty = false;
  c.ncreated = 0; c.pel_dirty = false; c.pel_loaded = false; c.pel_all = false; PEL_CLEAN(c);
}
#define AG_DISPATCH_NS(ns, CALL) do { if (e->all_vis) { switch (ns) { case 4: CALL(4, true); break; case 8: CALL(8, true); break; case 16: CALL(16, true); break; default: CALL(32, true); break; } } \
  else { switch (ns) { case 4: CALL(4, false); break; case 8: CALL(8, false); break; case 16: CALL(16, false); break; default: CALL(32, false); break; } } } while (0)
#ifdef AGAR_CPU_EMU
template <int NS, bool AV, class F> static void for_each_arena_ns(agarcl_env *e, F f) {
  std::vector<unsigned char> lds(e->lds_bytes + 64);
  for (int a = 0; a < e->d.A; a++) {
    AgCtx<NS, AV> *c = new AgCtx<NS, AV>();
    ag_ctx_init(*c, &e->s, a, lds.data(), e->act_dxdy, e->act, e->slot);
    f(*c);
    delete c;
  }
}
#else
extern __shared__ __align__(16) unsigned char ag_lds[];
// k_step is register-hungry (pellet registers + every rule inlined: ~210 VGPRs => 2 waves/SIMD).  Measured on the full
// rule set (C3 / mode 6, 4096 arenas): capping it at 128 VGPRs (4 waves/SIMD = all 4096 arenas resident at once; the
// compiler spills ~290 rarely-live registers to scratch) is 1.32x faster, 3 waves 1.11x, 5 waves same as 4, 6 and 8 slower;
// the quiet-dominated C2 step is unaffected (it runs in k_quiet).
#ifndef AG_KSTEP_ATTR
// Register budgets by instantiation (round 6, scripts/gpu_config_sweep.py): the several-player kernels with 16 or 32 pellet slots do not fit 128
// registers -- 100-800 bytes of scratch per lane with 16 slots, 1.5-4 KB with 32 (1096 spilled registers) -- and scratch is what their step then costs:
// agent + 3 bots in an 1100 x 1100 arena (16 slots, pellet grid larger than 2 x 2) 432 us per step at 4 wavefronts per SIMD, 135 us at 3; agent + 4 bots
// with 1500 pellets (32 slots) 1155 us at 4, 143 us at 2.  Several players hold 12-58 KB of LDS per arena anyway, so the wave slots given up were
// mostly not there to lose.  Up to 8 slots (C1, the paper's tasks: 0-12 bytes of scratch) and the single-player kernels keep 4 (the single-player
// 32-slot form at 2: 397 -> 509 us).
#ifndef AG_K32_SINGLE
#define AG_K32_SINGLE false   // (measurement switch: the single-player 32-slot instantiation at 2 wavefronts per SIMD as well)
#endif
#ifndef AG_K32MP_WAVES
#define AG_K32MP_WAVES 2
#endif
#ifndef AG_K16MP_WAVES
#define AG_K16MP_WAVES 3
#endif
#define AG_KSTEP_WAVES ((NS == 32 && MP) ? AG_K32MP_WAVES : (NS == 32 && AG_K32_SINGLE) ? 2 : (NS == 16 && MP) ? AG_K16MP_WAVES : 4)
#define AG_KSTEP_ATTR __attribute__((amdgpu_waves_per_eu(AG_KSTEP_WAVES, AG_KSTEP_WAVES)))
#endif
#define AG_KERNEL_PROLOGUE AgCtx<NS, AV> c; ag_ctx_init(c, gs, (int)blockIdx.x, ag_lds, act_dxdy, act);
// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), and the per-arena word arrays are tile-transposed
// (agar_types.h): 64 consecutive arenas share their cache lines.  This bijective remap gives every XCD one contiguous range
// of workgroup indices -- hence of arenas -- so that a tile's lines live in ONE L2 instead of up to eight (speed only).
__device__ __forceinline__ int ag_xcd_swizzle(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// One wavefront per arena, grid-stride: the grid is min(A, 4096) single-wave workgroups -- 4 per SIMD is all the register
// budget admits, so a larger grid would only queue -- and every workgroup walks its share of the arenas.
// use_q: k_quiet ran in front of this launch and left a work list (qlist / qcount) of the arenas it did not finish; only
// those are visited, resuming where the front part stopped.  A quiet-dominated step therefore costs this launch one
// scalar load per workgroup whatever the arena count.
// MP: the arenas hold several players (or none).  The single-player instantiation tells the compiler so (c.P = 1 below): everything that exists for
// several players only -- the batched simple turns, move_all_players, players_collision, the bots, the per-player loops -- folds away instead of
// sitting, never executed, in the register allocation of the single-player engine (with simple_turns inlined into ONE k_step the 16-slot
// instantiation went from 20 to 88 bytes of scratch per lane and wrote 26 MB instead of 9 MB per 4096-arena launch of mode 6: PMC WRITE_SIZE).
template <int NS, bool AV, int TSLG, bool MP> __global__ void __launch_bounds__(64) AG_KSTEP_ATTR k_step(const AgState *__restrict__ gs, const float *act_dxdy, const int32_t *act, int ticks, int with_env, int slot, int use_q, int parity, int sp, const int32_t *order) {
  const int A = gs->d.A;
  int total = A;
  if (use_q) {
    auto qc = (AG_GLOBAL int32_t *)gs->qcount;
    total = qc[parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) qc[parity ^ 1] = 0;   // re-arm the other parity's counter for the next step's k_quiet
  }
  // Work beyond the grid is handed out dynamically: a workgroup that has finished its arena draws the next item from a counter (sched[sp];
  // this launch re-arms sched[sp ^ 1], which the previous launch used and the next one will).  With a fixed stride every wavefront slot owned
  // arenas it, it + 4096, ... whatever they cost, and an arena-step of the full rule set costs anything between 0.4 and 2.4 times the mean
  // (DESIGN.md section 4.4): at 32768 arenas the launch waited for the slot whose eight arenas happened to be dense clumps.
  auto sched = (AG_GLOBAL int32_t *)gs->sched;
  if (blockIdx.x == 0 && threadIdx.x == 0) sched[sp ^ 1] = 0;
  for (int it = (TSLG && !order) ? ag_xcd_swizzle((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x; it < total;
#ifdef AG_KSTEP_STATIC   // (measurement builds only: the fixed stride)
       it += (int)gridDim.x) {
#else
       it = (int)gridDim.x + (total > (int)gridDim.x ? ag_uni(threadIdx.x == 0 ? __hip_atomic_fetch_add(sched + sp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0) : total)) {
#endif
    // order: the arenas by descending cost of their last visit (k_order).  The expensive ones go first, so that the launch does not end with
    // a dense clump that was started last (longest-processing-time-first list scheduling)
    int arena = order ? ((const AG_GLOBAL int32_t *)order)[it] : it, q_done = -1, q_before = 0;
    const unsigned long long t_begin = __builtin_readcyclecounter();
    if (use_q) {
      arena = ((const AG_GLOBAL int32_t *)gs->qlist)[(size_t)parity * A + it];
      auto qi = (const AG_GLOBAL int32_t *)(gs->qinfo + (size_t)arena * 2);
      q_done = qi[0]; q_before = qi[1];
    }
    AgCtx<NS, AV> c; ag_ctx_init(c, gs, arena, ag_lds, act_dxdy, act, slot);
    c.ts_lg = TSLG;   // (== gs->d.ts_lg, as a constant: the array strides fold into the address arithmetic)
    if constexpr (!MP) c.P = 1;   // (== gs->d.P: launch_step picks the instantiation)
#ifdef AGAR_PROFILE
    for (int i = 0; i < AG_NPROF; i++) c.tacc[i] = 0;
    c.tlast = ag_clock32();
#endif
    // The pellets are NOT part of this load although a general tick is certain: a launch starts with every wavefront of the device loading at
    // once, which is a bandwidth burst (45 MB at 4096 arenas), not a latency -- so the 8 KB of pellets are requested by the first tick
    // (ensure_pellets) and stream in behind its kinematics, relaxation and virus test, none of which reads them; the compiler's wait sits in
    // front of the first pellet scan.  Measured against "everything in one round trip": mid-game 67.7 -> 65.2 us, mode 6 357 -> 352, C1 208.8 -> 205.6.
#ifndef AG_KSTEP_WANT_PELLETS
#define AG_KSTEP_WANT_PELLETS false   // (measurement switch)
#endif
    arena_load(c, AG_KSTEP_WANT_PELLETS);
    AG_T(c, 0);
    env_step<NS, AV, false>(c, ticks, with_env != 0, q_done, q_before);   // (general ticks only: see env_step)
    AG_T(c, 10);
    arena_store(c);
    AG_T(c, 11);
#ifdef AGAR_PROFILE
    if (threadIdx.x == 0 && gs->prof) for (int i = 0; i < AG_NPROF; i++) gs->prof[(size_t)arena * AG_NPROF + i] += c.tacc[i];
#endif
    if (threadIdx.x == 0) ((AG_GLOBAL uint32_t *)gs->cost)[arena] = (uint32_t)(__builtin_readcyclecounter() - t_begin);
    ag_lds_order();
  }
}
// 256 threads = 256 / QG arenas per workgroup, QG lanes each (agar_quiet.inl)
// k_quiet stands alone only in the two-kernel step, which is chosen where the batch does not fit k_fused's 2048 wavefronts
// (> 131072 arenas) or the arenas are not quiet.  Register budget of 3 waves per SIMD (168 VGPRs).  Round 2 capped it at 128 (4 waves: 80
// bytes of scratch then, and 65 536 arenas 52.3 -> 49.6 us); with the tracked pellet in the tick loop's state the 16-slot variants need
// 156-159 registers, and at 128 every launch pushed 112-192 bytes per lane through scratch -- at 262 144 arenas 66 of the 198 MB a step
// moved.  Measured with the budget of 3 waves (no scratch; the 4- and 8-slot variants still fit 128 registers and keep 4 waves): 262 144
// arenas 45.8 -> 36.6 us per step, 524 288: 81.6 -> 55.0, the two-kernel step at 4096 / 65 536 / 131 072 arenas 15.3 -> 11.0 /
// 25.0 -> 18.2 / 33.1 -> 24.2 us (the single launch k_fused still wins up to 131 072: 16.7 and 22.6 us at 65 536 and 131 072).
#ifndef AG_KQUIET_ATTR
#define AG_KQUIET_ATTR __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif
template <int NS, bool AV, int QG, int TSLG> __global__ void __launch_bounds__(256) AG_KQUIET_ATTR k_quiet(const AgHot hot, const AgState *__restrict__ gs, const float *act_dxdy, const int32_t *act, int ticks, int with_env, int slot, int parity) {
  int arena = (TSLG ? ag_xcd_swizzle((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x) * (256 / QG) + (int)threadIdx.x / QG;
  const int A = gs->d.A; const bool valid = arena < A;
  if (!valid) arena = A - 1;
  quiet_arena<NS, AV, QG, TSLG>(hot, gs, arena, (int)threadIdx.x % QG, valid, (const AG_GLOBAL float *)act_dxdy, (const AG_GLOBAL int32_t *)act, ticks, with_env != 0, slot, parity);
}
// The general engine for ONE arena as a real call: k_fused's front part (every lane-group size, both layouts) shares one copy
// of it per (NS, AV, layout) and keeps its own, small register allocation around the rare call.
// The wave's LDS block travels as a BYTE OFFSET into ag_lds, not as a pointer: a pointer parameter is a generic (flat) address to the
// compiler, every LDS access of the callee becomes a flat_* instruction, and the hardware picks the aperture of a flat access from the
// address REGISTER alone -- the instruction's immediate offset is added afterwards.  hipcc folds constant parts of an index into that
// immediate, so `cand[4 * b + 4]` with b == -1 (the insertion sort of pellets_eat placing a record at the front) became
// flat_store(vaddr = lds + L_CAND*0 - 16, offset: L_CAND + 16): for the workgroup's first wavefront, whose block starts at LDS offset 0,
// vaddr lies 16 bytes BELOW the LDS aperture and the queue dies with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION (found by the GPU soak,
// located with rocgdb: profiles/r03_fused_fault_rocgdb.txt).  `ag_lds + offset` formed inside the callee keeps the LDS
// address space: ds_* instructions, whose 32-bit address arithmetic has no aperture to leave (tests/test_isa_lds_access.py keeps it so).
template <int NS, bool AV, int TSLG> __device__ __attribute__((noinline)) void general_arena_step(const AgState *gs, int arena, int lds_off, const float *act_dxdy, const int32_t *act, int slot, int ticks, int with_env, int qd, int qb) {
  AgCtx<NS, AV> c; ag_ctx_init(c, gs, arena, ag_lds + lds_off, act_dxdy, act, slot);
  c.ts_lg = TSLG;
  c.P = 1;   // (k_fused serves single-player envs only -- fused_ok --: the several-player code folds away, as in k_step<.., MP = false>)
  arena_load(c, true);
  // general ticks only, as k_step runs them (r05): the front part has played the quiet ticks it could; with the quiet run inlined here as well the
  // callee needed 860 bytes of scratch per lane instead of 484 for the same times (C2 at 4096 / 65 536 / 131 072 arenas: 9.10 / 16.7 / 22.5 us either way)
  env_step<NS, AV, false>(c, ticks, with_env != 0, qd, qb);
  arena_store(c);
  ag_lds_order();
}
#ifndef AG_KFUSED_ATTR
#define AG_KFUSED_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
// Fused step for quiet-dominated single-player envs: ONE launch.  The front part is k_quiet's (QG lanes per arena, either
// layout of the word arrays); whatever it leaves unfinished (in the C2 workload: 0.004 arenas per 4096-arena step) is completed
// right here by the same wavefront calling the general engine (general_arena_step), one arena after the other.  Saves the second dependent launch (>= 3.4 us:
// scripts/microbench/launch_floor.hip) at the price of serialising a wavefront's unfinished arenas -- the wrong trade
// when most arenas need the general path every step (mass-1000 modes), where the two-kernel step is used.
template <int NS, bool AV, int QG, int TSLG> __global__ void __launch_bounds__(256) AG_KFUSED_ATTR k_fused(const AgHot hot, const AgState *__restrict__ gs, const float *act_dxdy, const int32_t *act, int ticks, int with_env, int slot, int lds_per_wave) {
  int arena = (TSLG ? ag_xcd_swizzle((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x) * ((int)blockDim.x / QG) + (int)threadIdx.x / QG;
  const int A = gs->d.A; const bool valid = arena < A;
  if (!valid) arena = A - 1;
  const int sub = (int)threadIdx.x % QG;
  QHandOver h = quiet_arena<NS, AV, QG, TSLG>(hot, gs, arena, sub, valid, (const AG_GLOBAL float *)act_dxdy, (const AG_GLOBAL int32_t *)act, ticks, with_env != 0, slot, -1);
  unsigned long long todo = __ballot(valid && sub == 0 && h.done != ticks);
  if (!todo) return;
  ag_mem_fence();  // the front part's stores precede the general part's loads of the same arena
  const int lds = ((int)threadIdx.x >> 6) * lds_per_wave;   // (a byte offset into ag_lds: see general_arena_step)
  while (todo) {
    const int src = (int)__builtin_ctzll(todo); todo &= todo - 1ull;
    const int ar = __builtin_amdgcn_readlane(arena, src), qd = __builtin_amdgcn_readlane(h.done, src), qb = __builtin_amdgcn_readlane(h.before, src);
    general_arena_step<NS, AV, TSLG>(gs, ar, lds, act_dxdy, act, slot, ticks, with_env, qd, qb);
  }
}
// a workgroup (one wavefront) per arena: lane 0 settles the arena's episode; the wavefront resets the arena if the episode ended and leaves
// at once otherwise -- bookkeeping and same-step auto-reset of agarcl_vec_step in ONE launch
template <int NS, bool AV> __global__ void __launch_bounds__(64) k_vec_post_reset(const AgState *__restrict__ gs, const AgVecPost v, int reset_ids) {
  int any = 0;
  if (threadIdx.x == 0) any = ag_vec_post_arena(v, (int)blockIdx.x) ? 1 : 0;
  if (!__builtin_amdgcn_readfirstlane(any)) return;
  const float *act_dxdy = nullptr; const int32_t *act = nullptr;
  AG_KERNEL_PROLOGUE
  arena_load(c);
  env_reset(c, reset_ids);
  arena_store(c);
}
template <int NS, bool AV> __global__ void __launch_bounds__(64) k_reset(const AgState *__restrict__ gs, const uint8_t *mask, int reset_ids) {
  if (mask && !mask[blockIdx.x]) {
    // the flag watch word (qstat[1]) was zeroed in front of this launch (agarcl_reset / agarcl_reset_device): arenas that are NOT reset
    // put their flags back, so the watch becomes the OR over the arenas that are still flagged after the masked reset
    if (threadIdx.x == 0) { const int ag_ts_lg = gs->d.ts_lg; const int fl = ((const AG_GLOBAL int32_t *)gs->ar)[AG_TILE_BASE(blockIdx.x, AR_WORDS) + AG_TW(AR_FLAGS)]; if (fl) ag_atomic_or(gs->qstat + 1, fl); }
    return;
  }
  const float *act_dxdy = nullptr; const int32_t *act = nullptr;
  AG_KERNEL_PROLOGUE
  arena_load(c);
  env_reset(c, reset_ids);   // (marks every pellet slot dirty)
  arena_store(c);
}
template <int NS, bool AV> __global__ void __launch_bounds__(64) k_respawn(const AgState *__restrict__ gs) {
  const float *act_dxdy = nullptr; const int32_t *act = nullptr;
  AG_KERNEL_PROLOGUE
  arena_load(c);
  respawn_dead(c);
  arena_store(c);
}
#ifndef AG_PART_NS   // (split build: plain kernels live in the main unit only)
// order[] = the arenas by descending cost (one 1024-thread workgroup; a counting sort over 1024 cost classes scaled to the largest cost --
// within a class the order is whatever the atomics give: it only steers scheduling, never results)
__global__ void __launch_bounds__(1024) k_order(const uint32_t *__restrict__ cost, int32_t *__restrict__ order, int A) {
  __shared__ unsigned hist[1024]; __shared__ unsigned mx;
  const int t = (int)threadIdx.x;
  hist[t] = 0u; if (t == 0) mx = 1u;
  __syncthreads();
  unsigned m = 0u; for (int a = t; a < A; a += 1024) { const unsigned c = cost[a]; m = c > m ? c : m; }
  atomicMax(&mx, m);
  __syncthreads();
  const unsigned long long top = mx;
  auto cls = [&](unsigned c) -> int { return 1023 - (int)(((unsigned long long)c * 1023ull) / top); };   // class 0 = the most expensive
  for (int a = t; a < A; a += 1024) atomicAdd(&hist[cls(cost[a])], 1u);
  __syncthreads();
  if (t == 0) { unsigned run = 0u; for (int k = 0; k < 1024; k++) { const unsigned n = hist[k]; hist[k] = run; run += n; } }   // (1024 serial adds: ~2 us, every 8th step)
  __syncthreads();
  for (int a = t; a < A; a += 1024) order[atomicAdd(&hist[cls(cost[a])], 1u)] = a;
}
__global__ void k_set_ar_word(int32_t *ar, int ag_ts_lg, int word, int n, int value) {  // one arena word of every arena
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ar[AG_TILE_BASE(i, AR_WORDS) + AG_TW(word)] = value;
}
// one arena's block of a tile-transposed array <-> a contiguous staging buffer (host copies: dumps, snapshots, introspection)
__global__ void k_tile_gather(const uint32_t *src, uint32_t *dst, int R, int ag_ts_lg) { int w = blockIdx.x * blockDim.x + threadIdx.x; if (w < R) dst[w] = src[AG_TW(w)]; }
__global__ void k_tile_scatter(uint32_t *dst, const uint32_t *src, int R, int ag_ts_lg) { int w = blockIdx.x * blockDim.x + threadIdx.x; if (w < R) dst[AG_TW(w)] = src[w]; }
#endif
#endif

// ---- split build (build.py): the step / reset kernels of ONE (NS, AV) pair per translation unit -----------------------------------------
// The 8 (pellet slots, all-visible) pairs x (2 layouts x 5 lane-group sizes) instantiations of the step kernels are what makes this file
// slow to compile (3 m 47 s in one piece).  build.py therefore compiles it seventeen times in parallel: per pair a "step" unit
// (-DAG_PART_NS=<4|8|16|32> -DAG_PART_AV=<0|1> -DAG_PART_KIND=0: k_step, k_reset, k_respawn) and a "front" unit (-DAG_PART_KIND=1: k_quiet and
// k_fused with the general engine behind it) -- only the kernel templates above and their explicit instantiations, the rest of the file is
// skipped -- and the main unit (-DAG_SPLIT_BUILD: everything else, with the step kernels declared `extern template`), linked into one
// libagarcl_hip.so.  The step units are compiled with machine-level loop-invariant code motion off (build.py STEP_FLAGS: what it hoists in
// front of k_step's loops does not fit 128 registers and is spilled on the spot).  The same source without these macros still builds as a
// single unit.
#if !defined(AGAR_CPU_EMU) && (defined(AG_PART_NS) || defined(AG_SPLIT_BUILD))
#define AG_INST_STEP1(X, N, V, T, M) X template __global__ void k_step<N, V, T, M>(const AgState *__restrict__, const float *, const int32_t *, int, int, int, int, int, int, const int32_t *);
#define AG_INST_STEP(X, N, V, T) AG_INST_STEP1(X, N, V, T, false) AG_INST_STEP1(X, N, V, T, true)
#define AG_INST_FRONT(X, N, V, Q, T) \
  X template __global__ void k_quiet<N, V, Q, T>(const AgHot, const AgState *__restrict__, const float *, const int32_t *, int, int, int, int); \
  X template __global__ void k_fused<N, V, Q, T>(const AgHot, const AgState *__restrict__, const float *, const int32_t *, int, int, int, int);
#define AG_INST_FRONTS(X, N, V, T) AG_INST_FRONT(X, N, V, 1, T) AG_INST_FRONT(X, N, V, 2, T) AG_INST_FRONT(X, N, V, 4, T) AG_INST_FRONT(X, N, V, 8, T) AG_INST_FRONT(X, N, V, 16, T)
#define AG_INST_KIND0(X, N, V) AG_INST_STEP(X, N, V, 0) AG_INST_STEP(X, N, V, 6) \
  X template __global__ void k_reset<N, V>(const AgState *__restrict__, const uint8_t *, int); \
  X template __global__ void k_respawn<N, V>(const AgState *__restrict__); \
  X template __global__ void k_vec_post_reset<N, V>(const AgState *__restrict__, const AgVecPost, int);
#define AG_INST_KIND1(X, N, V) AG_INST_FRONTS(X, N, V, 0) AG_INST_FRONTS(X, N, V, 6)
#ifdef AG_PART_NS
#if AG_PART_KIND == 0
AG_INST_KIND0(, AG_PART_NS, (AG_PART_AV != 0))
#else
AG_INST_KIND1(, AG_PART_NS, (AG_PART_AV != 0))
#endif
#else
#define AG_INST_ALL(N) AG_INST_KIND0(extern, N, true) AG_INST_KIND0(extern, N, false) AG_INST_KIND1(extern, N, true) AG_INST_KIND1(extern, N, false)
AG_INST_ALL(4) AG_INST_ALL(8) AG_INST_ALL(16) AG_INST_ALL(32)
#endif
#endif
#ifndef AG_PART_NS   // (a part unit ends here)
// host copy of ONE arena's block of a tile-transposed array (32-bit words): device gather / scatter through a staging buffer
template <class T> static int pull_t(agarcl_env *e, std::vector<T> &h, const T *dev, size_t a, size_t R) {
  static_assert(sizeof(T) == 4, "32-bit words");
  h.resize(R);
  if (!R) return 0;
  const int ag_ts_lg = e->d.ts_lg;
#ifdef AGAR_CPU_EMU
  for (size_t w = 0; w < R; w++) h[w] = dev[tix(ag_ts_lg, a, R, w)];
  return 0;
#else
  if (R > e->stage_words) return -1;
  hipLaunchKernelGGL(k_tile_gather, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, e->stream, (const uint32_t *)(dev + AG_TILE_BASE(a, R)), e->d_stage, (int)R, ag_ts_lg);
  return d2h(h.data(), e->d_stage, R * 4, e->stream);
#endif
}
template <class T> static int push_t(agarcl_env *e, const std::vector<T> &h, T *dev, size_t a) {
  static_assert(sizeof(T) == 4, "32-bit words");
  const size_t R = h.size();
  if (!R) return 0;
  const int ag_ts_lg = e->d.ts_lg;
#ifdef AGAR_CPU_EMU
  for (size_t w = 0; w < R; w++) dev[tix(ag_ts_lg, a, R, w)] = h[w];
  return 0;
#else
  if (R > e->stage_words || h2d(e->d_stage, h.data(), R * 4, e->stream)) return -1;
  hipLaunchKernelGGL(k_tile_scatter, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, e->stream, (uint32_t *)(dev + AG_TILE_BASE(a, R)), (const uint32_t *)e->d_stage, (int)R, ag_ts_lg);
  return hipGetLastError() == hipSuccess ? 0 : -1;
#endif
}

#ifndef AGAR_CPU_EMU
// Asynchronous statistics.  The kernels keep two running words in HBM (AgState::qstat): [0] the arena-steps the front
// part left unfinished, [1] the OR of every capacity flag raised.  Every 64 steps the host asks for them with an
// asynchronous copy into pinned memory and, once the copy has landed (event query, never a wait), uses them:
//   * flag watch (agarcl_poll_flags): a diverged arena becomes visible to the host without a synchronising query;
//   * fused or two-kernel step: a wavefront of k_fused completes its unfinished arenas one after the other, which only
//     pays while they are rare.  Results do not depend on the choice, only the time does.
static void poll_stats(agarcl_env *e, bool adapt) {
  e->step_no++;
  if (!e->h_stat) return;
  if (e->stat_pending) {
    if (hipEventQuery((hipEvent_t)e->stat_ev) != hipSuccess) return;
    e->stat_pending = false;
    if (!e->stat_stale_flags) e->flags_seen |= (uint32_t)e->h_stat[1];
    e->stat_stale_flags = false;
    const long steps = e->stat_req_front - e->stat_last_front;  // steps in which the front part actually ran
    if (adapt && steps > 0) {
      const double frac = (double)(uint32_t)(e->h_stat[0] - e->stat_last_total) / ((double)steps * (double)e->d.A);
      // (fused_ok: the batch fits the 2048 wavefronts k_fused keeps resident at its best lane-group size, see agarcl_create)
      if (e->d.A <= e->kstep_grid) {
        // Every arena fits ONE round of k_step: the general engine over all of them lasts as long as over the unfinished ones alone -- as long as its slowest
        // arena -- so the work list buys nothing and the front launch in front of it only costs (the paper's task 1, 4096 arenas: k_quiet 18 + k_step 36 us
        // against k_step alone 40; tasks 3 / 4 while the agents grow and split: k_fused 9.5 us when nothing is left over, 25-40 when one wavefront has to
        // complete an arena or two, k_step alone 21).  The single launch pays only while a step usually leaves NO arena unfinished.
        const double unfinished = frac * (double)e->d.A;   // arenas per step
        if (e->fused && unfinished > 0.5) e->fused = false; else if (!e->fused && unfinished < 0.1) e->fused = e->fused_ok;   // (the tail of a single launch with one left-over arena: +15-30 us)
        e->front_off = !e->fused;
      } else {
        if (e->fused && frac > 0.05) e->fused = false; else if (!e->fused && frac < 0.01) e->fused = e->fused_ok;
        // the two-kernel step's front launch is pure overhead when it finishes (almost) nothing: mass-1000 modes
        e->front_off = !e->fused && frac > 0.99;
      }
      e->few_unfinished = frac * (double)e->d.A < 64.0;   // k_step's work list is short: a small grid dispatches faster
    }
    e->stat_last_total = e->h_stat[0]; e->stat_last_front = e->stat_req_front;
  } else if (e->step_no >= e->next_poll) {
    // every 64 steps -- but 2, 8 and 32 steps after a reset or a state load, whose arenas may behave quite unlike what the starting form
    // was chosen for (grown agents under the single launch: every arena-step through its serialising general tail, 3x the two-kernel time)
    if (hipMemcpyAsync(e->h_stat, e->s.qstat, 8, hipMemcpyDeviceToHost, e->stream) != hipSuccess) return;
    if (hipEventRecord((hipEvent_t)e->stat_ev, e->stream) != hipSuccess) return;
    e->stat_pending = true; e->stat_req_front = e->front_runs;
    e->poll_gap = e->poll_gap < 64 ? e->poll_gap * 4 : 64; if (e->poll_gap > 64) e->poll_gap = 64;
    e->next_poll = e->step_no + e->poll_gap;
  }
}
#endif
static int launch_step(agarcl_env *e, int ticks, int with_env) {
#ifdef AGAR_CPU_EMU
  const int use_q = e->d.P == 1 && !e->no_front;  // the lean front kernel handles single-player arenas' quiet steps
#define CALL(N, V) for_each_arena_ns<N, V>(e, [&](AgCtx<N, V> &c) { \
    int qd = -1, qb = 0; \
    if (use_q) { const AgHot hot_{c.gs->ar, c.gs->pl, c.gs->cells}; if (c.ts_lg) quiet_arena<N, V, AG_QG, 6>(hot_, c.gs, c.arena, 0, true, c.act_dxdy, c.act, ticks, with_env != 0, c.slot); else quiet_arena<N, V, AG_QG, 0>(hot_, c.gs, c.arena, 0, true, c.act_dxdy, c.act, ticks, with_env != 0, c.slot); qd = c.gs->qinfo[2 * c.arena]; qb = c.gs->qinfo[2 * c.arena + 1]; if (qd == ticks) return; } \
    arena_load(c); env_step(c, ticks, with_env != 0, qd, qb); arena_store(c); })
  AG_DISPATCH_NS(e->ns, CALL);
#undef CALL
#else
  const bool front_ok = e->d.P == 1 && !e->no_front;
  poll_stats(e, front_ok && !e->fused_fixed);
  // front_off: the statistics say the front part finishes < 1 % of the arena-steps, so its launch is skipped; every 256th
  // step it runs again so that the statistics notice when the arenas have become quiet
  // (one round of k_step: every 32nd step, the switch back to the single launch is worth noticing early)
  const int use_q = front_ok && !(e->front_off && !e->fused && (e->step_no & (e->d.A <= e->kstep_grid ? 31 : 255)) != 0);
  if (use_q) e->front_runs++;
  const AgHot hot{e->s.ar, e->s.pl, e->s.cells};
  const bool tiled = e->d.ts_lg != 0;   // (0 or 6, agarcl_create)
  if (use_q && e->fused) {
    const unsigned lpw = (unsigned)((e->lds_bytes + 15) & ~(size_t)15);
    const int wg = e->fused_wg, apw = wg / e->fused_qg;  // threads and arenas per workgroup
#define CALLQ(N, V, Q) hipLaunchKernelGGL((k_fused<N, V, Q, T>), dim3((e->d.A + apw - 1) / apw), dim3(wg), (wg / 64) * lpw, e->stream, hot, e->d_state, e->act_dxdy, e->act, ticks, with_env, e->slot, (int)lpw)
#define CALL(N, V) do { switch (e->fused_qg) { case 1: CALLQ(N, V, 1); break; case 2: CALLQ(N, V, 2); break; case 4: CALLQ(N, V, 4); break; case 8: CALLQ(N, V, 8); break; default: CALLQ(N, V, 16); break; } } while (0)
#define T 6
    if (tiled) AG_DISPATCH_NS(e->ns, CALL);
#undef T
#define T 0
    if (!tiled) AG_DISPATCH_NS(e->ns, CALL);
#undef T
#undef CALL
#undef CALLQ
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (use_q) {
#define CALLQ(N, V, Q) hipLaunchKernelGGL((k_quiet<N, V, Q, T>), dim3((e->d.A + 256 / Q - 1) / (256 / Q)), dim3(256), 0, e->stream, hot, e->d_state, e->act_dxdy, e->act, ticks, with_env, e->slot, e->parity)
#define CALL(N, V) do { switch (e->quiet_qg) { case 1: CALLQ(N, V, 1); break; case 2: CALLQ(N, V, 2); break; case 4: CALLQ(N, V, 4); break; case 8: CALLQ(N, V, 8); break; default: CALLQ(N, V, 16); break; } } while (0)
#define T 6
    if (tiled) AG_DISPATCH_NS(e->ns, CALL);
#undef T
#define T 0
    if (!tiled) AG_DISPATCH_NS(e->ns, CALL);
#undef T
#undef CALL
#undef CALLQ
  }
#define CALLM(N, V, M) hipLaunchKernelGGL((k_step<N, V, T, M>), dim3(kgrid), dim3(64), e->lds_bytes, e->stream, e->d_state, e->act_dxdy, e->act, ticks, with_env, e->slot, use_q, e->parity, e->sched_parity, order)
#define CALL(N, V) do { if (e->d.P == 1) CALLM(N, V, false); else CALLM(N, V, true); } while (0)
  // grid of k_step: every arena (grid-stride from 4096 workgroups on), or -- working off a list the statistics say is short --
  // 256 workgroups, which dispatch faster (the loop still visits every listed arena if the list is long after all)
  const int kfull = e->d.A < e->kstep_grid ? e->d.A : e->kstep_grid, kgrid = use_q && e->few_unfinished && kfull > AG_KSTEP_SMALL_GRID ? AG_KSTEP_SMALL_GRID : kfull;
  // a batch beyond the grid visiting every arena: expensive arenas first (k_order over the cycle counts of their last visits, refreshed
  // every 8th step -- the same arenas are expensive step after step)
  const int32_t *order = nullptr;
  if (!use_q && e->d.A > kfull && !e->no_order) {
    if (e->order_age >= 1 && (e->order_age & 7) == 1) { hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, e->stream, (const uint32_t *)e->s.cost, e->s.order, e->d.A); e->order_ready = true; }
    e->order_age++;
    if (e->order_ready) order = e->s.order;
  }
#define T 6
  if (tiled) AG_DISPATCH_NS(e->ns, CALL);
#undef T
#define T 0
  if (!tiled) AG_DISPATCH_NS(e->ns, CALL);
#undef T
#undef CALL
#undef CALLM
  if (use_q) e->parity ^= 1;
  e->sched_parity ^= 1;
  HIPCHK(hipGetLastError());
#endif
  return 0;
}
static int launch_reset(agarcl_env *e, const uint8_t *mask_dev, const uint8_t *mask_host, int reset_ids) {
#ifdef AGAR_CPU_EMU
  (void)mask_dev;
  e->s.qstat[1] = 0;
#define CALL(N, V) for_each_arena_ns<N, V>(e, [&](AgCtx<N, V> &c) { if (mask_host && !mask_host[c.arena]) { e->s.qstat[1] |= e->s.ar[tix(e->d.ts_lg, (size_t)c.arena, AR_WORDS, AR_FLAGS)]; return; } arena_load(c); env_reset(c, reset_ids); arena_store(c); })
  AG_DISPATCH_NS(e->ns, CALL);
#undef CALL
#else
  (void)mask_host;
#define CALL(N, V) hipLaunchKernelGGL((k_reset<N, V>), dim3(e->d.A), dim3(64), e->lds_bytes, e->stream, e->d_state, mask_dev, reset_ids)
  AG_DISPATCH_NS(e->ns, CALL);
#undef CALL
  HIPCHK(hipGetLastError());
#endif
  return 0;
}
static int launch_respawn(agarcl_env *e) {
#ifdef AGAR_CPU_EMU
#define CALL(N, V) for_each_arena_ns<N, V>(e, [&](AgCtx<N, V> &c) { arena_load(c); respawn_dead(c); arena_store(c); })
  AG_DISPATCH_NS(e->ns, CALL);
#undef CALL
#else
#define CALL(N, V) hipLaunchKernelGGL((k_respawn<N, V>), dim3(e->d.A), dim3(64), e->lds_bytes, e->stream, e->d_state)
  AG_DISPATCH_NS(e->ns, CALL);
#undef CALL
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

// ---- host helpers -------------------------------------------------------------------------------------
template <class T> static T *alloc(agarcl_env *e, size_t n) { T *p = (T *)dmalloc(n * sizeof(T), e->stream); if (p) { e->allocs.push_back(p); e->alloc_bytes += n * sizeof(T); } else e->alloc_failed = true; return p; }

// glibc srand(): TYPE_3 additive-feedback state after the 310 discarded outputs, as the 34-word ring the device
// continues (agar_multi.inl: ag_rand_next).  Engine::seed calls std::srand(s) (Engine.hpp:242-245).
static void glibc_srand_host(int32_t *st, unsigned seed) {
  int32_t r[344];
  if (seed == 0) seed = 1;
  r[0] = (int32_t)seed;
  for (int i = 1; i < 31; i++) { long long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773; long long w = 16807 * lo - 2836 * hi; if (w < 0) w += 2147483647; r[i] = (int32_t)w; }
  for (int i = 31; i < 34; i++) r[i] = r[i - 31];
  for (int i = 34; i < 344; i++) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
  for (int i = 0; i < 34; i++) st[i] = r[310 + i];
  st[34] = 0;
}
static void mt_seed_host(uint64_t *mt, uint64_t seed) {  // std::mt19937_64::seed
  mt[0] = seed;
  for (int i = 1; i < 312; i++) mt[i] = 6364136223846793005ULL * (mt[i - 1] ^ (mt[i - 1] >> 62)) + (uint64_t)i;
}

static int set_mode(AgParams &g, int mode) {  // R: Engine.hpp:367-416
  switch (mode) {
    case 0: case 4: g.mass_decay = 1; g.squared = 0; g.regen = 1; g.agent_mass = 25; return 0;
    case 1: g.mass_decay = 0; g.squared = 1; g.regen = 0; g.agent_mass = 25; return 0;
    case 2: g.mass_decay = 1; g.squared = 1; g.regen = 0; g.agent_mass = 25; return 0;
    case 3: g.mass_decay = 0; g.squared = 0; g.regen = 1; g.agent_mass = 25; return 0;
    case 5: set_mode(g, 2); g.agent_mass = 1000; return 0;
    case 6: set_mode(g, 4); g.agent_mass = 1000; return 0;
    case 7: case 8: case 9: case 10: set_mode(g, 4); return 0;
    default: return -1;
  }
}

extern "C" int agarcl_device_count(void) {
#ifdef AGAR_CPU_EMU
  return 1;
#else
  int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n;
#endif
}

extern "C" int agarcl_destroy(agarcl_env *e) {
  if (!e) return AGARCL_OK;
#ifndef AGAR_CPU_EMU
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  if (e->own_stream) (void)hipStreamDestroy(e->stream);
  if (e->stat_ev) (void)hipEventDestroy((hipEvent_t)e->stat_ev);
  for (int i = 0; i < 2; i++) if (e->timer_ev[i]) (void)hipEventDestroy((hipEvent_t)e->timer_ev[i]);
  for (int i = 0; i < 2; i++) if (e->order_ev[i]) (void)hipEventDestroy((hipEvent_t)e->order_ev[i]);
  if (e->h_stat) (void)hipHostFree(e->h_stat);
#endif
  for (void *p : e->allocs) dfree(p);
#ifndef AGAR_CPU_EMU
  if (e->obs_buf) (void)hipFree(e->obs_buf);
#endif
  delete e;
  return AGARCL_OK;
}

// (base_seed: arena i starts with seed base_seed + i; agarcl_create passes 5489, a pipe's sub-batch 5489 + its first arena)
static int create_env(const agarcl_config *cfg, int32_t num_arenas, int32_t device, uint32_t base_seed, agarcl_env **out) {
  if (!cfg || !out || num_arenas <= 0) return fail(AGARCL_E_INVALID, "agarcl_create: bad arguments");
  if (cfg->num_agents < 0 || cfg->example_bots < 0 || cfg->ticks_per_step < 1 || cfg->arena_size < 8 || cfg->num_pellets < 0 || cfg->num_viruses < 0 || cfg->num_bots < 0)
    return fail(AGARCL_E_INVALID, "agarcl_create: invalid environment arguments");
  AgParams g; memset(&g, 0, sizeof(g));
  if (set_mode(g, cfg->mode_number) != 0) return fail(AGARCL_E_MODE, "Invalid mode number");
#ifndef AGAR_CPU_EMU
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(AGARCL_E_HIP, "agarcl_create: no HIP device available (the HIP engine has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(AGARCL_E_INVALID, "agarcl_create: bad device index");
  HIPCHK(hipSetDevice(device));
#endif
  agarcl_env *e = new agarcl_env();
  e->cfg = *cfg; e->device = device; e->own_stream = true; e->d_act_dxdy = nullptr; e->d_act = nullptr;
  e->slot = 0; e->d_state = nullptr; e->act_dxdy = nullptr; e->act = nullptr; e->timer_ev[0] = e->timer_ev[1] = nullptr; e->order_ev[0] = e->order_ev[1] = nullptr; e->obs_buf = nullptr; e->obs_cap = 0; e->undo_sig = nullptr; e->undo_sig_g = 0; e->undo_list = e->undo_count = nullptr; e->undo_out = nullptr; e->undo_key = -1;
#ifdef AGAR_CPU_EMU
  e->stream = nullptr;
#else
  if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { delete e; return fail(AGARCL_E_HIP, "hipStreamCreate failed"); }
#endif
  double dt = cfg->dt > 0 ? cfg->dt : 1.0 / 30.0;
  g.W = (float)cfg->arena_size;
  g.target_pellets = cfg->num_pellets; g.target_viruses = cfg->num_viruses; g.mode = cfg->mode_number;
  g.dt = (float)dt; g.dt10 = (float)(dt * 10);
  g.recomb_ticks = (int)ceil(10.0 / dt - 1e-9);
  g.reward_type = cfg->reward_type != 0; g.c_death = cfg->c_death; g.screen_respawn = cfg->screen_respawn != 0;
  g.pel_r = (float)sqrt((double)AG_PELLET_MASS / 1.0 / 3.14159265358979323846);  // == lut_r[AG_PELLET_MASS] below
  // grid dims exactly as the reference computes them in float (Engine.hpp:964-965, 1210-1211)
  g.pgw = g.pgh = (int)((g.W + (float)AG_PELLET_GRID - 1.0f) / (float)AG_PELLET_GRID);
  g.vgw = g.vgh = (int)((g.W + (float)AG_VIRUS_GRID - 1.0f) / (float)AG_VIRUS_GRID);
  AgDims d;
  // bots exist only in mode 0 (num_bots of them, BaseEnvironment.hpp:374-399) and modes 7-10 (exactly one, :401-425)
  int nbots = cfg->mode_number == 0 ? cfg->num_bots : (cfg->mode_number > 6 ? 1 : 0);
  d.A = num_arenas; d.n_agents = cfg->num_agents; d.P = cfg->num_agents + nbots + cfg->example_bots;
  g.example_bots = cfg->example_bots;
  int npel = cfg->num_pellets;
  if (g.squared) { int pps = (int)(g.W / 2.0f); npel = 4 * pps; }
  d.PC = ((npel > 0 ? npel : 1) + 63) / 64 * 64;
  // (an arena created without viruses never grows one -- they come from the regeneration target and from feeding an existing virus --: 16 slots, for states
  // loaded into it, instead of 64; the 768 bytes are what a two-player arena needs to get under 10 KB of LDS, below)
  d.VC = cfg->cap_viruses > 0 ? cfg->cap_viruses : cfg->num_viruses > 0 ? cfg->num_viruses + 64 : 16;
  // (foods live in LDS during a launch, 16 bytes each: 128 keep the single-player layout within the 10 KB per wavefront that 16 resident
  // wavefronts per CU leave; nominal play stays far below -- <= ~60 ejected foods in 20k-tick mode-6 roll-outs)
  d.FC = cfg->cap_foods > 0 ? cfg->cap_foods : 128;
  // layout of the per-arena word arrays (agar_types.h): tiles of 64 arenas where the lean front kernel runs with one to four
  // lanes per arena (single-player batches from 32768 arenas on), arena-major otherwise.  AGARCL_TILE_LG=0/6 pins it.
  // (modes 5 and 6 start every agent at mass 1000: the front kernel never runs, and the general engine alone is 2-10 % faster arena-major)
  d.ts_lg = (d.P == 1 && d.A >= 32768 && cfg->mode_number <= 4) ? 6 : 0;
  { const char *t = getenv("AGARCL_TILE_LG"); if (t && (t[0] == '0' || t[0] == '6') && !t[1]) d.ts_lg = t[0] - '0'; }
  if (d.P > AG_MAX_PLAYERS) { agarcl_destroy(e); return fail(AGARCL_E_UNSUPPORTED, "too many players per arena"); }
  if (cfg->cap_cells != 0 && cfg->cap_cells != AG_CC) { agarcl_destroy(e); return fail(AGARCL_E_INVALID, "cap_cells is fixed at 32 in this build"); }
  // Pellet eat events of a tick (the reference's pellets_to_remove, Engine.hpp:976-1009: unbounded, one entry per EAT) and the candidate records
  // of one cell's ordered replay.  LDS holds 256 events and 256 candidates -- for every arena with <= 0.016 pellets per unit area (the
  // 1000 x 1000 / 1000-pellet default: 0.001; C1: 0.008) that is all there is, as before.  A DENSE arena (the gym "trivial" preset 50 x 50 / 200:
  // 0.08; the soak's 80 x 80 / 1300 corner: 0.2) gets candidates up to its pellet capacity (one cell cannot reach more pellets than exist) and a
  // spill area in HBM for the events beyond 256: a pellet under k overlapping cells is eaten k times in one tick (the stale-index quirk) -- a
  // mass-1000 agent split into 13 cells produced 16 861 events in ONE tick of an 80 x 80 / 1300 arena, four players 37 821 (measured on the
  // emulation).  Spill = 64 eats per pellet slot, at most 65 536 entries (256 KB per arena, only there).
  // A CROWDED arena spills too (no LDS cost): five mass-1000 agents and sixteen bots in a 150 x 150 / 200 arena (0.009 pellets per unit area, 22
  // players) overflowed the 256 events in the crowded soak -- pellets per unit area x (1 + players) > 0.06.
  { const double dens = (double)npel / ((double)g.W * (double)g.W); const bool dense = dens > 0.016, crowded = dens * (double)(1 + d.P) > 0.06;
    const int pc = d.PC <= 256 ? 256 : d.PC <= 512 ? 512 : d.PC <= 1024 ? 1024 : 2048;
    // (a crowded but sparse arena -- Tick/30, C1 -- cannot repeat the dense corner's counts: its events are bounded by the pellets its
    // cells cover, a few per pellet slot.  8 per slot there (Tick/30: 16 KB per arena instead of 128 KB, 64 MB instead of 0.5 GB at 4096 arenas);
    // flag 8 still reports an overflow, and the soak draws these configurations)
    d.EC = AG_EV_MIN; d.KC = dense ? pc : AG_EV_MIN; d.EX = dense ? (64 * pc < 65536 ? 64 * pc : 65536) : crowded ? 8 * pc : 0; }
  // ejected foods live in LDS during a launch, 16 bytes each: 128 keep the single-player layout within the 10 KB per wavefront that 16 resident
  // wavefronts per CU leave (nominal play: <= ~60 in 20k-tick mode-6 roll-outs); arenas with many players feed more: 16 per player
  if (cfg->cap_foods <= 0 && d.P * 16 > d.FC) d.FC = d.P * 16;
  // 16 arenas per compute unit -- 4096 resident at once, one round of k_step -- need <= 10 240 bytes of LDS each.  An arena a few hundred bytes above that
  // (agent + 1 bot, the paper's tasks 7-10: 10 576) runs 4096 arenas as 3840 + a second round: 52.9 us per step against 38.5 at 3584 arenas.  Its default food
  // capacity gives the difference back (128 -> 107 there; at most 32 slots, never below 96; an explicit cap_foods is left alone): 39.3 us.
  if (cfg->cap_foods <= 0) {
    const long over = (long)ag_lds_layout(d.P, d.VC, d.FC, d.EC, d.KC, nullptr) - 10240;
    if (over > 0 && over <= 16 * 32 && d.FC - (int)((over + 15) / 16) >= 96) d.FC -= (int)((over + 15) / 16);
  }
  e->d = d; e->g = g;
  e->lds_bytes = ag_lds_layout(d.P, d.VC, d.FC, d.EC, d.KC, nullptr);
  e->all_vis = g.pgw <= 2 && g.pgh <= 2;
  e->ns = d.PC <= 256 ? 4 : d.PC <= 512 ? 8 : d.PC <= 1024 ? 16 : 32;
  int pc_needed = d.PC;
  d.PC = e->ns * 64; e->d = d;  // pellet capacity == register file size: loads / stores need no bounds test
  if (pc_needed > 2048) { agarcl_destroy(e); return fail(AGARCL_E_UNSUPPORTED, "more than 2048 pellets per arena do not fit the pellet register file layout"); }
  if (e->lds_bytes > 160 * 1024) { agarcl_destroy(e); return fail(AGARCL_E_UNSUPPORTED, "arena does not fit the 160 KiB LDS of a CU (too many players, or virus / food capacities too large)"); }
  AgState &s = e->s; memset(&s, 0, sizeof(s));
  s.d = d; s.g = g;
  size_t A = (size_t)d.A;
  s.pel_xy = alloc<float>(e, A * d.PC * 2); s.pel_id = alloc<int32_t>(e, A * d.PC);
  s.vir_x = alloc<float>(e, A * d.VC); s.vir_y = alloc<float>(e, A * d.VC); s.vir_vx = alloc<float>(e, A * d.VC); s.vir_vy = alloc<float>(e, A * d.VC);
  s.vir_mass = alloc<int32_t>(e, A * d.VC); s.vir_hits = alloc<int32_t>(e, A * d.VC); s.vir_id = alloc<int32_t>(e, A * d.VC);
  s.food_x = alloc<float>(e, A * d.FC); s.food_y = alloc<float>(e, A * d.FC); s.food_vx = alloc<float>(e, A * d.FC); s.food_vy = alloc<float>(e, A * d.FC); s.food_id = alloc<int32_t>(e, A * d.FC);
  const int ag_ts_lg = d.ts_lg;
  const size_t At = AG_TILE_ARENAS(A);  // the tile-transposed arrays hold whole tiles
  e->stage_words = (size_t)d.P * CF_ALL * AG_CC; if (e->stage_words < 256) e->stage_words = 256;   // (also holds one arena's AR block when there are no players)
  e->d_stage = alloc<uint32_t>(e, e->stage_words);
  s.cells = alloc<uint32_t>(e, At * d.P * CF_ALL * AG_CC);
  s.pl = alloc<int32_t>(e, At * d.P * PL_WORDS); s.vticks = alloc<int32_t>(e, A * d.P * AG_VT_CAP); s.ar = alloc<int32_t>(e, At * AR_WORDS);
  s.mt = alloc<uint64_t>(e, A * 312); s.rnd = alloc<int32_t>(e, A * 35);
  s.scratch = d.P > 1 ? alloc<int32_t>(e, A * (size_t)AGM_WORDS) : nullptr;
  if (d.P > 1 && !s.scratch) { agarcl_destroy(e); return fail(AGARCL_E_NOMEM, "device allocation failed"); }
  s.rewards = alloc<double>(e, A * d.n_agents); s.dones = alloc<uint8_t>(e, A * d.n_agents); s.masses = alloc<int32_t>(e, A * d.n_agents);
  s.packed = alloc<float>(e, (size_t)AG_PACKED_SLOTS * A * d.n_agents * 2);
  s.counts = alloc<int32_t>(e, A * 4); s.ev_p = alloc<int32_t>(e, A * (size_t)(d.EC + d.EX)); s.ev_v = alloc<int32_t>(e, A * AG_EVV_CAP);
  e->d_act_dxdy = alloc<float>(e, A * d.n_agents * 2); e->d_act = alloc<int32_t>(e, A * d.n_agents);
  e->lut_r = alloc<float>(e, AG_LUT_SIZE); e->lut_ms = alloc<float>(e, AG_LUT_SIZE); e->lut_ss = alloc<float>(e, AG_LUT_SIZE); e->lut_anti = alloc<float>(e, AG_ANTI_LUT);
  if (!e->lut_anti || !s.ev_v || !s.mt || !s.cells || !s.pel_xy) { agarcl_destroy(e); return fail(AGARCL_E_NOMEM, "device allocation failed"); }
  s.lut_r = e->lut_r; s.lut_ms = e->lut_ms; s.lut_ss = e->lut_ss; s.lut_anti = e->lut_anti;
  s.act_dxdy = nullptr; s.act = nullptr;
  s.prof = alloc<unsigned long long>(e, (size_t)d.A * 16);
  s.qinfo = alloc<int32_t>(e, (size_t)d.A * 2);
  s.qcount = alloc<int32_t>(e, 2); e->parity = 0;
  s.sched = alloc<int32_t>(e, 2); e->sched_parity = 0;
  s.cost = alloc<uint32_t>(e, (size_t)d.A); s.order = alloc<int32_t>(e, (size_t)d.A); e->order_age = 0; e->order_ready = false;
  { const char *no = getenv("AGARCL_NO_ORDER"); e->no_order = no && no[0] == '1'; }
  // k_step's grid: 4096 single-wave workgroups is what stays resident (4 per SIMD).  AGARCL_KSTEP_GRID=<n> caps it lower: soaks and tests use it to
  // put small batches through the several-items-per-workgroup path (work counter, cost order)
#ifndef AG_KSTEP_MAXGRID
#define AG_KSTEP_MAXGRID 4096   // (measurement builds with another AG_KSTEP_ATTR: waves per SIMD x 1024)
#endif
  e->kstep_grid = AG_KSTEP_MAXGRID; { const char *kg = getenv("AGARCL_KSTEP_GRID"); if (kg) { int v = atoi(kg); if (v >= 1 && v <= AG_KSTEP_MAXGRID) e->kstep_grid = v; } }
  s.qlist = alloc<int32_t>(e, 2 * (size_t)d.A);
  s.qstat = alloc<int32_t>(e, 16); e->d_mask = alloc<uint8_t>(e, (size_t)d.A);
  // every array above is written through by some kernel (the event spill ev_p by pellets_eat / arena_store / k_quiet): one check for all
  if (e->alloc_failed) { agarcl_destroy(e); return fail(AGARCL_E_NOMEM, "device allocation failed"); }
  { const char *nf = getenv("AGARCL_NO_FRONT"); e->no_front = nf && nf[0] == '1'; }
  // modes 0-4 start agents at mass 25 (quiet-dominated); 5 and 6 start at mass 1000 (general path every step)
  // single launch (k_fused) wherever the batch fits 2048 wavefronts -- the 2 per SIMD its register footprint admits -- at
  // some lane-group size: up to 131072 arenas
  // (measured, C2, us per step by wavefront count: 2048 arenas 512 / 1024 / 2048 wavefronts 8.75 / 8.65 / 9.26; 4096 arenas
  // 9.47 / 9.23 / 9.81; 8192: 1024 / 2048: 10.23 / 10.63; 16384: 13.52 / 12.11 / 12.30 / 17.96 (512 ... 4096); 32768: 15.97 /
  // 14.09 / 13.83 / 20.20; 65536: 1024 / 2048 / 4096: 18.97 / 17.70 / 23.93; 131072: 2048 / 4096: 25.57 / 30.92)
  { const long target = d.A <= 16384 ? 1024L : AG_FUSED_WAVES;   // one wavefront per SIMD for the small batches, two beyond
    // (32 lanes per arena measured 1 % faster below 4096 arenas -- 2048 arenas: 8.75 -> 8.65 us -- not worth 16 more instantiations)
    e->fused_qg = 16; while (e->fused_qg > 1 && (long)d.A * e->fused_qg > target * 64L) e->fused_qg >>= 1; }
  { const char *w = getenv("AGARCL_FUSED_QG"); if (w) { int v = atoi(w); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) e->fused_qg = v; } }
  e->fused_ok = d.P == 1 && (long)d.A * e->fused_qg <= AG_FUSED_WAVES * 64L;
  e->fused = e->fused_ok && cfg->mode_number <= 4;  // starting point; poll_stats follows what the arenas actually do
  e->front_off = d.P == 1 && cfg->mode_number > 4; e->few_unfinished = false;
  e->work_step0 = e->work_front0 = 0; e->work_unf0 = 0; e->work_pass0 = 0;
  e->flags_seen = 0;
  e->fused_fixed = false; e->h_stat = nullptr; e->stat_ev = nullptr; e->next_poll = 2; e->poll_gap = 2; e->stat_pending = e->stat_stale_flags = false; e->step_no = e->front_runs = e->stat_req_front = e->stat_last_front = 0; e->stat_last_total = 0;
  { const char *fu = getenv("AGARCL_FUSED"); if (fu && (fu[0] == '0' || fu[0] == '1')) { e->fused = fu[0] == '1' && e->fused_ok; e->fused_fixed = true; } }
  // Lanes per arena of the lean front kernel.  The quiet tick is per-lane code that every lane of an arena's group carries
  // redundantly and a pellet pass is wave-wide whatever the group size, so the group size only sets how many wavefronts the
  // arenas make -- and from ~16k arenas on the launch is bound by issued wave-instructions.  Measured (MI355X, C2, us per
  // step, lanes per arena 16 / 8 / 4 / 2 / 1): 16384: 17.2 / 15.2 / 15.0 / 16.4 / 19.2, 65536: 42.0 / 30.7 / 26.3 / 25.4 /
  // 27.0, 262144: 137.7 / 90.4 / 71.6 / 66.6 / 64.8: best is ~2048 wavefronts (2 per SIMD) until one lane per arena is reached.
  e->quiet_qg = 8; while (e->quiet_qg > 1 && (long)d.A * e->quiet_qg > 2048L * 64) e->quiet_qg >>= 1;
  { const char *w = getenv("AGARCL_QUIET_QG"); if (w) { int v = atoi(w); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) e->quiet_qg = v; } }
  e->fused_wg = 256; { const char *w = getenv("AGARCL_FUSED_WG"); if (w) { int v = atoi(w); if (v == 64 || v == 128 || v == 256) e->fused_wg = v; } }
#ifndef AGAR_CPU_EMU
  {
    hipEvent_t ev;
    if (hipHostMalloc((void **)&e->h_stat, 8) == hipSuccess && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) { e->h_stat[0] = 0; e->h_stat[1] = 0; e->stat_ev = (void *)ev; }
    else { if (e->h_stat) (void)hipHostFree(e->h_stat); e->h_stat = nullptr; }
  }
#endif
#ifdef AGAR_CPU_EMU
  e->d_state = &e->s;
#else
  e->d_state = (AgState *)dmalloc(sizeof(AgState), e->stream);
  if (!e->d_state) { agarcl_destroy(e); return fail(AGARCL_E_NOMEM, "device allocation failed"); }
  e->allocs.push_back(e->d_state);
  if (h2d(e->d_state, &e->s, sizeof(AgState), e->stream)) { agarcl_destroy(e); return fail(AGARCL_E_HIP, "descriptor upload failed"); }
#endif
  {  // mass -> fp32 tables: the reference's double-precision islands, evaluated with the host libm
     // (core/utils.hpp:8-11; Engine.hpp:1296-1302; Engine.hpp:567)
    std::vector<float> r(AG_LUT_SIZE), ms(AG_LUT_SIZE), ss(AG_LUT_SIZE), an(AG_ANTI_LUT);
    for (int m = 0; m < AG_LUT_SIZE; m++) {
      double area = (double)(unsigned)m / 1.0;
      r[m] = (float)sqrt(area / 3.14159265358979323846);
      ms[m] = (float)(300.0 / pow((double)(unsigned)m, 0.439));
      double v = 3.0 * pow((double)ms[m], 1.2);
      double cl = (v < 130.0) ? v : 130.0; cl = (cl < 20.0) ? 20.0 : cl;
      ss[m] = (float)cl;
    }
    for (int k = 0; k < AG_ANTI_LUT; k++) an[k] = (float)pow(1.1, (double)k);
    if (h2d(e->lut_r, r.data(), r.size() * 4, e->stream) || h2d(e->lut_ms, ms.data(), ms.size() * 4, e->stream) ||
        h2d(e->lut_ss, ss.data(), ss.size() * 4, e->stream) || h2d(e->lut_anti, an.data(), an.size() * 4, e->stream)) { agarcl_destroy(e); return fail(AGARCL_E_HIP, "LUT upload failed"); }
  }
  {  // initial arena words: id counter 1 (first entity id 2, core/Ball.hpp:18,97), identity player order
    std::vector<int32_t> ar(AG_TILE_ARENAS(A) * AR_WORDS, 0);
    for (size_t a = 0; a < A; a++) { ar[tix(ag_ts_lg, a, AR_WORDS, AR_IDC)] = 1; ar[tix(ag_ts_lg, a, AR_WORDS, AR_MTIDX)] = 312; for (int p = 0; p < d.P; p++) ar[tix(ag_ts_lg, a, AR_WORDS, AR_ORDER0 + p)] = p; }
    if (h2d(s.ar, ar.data(), ar.size() * 4, e->stream)) { agarcl_destroy(e); return fail(AGARCL_E_HIP, "state upload failed"); }
  }
  *out = e;
  // the reference constructor resets once (BaseEnvironment.hpp:66)
  int rc = agarcl_seed(e, nullptr, base_seed);
  if (rc == 0) rc = agarcl_reset(e, nullptr, 1);
  if (rc != 0) { agarcl_destroy(e); *out = nullptr; return rc; }
  return AGARCL_OK;
}
extern "C" int agarcl_create(const agarcl_config *cfg, int32_t num_arenas, int32_t device, agarcl_env **out) { return create_env(cfg, num_arenas, device, 5489u, out); }

extern "C" int agarcl_set_stream(agarcl_env *e, void *hip_stream) {
  if (!e) return fail(AGARCL_E_INVALID, "null env");
#ifndef AGAR_CPU_EMU
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  // adopt the caller's stream; NULL is the legacy default stream (what torch.cuda.current_stream() is by default)
  if (e->own_stream) (void)hipStreamDestroy(e->stream);
  e->stream = (hipStream_t)hip_stream; e->own_stream = false;
#else
  (void)hip_stream;
#endif
  return AGARCL_OK;
}
extern "C" int agarcl_sync(agarcl_env *e) {
  if (!e) return fail(AGARCL_E_INVALID, "null env");
  return dsync(e->stream) ? fail(AGARCL_E_HIP, "stream synchronize failed") : AGARCL_OK;
}

// Two timing events on the env's stream, created once and without the system-scope fence an ordinary event record carries
// (hipEventDisableSystemFence): mark 0 = start, 1 = stop; elapsed = milliseconds between them (after the stream has passed both).
extern "C" int agarcl_timer_mark(agarcl_env *e, int32_t which) {
  if (!e || which < 0 || which > 1) return fail(AGARCL_E_INVALID, "agarcl_timer_mark: bad arguments");
#ifdef AGAR_CPU_EMU
  return AGARCL_OK;
#else
  HIPCHK(hipSetDevice(e->device));
  if (!e->timer_ev[0]) {
    for (int i = 0; i < 2; i++) {
      hipEvent_t ev = nullptr;
      if (hipEventCreateWithFlags(&ev, hipEventDisableSystemFence) != hipSuccess) return fail(AGARCL_E_HIP, "agarcl_timer_mark: event creation failed");
      e->timer_ev[i] = (void *)ev;
    }
  }
  HIPCHK(hipEventRecord((hipEvent_t)e->timer_ev[which], e->stream));
  return AGARCL_OK;
#endif
}
extern "C" int agarcl_timer_elapsed_ms(agarcl_env *e, float *ms) {
  if (!e || !ms) return fail(AGARCL_E_INVALID, "agarcl_timer_elapsed_ms: null pointer");
  *ms = 0.0f;
#ifndef AGAR_CPU_EMU
  if (!e->timer_ev[0]) return fail(AGARCL_E_INVALID, "agarcl_timer_elapsed_ms: no marks recorded");
  HIPCHK(hipEventSynchronize((hipEvent_t)e->timer_ev[1]));
  HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)e->timer_ev[0], (hipEvent_t)e->timer_ev[1]));
#endif
  return AGARCL_OK;
}

extern "C" int agarcl_seed(agarcl_env *e, const uint32_t *seeds_host, uint32_t base_seed) {
  if (!e) return fail(AGARCL_E_INVALID, "null env");
  size_t A = (size_t)e->d.A;
  e->seeds.resize(A);
  for (size_t a = 0; a < A; a++) e->seeds[a] = seeds_host ? seeds_host[a] : base_seed + (uint32_t)a;
  std::vector<uint64_t> mt(A * 312);
  for (size_t a = 0; a < A; a++) mt_seed_host(&mt[a * 312], (uint64_t)(seeds_host ? seeds_host[a] : base_seed + (uint32_t)a));
  if (h2d(e->s.mt, mt.data(), mt.size() * 8, e->stream)) return fail(AGARCL_E_HIP, "seed upload failed");
  std::vector<int32_t> rnd(A * 35);
  for (size_t a = 0; a < A; a++) glibc_srand_host(&rnd[a * 35], seeds_host ? seeds_host[a] : base_seed + (uint32_t)a);
  if (h2d(e->s.rnd, rnd.data(), rnd.size() * 4, e->stream)) return fail(AGARCL_E_HIP, "seed upload failed");
#ifdef AGAR_CPU_EMU
  for (size_t a = 0; a < A; a++) e->s.ar[tix(e->d.ts_lg, a, AR_WORDS, AR_MTIDX)] = 312;
#else
  HIPCHK(hipSetDevice(e->device));
  hipLaunchKernelGGL(k_set_ar_word, dim3((e->d.A + 255) / 256), dim3(256), 0, e->stream, e->s.ar, e->d.ts_lg, (int)AR_MTIDX, e->d.A, 312);
  HIPCHK(hipGetLastError());
#endif
  return AGARCL_OK;
}

// Engine::seed for ONE arena (Engine.hpp:242-245): used by snapshot loading, which ends with seed(json["seed"]).
extern "C" int agarcl_seed_arena(agarcl_env *e, int32_t arena, uint32_t seed) {
  if (!e || arena < 0 || arena >= e->d.A) return fail(AGARCL_E_INVALID, "agarcl_seed_arena: bad arguments");
  std::vector<uint64_t> mt(312); mt_seed_host(mt.data(), (uint64_t)seed);
  std::vector<int32_t> rnd(35); glibc_srand_host(rnd.data(), seed);
  int32_t idx = 312;
  if (h2d(e->s.mt + (size_t)arena * 312, mt.data(), 312 * 8, e->stream) || h2d(e->s.rnd + (size_t)arena * 35, rnd.data(), 35 * 4, e->stream) ||
      h2d(e->s.ar + tix(e->d.ts_lg, (size_t)arena, AR_WORDS, AR_MTIDX), &idx, 4, e->stream)) return fail(AGARCL_E_HIP, "seed upload failed");
  // (BaseEnvironment::seed_ -- what a later save writes as "seed" -- is NOT touched by a load: Engine::seed is called
  // directly, Engine.hpp:347; e->seeds therefore keeps the last agarcl_seed value)
  return AGARCL_OK;
}
extern "C" int agarcl_get_seeds(agarcl_env *e, uint32_t *out) {
  if (!e || !out) return fail(AGARCL_E_INVALID, "agarcl_get_seeds: null pointer");
  for (int a = 0; a < e->d.A; a++) out[a] = (size_t)a < e->seeds.size() ? e->seeds[(size_t)a] : 0u;
  return AGARCL_OK;
}
// Raw per-arena words (AR_*) and per-player words (PL_*, slot-major): introspection for the snapshot code and tests.
static_assert(AR_WORDS == AGARCL_ARENA_WORDS && PL_WORDS == AGARCL_PLAYER_WORDS, "include/agarcl_batch.h must state the word counts of agar_types.h");
extern "C" int agarcl_player_words(void) { return PL_WORDS; }
extern "C" int agarcl_get_arena_words(agarcl_env *e, int32_t arena, int32_t *ar_out, int32_t *pl_out) {
  if (!e || arena < 0 || arena >= e->d.A) return fail(AGARCL_E_INVALID, "agarcl_get_arena_words: bad arguments");
  std::vector<int32_t> t;
  if (ar_out) { if (pull_t(e, t, e->s.ar, (size_t)arena, AR_WORDS)) return fail(AGARCL_E_HIP, "copy failed"); memcpy(ar_out, t.data(), t.size() * 4); }
  if (pl_out) { if (pull_t(e, t, e->s.pl, (size_t)arena, (size_t)e->d.P * PL_WORDS)) return fail(AGARCL_E_HIP, "copy failed"); memcpy(pl_out, t.data(), t.size() * 4); }
  return AGARCL_OK;
}

#ifndef AGAR_CPU_EMU
// Every reset restarts the flag watch (agarcl_poll_flags): the device word is zeroed in front of k_reset, whose not-reset arenas OR their
// flags back in (a full reset leaves none), the host's accumulated view is dropped, and a statistics sample that is still in flight --
// taken before this reset -- must not bring the old flags back.
// `resample`: the reset covers every arena (or a state was loaded), so the arenas may behave quite unlike what the step form was chosen for:
// the step-form statistics are sampled 2, 8 and 32 steps later.  A MASKED reset leaves the sampling cadence alone -- a learner that calls
// reset(mask = dones) after every step would otherwise pay a copy and an event record every other step for ever, and feed the fused /
// two-kernel choice from 2-step windows.
static int restart_flag_watch(agarcl_env *e, bool resample) {
  if (hipMemsetAsync(e->s.qstat + 1, 0, 4, e->stream) != hipSuccess) return fail(AGARCL_E_HIP, "flag watch reset failed");
  e->flags_seen = 0; e->stat_stale_flags = e->stat_pending;
  if (resample) { e->poll_gap = 2; e->next_poll = e->step_no + 2; }
  return 0;
}
#endif
extern "C" int agarcl_reset(agarcl_env *e, const uint8_t *mask_host, int32_t reset_ids) {
  if (!e) return fail(AGARCL_E_INVALID, "null env");
  uint8_t *mask_dev = nullptr;
#ifndef AGAR_CPU_EMU
  HIPCHK(hipSetDevice(e->device));
  if (mask_host) {  // stream-ordered upload into the env's own mask buffer: no allocation, no device-wide sync
    mask_dev = e->d_mask;
    HIPCHK(hipMemcpyAsync(mask_dev, mask_host, (size_t)e->d.A, hipMemcpyHostToDevice, e->stream));
  }
  bool most = !mask_host;   // an unmasked reset, or a mask that covers at least a quarter of the arenas, changes what the batch looks like
  if (mask_host) { size_t n = 0; for (int a = 0; a < e->d.A; a++) n += mask_host[a] != 0; most = 4 * n >= (size_t)e->d.A; }
  if (restart_flag_watch(e, most)) return AGARCL_E_HIP;
#endif
  return launch_reset(e, mask_dev, mask_host, reset_ids);
}
extern "C" int agarcl_reset_device(agarcl_env *e, const uint8_t *mask_dev, int32_t reset_ids) {
  if (!e || !mask_dev) return fail(AGARCL_E_INVALID, "agarcl_reset_device: null pointer");
#ifdef AGAR_CPU_EMU
  return launch_reset(e, nullptr, mask_dev, reset_ids);  // (test-only host build: "device" memory is host memory)
#else
  HIPCHK(hipSetDevice(e->device));
  if (restart_flag_watch(e, false)) return AGARCL_E_HIP;   // (a device mask is the auto-reset loop's: the sampling cadence stays)
  return launch_reset(e, mask_dev, nullptr, reset_ids);
#endif
}

extern "C" int agarcl_set_actions(agarcl_env *e, const float *dxdy, const int32_t *act, int32_t on_device) {
  if (!e || !dxdy || !act) return fail(AGARCL_E_INVALID, "agarcl_set_actions: null pointer");
  size_t n = (size_t)e->d.A * e->d.n_agents;
  if (on_device) { e->act_dxdy = dxdy; e->act = act; return AGARCL_OK; }
  if (h2d(e->d_act_dxdy, dxdy, n * 8, e->stream) || h2d(e->d_act, act, n * 4, e->stream)) return fail(AGARCL_E_HIP, "action upload failed");
  e->act_dxdy = e->d_act_dxdy; e->act = e->d_act;
  return AGARCL_OK;
}

extern "C" int agarcl_step(agarcl_env *e, int32_t ticks) {
  if (!e) return fail(AGARCL_E_INVALID, "null env");
#ifndef AGAR_CPU_EMU
  HIPCHK(hipSetDevice(e->device));
#endif
  int rc = launch_step(e, ticks > 0 ? ticks : e->cfg.ticks_per_step, 1);
  e->slot = (e->slot + 1) % AG_PACKED_SLOTS;
  return rc;
}
extern "C" int agarcl_step_actions(agarcl_env *e, const float *dxdy_dev, const int32_t *act_dev, int32_t ticks) {
  if (!e || !dxdy_dev || !act_dev) return fail(AGARCL_E_INVALID, "agarcl_step_actions: null pointer");
  e->act_dxdy = dxdy_dev; e->act = act_dev;
  return agarcl_step(e, ticks);
}
extern "C" const float *agarcl_packed_dev(agarcl_env *e, int32_t slot) { return e ? e->s.packed + (size_t)(((slot % AG_PACKED_SLOTS) + AG_PACKED_SLOTS) % AG_PACKED_SLOTS) * e->d.A * e->d.n_agents * 2 : nullptr; }
extern "C" int agarcl_last_slot(agarcl_env *e) { return e ? (e->slot + AG_PACKED_SLOTS - 1) % AG_PACKED_SLOTS : 0; }
extern "C" int agarcl_tick(agarcl_env *e, int32_t ticks) {
  if (!e) return fail(AGARCL_E_INVALID, "null env");
#ifndef AGAR_CPU_EMU
  HIPCHK(hipSetDevice(e->device));
#endif
  return launch_step(e, ticks > 0 ? ticks : 1, 0);
}

extern "C" int agarcl_set_targets(agarcl_env *e, const float *txy_host, const int32_t *act_host) {
  if (!e || !txy_host || !act_host) return fail(AGARCL_E_INVALID, "agarcl_set_targets: null pointer");
  const int ag_ts_lg = e->d.ts_lg;
  const size_t P = (size_t)e->d.P, n = (size_t)e->d.A * P, R = P * PL_WORDS;
  std::vector<int32_t> pl(AG_TILE_ARENAS(e->d.A) * R);
  if (d2h(pl.data(), e->s.pl, pl.size() * 4, e->stream)) return fail(AGARCL_E_HIP, "download failed");
  for (size_t i = 0; i < n; i++) {
    const size_t a = i / P, w0 = (i % P) * PL_WORDS;
    memcpy(&pl[tix(ag_ts_lg, a, R, w0 + PL_TX)], &txy_host[2 * i], 4); memcpy(&pl[tix(ag_ts_lg, a, R, w0 + PL_TY)], &txy_host[2 * i + 1], 4);
    pl[tix(ag_ts_lg, a, R, w0 + PL_ACTION)] = act_host[i];
  }
  if (h2d(e->s.pl, pl.data(), pl.size() * 4, e->stream)) return fail(AGARCL_E_HIP, "upload failed");
  return AGARCL_OK;
}

extern "C" int agarcl_respawn_dead(agarcl_env *e) {
  if (!e) return fail(AGARCL_E_INVALID, "null env");
#ifndef AGAR_CPU_EMU
  HIPCHK(hipSetDevice(e->device));
#endif
  return launch_respawn(e);
}

extern "C" const double *agarcl_rewards_dev(agarcl_env *e) { return e ? e->s.rewards : nullptr; }
extern "C" const uint8_t *agarcl_dones_dev(agarcl_env *e) { return e ? e->s.dones : nullptr; }
extern "C" const int32_t *agarcl_masses_dev(agarcl_env *e) { return e ? e->s.masses : nullptr; }

#define GETTER(name, field, type, count) \
  extern "C" int name(agarcl_env *e, type *out) { \
    if (!e || !out) return fail(AGARCL_E_INVALID, #name ": null pointer"); \
    return d2h(out, e->s.field, (size_t)(count) * sizeof(type), e->stream) ? fail(AGARCL_E_HIP, #name ": copy failed") : AGARCL_OK; }
GETTER(agarcl_get_rewards, rewards, double, e->d.A * e->d.n_agents)
GETTER(agarcl_get_dones, dones, uint8_t, e->d.A * e->d.n_agents)
GETTER(agarcl_get_masses, masses, int32_t, e->d.A * e->d.n_agents)
GETTER(agarcl_get_counts, counts, int32_t, e->d.A * 4)

extern "C" int agarcl_get_flags(agarcl_env *e, uint32_t *out) {
  if (!e || !out) return fail(AGARCL_E_INVALID, "agarcl_get_flags: null pointer");
  const int ag_ts_lg = e->d.ts_lg;
  std::vector<int32_t> ar(AG_TILE_ARENAS(e->d.A) * AR_WORDS);
  if (d2h(ar.data(), e->s.ar, ar.size() * 4, e->stream)) return fail(AGARCL_E_HIP, "copy failed");
  for (int a = 0; a < e->d.A; a++) out[a] = (uint32_t)ar[tix(ag_ts_lg, (size_t)a, AR_WORDS, AR_FLAGS)];
  return AGARCL_OK;
}

extern "C" int agarcl_poll_flags(agarcl_env *e, uint32_t *out) {
  if (!e || !out) return fail(AGARCL_E_INVALID, "agarcl_poll_flags: null pointer");
#ifdef AGAR_CPU_EMU
  *out = (uint32_t)e->s.qstat[1];
#else
  if (e->stat_pending && !e->stat_stale_flags && hipEventQuery((hipEvent_t)e->stat_ev) == hipSuccess) e->flags_seen |= (uint32_t)e->h_stat[1];
  *out = e->flags_seen;
#endif
  return AGARCL_OK;
}

extern "C" int agarcl_get_events(agarcl_env *e, int32_t *n_events_host, int32_t *pellet_idx_host, int32_t cap, int32_t *virus_idx_host, int32_t cap_v) {
  if (!e || !n_events_host) return fail(AGARCL_E_INVALID, "agarcl_get_events: null pointer");
  size_t A = (size_t)e->d.A; const int ag_ts_lg = e->d.ts_lg;
  const size_t EC = (size_t)(e->d.EC + e->d.EX);   // (the arena's stride: exported LDS events, then the spill area)
  // only the first min(cap, stride) entries of every arena travel (one strided copy): the spill area of a dense batch is gigabytes, the caller's
  // buffer says how much of it is wanted
  const size_t W = !pellet_idx_host || cap <= 0 ? 0 : ((size_t)cap < EC ? (size_t)cap : EC);
  std::vector<int32_t> ar(AG_TILE_ARENAS(A) * AR_WORDS), evp(A * W), evv(A * AG_EVV_CAP);
  if (d2h(ar.data(), e->s.ar, ar.size() * 4, e->stream) || d2h(evv.data(), e->s.ev_v, evv.size() * 4, e->stream)) return fail(AGARCL_E_HIP, "copy failed");
  if (W) {
#ifdef AGAR_CPU_EMU
    for (size_t a = 0; a < A; a++) memcpy(&evp[a * W], e->s.ev_p + a * EC, W * 4);
#else
    if (hipMemcpy2DAsync(evp.data(), W * 4, e->s.ev_p, EC * 4, W * 4, A, hipMemcpyDeviceToHost, e->stream) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
      return fail(AGARCL_E_HIP, "copy failed");
#endif
  }
  for (size_t a = 0; a < A; a++) {
    int np = ar[tix(ag_ts_lg, a, AR_WORDS, AR_NEVP)], nv = ar[tix(ag_ts_lg, a, AR_WORDS, AR_NEVV)];
    n_events_host[2 * a] = np; n_events_host[2 * a + 1] = nv;
    if (pellet_idx_host) for (int i = 0; i < np && i < (int)W; i++) pellet_idx_host[a * cap + i] = evp[a * W + i];
    if (virus_idx_host) for (int i = 0; i < nv && i < cap_v && i < AG_EVV_CAP; i++) virus_idx_host[a * cap_v + i] = evv[a * AG_EVV_CAP + i];
  }
  return AGARCL_OK;
}

extern "C" int agarcl_debug_prof(agarcl_env *e, unsigned long long *out16, int reset) {
  if (!e || !out16) return AGARCL_E_INVALID;
  std::vector<unsigned long long> h((size_t)e->d.A * 16);
  if (d2h(h.data(), e->s.prof, h.size() * 8, e->stream)) return AGARCL_E_HIP;
  for (int i = 0; i < 16; i++) { out16[i] = 0; for (int a = 0; a < e->d.A; a++) out16[i] += h[(size_t)a * 16 + i]; }
  if (reset) { std::fill(h.begin(), h.end(), 0ull); if (h2d(e->s.prof, h.data(), h.size() * 8, e->stream)) return AGARCL_E_HIP; }
  return AGARCL_OK;
}
extern "C" int agarcl_debug_fused(agarcl_env *e) { return e ? (e->fused ? 1 : 0) : -1; }  // current step mode (diagnostics / tests)
extern "C" int agarcl_debug_qinfo(agarcl_env *e, int32_t *out) {  // [A][2]: the front kernel's hand-over words of the last step (diagnostics)
  if (!e || !out) return AGARCL_E_INVALID;
  if (d2h(out, e->s.qinfo, (size_t)e->d.A * 2 * 4, e->stream)) return AGARCL_E_HIP;
  return AGARCL_OK;
}
extern "C" int agarcl_debug_prof_raw(agarcl_env *e, unsigned long long *out) {  // [A][16], diagnostic builds
  if (!e || !out) return AGARCL_E_INVALID;
  if (d2h(out, e->s.prof, (size_t)e->d.A * 16 * 8, e->stream)) return AGARCL_E_HIP;
  return AGARCL_OK;
}
// Work the step kernels did since the last reset of these counters, counted by the kernels themselves (bench.py turns it
// into "bytes requested": DESIGN.md section 5).  [0] arena-steps the lean front part finished, [1] arena-steps that needed the
// general engine, [2] pellet passes (each reads the arena's pellet array once), [3] spare.
extern "C" int agarcl_debug_work(agarcl_env *e, int64_t *out4, int reset) {
  if (!e || !out4) return AGARCL_E_INVALID;
  const int ag_ts_lg = e->d.ts_lg;
  const size_t R = (size_t)e->d.P * PL_WORDS, n = AG_TILE_ARENAS(e->d.A) * R;
  std::vector<int32_t> pl(n); int32_t st[4];
  if (d2h(pl.data(), e->s.pl, n * 4, e->stream) || d2h(st, e->s.qstat, 16, e->stream)) return AGARCL_E_HIP;
  int64_t passes = 0;
  for (size_t a = 0; a < (size_t)e->d.A; a++) for (size_t p = 0; p < (size_t)e->d.P; p++) passes += (uint32_t)pl[tix(ag_ts_lg, a, R, p * PL_WORDS + PL_PASSES)];
  const int64_t steps = (int64_t)(e->step_no - e->work_step0) * e->d.A, unfinished = (int64_t)(uint32_t)(st[0] - e->work_unf0) + (int64_t)(e->step_no - e->work_step0 - (e->front_runs - e->work_front0)) * e->d.A;
  out4[0] = steps - unfinished; out4[1] = unfinished; out4[2] = passes - e->work_pass0; out4[3] = 0;
  if (reset) { e->work_step0 = e->step_no; e->work_front0 = e->front_runs; e->work_unf0 = st[0]; e->work_pass0 = passes; }
  return AGARCL_OK;
}
extern "C" int agarcl_debug_qstat(agarcl_env *e, int32_t *out16) {  // raw statistics words (AgState::qstat); [2..11] only in diagnostic builds
  if (!e || !out16) return AGARCL_E_INVALID;
  return d2h(out16, e->s.qstat, 64, e->stream) ? AGARCL_E_HIP : AGARCL_OK;
}
// Exhaustive check of the relaxation's short square root (agar_core.inl ag_sqrtf_lean) against the compiler's correctly rounded sqrtf: all
// 2^32 bit patterns, compared bit for bit.  out2[0] = patterns that differ, out2[1] = the lowest such pattern (valid when out2[0] != 0).
#ifndef AGAR_CPU_EMU
__global__ void k_sqrt_check(unsigned long long *out) {
  unsigned long long bad = 0ull; unsigned first = 0xffffffffu;
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;   // (nth divides 2^32: every wave makes the same number of rounds)
  for (unsigned long long i = tid; i < (1ull << 32); i += nth) {
    const float x = __int_as_float((int)(unsigned)i);
    const unsigned a = (unsigned)__float_as_int(ag_sqrtf_lean(x)), b = (unsigned)__float_as_int(__builtin_sqrtf(x));
    if (a != b) { bad++; if ((unsigned)i < first) first = (unsigned)i; }
  }
  if (bad) { atomicAdd(out, bad); atomicMin(out + 1, (unsigned long long)first); }
}
#endif
extern "C" int agarcl_debug_sqrt_check(agarcl_env *e, unsigned long long *out2) {
  if (!e || !out2) return AGARCL_E_INVALID;
  out2[0] = 0ull; out2[1] = ~0ull;
#ifndef AGAR_CPU_EMU   // (the host emulation has one square root only)
  unsigned long long *d = (unsigned long long *)dmalloc(16, e->stream);   // (freed below: repeated calls must not grow the env's allocation list)
  if (!d || h2d(d, out2, 16, e->stream)) { dfree(d); return fail(AGARCL_E_HIP, "sqrt check: allocation failed"); }
  hipLaunchKernelGGL(k_sqrt_check, dim3(4096), dim3(256), 0, e->stream, d);
  const int rc_ = d2h(out2, d, 16, e->stream);
  dfree(d);
  if (rc_) return fail(AGARCL_E_HIP, "sqrt check failed");
#endif
  return AGARCL_OK;
}
extern "C" int agarcl_num_arenas(agarcl_env *e) { return e ? e->d.A : 0; }
extern "C" int agarcl_players_per_arena(agarcl_env *e) { return e ? e->d.P : 0; }

// ---- state blobs (oracle/BLOB_FORMAT.md) ------------------------------------------------------------------
namespace {
struct ArenaHost {
  std::vector<int32_t> ar, pl, vt;
  std::vector<float> pxy, vx, vy, vvx, vvy, fx, fy, fvx, fvy;
  std::vector<int32_t> pid, vm, vh, vid, fid; std::vector<uint32_t> cells;
};
template <class T> int pull(agarcl_env *e, std::vector<T> &h, const T *dev, size_t off, size_t n) { h.resize(n); return n ? d2h(h.data(), dev + off, n * sizeof(T), e->stream) : 0; }
template <class T> int push(agarcl_env *e, const std::vector<T> &h, T *dev, size_t off) { return h.empty() ? 0 : h2d(dev + off, h.data(), h.size() * sizeof(T), e->stream); }
}  // namespace

extern "C" int agarcl_dump_arena(agarcl_env *e, int32_t arena, uint32_t *buf, int32_t cap) {
  if (!e || !buf || arena < 0 || arena >= e->d.A) return fail(AGARCL_E_INVALID, "agarcl_dump_arena: bad arguments");
  const AgDims &d = e->d; const AgState &s = e->s; size_t a = (size_t)arena; ArenaHost h; int rc = 0;
  rc |= pull_t(e, h.ar, s.ar, a, AR_WORDS); rc |= pull_t(e, h.pl, s.pl, a, (size_t)d.P * PL_WORDS); rc |= pull(e, h.vt, s.vticks, a * d.P * AG_VT_CAP, (size_t)d.P * AG_VT_CAP);
  if (rc) return fail(AGARCL_E_HIP, "copy failed");
  size_t np = (size_t)h.ar[AR_NPEL], nv = (size_t)h.ar[AR_NVIR], nf = (size_t)h.ar[AR_NFOOD];
  rc |= pull(e, h.pxy, s.pel_xy, a * d.PC * 2, np * 2); rc |= pull(e, h.pid, s.pel_id, a * d.PC, np);
  rc |= pull(e, h.vx, s.vir_x, a * d.VC, nv); rc |= pull(e, h.vy, s.vir_y, a * d.VC, nv); rc |= pull(e, h.vvx, s.vir_vx, a * d.VC, nv); rc |= pull(e, h.vvy, s.vir_vy, a * d.VC, nv);
  rc |= pull(e, h.vm, s.vir_mass, a * d.VC, nv); rc |= pull(e, h.vh, s.vir_hits, a * d.VC, nv); rc |= pull(e, h.vid, s.vir_id, a * d.VC, nv);
  rc |= pull(e, h.fx, s.food_x, a * d.FC, nf); rc |= pull(e, h.fy, s.food_y, a * d.FC, nf); rc |= pull(e, h.fvx, s.food_vx, a * d.FC, nf); rc |= pull(e, h.fvy, s.food_vy, a * d.FC, nf); rc |= pull(e, h.fid, s.food_id, a * d.FC, nf);
  size_t nc = (size_t)d.P * CF_ALL * AG_CC;
  rc |= pull_t(e, h.cells, s.cells, a, nc);
  if (rc) return fail(AGARCL_E_HIP, "copy failed");
  std::vector<uint32_t> o;
  auto F = [&](float f) { uint32_t u; memcpy(&u, &f, 4); o.push_back(u); };
  o.push_back(0x31524741u); o.push_back((uint32_t)h.ar[AR_TICKS]); o.push_back((uint32_t)h.ar[AR_IDC]); o.push_back((uint32_t)h.ar[AR_NEXT_PID]);
  o.push_back((uint32_t)np); o.push_back((uint32_t)nv); o.push_back((uint32_t)nf); o.push_back((uint32_t)d.P);
  for (size_t i = 0; i < np; i++) F(h.pxy[2 * i]);
  for (size_t i = 0; i < np; i++) F(h.pxy[2 * i + 1]);
  for (int32_t v : h.pid) o.push_back((uint32_t)v);
  for (float v : h.vx) F(v); for (float v : h.vy) F(v); for (float v : h.vvx) F(v); for (float v : h.vvy) F(v);
  for (int32_t v : h.vm) o.push_back((uint32_t)v); for (int32_t v : h.vh) o.push_back((uint32_t)v); for (int32_t v : h.vid) o.push_back((uint32_t)v);
  for (float v : h.fx) F(v); for (float v : h.fy) F(v); for (float v : h.fvx) F(v); for (float v : h.fvy) F(v); for (int32_t v : h.fid) o.push_back((uint32_t)v);
  uint32_t clock = (uint32_t)h.ar[AR_CLOCK];
  for (int k = 0; k < d.P; k++) {
    int p = h.ar[AR_ORDER0 + k]; const int32_t *P = &h.pl[(size_t)p * PL_WORDS];
    o.push_back((uint32_t)P[PL_PID]); o.push_back(P[PL_KIND] != 0 ? 1u : 0u); o.push_back((uint32_t)P[PL_NCELLS]); o.push_back((uint32_t)P[PL_ACTION]);
    o.push_back((uint32_t)P[PL_TX]); o.push_back((uint32_t)P[PL_TY]); o.push_back((uint32_t)P[PL_SPLIT_CD]); o.push_back((uint32_t)P[PL_FEED_CD]);
    o.push_back((uint32_t)P[PL_ELAPSED]); o.push_back((uint32_t)P[PL_LAST_DECAY]); o.push_back((uint32_t)P[PL_ANTI_TEAM]);
    o.push_back((uint32_t)P[PL_FOOD_EATEN]); o.push_back((uint32_t)P[PL_HIGHEST_MASS]); o.push_back((uint32_t)P[PL_CELLS_EATEN]); o.push_back((uint32_t)P[PL_VIRUSES_EATEN]);
    o.push_back((uint32_t)P[PL_MIN_MASS]); o.push_back((uint32_t)P[PL_NVTICKS]);
    for (int i = 0; i < P[PL_NVTICKS]; i++) o.push_back((uint32_t)h.vt[(size_t)p * AG_VT_CAP + i]);
    const uint32_t *C = &h.cells[(size_t)p * CF_ALL * AG_CC];
    for (int i = 0; i < P[PL_NCELLS]; i++) {
      for (int f = CF_X; f <= CF_ID; f++) o.push_back(C[AG_CELL_HW(f, i)]);
      uint32_t dl = C[AG_CELL_HW(CF_DL, i)];
      o.push_back(dl > clock ? dl - clock : 0u);
    }
  }
  if ((int64_t)o.size() > cap) return -(int)o.size() - 1000;
  memcpy(buf, o.data(), o.size() * 4);
  return (int)o.size();
}

// kinds == nullptr: the blob's players must be the arena's players (same pids, same iteration order).
// kinds != nullptr ("adopt"): the blob brings a NEW player set, as Engine::load_env_state creates one -- players in the
// map's iteration order with fresh pids; kinds[k] = AG_KIND_* of the k-th blob player.  Non-bots take slots 0.. in that
// order (BaseEnvironment::load_env_state rebuilds pids_ from the map, BaseEnvironment.hpp:324-335), bots follow.
static int load_arena_impl(agarcl_env *e, int32_t arena, const uint32_t *b, int32_t words, const int32_t *kinds, int hm_buckets, int hm_next_resize) {
  if (!e || !b || arena < 0 || arena >= e->d.A || words < 8 || b[0] != 0x31524741u) return fail(AGARCL_E_INVALID, "agarcl_load_arena: bad arguments");
  const AgDims &d = e->d; const AgState &s = e->s; size_t a = (size_t)arena; ArenaHost h;
  uint32_t np = b[4], nv = b[5], nf = b[6], npl = b[7];
  if ((int)npl != d.P) return fail(AGARCL_E_INVALID, "agarcl_load_arena: player count mismatch");
  if (np > (uint32_t)d.PC || nv > (uint32_t)d.VC || nf > (uint32_t)d.FC) return fail(AGARCL_E_CAPACITY, "agarcl_load_arena: blob exceeds arena capacities");
  // the entity tables and every player's 17 header words must lie inside the blob before anything of them is read
  if ((uint64_t)8 + 3ull * np + 7ull * nv + 5ull * nf + 17ull * npl > (uint64_t)words) return fail(AGARCL_E_INVALID, "agarcl_load_arena: blob length mismatch");
  const uint32_t *const b_end = b + words;
  if (pull_t(e, h.ar, s.ar, a, AR_WORDS) || pull_t(e, h.pl, s.pl, a, (size_t)d.P * PL_WORDS)) return fail(AGARCL_E_HIP, "copy failed");
  h.vt.assign((size_t)d.P * AG_VT_CAP, 0);
  auto U2F = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
  const uint32_t *p = b + 8;
  h.ar[AR_TICKS] = (int32_t)b[1]; h.ar[AR_IDC] = (int32_t)b[2]; h.ar[AR_NEXT_PID] = (int32_t)b[3];
  h.ar[AR_NPEL] = (int32_t)np; h.ar[AR_NVIR] = (int32_t)nv; h.ar[AR_NFOOD] = (int32_t)nf; h.ar[AR_SAFE] = 0;
  for (uint32_t i = 0; i < np; i++) { h.pxy.push_back(U2F(p[i])); h.pxy.push_back(U2F(p[np + i])); h.pid.push_back((int32_t)p[2 * np + i]); }
  h.pxy.resize((size_t)d.PC * 2, AG_PEL_SENTINEL);  // HBM invariant: sentinel at every index >= n_pellets
  p += 3 * np;
  for (uint32_t i = 0; i < nv; i++) { h.vx.push_back(U2F(p[i])); h.vy.push_back(U2F(p[nv + i])); h.vvx.push_back(U2F(p[2 * nv + i])); h.vvy.push_back(U2F(p[3 * nv + i]));
    h.vm.push_back((int32_t)p[4 * nv + i]); h.vh.push_back((int32_t)p[5 * nv + i]); h.vid.push_back((int32_t)p[6 * nv + i]); }
  p += 7 * nv;
  for (uint32_t i = 0; i < nf; i++) { h.fx.push_back(U2F(p[i])); h.fy.push_back(U2F(p[nf + i])); h.fvx.push_back(U2F(p[2 * nf + i])); h.fvy.push_back(U2F(p[3 * nf + i])); h.fid.push_back((int32_t)p[4 * nf + i]); }
  p += 5 * nf;
  size_t nc = (size_t)d.P * CF_ALL * AG_CC;
  h.cells.assign(nc, 0);  // cache words 0 => invalid (no cell has mass 0): the first tick refreshes it
  uint32_t clock = (uint32_t)h.ar[AR_CLOCK];
  if (kinds) {
    int nonbots = 0; for (int k = 0; k < d.P; k++) nonbots += kinds[k] == 0;
    if (nonbots != d.n_agents) return fail(AGARCL_E_INVALID, "agarcl_adopt_arena: number of non-bot players differs from num_agents");
    int next_agent = 0, next_bot = d.n_agents;
    for (int k = 0; k < d.P; k++) h.ar[AR_ORDER0 + k] = kinds[k] == 0 ? next_agent++ : next_bot++;
    h.ar[AR_HM_BUCKETS] = hm_buckets; h.ar[AR_HM_RESIZE] = hm_next_resize;
    h.ar[AR_DONE] = 0; h.ar[AR_RESPAWNED] = 0; h.ar[AR_FLAGS] = 0; h.ar[AR_NEVP] = 0; h.ar[AR_NEVV] = 0;
  }
  for (int k = 0; k < d.P; k++) {
    int slot = h.ar[AR_ORDER0 + k]; int32_t *P = &h.pl[(size_t)slot * PL_WORDS];
    if (kinds) { for (int w = 0; w < PL_WORDS; w++) P[w] = 0; P[PL_PID] = (int32_t)p[0]; P[PL_KIND] = kinds[k]; }
    else if ((int32_t)p[0] != P[PL_PID]) return fail(AGARCL_E_INVALID, "agarcl_load_arena: pid / iteration order mismatch");
    if (p + 17 > b_end) return fail(AGARCL_E_INVALID, "agarcl_load_arena: blob length mismatch");
    uint32_t ncell = p[2];
    if ((int)ncell > AG_CC) return fail(AGARCL_E_CAPACITY, "agarcl_load_arena: too many cells");
    P[PL_NCELLS] = (int32_t)ncell; P[PL_ACTION] = (int32_t)p[3]; P[PL_TX] = (int32_t)p[4]; P[PL_TY] = (int32_t)p[5]; P[PL_SPLIT_CD] = (int32_t)p[6]; P[PL_FEED_CD] = (int32_t)p[7];
    P[PL_ELAPSED] = (int32_t)p[8]; P[PL_LAST_DECAY] = (int32_t)p[9]; P[PL_ANTI_TEAM] = (int32_t)p[10]; P[PL_FOOD_EATEN] = (int32_t)p[11]; P[PL_HIGHEST_MASS] = (int32_t)p[12];
    P[PL_CELLS_EATEN] = (int32_t)p[13]; P[PL_VIRUSES_EATEN] = (int32_t)p[14]; P[PL_MIN_MASS] = (int32_t)p[15];
    uint32_t nt = p[16];
    if (nt > AG_VT_CAP) return fail(AGARCL_E_CAPACITY, "agarcl_load_arena: too many virus ticks");
    if (p + 17 + nt + 9ull * ncell > b_end) return fail(AGARCL_E_INVALID, "agarcl_load_arena: blob length mismatch");
    P[PL_NVTICKS] = (int32_t)nt; P[PL_CAND_IDX] = -1;   // (AR_SAFE = 0 above: a loaded state has neither a pellet-free disc nor a tracked pellet)
    for (uint32_t i = 0; i < nt; i++) h.vt[(size_t)slot * AG_VT_CAP + i] = (int32_t)p[17 + i];
    p += 17 + nt;
    uint32_t *C = &h.cells[(size_t)slot * CF_ALL * AG_CC];
    for (uint32_t i = 0; i < ncell; i++, p += 9) {
      for (int f = CF_X; f <= CF_SY; f++) C[AG_CELL_HW(f, i)] = p[f];
      C[AG_CELL_HW(CF_M, i)] = p[6] > AG_CELL_MIN_SIZE ? p[6] : AG_CELL_MIN_SIZE; C[AG_CELL_HW(CF_ID, i)] = p[7]; C[AG_CELL_HW(CF_DL, i)] = clock + p[8];
    }
  }
  if (p - b != words) return fail(AGARCL_E_INVALID, "agarcl_load_arena: blob length mismatch");
  int rc = 0;
  rc |= push_t(e, h.ar, s.ar, a); rc |= push_t(e, h.pl, s.pl, a); rc |= push(e, h.vt, s.vticks, a * d.P * AG_VT_CAP);
  rc |= push(e, h.pxy, s.pel_xy, a * d.PC * 2); rc |= push(e, h.pid, s.pel_id, a * d.PC);
  rc |= push(e, h.vx, s.vir_x, a * d.VC); rc |= push(e, h.vy, s.vir_y, a * d.VC); rc |= push(e, h.vvx, s.vir_vx, a * d.VC); rc |= push(e, h.vvy, s.vir_vy, a * d.VC);
  rc |= push(e, h.vm, s.vir_mass, a * d.VC); rc |= push(e, h.vh, s.vir_hits, a * d.VC); rc |= push(e, h.vid, s.vir_id, a * d.VC);
  rc |= push(e, h.fx, s.food_x, a * d.FC); rc |= push(e, h.fy, s.food_y, a * d.FC); rc |= push(e, h.fvx, s.food_vx, a * d.FC); rc |= push(e, h.fvy, s.food_vy, a * d.FC); rc |= push(e, h.fid, s.food_id, a * d.FC);
  rc |= push_t(e, h.cells, s.cells, a);
  e->poll_gap = 2; e->next_poll = e->step_no + 2;   // a loaded state may need the other step form: sample the statistics soon
  return rc ? fail(AGARCL_E_HIP, "upload failed") : AGARCL_OK;
}
extern "C" int agarcl_load_arena(agarcl_env *e, int32_t arena, const uint32_t *b, int32_t words) { return load_arena_impl(e, arena, b, words, nullptr, 0, 0); }
extern "C" int agarcl_adopt_arena(agarcl_env *e, int32_t arena, const uint32_t *b, int32_t words, const int32_t *kinds, int32_t hm_buckets, int32_t hm_next_resize) {
  if (!kinds) return fail(AGARCL_E_INVALID, "agarcl_adopt_arena: kinds is null");
  return load_arena_impl(e, arena, b, words, kinds, hm_buckets, hm_next_resize);
}

extern "C" int64_t agarcl_state_bytes(agarcl_env *e) {
  if (!e) return 0;
  // streaming model (DESIGN.md / SURVEY.md 8d): pellet (x,y) read + cells r/w + player r/w + action in / result out
  return 8LL * e->cfg.num_pellets + 12LL * e->cfg.num_viruses + 72LL * 1 + 112LL * e->d.P + 24LL * e->d.n_agents;
}

// HBM the env holds (every array of agarcl_create incl. the event spill of dense / crowded arenas; observation scratch is added when first used)
extern "C" int64_t agarcl_device_bytes(agarcl_env *e) { return e ? (int64_t)e->alloc_bytes : 0; }

#ifndef AGAR_CPU_EMU
#ifndef AG_GRID_WAVES
#define AG_GRID_WAVES 6   // (waves per SIMD the register budget is cut for: 80 VGPRs, no spills; 8 -- all that LDS admits -- spills 20 registers for the same time, the uncapped 111 VGPRs leave 4 waves and cost 6 us)
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(AG_GRID_WAVES, AG_GRID_WAVES))) k_grid_obs(const AgState *__restrict__ gs, AgObsCfg o, int32_t *out, int zero_fill, AgObsUndo un) {
  int b = (int)blockIdx.x, na = gs->d.n_agents;
  grid_obs_agent(gs, b / na, b % na, o, out + (size_t)b * obs_channels(o) * o.G * o.G, zero_fill != 0, un, b);
}
// zeros for channels 1 .. C-1 of every frame: a plain streaming fill (grid-stride, 16 bytes per lane), which reaches the
// write bandwidth of a memset; channel 0 and the entities are written afterwards by k_grid_obs
__global__ void __launch_bounds__(256) k_grid_zero(int32_t *out, size_t frames, size_t frame_v4, size_t skip_v4) {
  typedef int32_t v4 __attribute__((ext_vector_type(4)));
  (void)frames;  // blockIdx.x = frame * 8 + slice of that frame's zero region (no 65535-frame limit of gridDim.y)
  const unsigned per = (unsigned)(frame_v4 - skip_v4), frame = blockIdx.x >> 3, slice = blockIdx.x & 7u;
  const v4 z = {0, 0, 0, 0};
  v4 *o4 = (v4 *)out + (size_t)frame * frame_v4 + skip_v4;
  for (unsigned off = slice * blockDim.x + threadIdx.x; off < per; off += 8u * blockDim.x) __builtin_nontemporal_store(z, &o4[off]);
}
#endif

#ifndef AGAR_CPU_EMU
__global__ void __launch_bounds__(64) k_gobigger_obs(const AgState *__restrict__ gs, AgGbCfg o, int32_t *hdr, float *food, float *virus, float *spore, float *clone) {
  const int P = gs->d.P, b = (int)blockIdx.x;
  gobigger_player(gs, b / P, b % P, o, hdr, food, virus, spore, clone);
}
#endif
extern "C" int agarcl_gobigger_obs(agarcl_env *e, int32_t grid_size, int32_t cap_food, int32_t cap_virus, int32_t cap_spore, int32_t cap_clone,
                                   int32_t *hdr, float *food, float *virus, float *spore, float *clone, int32_t on_device) {
  if (!e || !hdr || !food || !virus || !spore || !clone) return fail(AGARCL_E_INVALID, "agarcl_gobigger_obs: null pointer");
  if (grid_size < 1 || cap_food < 1 || cap_virus < 1 || cap_spore < 1 || cap_clone < 1) return fail(AGARCL_E_INVALID, "agarcl_gobigger_obs: grid size and capacities must be positive");
  AgGbCfg o; o.G = grid_size; o.KF = cap_food; o.KV = cap_virus; o.KS = cap_spore; o.KC = cap_clone;
  const size_t rows = (size_t)e->d.A * e->d.P;
#ifdef AGAR_CPU_EMU
  (void)on_device;
  for (size_t b = 0; b < rows; b++) gobigger_player(&e->s, (int)(b / e->d.P), (int)(b % e->d.P), o, hdr, food, virus, spore, clone);
  return AGARCL_OK;
#else
  HIPCHK(hipSetDevice(e->device));
  const size_t nb[5] = {rows * 8 * 4, rows * o.KF * 16, rows * o.KV * 16, rows * o.KS * 16, rows * o.KC * 28};
  void *host[5] = {hdr, food, virus, spore, clone}, *dev[5] = {hdr, food, virus, spore, clone};
  if (!on_device) {  // one staging block for the five tensors
    size_t words = 0; for (int i = 0; i < 5; i++) words += (nb[i] + 15) / 16 * 4;
    if (e->obs_cap < words) {
      if (e->obs_buf) { HIPCHK(hipStreamSynchronize(e->stream)); (void)hipFree(e->obs_buf); e->obs_buf = nullptr; e->obs_cap = 0; }
      if (hipMalloc((void **)&e->obs_buf, words * 4) != hipSuccess) return fail(AGARCL_E_NOMEM, "agarcl_gobigger_obs: staging allocation failed");
      e->obs_cap = words;
    }
    unsigned char *q = (unsigned char *)e->obs_buf;
    e->undo_out = nullptr;   // the shared staging buffer no longer holds a grid observation (agarcl_grid_obs: incremental clearing)
    for (int i = 0; i < 5; i++) { dev[i] = q; q += (nb[i] + 15) / 16 * 16; }
  }
  hipLaunchKernelGGL(k_gobigger_obs, dim3((unsigned)rows), dim3(64), 0, e->stream, e->d_state, o, (int32_t *)dev[0], (float *)dev[1], (float *)dev[2], (float *)dev[3], (float *)dev[4]);
  HIPCHK(hipGetLastError());
  if (!on_device) for (int i = 0; i < 5; i++) if (d2h(host[i], dev[i], nb[i], e->stream)) return fail(AGARCL_E_HIP, "agarcl_gobigger_obs: copy failed");
  return AGARCL_OK;
#endif
}

#ifndef AGAR_CPU_EMU
__global__ void __launch_bounds__(64) k_ram_obs(const AgState *__restrict__ gs, AgRamCfg o, float *out) {
  const int na = gs->d.n_agents, b = (int)blockIdx.x;
  ram_obs_agent(gs, b / na, b % na, o, out);
}
#endif
extern "C" int agarcl_ram_obs(agarcl_env *e, int32_t k_cells, int32_t k_pellets, int32_t k_viruses, int32_t k_others, float *out, int32_t on_device, int32_t *dim) {
  if (!e) return fail(AGARCL_E_INVALID, "null env");
  if (k_cells < 0 || k_pellets < 0 || k_viruses < 0 || k_others < 0 || k_cells > AG_CC || k_pellets > 2048 || k_viruses > 1024 || k_others > 512)
    return fail(AGARCL_E_INVALID, "agarcl_ram_obs: row counts out of range");
  AgRamCfg o; o.KC = k_cells; o.KP = k_pellets; o.KV = k_viruses; o.KO = k_others;
  const int D = ag_ram_dim(o);
  if (dim) *dim = D;
  if (!out) return AGARCL_OK;
#ifdef AGAR_CPU_EMU
  (void)on_device;
  return fail(AGARCL_E_UNSUPPORTED, "agarcl_ram_obs: the ram observation exists only as a HIP kernel");
#else
  HIPCHK(hipSetDevice(e->device));
  const size_t n = (size_t)e->d.A * e->d.n_agents, words = n * (size_t)D;
  float *dst = out;
  if (!on_device) {
    if (e->obs_cap < words) {
      if (e->obs_buf) { HIPCHK(hipStreamSynchronize(e->stream)); (void)hipFree(e->obs_buf); e->obs_buf = nullptr; e->obs_cap = 0; }
      if (hipMalloc((void **)&e->obs_buf, words * 4) != hipSuccess) return fail(AGARCL_E_NOMEM, "agarcl_ram_obs: staging allocation failed");
      e->obs_cap = words;
    }
    dst = (float *)e->obs_buf;
    e->undo_out = nullptr;   // the shared staging buffer no longer holds a grid observation
  }
  hipLaunchKernelGGL(k_ram_obs, dim3((unsigned)n), dim3(64), 0, e->stream, e->d_state, o, dst);
  HIPCHK(hipGetLastError());
  if (!on_device && d2h(out, dst, words * 4, e->stream)) return fail(AGARCL_E_HIP, "agarcl_ram_obs: copy failed");
  return AGARCL_OK;
#endif
}

extern "C" int agarcl_screen_obs(agarcl_env *e, int32_t width, int32_t height, int32_t agent_view, uint8_t *out, int32_t on_device) {
  if (!e || !out) return fail(AGARCL_E_INVALID, "agarcl_screen_obs: null pointer");
  if (width < 1 || height < 1 || width > 1024 || height > 1024) return fail(AGARCL_E_INVALID, "agarcl_screen_obs: screen size must be in [1, 1024]");
#ifdef AGAR_CPU_EMU
  (void)on_device;
  return fail(AGARCL_E_UNSUPPORTED, "agarcl_screen_obs: the screen rasteriser exists only as a HIP kernel");
#else
  HIPCHK(hipSetDevice(e->device));
  size_t n = (size_t)e->d.A * e->d.n_agents, bytes = n * (size_t)width * height * (agent_view ? 4 : 3);
  uint8_t *dst = out;
  if (!on_device) {
    size_t words = (bytes + 3) / 4;
    if (e->obs_cap < words) {
      if (e->obs_buf) { HIPCHK(hipStreamSynchronize(e->stream)); (void)hipFree(e->obs_buf); e->obs_buf = nullptr; e->obs_cap = 0; }
      if (hipMalloc((void **)&e->obs_buf, words * 4) != hipSuccess) return fail(AGARCL_E_NOMEM, "agarcl_screen_obs: staging allocation failed");
      e->obs_cap = words;
    }
    dst = (uint8_t *)e->obs_buf;
    e->undo_out = nullptr;   // (as above)
  }
  AgScreenCfg o; o.W = width; o.H = height; o.agent_view = agent_view != 0; scr_cfg_geometry(o);
#ifdef AG_SCR_ABL
  { const char *ab = getenv("AGARCL_SCR_ABL"); o.abl = ab ? atoi(ab) : 0; }
#endif
  { const char *pw = getenv("AGARCL_SCREEN_PIXELWISE");   // the pixel-wise kernel: cross-check only (agar_screen.inl)
    if (pw && pw[0] == '1') hipLaunchKernelGGL(k_screen_obs_pixelwise, dim3((unsigned)n), dim3(256), 0, e->stream, e->d_state, o, dst);
    else if (o.W <= 256 && o.H <= 256) { if (o.agent_view) hipLaunchKernelGGL((k_screen_obs<256, true>), dim3((unsigned)n), dim3(256), 0, e->stream, e->s, o, dst); else hipLaunchKernelGGL((k_screen_obs<256, false>), dim3((unsigned)n), dim3(256), 0, e->stream, e->s, o, dst); }
    else { if (o.agent_view) hipLaunchKernelGGL((k_screen_obs<1024, true>), dim3((unsigned)n), dim3(256), 0, e->stream, e->s, o, dst); else hipLaunchKernelGGL((k_screen_obs<1024, false>), dim3((unsigned)n), dim3(256), 0, e->stream, e->s, o, dst); } }
  HIPCHK(hipGetLastError());
  if (!on_device && d2h(out, dst, bytes, e->stream)) return fail(AGARCL_E_HIP, "agarcl_screen_obs: copy failed");
  return AGARCL_OK;
#endif
}

extern "C" int agarcl_grid_obs(agarcl_env *e, int32_t G, int32_t cells, int32_t others, int32_t viruses, int32_t pellets,
                               int32_t *out, int32_t on_device, int32_t *channels) {
  if (!e) return fail(AGARCL_E_INVALID, "null env");
  if (G < 1 || G > 1024) return fail(AGARCL_E_INVALID, "agarcl_grid_obs: grid_size must be in [1, 1024]");
  AgObsCfg o; o.G = G; o.cells = cells != 0; o.others = others != 0; o.viruses = viruses != 0; o.pellets = pellets != 0;
  int C = 1 + o.cells + 2 * o.others + 2 * o.viruses + 2 * o.pellets;
  if (channels) *channels = C;
  if (!out) return AGARCL_OK;
  size_t n = (size_t)e->d.A * e->d.n_agents, words = n * C * G * G;
#ifdef AGAR_CPU_EMU
  (void)on_device;
  for (size_t b = 0; b < n; b++) grid_obs_agent(&e->s, (int)(b / e->d.n_agents), (int)(b % e->d.n_agents), o, out + b * C * G * G);
  return AGARCL_OK;
#else
  HIPCHK(hipSetDevice(e->device));
  int32_t *dst = out;
  if (!on_device) {
    if (e->obs_cap < words) {
      if (e->obs_buf) { HIPCHK(hipStreamSynchronize(e->stream)); (void)hipFree(e->obs_buf); e->obs_buf = nullptr; e->obs_cap = 0; }
      if (hipMalloc((void **)&e->obs_buf, words * 4) != hipSuccess) return fail(AGARCL_E_NOMEM, "agarcl_grid_obs: staging allocation failed");
      e->obs_cap = words; e->undo_out = nullptr;   // a fresh buffer (possibly at the old address): nothing to undo
    }
    dst = e->obs_buf;
  }
  if (on_device == 1 && dst == e->undo_out) e->undo_out = nullptr;   // a plain call into the persistent buffer: its undo list no longer describes what the buffer holds
  const size_t GG = (size_t)G * G;
  // on_device == 2: the caller's buffer still holds this env's previous observation (same grid, same channels): clear only
  // what was written then.  The first such call (or a different buffer / configuration) clears everything and starts the list.
  AgObsUndo un{nullptr, nullptr, 0, 0, nullptr};
  if ((on_device == 2 || !on_device) && C > 1) {   // (host copies go through the engine's own staging buffer: persistent by construction)
    const int ucap = OBS_UNDO_CAP(e->d.PC), key = G * 16 + o.cells + 2 * o.others + 4 * o.viruses + 8 * o.pellets;
    if (!e->undo_list) { e->undo_list = alloc<int32_t>(e, n * (size_t)ucap); e->undo_count = alloc<int32_t>(e, n); }
    if (e->undo_list && e->undo_count) {
      un.list = e->undo_list; un.count = e->undo_count; un.cap = ucap;
      un.clear = e->undo_out == dst && e->undo_key == key;
      e->undo_out = dst; e->undo_key = key;
      if (e->undo_sig_g < G) {   // (a larger grid than any before: the signature array grows; the old one stays allocated until the env goes)
        e->undo_sig = alloc<uint8_t>(e, n * 2 * (size_t)G); e->undo_sig_g = e->undo_sig ? G : 0; un.clear = 0;
      }
      un.sig = e->undo_sig;   // (null: every call stores the whole channel, as before)
    }
  }
  const bool split = !un.clear && (GG & 3) == 0 && C > 1 && (((size_t)dst) & 15) == 0;
  if (split) hipLaunchKernelGGL(k_grid_zero, dim3(8u * (unsigned)n), dim3(256), 0, e->stream, dst, n, (size_t)C * GG / 4, GG / 4);
  hipLaunchKernelGGL(k_grid_obs, dim3((unsigned)n), dim3(256), 0, e->stream, e->d_state, o, dst, (split || un.clear) ? 0 : 1, un);
  HIPCHK(hipGetLastError());
  if (!on_device && d2h(out, dst, words * 4, e->stream)) return fail(AGARCL_E_HIP, "agarcl_grid_obs: copy failed");
  return AGARCL_OK;
#endif
}

// ---- stream ordering (include/agarcl_batch.h) ---------------------------------------------------------------------------------------
extern "C" void *agarcl_get_stream(agarcl_env *e) { return e ? (void *)e->stream : nullptr; }
#ifndef AGAR_CPU_EMU
// everything enqueued so far on `from` happens before whatever is enqueued on `to` from now on (one event per env and direction, re-recorded:
// a wait captures the record that precedes it)
static int order_streams(agarcl_env *e, int which, hipStream_t from, hipStream_t to) {
  if (from == to) return AGARCL_OK;
  HIPCHK(hipSetDevice(e->device));
  // (no timing, and no system-scope fence when the event completes: producer and consumer are streams of ONE device, an agent-scope release is what
  // they need -- with the default flags every record flushed the caches to system scope: 147 us per quiet 4096-arena step instead of 15)
  if (!e->order_ev[which]) { hipEvent_t ev = nullptr; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence)); e->order_ev[which] = (void *)ev; }
  HIPCHK(hipEventRecord((hipEvent_t)e->order_ev[which], from));
  HIPCHK(hipStreamWaitEvent(to, (hipEvent_t)e->order_ev[which], 0));
  return AGARCL_OK;
}
#endif
extern "C" int agarcl_stream_wait(agarcl_env *e, void *producer_stream) {
  if (!e) return fail(AGARCL_E_INVALID, "null env");
#ifdef AGAR_CPU_EMU
  (void)producer_stream; return AGARCL_OK;
#else
  return order_streams(e, 0, (hipStream_t)producer_stream, e->stream);
#endif
}
extern "C" int agarcl_stream_signal(agarcl_env *e, void *consumer_stream) {
  if (!e) return fail(AGARCL_E_INVALID, "null env");
#ifdef AGAR_CPU_EMU
  (void)consumer_stream; return AGARCL_OK;
#else
  return order_streams(e, 1, e->stream, (hipStream_t)consumer_stream);
#endif
}

// ---- one host call per vector step (include/agarcl_vec.h) ---------------------------------------------------------------------------
static int vec_args(agarcl_env *e, const agarcl_vec_spec *sp, const agarcl_vec_buffers *b, AgVecPost &v, const char *who) {
  if (!e || !sp || !b) return fail(AGARCL_E_INVALID, std::string(who) + ": null pointer");
  if (e->d.n_agents < 1) return fail(AGARCL_E_INVALID, std::string(who) + ": the env has no agents");
  if (!b->steps || !b->reward || !b->done || !b->truncated || !b->ended || !b->ep_return || !b->final_return || !b->final_length)
    return fail(AGARCL_E_INVALID, std::string(who) + ": a bookkeeping buffer is null");
  if (sp->obs_kind < AGARCL_OBS_NONE || sp->obs_kind > AGARCL_OBS_RAM || (sp->obs_kind != AGARCL_OBS_NONE && !b->obs))
    return fail(AGARCL_E_INVALID, std::string(who) + ": bad observation kind / null observation buffer");
  v.dones = e->s.dones; v.rewards = e->s.rewards; v.n = e->d.n_agents; v.number_steps = sp->number_steps; v.episodic = sp->episodic;
  v.steps = b->steps; v.reward = b->reward; v.done = b->done; v.trunc = b->truncated; v.ended = b->ended;
  v.ep_return = b->ep_return; v.final_return = b->final_return; v.final_length = b->final_length;
  v.ar = e->s.ar; v.ts_lg = e->d.ts_lg; v.reset_flagged = sp->reset_flagged != 0;
  return AGARCL_OK;
}
static int vec_observe(agarcl_env *e, const agarcl_vec_spec *sp, const agarcl_vec_buffers *b) {
  const int32_t *a = sp->obs_arg; int32_t q = 0;
  switch (sp->obs_kind) {
    case AGARCL_OBS_GRID: return agarcl_grid_obs(e, a[0], a[1], a[2], a[3], a[4], (int32_t *)b->obs, 2, &q);
    case AGARCL_OBS_SCREEN: return agarcl_screen_obs(e, a[0], a[1], a[2], (uint8_t *)b->obs, 1);
    case AGARCL_OBS_RAM: return agarcl_ram_obs(e, a[0], a[1], a[2], a[3], (float *)b->obs, 1, &q);
    default: return AGARCL_OK;
  }
}
extern "C" int agarcl_vec_reset(agarcl_env *e, const agarcl_vec_spec *sp, const agarcl_vec_buffers *b) {
  AgVecPost v; int rc = vec_args(e, sp, b, v, "agarcl_vec_reset");
  if (rc) return rc;
  if ((rc = agarcl_reset(e, nullptr, sp->reset_ids)) != 0) return rc;
  const size_t A = (size_t)e->d.A, n = (size_t)e->d.n_agents;
#ifdef AGAR_CPU_EMU
  memset(b->steps, 0, A * 4); memset(b->final_length, 0, A * 4); memset(b->ended, 0, A);
  memset(b->reward, 0, A * n * 4); memset(b->ep_return, 0, A * n * 4); memset(b->final_return, 0, A * n * 4); memset(b->done, 0, A * n); memset(b->truncated, 0, A * n);
#else
  HIPCHK(hipMemsetAsync(b->steps, 0, A * 4, e->stream)); HIPCHK(hipMemsetAsync(b->final_length, 0, A * 4, e->stream)); HIPCHK(hipMemsetAsync(b->ended, 0, A, e->stream));
  HIPCHK(hipMemsetAsync(b->reward, 0, A * n * 4, e->stream)); HIPCHK(hipMemsetAsync(b->ep_return, 0, A * n * 4, e->stream)); HIPCHK(hipMemsetAsync(b->final_return, 0, A * n * 4, e->stream));
  HIPCHK(hipMemsetAsync(b->done, 0, A * n, e->stream)); HIPCHK(hipMemsetAsync(b->truncated, 0, A * n, e->stream));
#endif
  return vec_observe(e, sp, b);
}
extern "C" int agarcl_vec_step(agarcl_env *e, const agarcl_vec_spec *sp, const agarcl_vec_buffers *b, const float *dxdy_dev, const int32_t *act_dev, uint32_t *flags_seen) {
  AgVecPost v; int rc = vec_args(e, sp, b, v, "agarcl_vec_step");
  if (rc) return rc;
  if (!dxdy_dev || !act_dev) return fail(AGARCL_E_INVALID, "agarcl_vec_step: null action pointer");
#ifndef AGAR_CPU_EMU
  HIPCHK(hipSetDevice(e->device));
#endif
  e->act_dxdy = dxdy_dev; e->act = act_dev;
  rc = launch_step(e, sp->ticks > 0 ? sp->ticks : e->cfg.ticks_per_step, 1);
  e->slot = (e->slot + 1) % AG_PACKED_SLOTS;
  if (rc) return rc;
#ifdef AGAR_CPU_EMU
  { std::vector<uint8_t> mask((size_t)e->d.A, 0); bool some = false;
    for (int a = 0; a < e->d.A; a++) if (ag_vec_post_arena(v, a)) { mask[(size_t)a] = 1; some = true; }
    if (some) { const int32_t keep = e->s.qstat[1]; rc = launch_reset(e, nullptr, mask.data(), sp->reset_ids); e->s.qstat[1] |= keep; if (rc) return rc; } }
#else
#define CALL(N, V) hipLaunchKernelGGL((k_vec_post_reset<N, V>), dim3(e->d.A), dim3(64), e->lds_bytes, e->stream, e->d_state, v, (int)sp->reset_ids)
  AG_DISPATCH_NS(e->ns, CALL);
#undef CALL
  HIPCHK(hipGetLastError());
#endif
  if ((rc = vec_observe(e, sp, b)) != 0) return rc;
  if (flags_seen) return agarcl_poll_flags(e, flags_seen);
  return AGARCL_OK;
}

// ---- sub-batch pipelining (include/agarcl_batch.h) ------------------------------------------------------------------------------------
struct agarcl_pipe { std::vector<agarcl_env *> envs; std::vector<int32_t> first; int32_t A; int device; int concurrent; void *fork_ev;
  // fork / join through flag words in HBM instead of HIP events (see agarcl_pipe_fork): [0] the fork's epoch, [1 + j] sub-batch j's join epoch, [65] a spin ran into its time bound
  int32_t *d_sync; int32_t epoch; void *spin_for; bool spin_ok, spin_off; };
#ifndef AGAR_CPU_EMU
// Do two streams execute concurrently?  The runtime multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4) and two
// streams that share a queue run one after the other whatever the API says -- measured on MI355X: two 2048-arena sub-batches on streams that
// happened to share a queue took 590 us per step instead of 313 (scripts/gpu_pipe_probe.py).  The probe: a one-lane kernel on `a` waits (at most
// `ticks` of the 100 MHz wall clock) for a flag that a kernel on `b` sets; it sees the flag iff `b`'s kernel ran while it was waiting.
__global__ void k_pipe_wait(int *flag, int *seen, long long ticks) {
  const long long t0 = (long long)wall_clock64(); int f;
  do { f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!f && (long long)wall_clock64() - t0 < ticks);
  *seen = f;
}
__global__ void k_pipe_set(int *flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static int streams_concurrent(hipStream_t a, hipStream_t b, int *d_two) {   // 1 yes, 0 no, -1 error
  if (a == b) return 0;
  if (hipMemsetAsync(d_two, 0, 8, a) != hipSuccess || hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
  hipLaunchKernelGGL(k_pipe_wait, dim3(1), dim3(1), 0, a, d_two, d_two + 1, 200000LL);   // <= 2 ms
  hipLaunchKernelGGL(k_pipe_set, dim3(1), dim3(1), 0, b, d_two);
  int seen = 0;
  if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess || hipMemcpy(&seen, d_two + 1, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return seen ? 1 : 0;
}
#endif
extern "C" int agarcl_pipe_destroy(agarcl_pipe *p) {
  if (!p) return AGARCL_OK;
#ifndef AGAR_CPU_EMU
  if (p->fork_ev) (void)hipEventDestroy((hipEvent_t)p->fork_ev);
  if (p->d_sync) { for (agarcl_env *e : p->envs) (void)hipStreamSynchronize(e->stream); (void)hipFree(p->d_sync); }
#endif
  for (agarcl_env *e : p->envs) agarcl_destroy(e);
  delete p;
  return AGARCL_OK;
}
extern "C" int agarcl_pipe_create(const agarcl_config *cfg, int32_t num_arenas, int32_t sub_batches, int32_t device, agarcl_pipe **out) {
  if (!cfg || !out || num_arenas <= 0 || sub_batches < 1 || sub_batches > num_arenas || sub_batches > 64) return fail(AGARCL_E_INVALID, "agarcl_pipe_create: bad arguments (1 <= sub_batches <= min(num_arenas, 64))");
  agarcl_pipe *p = new agarcl_pipe(); p->A = num_arenas; p->device = device; p->concurrent = 1; p->fork_ev = nullptr;
  p->d_sync = nullptr; p->epoch = 0; p->spin_for = nullptr; p->spin_ok = false; { const char *pe = getenv("AGARCL_PIPE_EVENTS"); p->spin_off = pe && pe[0] == '1'; }
  const int32_t base = num_arenas / sub_batches, rem = num_arenas % sub_batches;   // contiguous ranges, the first ones take the remainder (agarcl_amd/dist.py shard_bounds)
  for (int32_t j = 0; j < sub_batches; j++) {
    const int32_t lo = j * base + (j < rem ? j : rem), n = base + (j < rem ? 1 : 0);
    agarcl_env *e = nullptr;
    // one engine over all arenas gives arena i the default seed 5489 + i (agarcl_create): so do the sub-batches, by global index
    int rc = create_env(cfg, n, device, 5489u + (uint32_t)lo, &e);
    if (rc != 0) { if (e) agarcl_destroy(e); agarcl_pipe_destroy(p); return rc; }
    p->envs.push_back(e); p->first.push_back(lo);
  }
#ifndef AGAR_CPU_EMU
  if (sub_batches > 1) {
    // every sub-batch stream must run concurrently with the ones before it; a stream that shares a hardware queue with one of them is
    // replaced (the rejected candidates stay alive until the end, so that the runtime does not hand the same queue out again)
    int *d_two = nullptr; std::vector<hipStream_t> rejected;
    if (hipMalloc((void **)&d_two, 8) != hipSuccess) { agarcl_pipe_destroy(p); return fail(AGARCL_E_NOMEM, "agarcl_pipe_create: allocation failed"); }
    for (int32_t j = 1; j < sub_batches; j++) {
      agarcl_env *e = p->envs[(size_t)j];
      bool ok = false;
      for (int attempt = 0; attempt < 12 && !ok; attempt++) {
        ok = true;
        for (int32_t i = 0; i < j && ok; i++) ok = streams_concurrent(p->envs[(size_t)i]->stream, e->stream, d_two) == 1;
        if (!ok && attempt < 11) {
          hipStream_t ns = nullptr;
          if (hipStreamCreateWithFlags(&ns, hipStreamNonBlocking) != hipSuccess) break;
          (void)hipStreamSynchronize(e->stream);
          rejected.push_back(e->stream); e->stream = ns;
        }
      }
      if (ok) p->concurrent++;
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    (void)hipFree(d_two);
  }
#else
  p->concurrent = sub_batches;
#endif
  *out = p;
  return AGARCL_OK;
}
extern "C" int agarcl_pipe_sub_batches(agarcl_pipe *p) { return p ? (int)p->envs.size() : 0; }
extern "C" agarcl_env *agarcl_pipe_env(agarcl_pipe *p, int32_t j) { return p && j >= 0 && (size_t)j < p->envs.size() ? p->envs[(size_t)j] : nullptr; }
extern "C" int agarcl_pipe_range(agarcl_pipe *p, int32_t j, int32_t *first_arena, int32_t *count) {
  if (!p || j < 0 || (size_t)j >= p->envs.size()) return fail(AGARCL_E_INVALID, "agarcl_pipe_range: bad arguments");
  if (first_arena) *first_arena = p->first[(size_t)j];
  if (count) *count = p->envs[(size_t)j]->d.A;
  return AGARCL_OK;
}
extern "C" int agarcl_pipe_seed(agarcl_pipe *p, const uint32_t *seeds_host, uint32_t base_seed) {
  if (!p) return fail(AGARCL_E_INVALID, "null pipe");
  for (size_t j = 0; j < p->envs.size(); j++) {
    const int rc = agarcl_seed(p->envs[j], seeds_host ? seeds_host + p->first[j] : nullptr, base_seed + (uint32_t)p->first[j]);
    if (rc) return rc;
  }
  return AGARCL_OK;
}
extern "C" int agarcl_pipe_concurrent(agarcl_pipe *p) { return p ? p->concurrent : 0; }
#ifndef AGAR_CPU_EMU
// Ordering the sub-batches against the caller's stream with HIP events costs ~40 us of device-side latency per hand-over on this stack (r05: fork + join
// around a 10 us step took 170 us; round 6, k = 4: a mode-6 vector step 366 -> 731 us), which is why the full-batch step() of a pipelined env lost to
// the lock-step one.  The same ordering through flag words: the producer's stream runs a one-lane kernel that publishes an epoch, every consumer stream
// starts with a one-lane kernel that spins until it sees it.  Kernels of one stream run in order and a kernel's end releases what it wrote, so what
// comes behind the spin sees what came in front of the flag.  A spinning kernel must never sit in front of the kernel it waits for in the SAME hardware
// queue: the caller's stream is verified once to run beside every sub-batch stream (as the sub-batch streams were verified against each other);
// otherwise, or with AGARCL_PIPE_EVENTS=1, events are used.  Every spin is bounded (200 ms): a time-out sets a word that the next call reports.
__global__ void k_flag_set(int32_t *w, int32_t v) { __hip_atomic_store((AG_GLOBAL int32_t *)w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_flag_wait(const int32_t *w, int n, int32_t v, int32_t *err) {   // lane i < n waits for w[i] to reach v (wrap-safe)
  const int i = (int)threadIdx.x;
  if (i < n) {
    const long long t0 = (long long)wall_clock64();
    while ((int32_t)(__hip_atomic_load((const AG_GLOBAL int32_t *)w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v) < 0) {
      if ((long long)wall_clock64() - t0 > 20000000LL) { (void)__hip_atomic_fetch_or((AG_GLOBAL int32_t *)err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
}
static bool pipe_spin_ready(agarcl_pipe *p, hipStream_t caller) {
  if (p->spin_off) return false;
  if (p->spin_for == (void *)caller && p->d_sync) return p->spin_ok;
  if (!p->d_sync) { if (hipMalloc((void **)&p->d_sync, 66 * 4) != hipSuccess) { p->spin_off = true; return false; } (void)hipMemset(p->d_sync, 0, 66 * 4); }
  int *d_two = nullptr; bool ok = hipMalloc((void **)&d_two, 8) == hipSuccess;
  for (size_t j = 0; ok && j < p->envs.size(); j++) ok = p->envs[j]->stream != caller && streams_concurrent(caller, p->envs[j]->stream, d_two) == 1;
  if (d_two) (void)hipFree(d_two);
  p->spin_for = (void *)caller; p->spin_ok = ok;
  return ok;
}
#endif
extern "C" int agarcl_pipe_fork(agarcl_pipe *p, void *producer_stream) {
  if (!p) return fail(AGARCL_E_INVALID, "null pipe");
#ifndef AGAR_CPU_EMU
  HIPCHK(hipSetDevice(p->device));
  if (p->envs.size() <= 64 && pipe_spin_ready(p, (hipStream_t)producer_stream)) {
    p->epoch += 1;
    hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(1), 0, (hipStream_t)producer_stream, p->d_sync, p->epoch);
    for (agarcl_env *e : p->envs) hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(1), 0, e->stream, (const int32_t *)p->d_sync, 1, p->epoch, p->d_sync + 65);
    HIPCHK(hipGetLastError());
    return AGARCL_OK;
  }
  if (!p->fork_ev) { hipEvent_t ev = nullptr; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence)); p->fork_ev = (void *)ev; }
  HIPCHK(hipEventRecord((hipEvent_t)p->fork_ev, (hipStream_t)producer_stream));
  for (agarcl_env *e : p->envs) if (e->stream != (hipStream_t)producer_stream) HIPCHK(hipStreamWaitEvent(e->stream, (hipEvent_t)p->fork_ev, 0));
#else
  (void)producer_stream;
#endif
  return AGARCL_OK;
}
extern "C" int agarcl_pipe_join(agarcl_pipe *p, void *consumer_stream) {
  if (!p) return fail(AGARCL_E_INVALID, "null pipe");
#ifndef AGAR_CPU_EMU
  HIPCHK(hipSetDevice(p->device));
  if (p->envs.size() <= 64 && pipe_spin_ready(p, (hipStream_t)consumer_stream)) {
    // (the join's epoch is the last fork's: a join without a fork in front of it -- after a reset -- takes a fresh one)
    p->epoch += 1;
    for (size_t j = 0; j < p->envs.size(); j++) hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(1), 0, p->envs[j]->stream, p->d_sync + 1 + j, p->epoch);
    hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, (hipStream_t)consumer_stream, (const int32_t *)(p->d_sync + 1), (int)p->envs.size(), p->epoch, p->d_sync + 65);
    HIPCHK(hipGetLastError());
    return AGARCL_OK;
  }
#endif
  for (agarcl_env *e : p->envs) { const int rc = agarcl_stream_signal(e, consumer_stream); if (rc) return rc; }
  return AGARCL_OK;
}
// 1 if a spin of the flag-word fork / join ever ran into its time bound (an ordering that was NOT enforced: a bug or two streams of one hardware queue
// after all), else 0; synchronises the pipe (diagnostics / tests)
extern "C" int agarcl_pipe_spin_timeouts(agarcl_pipe *p) {
  if (!p) return -1;
#ifndef AGAR_CPU_EMU
  if (!p->d_sync) return 0;
  for (agarcl_env *e : p->envs) (void)hipStreamSynchronize(e->stream);
  int32_t w = 0;
  if (hipMemcpy(&w, p->d_sync + 65, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return w != 0;
#else
  return 0;
#endif
}
extern "C" int agarcl_pipe_sync(agarcl_pipe *p) {
  if (!p) return fail(AGARCL_E_INVALID, "null pipe");
  for (agarcl_env *e : p->envs) { const int rc = agarcl_sync(e); if (rc) return rc; }
  return AGARCL_OK;
}
#endif  // AG_PART_NS
