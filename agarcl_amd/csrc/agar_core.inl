// Wave-level Agar.io engine: ONE 64-lane wavefront simulates ONE arena.
//
// Programming discipline (what makes the code both a CDNA4 kernel and checkable on a host):
//   * code outside AG_LANES / AG_SERIAL is *wave-uniform* scalar code (same value in every lane).
//     Per-arena and per-player scalars live in "uniform blocks" (UBlock): ONE VGPR whose lane k holds
//     word k, read with v_readlane / written with v_writelane -- no LDS round trip on the hot path;
//   * AG_LANES(i, n) bodies are the data-parallel parts: lane-private temporaries only, all
//     communication through LDS / HBM;
//   * collectives are the only cross-lane ops: wave_count / wave_any (ballot + popcount), wave_compact
//     (ballot + popcount(mask & lanemask_lt): ordered stream compaction), wave_min/max/sum (DPP row_shr +
//     row_bcast scan, result read from lane 63);
//   * AG_SERIAL sections replay the reference's order-dependent semantics on lane 0 and hand scalar
//     results back through a small LDS mailbox (rare paths only).
// The reference semantics being reproduced are cited as "R:" (paths under /root/reference).
//
// Compiled for gfx950 by agar_engine.hip.  tests/emu/ compiles the same text with -DAGAR_CPU_EMU into a
// *test-only* host library (lanes become loops) so that kernel logic can be diffed against the oracle
// without a GPU; the product never loads that build.
#pragma once
#include "agar_types.h"
#include <limits.h>
#include <math.h>

#pragma clang fp contract(off)

#ifdef AGAR_CPU_EMU
#define AG_DEV static inline
#define AG_MEM inline
#define AG_LANES(i, n) for (int i = 0; i < (n); ++i)
#define AG_LANE_OR_0 0
#define AG_LANE_STEP 1
#define AG_SERIAL if (true)
AG_DEV int ag_uni(int v) { return v; }
AG_DEV unsigned ag_uniu(unsigned v) { return v; }
AG_DEV float ag_unif(float v) { return v; }
AG_DEV void ag_mem_fence() {}
AG_DEV void ag_lds_order() {}
AG_DEV float ag_sqrtf(float x) { return sqrtf(x); }
AG_DEV float ag_sqrtf_lean(float x) { return sqrtf(x); }
AG_DEV float ag_divf(float a, float b) { return a / b; }
AG_DEV void ag_atomic_or(int32_t *p, int v) { *p |= v; }
AG_DEV bool ag_any(bool p) { return p; }
#else
#define AG_DEV __device__ __forceinline__
// lane within the wavefront (kernels may pack several wavefronts = several arenas into one workgroup)
#define AG_LANE ((int)threadIdx.x & 63)
#define AG_MEM __device__ __forceinline__
#define AG_MEM_NOINLINE __device__   // (with __attribute__((noinline)): a real call, for rare code whose registers must stay out of a hot loop's budget)
#define AG_LANES(i, n) for (int i = AG_LANE; i < (n); i += 64)
#define AG_LANE_OR_0 AG_LANE   // (lane-strided loops with another start: for (i = start + AG_LANE_OR_0; i < n; i += AG_LANE_STEP))
#define AG_LANE_STEP 64
#define AG_SERIAL if (AG_LANE == 0)
AG_DEV int ag_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
AG_DEV unsigned ag_uniu(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
AG_DEV float ag_unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// HBM written by one lane and read by another lane of the same wave later on: drain the wave's
// outstanding vector-memory operations first (single-wave workgroups: same CU, same L1)
AG_DEV void ag_mem_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
// LDS traffic of one wave is processed in program order; only the compiler must not reorder
AG_DEV void ag_lds_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
// IEEE correctly rounded: built with -fhip-fp32-correctly-rounded-divide-sqrt (build.py); NOT __fsqrt_rn,
// which ROCm's headers map to the approximate __ocml_native_sqrt_f32.
AG_DEV float ag_sqrtf(float x) { return __builtin_sqrtf(x); }
AG_DEV float ag_divf(float a, float b) { return a / b; }
AG_DEV void ag_atomic_or(int32_t *p, int v) { (void)__hip_atomic_fetch_or((__attribute__((address_space(1))) int32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // (p is HBM: a global, not a flat, atomic)
AG_DEV bool ag_any(bool p) { return __ballot(p) != 0ull; }   // does any ACTIVE lane of the wave want it?
// The same correctly rounded square root in 9 instead of 22 instructions, for the relaxation's dependent chain (three per pair visit).
// The compiler's expansion of sqrtf is: scale x by 2^32 when x < 2^-96, v_sqrt_f32 (1 ulp), two residual tests (one ulp down / up, each an
// FMA), undo the scaling, pass +-0 / +inf through.  Without the scaling the residual FMAs are still exact for x >= 2^-96, and 0, +inf and
// NaN come out of the two tests unchanged (NaN residuals compare false) -- so only 0 < |x| < 2^-96 needs the long form, and a squared
// distance between two cells never is (but a loaded snapshot may hold anything): wave-uniform fallback.  agarcl_debug_sqrt_check runs all
// 2^32 bit patterns through both.
AG_DEV float ag_sqrtf_lean(float x) {
#ifdef AG_NO_LEAN_SQRT
  return __builtin_sqrtf(x);
#else
  if (__builtin_expect(ag_any(fabsf(x) < 0x1p-96f && x != 0.0f), 0)) return __builtin_sqrtf(x);   // (v_sqrt_f32 takes a denormal for a zero)
  const float s = __builtin_amdgcn_sqrtf(x);
  const float dn = __int_as_float(__float_as_int(s) - 1), up = __int_as_float(__float_as_int(s) + 1);
  const float rdn = __builtin_fmaf(-dn, s, x), rup = __builtin_fmaf(-up, s, x);
  float r = (0.0f >= rdn) ? dn : s;
  r = (0.0f < rup) ? up : r;
  return r;
#endif
}
#endif

#include "agar_libm.inl"

#if defined(AGAR_CPU_EMU) && defined(AG_DEBUG_FLAG8)
#define AG_DBG8(what, a, b) fprintf(stderr, "flag 8: %s %d > %d\n", what, (int)(a), (int)(b))
#else
#define AG_DBG8(what, a, b) do { } while (0)
#endif
#if defined(AGAR_CPU_EMU) && defined(AG_DEBUG_FLAG8)
static int ag_dbg8_max = 0;
#define AG_DBG8MAX(v) do { if ((int)(v) > ag_dbg8_max) { ag_dbg8_max = (int)(v); if (ag_dbg8_max > 256) fprintf(stderr, "max events %d\n", ag_dbg8_max); } } while (0)
#else
#define AG_DBG8MAX(v) do { } while (0)
#endif
#define AG_RARE(x) __builtin_expect(!!(x), 0)  // keeps rare branches out of the hot instruction stream

// Pointers fetched from the HBM-resident descriptor are generic to the compiler (flat_load/flat_store); every
// accessor below casts them to the global address space so they become global_load / global_store.
#ifdef AGAR_CPU_EMU
#define AG_GLOBAL
#else
#define AG_GLOBAL __attribute__((address_space(1)))
#endif

// ---- collectives --------------------------------------------------------------------------------
#ifdef AGAR_CPU_EMU
// (no short-circuiting: functors may have side effects, exactly like on the device)
template <class F> AG_DEV int wave_sum(int n, F f) { int s = 0; for (int i = 0; i < n; i++) s += f(i); return s; }
template <class F> AG_DEV int wave_count(int n, F f) { int s = 0; for (int i = 0; i < n; i++) s += f(i) ? 1 : 0; return s; }
template <class F> AG_DEV unsigned wave_max(int n, F f) { unsigned s = 0; for (int i = 0; i < n; i++) { unsigned v = f(i); if (v > s) s = v; } return s; }
template <class F> AG_DEV unsigned wave_min(int n, F f) { unsigned s = UINT_MAX; for (int i = 0; i < n; i++) { unsigned v = f(i); if (v < s) s = v; } return s; }
template <class F> AG_DEV bool wave_any(int n, F f) { bool a = false; for (int i = 0; i < n; i++) a = a | (bool)f(i); return a; }
template <class F> AG_DEV unsigned wave_or(int n, F f) { unsigned s = 0; for (int i = 0; i < n; i++) s |= f(i); return s; }
template <class P, class S> AG_DEV int wave_compact(int n, P pred, S sink) { int c = 0; for (int i = 0; i < n; i++) if (pred(i)) { sink(i, c); c++; } return c; }
#else
// DPP scan (row_shr 1,2,4,8 + row_bcast15/31): lane 63 ends up with the reduction of all 64 lanes.
#define AG_DPP_STEP(v, ident, OP, ctrl, rmask) { int t_ = __builtin_amdgcn_update_dpp((int)(ident), (int)(v), ctrl, rmask, 0xf, false); v = OP(v, t_); }
#define AG_DPP_REDUCE(v, ident, OP) \
  AG_DPP_STEP(v, ident, OP, 0x111, 0xf) AG_DPP_STEP(v, ident, OP, 0x112, 0xf) AG_DPP_STEP(v, ident, OP, 0x114, 0xf) \
  AG_DPP_STEP(v, ident, OP, 0x118, 0xf) AG_DPP_STEP(v, ident, OP, 0x142, 0xa) AG_DPP_STEP(v, ident, OP, 0x143, 0xc)
#define AG_OP_ADD(a, b) ((a) + (b))
#define AG_OP_UMAX(a, b) ((unsigned)(a) > (unsigned)(b) ? (a) : (b))
#define AG_OP_UMIN(a, b) ((unsigned)(a) < (unsigned)(b) ? (a) : (b))
#define AG_OP_OR(a, b) ((a) | (b))
AG_DEV int wred_add(int v) { AG_DPP_REDUCE(v, 0, AG_OP_ADD) return __builtin_amdgcn_readlane(v, 63); }
AG_DEV unsigned wred_max(unsigned u) { int v = (int)u; AG_DPP_REDUCE(v, 0, AG_OP_UMAX) return (unsigned)__builtin_amdgcn_readlane(v, 63); }
AG_DEV unsigned wred_min(unsigned u) { int v = (int)u; AG_DPP_REDUCE(v, -1, AG_OP_UMIN) return (unsigned)__builtin_amdgcn_readlane(v, 63); }
AG_DEV unsigned wred_or(unsigned u) { int v = (int)u; AG_DPP_REDUCE(v, 0, AG_OP_OR) return (unsigned)__builtin_amdgcn_readlane(v, 63); }
template <class F> AG_DEV unsigned wave_or(int n, F f) { unsigned s = 0; for (int i = AG_LANE; i < n; i += 64) s |= f(i); return wred_or(s); }
template <class F> AG_DEV int wave_sum(int n, F f) { int s = 0; for (int i = AG_LANE; i < n; i += 64) s += f(i); return wred_add(s); }
template <class F> AG_DEV int wave_count(int n, F f) {
  int c = 0;
  for (int base = 0; base < n; base += 64) { int i = base + AG_LANE; bool p = (i < n) && f(i); c += __popcll(__ballot(p)); }
  return c;
}
template <class F> AG_DEV unsigned wave_max(int n, F f) { unsigned s = 0; for (int i = AG_LANE; i < n; i += 64) { unsigned v = f(i); s = v > s ? v : s; } return wred_max(s); }
template <class F> AG_DEV unsigned wave_min(int n, F f) { unsigned s = UINT_MAX; for (int i = AG_LANE; i < n; i += 64) { unsigned v = f(i); s = v < s ? v : s; } return wred_min(s); }
template <class F> AG_DEV bool wave_any(int n, F f) { bool a = false; for (int i = AG_LANE; i < n; i += 64) a = a | (bool)f(i); return __ballot(a) != 0ull; }
// ordered stream compaction: sink(i, rank) for every i with pred(i), rank = number of earlier hits
template <class P, class S> AG_DEV int wave_compact(int n, P pred, S sink) {
  int count = 0;
  const unsigned long long lt = (1ull << AG_LANE) - 1ull;
  for (int base = 0; base < n; base += 64) {
    int i = base + AG_LANE;
    bool p = (i < n) && pred(i);
    unsigned long long m = __ballot(p);
    if (p) sink(i, count + __popcll(m & lt));
    count += __popcll(m);
  }
  return count;
}
#endif

// ---- uniform block: up to 64 wave-uniform 32-bit words kept in ONE VGPR (lane k = word k) ----------
#ifdef AGAR_CPU_EMU
struct UBlock { int w[64]; };
AG_DEV int ub_get(const UBlock &b, int k) { return b.w[k]; }
AG_DEV void ub_set(UBlock &b, int k, int v) { b.w[k] = v; }
template <class PT> AG_DEV void ub_load(UBlock &b, PT src, int n) { for (int i = 0; i < 64; i++) b.w[i] = i < n ? src[i] : 0; }
template <class F> AG_DEV void ub_fill(UBlock &b, F f) { for (int i = 0; i < 64; i++) b.w[i] = f(i); }   // word i = f(i)
template <class PT> AG_DEV void ub_store(const UBlock &b, PT dst, int n) { for (int i = 0; i < n; i++) dst[i] = b.w[i]; }
template <class PT> AG_DEV void ub_load_t(UBlock &b, PT src, int n, int ag_ts_lg) { for (int i = 0; i < 64; i++) b.w[i] = i < n ? src[AG_TW(i)] : 0; }   // tile-transposed source
template <class PT> AG_DEV void ub_store_t(const UBlock &b, PT dst, int n, int ag_ts_lg) { for (int i = 0; i < n; i++) dst[AG_TW(i)] = b.w[i]; }
#else
struct UBlock { int v; };
AG_DEV int ub_get(const UBlock &b, int k) { return __builtin_amdgcn_readlane(b.v, k); }
AG_DEV void ub_set(UBlock &b, int k, int v) { b.v = (AG_LANE == k) ? v : b.v; }
template <class PT> AG_DEV void ub_load(UBlock &b, PT src, int n) { int l = AG_LANE; b.v = l < n ? src[l] : 0; }
template <class F> AG_DEV void ub_fill(UBlock &b, F f) { b.v = f(AG_LANE); }   // word i = f(i)
template <class PT> AG_DEV void ub_store(const UBlock &b, PT dst, int n) { int l = AG_LANE; if (l < n) dst[l] = b.v; }
template <class PT> AG_DEV void ub_load_t(UBlock &b, PT src, int n, int ag_ts_lg) { int l = AG_LANE; b.v = l < n ? src[AG_TW(l)] : 0; }   // tile-transposed source
template <class PT> AG_DEV void ub_store_t(const UBlock &b, PT dst, int n, int ag_ts_lg) { int l = AG_LANE; if (l < n) dst[AG_TW(l)] = b.v; }
#endif

// ---- numerics: C++ std::min/max/clamp on floats with their NaN behaviour (R: core/utils.hpp:19-21)
AG_DEV float smaxf(float a, float b) { return (a < b) ? b : a; }
AG_DEV float sminf(float a, float b) { return (b < a) ? b : a; }
AG_DEV float clampf(float x, float lo, float hi) { return smaxf(sminf(x, hi), lo); }
// static_cast<int>(float) as the reference build behaves on x86-64 (NaN / out of range -> INT_MIN)
AG_DEV int f2i(float f) { if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT_MIN; return (int)f; }
AG_DEV float u2f(int u) { union { int i; float f; } c; c.i = u; return c.f; }
AG_DEV int f2u(float f) { union { int i; float f; } c; c.f = f; return c.i; }
AG_DEV float vmag(float dx, float dy) { float a = dx * dx, b = dy * dy; return ag_sqrtf(a + b); }
AG_DEV float sqr_dist(float ax, float ay, float bx, float by) { float dx = fabsf(ax - bx), dy = fabsf(ay - by); float a = dx * dx, b = dy * dy; return a + b; }
// R: core/Ball.hpp:31-43 (pow(r,2) == r*r in fp32, see oracle/agar_oracle.c)
AG_DEV bool collides(float ax, float ay, float ar, float bx, float by, float br) { float r = smaxf(ar, br); float rr = r * r; return rr >= sqr_dist(ax, ay, bx, by); }
AG_DEV bool touches(float ax, float ay, float ar, float bx, float by, float br) { float r = ar + br; float rr = r * r; float d = sqr_dist(ax, ay, bx, by) + 0.0f; return rr >= d; }
AG_DEV bool can_eat_mass(unsigned a, unsigned b) { return (double)a > (double)b * 1.1; }  // R: Ball.hpp:45-47
// double -> uint32 as the reference's x86-64 build does it (cvttsd2si to 64 bits, low half kept): a NEGATIVE value wraps instead of
// saturating to 0.  Entities.hpp:199-202: after >= 66 virus meals inside the anti-team window the decay factor 1 - 0.002 * 1.1^k is
// negative and the reference's cell ends up with mass 2^32 - x; the device's own conversion would give 0.
AG_DEV unsigned d2u_x86(double v) { return (unsigned)(long long)v; }
AG_DEV unsigned clamp_mass(unsigned m) { return m > AG_CELL_MIN_SIZE ? m : AG_CELL_MIN_SIZE; }  // R: Entities.hpp:171-177

// ---- LDS layout (bytes).  Everything but the pellet base is a compile-time constant for P == 1 -----
#define L_EVV 0                                    // int[AG_EVV_CAP]  virus eat events of the tick
#define L_TMP (L_EVV + 4 * AG_EVV_CAP)             // int[128] mailbox / scratch
#define L_NEW (L_TMP + 4 * 128)                    // created cells [CF_FIELDS][AG_CC]; also the RNG draw buffer (128 x u64)
#define L_PLS (L_NEW + 4 * CF_FIELDS * AG_CC)     // int[P][PL_LDS_STRIDE]
// (+ 4 bytes: without them every player's block starts in the same LDS bank, and a lane-per-player access -- move_all_players, simple_turns, the
// collision pre-test: field f of cell 0 of thirty players -- is a thirty-way bank conflict)
#define CELL_STRIDE (4 * (CF_FIELDS + 3) * AG_CC + 4) // per player: 9 fields + (cached-for mass, radius, max speed)
#define PL_LDS_STRIDE (PL_WORDS + 1)                // the player words in LDS: 25 words apart (24 would put eight players on each of four banks)
#ifdef AGAR_CPU_EMU
static inline
#else
__host__ __device__ inline
#endif
// behind the cells: the arena's viruses (x, y, radius, mass: a read cache, see stage_viruses) and ejected foods (x, y, vx, vy: the working
// copy of a launch, see Foods) -- VC and FC entries each (AgDims) --, then the two arrays whose size follows the arena's pellet DENSITY
// (AgDims::EC / KC): the pellet eat events of the tick (int[EC]) and the candidate records of one cell's ordered replay (2 words x KC; also the 64
// (x, y) pairs staged by add_pellets).  The reference's containers are unbounded (Engine.hpp:976-1009: pellets_to_remove); a mass-1000 cell
// in an 80 x 80 arena with 1300 pellets has 200 of them within its radius and a handful of such cells exceeded the 256 events these arrays
// held at fixed offsets until round 4 -- 3-5 % of the soak's default draw ended flagged.  At the end of the block their size costs the
// default arenas nothing (EC = 256 as before) and the dense ones LDS, not correctness.
size_t ag_lds_layout(int P, int VC, int FC, int EC, int KC, int *cells_off, int *vir_off = nullptr, int *food_off = nullptr, int *evp_off = nullptr, int *cand_off = nullptr) {
  int co = L_PLS + 4 * P * PL_LDS_STRIDE;
  if (cells_off) *cells_off = co;
  int vo = (co + P * CELL_STRIDE + 15) & ~15, fo = vo + 16 * VC, eo = fo + 16 * FC, ko = eo + 4 * EC;
  if (vir_off) *vir_off = vo;
  if (food_off) *food_off = fo;
  if (evp_off) *evp_off = eo;
  if (cand_off) *cand_off = ko;
  return (size_t)ko + (size_t)8 * (size_t)KC;
}

struct Cells {  // LDS arrays of one player
  float *x, *y, *vx, *vy, *sx, *sy; unsigned *m; int *id; unsigned *dl;
  unsigned *cmc; float *crad, *cms;  // radius / max-speed cache, valid for cell i iff cmc[i] == m[i]
};

// diagnostic build only (-DAGAR_PROFILE): per-phase shader-clock accumulation, summed into gs->prof
#if defined(AGAR_PROFILE) && !defined(AGAR_CPU_EMU)
#define AG_NPROF 16
// (volatile asm with a memory clobber: __builtin_readcyclecounter may be scheduled across the phase it is meant to close)
__device__ __forceinline__ unsigned ag_clock32() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory"); return (unsigned)t; }
#ifdef AGAR_PROFILE_LEVELS   // slots 13..15 count relaxation levels visited / with a touching pair / touch-bit passes instead (their phases go to 12)
#define AG_T(c, k) do { unsigned t_ = ag_clock32(); (c).tacc[(k) >= 13 ? 12 : (k)] += t_ - (c).tlast; (c).tlast = t_; } while (0)
#else
#define AG_T(c, k) do { unsigned t_ = ag_clock32(); (c).tacc[k] += t_ - (c).tlast; (c).tlast = t_; } while (0)
#endif
#else
#define AG_T(c, k) do { } while (0)
#endif

// pellets live in registers: lane l, slot s holds pellet s*64+l; unused entries hold a far-away sentinel
// so the scan needs no bounds test.  (test-only host build: plain arrays)
#define AG_PEL_SENTINEL 3.0e38f
// `dirty`: bit s set <=> this lane's slot s differs from HBM (arena_store writes back only those: one eaten pellet is two 8-byte
// stores, not the whole array)
template <int NS> struct Pel {
#ifdef AGAR_CPU_EMU
  float x[NS][64], y[NS][64]; unsigned dirty[64];
#else
  float x[NS], y[NS]; unsigned dirty;
#endif
};
#ifdef AGAR_CPU_EMU
#define AG_PEL_FOR(s, lane, i) for (int s = 0; s < NS; s++) for (int lane = 0, i = s * 64; lane < 64; lane++, i++)
#define PELX(c, s, lane) (c).pel.x[s][lane]
#define PELY(c, s, lane) (c).pel.y[s][lane]
#define PEL_MARK(c, s, lane) ((c).pel.dirty[lane] |= 1u << (s))
#define PEL_DIRTY(c, s, lane) (((c).pel.dirty[lane] >> (s)) & 1u)
#define PEL_CLEAN(c) do { for (int l_ = 0; l_ < 64; l_++) (c).pel.dirty[l_] = 0u; } while (0)
#else
#define AG_PEL_FOR(s, lane, i) _Pragma("unroll") for (int s = 0; s < NS; s++) for (int lane = AG_LANE, i = s * 64 + AG_LANE, once_ = 1; once_; once_ = 0)
#define PELX(c, s, lane) (c).pel.x[s]
#define PELY(c, s, lane) (c).pel.y[s]
#define PEL_MARK(c, s, lane) ((c).pel.dirty |= 1u << (s))
#define PEL_DIRTY(c, s, lane) (((c).pel.dirty >> (s)) & 1u)
#define PEL_CLEAN(c) ((c).pel.dirty = 0u)
#endif

// AV ("all visible"): the pellet grid is at most 2x2 buckets (arena <= 1020), so every bucket is within +-1 of
// every other and the reference's 3x3 bucket walk visits all pellets: no per-pellet bucket test is needed.
template <int NS, bool AV> struct AgCtx {
  Pel<NS> pel;
#if defined(AGAR_PROFILE) && !defined(AGAR_CPU_EMU)
  unsigned tacc[AG_NPROF]; unsigned tlast;
#endif
  const AgState *gs;
  const AG_GLOBAL float *act_dxdy; const AG_GLOBAL int32_t *act;
  int arena, P, PC, cells_off, slot, ts_lg;
  int vir_off, food_off, VC, FC;   // LDS blocks of the virus cache and the food working copy, their capacities (AgDims::VC / FC)
  bool food_dirty;                 // the LDS foods differ from HBM (arena_store writes them back)
  unsigned char *lds;
  UBlock S;    // arena words (AR_*): register-resident for the whole launch
  UBlock PB;   // current player's words (PL_*): valid inside tick_player only; LDS PLS is its home
  int ncreated;
  // end-of-tick table check (arena_tick): lut_risk = some cell had at least half the table size when its player moved this tick (an exact
  // per-cell test folded into move_player's pass); mass_sum = the players' total masses of this tick (they cannot wrap while lut_risk is off)
  bool lut_risk; unsigned mass_sum;
  bool moved_all;   // this tick's kinematics of ALL players have been done in one lane-parallel pass (move_all_players)
  bool pel_dirty, pel_loaded;   // pel_dirty: some slot is marked in pel.dirty
  bool pel_all;                 // every slot is dirty (reset): the store skips the per-slot tests
};

template <int NS, bool AV> AG_DEV int *L_I(const AgCtx<NS, AV> &c, int off) { return (int *)(c.lds + off); }
// the two density-sized arrays behind the foods (ag_lds_layout): formed where they are used -- rare paths -- instead of living in the context
template <int NS, bool AV> AG_DEV int ag_evp_off(const AgCtx<NS, AV> &c) { return c.food_off + 16 * c.FC; }
template <int NS, bool AV> AG_DEV int ag_cand_off(const AgCtx<NS, AV> &c) { return c.food_off + 16 * c.FC + 4 * c.gs->d.EC; }
// the arena's event list in HBM: [0, EC) = the LDS events as arena_store exports them (agarcl_get_events), [EC, EC + EX) = the spill area
template <int NS, bool AV> AG_DEV AG_GLOBAL int32_t *g_evp(const AgCtx<NS, AV> &c) { return (AG_GLOBAL int32_t *)(c.gs->ev_p + (size_t)c.arena * (size_t)(c.gs->d.EC + c.gs->d.EX)); }
template <int NS, bool AV> AG_DEV int *PLS(const AgCtx<NS, AV> &c, int p) { return (int *)(c.lds + L_PLS) + p * PL_LDS_STRIDE; }
template <int NS, bool AV> AG_DEV Cells cells_of(const AgCtx<NS, AV> &c, int p) {
  float *b = (float *)(c.lds + c.cells_off + p * CELL_STRIDE);
  Cells k;
  k.x = b; k.y = b + AG_CC; k.vx = b + 2 * AG_CC; k.vy = b + 3 * AG_CC; k.sx = b + 4 * AG_CC; k.sy = b + 5 * AG_CC;
  k.m = (unsigned *)(b + 6 * AG_CC); k.id = (int *)(b + 7 * AG_CC); k.dl = (unsigned *)(b + 8 * AG_CC);
  k.cmc = (unsigned *)(b + 9 * AG_CC); k.crad = b + 10 * AG_CC; k.cms = b + 11 * AG_CC;
  return k;
}
template <int NS, bool AV> AG_DEV Cells created_of(const AgCtx<NS, AV> &c) {
  float *b = (float *)(c.lds + L_NEW);
  Cells k;
  k.x = b; k.y = b + AG_CC; k.vx = b + 2 * AG_CC; k.vy = b + 3 * AG_CC; k.sx = b + 4 * AG_CC; k.sy = b + 5 * AG_CC;
  k.m = (unsigned *)(b + 6 * AG_CC); k.id = (int *)(b + 7 * AG_CC); k.dl = (unsigned *)(b + 8 * AG_CC);
  k.cmc = nullptr; k.crad = nullptr; k.cms = nullptr;
  return k;
}
// LDS-staged entities (north star: "LDS-staged ... per arena tile").  A mid-game tick used to make three dependent round trips to L2 / HBM
// for them -- the viruses and their radii in the virus test, the foods in the food test, foods and viruses again in the food move -- and 48 %
// of a mid-game wavefront's cycles were waits.  Now:
//   Foods  x, y, vx, vy of every ejected food: THE working copy for the whole launch (loaded once by arena_load, every rule reads and writes
//          LDS, arena_store writes back what is live when anything changed).  Ids stay in HBM: only appends, erases and swaps touch them.
//   Virs   x, y, radius (the mass table's entry: Engine.hpp:1223-1252 needs it per test), mass of every virus: a READ cache.  Viruses change
//          only on events (eaten, fed, regenerated); those paths keep working on HBM -- which also holds velocity, hits and id -- and call
//          stage_viruses() afterwards.
struct Foods { float *x, *y, *vx, *vy; };
struct Virs { float *x, *y, *r; unsigned *m; };
template <int NS, bool AV> AG_DEV Foods foods_of(const AgCtx<NS, AV> &c) { float *b = (float *)(c.lds + c.food_off); Foods f; f.x = b; f.y = b + c.FC; f.vx = b + 2 * c.FC; f.vy = b + 3 * c.FC; return f; }
template <int NS, bool AV> AG_DEV Virs virs_of(const AgCtx<NS, AV> &c) { float *b = (float *)(c.lds + c.vir_off); Virs v; v.x = b; v.y = b + c.VC; v.r = b + 2 * c.VC; v.m = (unsigned *)(b + 3 * c.VC); return v; }
// HBM slices of this arena (pointer + capacity fetched with scalar loads when a rare path needs them)
#define G_SLICE(name, type, field, cap) template <int NS, bool AV> AG_DEV AG_GLOBAL type *name(const AgCtx<NS, AV> &c) { return (AG_GLOBAL type *)(c.gs->field + (size_t)c.arena * (size_t)(cap)); }
G_SLICE(g_pxy, float, pel_xy, 2 * c.PC) G_SLICE(g_pid, int32_t, pel_id, c.PC)
G_SLICE(g_vx, float, vir_x, c.VC) G_SLICE(g_vy, float, vir_y, c.VC) G_SLICE(g_vvx, float, vir_vx, c.VC) G_SLICE(g_vvy, float, vir_vy, c.VC)
G_SLICE(g_vm, int32_t, vir_mass, c.VC) G_SLICE(g_vh, int32_t, vir_hits, c.VC) G_SLICE(g_vid, int32_t, vir_id, c.VC)
G_SLICE(g_fx, float, food_x, c.FC) G_SLICE(g_fy, float, food_y, c.FC) G_SLICE(g_fvx, float, food_vx, c.FC) G_SLICE(g_fvy, float, food_vy, c.FC)
G_SLICE(g_fid, int32_t, food_id, c.FC)
G_SLICE(g_mt, uint64_t, mt, 312)
// tile-transposed arrays (agar_types.h): word w of the returned block is [AG_TW(w)]
template <int NS, bool AV> AG_DEV AG_GLOBAL int32_t *g_ar(const AgCtx<NS, AV> &c) { const int ag_ts_lg = c.ts_lg; return (AG_GLOBAL int32_t *)(c.gs->ar + AG_TILE_BASE(c.arena, AR_WORDS)); }
template <int NS, bool AV> AG_DEV AG_GLOBAL int32_t *g_pl(const AgCtx<NS, AV> &c) { const int ag_ts_lg = c.ts_lg; return (AG_GLOBAL int32_t *)(c.gs->pl + AG_TILE_BASE(c.arena, c.P * PL_WORDS)); }
template <int NS, bool AV> AG_DEV AG_GLOBAL int32_t *g_vt(const AgCtx<NS, AV> &c, int p) { return (AG_GLOBAL int32_t *)(c.gs->vticks + ((size_t)c.arena * c.P + p) * AG_VT_CAP); }
template <int NS, bool AV> AG_DEV AG_GLOBAL uint32_t *g_cells(const AgCtx<NS, AV> &c, int p) { const int ag_ts_lg = c.ts_lg; return (AG_GLOBAL uint32_t *)(c.gs->cells + AG_TILE_BASE(c.arena, c.P * (CF_ALL * AG_CC)) + AG_TW(p * (CF_ALL * AG_CC))); }
template <int NS, bool AV> AG_DEV const AG_GLOBAL float *g_lut_r(const AgCtx<NS, AV> &c) { return (const AG_GLOBAL float *)c.gs->lut_r; }
template <int NS, bool AV> AG_DEV const AG_GLOBAL float *g_lut_ms(const AgCtx<NS, AV> &c) { return (const AG_GLOBAL float *)c.gs->lut_ms; }
template <int NS, bool AV> AG_DEV const AG_GLOBAL float *g_lut_ss(const AgCtx<NS, AV> &c) { return (const AG_GLOBAL float *)c.gs->lut_ss; }

template <int NS, bool AV> AG_DEV int SR(const AgCtx<NS, AV> &c, int k) { return ub_get(c.S, k); }
template <int NS, bool AV> AG_DEV void SW(AgCtx<NS, AV> &c, int k, int v) { ub_set(c.S, k, v); }
template <int NS, bool AV> AG_DEV int PR(const AgCtx<NS, AV> &c, int k) { return ub_get(c.PB, k); }
template <int NS, bool AV> AG_DEV float PRF(const AgCtx<NS, AV> &c, int k) { return u2f(ub_get(c.PB, k)); }
template <int NS, bool AV> AG_DEV void PW(AgCtx<NS, AV> &c, int k, int v) { ub_set(c.PB, k, v); }
template <int NS, bool AV> AG_DEV void flag(AgCtx<NS, AV> &c, unsigned f) { SW(c, AR_FLAGS, SR(c, AR_FLAGS) | (int)f); }
template <class PT> AG_DEV float lut(PT t, unsigned m) { return t[m < (unsigned)AG_LUT_SIZE ? m : (unsigned)AG_LUT_SIZE - 1u]; }
template <int NS, bool AV> AG_DEV float radius_of(const AgCtx<NS, AV> &c, unsigned m) { return lut(g_lut_r(c), m); }
// lane-level radius of LDS cell k of `cs`, through the per-cell cache when it is valid
template <int NS, bool AV> AG_DEV float cell_rad(const AgCtx<NS, AV> &c, const Cells &cs, int k) { unsigned m = cs.m[k]; return cs.cmc[k] == m ? cs.crad[k] : radius_of(c, m); }

// ---- pellet-register collectives: f(x, y, i) over every pellet slot, i = global pellet index ----------
template <int NS, bool AV, class F> AG_DEV bool pel_any(const AgCtx<NS, AV> &c, F f) {
  bool a = false;
  AG_PEL_FOR(s, lane, i) { a = a | (bool)f(PELX(c, s, lane), PELY(c, s, lane), i); }
#ifdef AGAR_CPU_EMU
  return a;
#else
  return __ballot(a) != 0ull;
#endif
}
template <int NS, bool AV, class F> AG_DEV int pel_count(const AgCtx<NS, AV> &c, F f) {
#ifdef AGAR_CPU_EMU
  int n = 0;
  AG_PEL_FOR(s, lane, i) { n += f(PELX(c, s, lane), PELY(c, s, lane), i) ? 1 : 0; }
  return n;
#else
  int n = 0;
  AG_PEL_FOR(s, lane, i) { n += __popcll(__ballot((bool)f(PELX(c, s, lane), PELY(c, s, lane), i))); }
  return n;
#endif
}
// ordered compaction in ascending pellet index: sink(x, y, i, rank)
template <int NS, bool AV, class F, class S> AG_DEV int pel_compact(const AgCtx<NS, AV> &c, F pred, S sink) {
  int count = 0;
#ifdef AGAR_CPU_EMU
  AG_PEL_FOR(s, lane, i) { if (pred(PELX(c, s, lane), PELY(c, s, lane), i)) { sink(PELX(c, s, lane), PELY(c, s, lane), i, count); count++; } }
#else
  const unsigned long long lt = (1ull << AG_LANE) - 1ull;
  AG_PEL_FOR(s, lane, i) {
    bool p = pred(PELX(c, s, lane), PELY(c, s, lane), i);
    unsigned long long m = __ballot(p);
    if (p) sink(PELX(c, s, lane), PELY(c, s, lane), i, count + __popcll(m & lt));
    count += __popcll(m);
  }
#endif
  return count;
}
// Optimisation barrier: tells the compiler the pellet registers "changed" (no instruction is emitted) so that it
// recomputes per-pellet distances in each rare-path pass instead of keeping 16+ of them alive across passes.
template <int NS, bool AV> AG_DEV void pel_launder(AgCtx<NS, AV> &c) {
#ifndef AGAR_CPU_EMU
  _Pragma("unroll") for (int s = 0; s < NS; s++) asm volatile("" : "+v"(c.pel.x[s]), "+v"(c.pel.y[s]));
#endif
}
// pellet[dst] = pellet[src] (wave-level; the swap half of the reference's swap-pop that survives)
template <int NS, bool AV> AG_DEV void pel_move(AgCtx<NS, AV> &c, int dst, int src) {
#ifdef AGAR_CPU_EMU
  c.pel.x[dst >> 6][dst & 63] = c.pel.x[src >> 6][src & 63]; c.pel.y[dst >> 6][dst & 63] = c.pel.y[src >> 6][src & 63];
  PEL_MARK(c, dst >> 6, dst & 63);
#else
  int ss = src >> 6, sl = src & 63, ds = dst >> 6, dl = dst & 63; float vx = 0.0f, vy = 0.0f;
  _Pragma("unroll") for (int s = 0; s < NS; s++) if (s == ss) { vx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.pel.x[s]), sl)); vy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.pel.y[s]), sl)); }
  _Pragma("unroll") for (int s = 0; s < NS; s++) if (s == ds && AG_LANE == dl) { c.pel.x[s] = vx; c.pel.y[s] = vy; PEL_MARK(c, s, 0); }
#endif
}
// uniform read of pellet i
template <int NS, bool AV> AG_DEV void pel_get(const AgCtx<NS, AV> &c, int i, float &x, float &y) {
#ifdef AGAR_CPU_EMU
  x = c.pel.x[i >> 6][i & 63]; y = c.pel.y[i >> 6][i & 63];
#else
  int ss = i >> 6, sl = i & 63; x = 0.0f; y = 0.0f;
  _Pragma("unroll") for (int s = 0; s < NS; s++) if (s == ss) { x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.pel.x[s]), sl)); y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.pel.y[s]), sl)); }
#endif
}

#ifndef AGAR_CPU_EMU
// one 4-byte word per active lane from its own global address into LDS at (wave-uniform base) + lane * 4, without a destination register
template <class GT, class LT> AG_DEV void ag_glds4(GT *g, LT *lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)lds_base, 4, 0, 0);
}
#endif
// ---- load / store arena state between HBM and LDS / registers -------------------------------------
template <int NS, bool AV> AG_DEV void arena_load(AgCtx<NS, AV> &c, bool want_pellets = false) {
  // Every load below is independent of every other (no count is needed to form an address), so the arena arrives in
  // ONE round trip to HBM: arena words, player words, and all AG_CC cell slots of every player incl. the persisted
  // radius / speed cache (slots >= n_cells are never read).  (More than two players: their LIVE cells in a second trip, below.)
  // (pellets are NOT loaded here: ensure_pellets() fetches them on first use, and a launch whose cell provably stays
  // out of reach of every pellet -- AR_SAFE budget -- never touches them)
  // (want_pellets: the caller knows a general tick is coming -- k_step resuming after k_quiet -- so the pellet loads
  // join the same round trip)
  const int ag_ts_lg = c.ts_lg;
  c.pel_loaded = false;
  if (want_pellets) { auto gxy = g_pxy(c); AG_PEL_FOR(s, lane, i) { PELX(c, s, lane) = gxy[2 * i]; PELY(c, s, lane) = gxy[2 * i + 1]; } c.pel_loaded = true; }
  ub_load_t(c.S, g_ar(c), AR_WORDS, ag_ts_lg);
  auto gpl = g_pl(c);
  AG_LANES(i, c.P * PL_WORDS) PLS(c, 0)[i + i / PL_WORDS] = gpl[AG_TW(i)];   // (word w of player p: LDS word p * PL_LDS_STRIDE + w)
#ifndef AGAR_CPU_EMU
  if (c.P <= 2)
#endif
  for (int p = 0; p < c.P; p++) {
    auto g = g_cells(c, p); uint32_t *l = (uint32_t *)(c.lds + c.cells_off + p * CELL_STRIDE);
    AG_LANES(i, AG_CC) {
#ifndef AGAR_CPU_EMU
#pragma unroll
#endif
      for (int f = 0; f < CF_ALL; f++) l[f * AG_CC + i] = g[AG_CELL_W(f, i)];
    }
  }
#ifndef AGAR_CPU_EMU
  else if (AG_LANE < c.P) {
    // more than two players: cell 0 of every player, a lane each, in THIS round trip -- its address needs no count, and it is the only cell of
    // most players (all of bench/main.cpp's); the cells behind it follow in the second trip below, if there are any
    auto g = g_cells(c, AG_LANE); uint32_t *l = (uint32_t *)(c.lds + c.cells_off + AG_LANE * CELL_STRIDE);
#pragma unroll
    for (int f = 0; f < CF_ALL; f++) l[f * AG_CC] = g[AG_CELL_W(f, 0)];
  }
#endif
  // viruses (x, y, mass -> radius) and the first 64 foods: their addresses need no count either, and on the device the words go STRAIGHT
  // into LDS (global_load_lds: wave-uniform LDS base + lane * 4, per-lane source address) -- no destination registers, so the arena's
  // other loads (46 registers in flight with 1000 pellets) keep theirs.  An unfed virus weighs 100, so its radius is the one table entry
  // requested here; only fed viruses (rare) look theirs up after the masses have arrived.
  {
    Virs V = virs_of(c); auto gx = g_vx(c); auto gy = g_vy(c); auto gm = g_vm(c);
    Foods F = foods_of(c); auto fx = g_fx(c); auto fy = g_fy(c); auto fvx = g_fvx(c); auto fvy = g_fvy(c);
    const float r100 = g_lut_r(c)[AG_VIRUS_MASS];
    const int first = c.FC < 64 ? c.FC : 64;
#if defined(AGAR_CPU_EMU) || defined(AG_NO_GLDS)   // (AG_NO_GLDS: measurement builds only -- through registers)
    AG_LANES(i, c.VC) { V.x[i] = gx[i]; V.y[i] = gy[i]; V.m[i] = (unsigned)gm[i]; }
    AG_LANES(i, first) { F.x[i] = fx[i]; F.y[i] = fy[i]; F.vx[i] = fvx[i]; F.vy[i] = fvy[i]; }
#else
    for (int base = 0; base < c.VC; base += 64) {
      const int i = base + AG_LANE;
      if (i < c.VC) { ag_glds4(gx + i, V.x + base); ag_glds4(gy + i, V.y + base); ag_glds4(gm + i, V.m + base); }
    }
    { const int i = AG_LANE; if (i < first) { ag_glds4(fx + i, F.x); ag_glds4(fy + i, F.y); ag_glds4(fvx + i, F.vx); ag_glds4(fvy + i, F.vy); } }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (an LDS-DMA is a pending LDS write on the VM counter)
#endif
    ag_lds_order();
#ifndef AGAR_CPU_EMU
    if (c.P > 2) {
      // Several players (r05): only the LIVE cells, a lane each over the flat list of all players' cells.  All AG_CC slots of every player in one
      // round trip (above) is right for one or two players; thirty single-cell players made it 360 load instructions and 46 KB per arena for 1.4 KB
      // of cells (23 % of a Tick/30 launch with the store, scripts/gpu_phase_multi.py).  The counts cost a second round trip: the player words
      // are in LDS by now.  (Slots >= n_cells are never read; a created cell is written whole.)
      const int ncl0 = AG_LANE < c.P ? PLS(c, AG_LANE)[PL_NCELLS] : 0;   // (P <= 32 players: a lane each)
      const int ncl = ncl0 > 1 ? ncl0 - 1 : 0;                            // cells behind cell 0 (which arrived with the first trip)
      if (__ballot(ncl != 0) != 0ull)
      for (int base = 0;; base += 64) {
        const int t = base + AG_LANE; int p = -1, i = 0, st = 0;
        for (int q = 0; q < c.P; q++) { const int nq = __builtin_amdgcn_readlane(ncl, q); if (t >= st && t < st + nq) { p = q; i = 1 + t - st; } st += nq; }
        if (p >= 0) {
          auto g = g_cells(c, p); uint32_t *l = (uint32_t *)(c.lds + c.cells_off + p * CELL_STRIDE);
#pragma unroll
          for (int f = 0; f < CF_ALL; f++) l[f * AG_CC + i] = g[AG_CELL_W(f, i)];
        }
        if (base + 64 >= st) break;
      }
    }
#endif
    AG_LANES(i, c.VC) { const unsigned m = V.m[i]; V.r[i] = m == AG_VIRUS_MASS ? r100 : lut(g_lut_r(c), m); }
    const int nf = SR(c, AR_NFOOD);
    if (AG_RARE(nf > 64)) { for (int i = 64 + AG_LANE_OR_0; i < nf; i += AG_LANE_STEP) { F.x[i] = fx[i]; F.y[i] = fy[i]; F.vx[i] = fvx[i]; F.vy[i] = fvy[i]; } }
    c.food_dirty = false;
  }
  c.pel_dirty = false; c.pel_all = false; PEL_CLEAN(c); c.ncreated = 0;
  ag_lds_order();
}
// Viruses changed in HBM (eaten, fed, regenerated, reset): the LDS cache is read again.  Lane-level stores to HBM made by this wave come first.
template <int NS, bool AV> AG_DEV void stage_viruses(AgCtx<NS, AV> &c) {
  ag_mem_fence();
  Virs V = virs_of(c); auto gx = g_vx(c); auto gy = g_vy(c); auto gm = g_vm(c);
  const int nv = SR(c, AR_NVIR);
  AG_LANES(i, nv) { const unsigned m = (unsigned)gm[i]; V.x[i] = gx[i]; V.y[i] = gy[i]; V.m[i] = m; V.r[i] = lut(g_lut_r(c), m); }
  ag_lds_order();
}
// Pellet capacity is exactly NS*64 and HBM keeps the sentinel at every index >= n_pellets, so the load is NS
// unconditional 8-byte-per-lane loads (512 B per wave-instruction), all in flight together.
template <int NS, bool AV> AG_DEV void ensure_pellets(AgCtx<NS, AV> &c) {
  if (c.pel_loaded) return;
  auto gxy = g_pxy(c);
  AG_PEL_FOR(s, lane, i) { PELX(c, s, lane) = gxy[2 * i]; PELY(c, s, lane) = gxy[2 * i + 1]; }
  c.pel_loaded = true;
}
// the pellet registers that differ from HBM go back (after a reset: the whole register file incl. sentinels)
template <int NS, bool AV> AG_DEV void pellets_store(AgCtx<NS, AV> &c) {
  if (!c.pel_dirty) return;
  auto gxy = g_pxy(c);
  if (c.pel_all) { AG_PEL_FOR(s, lane, i) { gxy[2 * i] = PELX(c, s, lane); gxy[2 * i + 1] = PELY(c, s, lane); } }
  else { AG_PEL_FOR(s, lane, i) { if (PEL_DIRTY(c, s, lane)) { gxy[2 * i] = PELX(c, s, lane); gxy[2 * i + 1] = PELY(c, s, lane); } } }
}
template <int NS, bool AV> AG_DEV void arena_store(AgCtx<NS, AV> &c) {
  ag_lds_order();
  const int ag_ts_lg = c.ts_lg;
  int np = SR(c, AR_NPEL);
  pellets_store(c);
  int total_cells = 0;
#ifndef AGAR_CPU_EMU
  if (c.P > 2) {   // several players: a lane per live cell over the flat list (as arena_load) instead of a dependent LDS read and twelve stores per player
    const int ncl = AG_LANE < c.P ? PLS(c, AG_LANE)[PL_NCELLS] : 0;
    const bool one_each = __ballot(AG_LANE < c.P && ncl != 1) == 0ull;
    for (int base = 0;; base += 64) {
      const int t = base + AG_LANE; int p = -1, i = 0, st = 0;
      if (one_each) { p = t < c.P ? t : -1; st = c.P; }   // (every player has exactly one cell -- bench/main.cpp's populations, most of C1's ticks: the flat list IS the player list)
      else for (int q = 0; q < c.P; q++) { const int nq = __builtin_amdgcn_readlane(ncl, q); if (t >= st && t < st + nq) { p = q; i = t - st; } st += nq; }
      if (p >= 0) {
        auto g = g_cells(c, p); const uint32_t *l = (const uint32_t *)(c.lds + c.cells_off + p * CELL_STRIDE);
#pragma unroll
        for (int f = 0; f < CF_ALL; f++) g[AG_CELL_W(f, i)] = l[f * AG_CC + i];
      }
      total_cells = st;
      if (base + 64 >= st) break;
    }
  } else
#endif
  for (int p = 0; p < c.P; p++) {
    int n = ag_uni(PLS(c, p)[PL_NCELLS]);
    total_cells += n;
    auto g = g_cells(c, p); const uint32_t *l = (const uint32_t *)(c.lds + c.cells_off + p * CELL_STRIDE);
    AG_LANES(i, n) {
#ifndef AGAR_CPU_EMU
#pragma unroll
#endif
      for (int f = 0; f < CF_ALL; f++) g[AG_CELL_W(f, i)] = l[f * AG_CC + i];
    }
  }
  if (c.food_dirty) {  // the live foods (ids are kept current in HBM by the rules themselves)
    Foods F = foods_of(c); auto fx = g_fx(c); auto fy = g_fy(c); auto fvx = g_fvx(c); auto fvy = g_fvy(c);
    AG_LANES(i, SR(c, AR_NFOOD)) { fx[i] = F.x[i]; fy[i] = F.y[i]; fvx[i] = F.vx[i]; fvy[i] = F.vy[i]; }
  }
  ub_store_t(c.S, g_ar(c), AR_WORDS, ag_ts_lg);
  {  // diagnostics (agarcl_debug_work): pellet-array transfers of this launch; flag watch word (agarcl_poll_flags)
    const int moved = (c.pel_loaded ? 1 : 0) + (c.pel_dirty && c.pel_all ? 1 : 0);   // (dirty-slot stores are a few 8-byte words: not counted as a transfer)
    if (moved) { AG_SERIAL { PLS(c, 0)[PL_PASSES] += moved; } ag_lds_order(); }
    const int fl = SR(c, AR_FLAGS);
    if (AG_RARE(fl != 0)) { AG_SERIAL { ag_atomic_or(c.gs->qstat + 1, fl); } }
  }
  auto gpl = g_pl(c);
  AG_LANES(i, c.P * PL_WORDS) gpl[AG_TW(i)] = PLS(c, 0)[i + i / PL_WORDS];
  int nevp = SR(c, AR_NEVP), nevv = SR(c, AR_NEVV);
  if (nevp > 0) { const int EC = c.gs->d.EC; auto ge = g_evp(c); int lim = nevp < EC ? nevp : EC; AG_LANES(i, lim) ge[i] = L_I(c, ag_evp_off(c))[i]; }
  if (nevv > 0) { auto ge = (AG_GLOBAL int32_t *)(c.gs->ev_v + (size_t)c.arena * AG_EVV_CAP); int lim = nevv < AG_EVV_CAP ? nevv : AG_EVV_CAP; AG_LANES(i, lim) ge[i] = L_I(c, L_EVV)[i]; }
  int nv = SR(c, AR_NVIR), nf = SR(c, AR_NFOOD);
  AG_SERIAL { auto cn = (AG_GLOBAL int32_t *)(c.gs->counts + (size_t)c.arena * 4); cn[0] = np; cn[1] = nv; cn[2] = nf; cn[3] = total_cells; }
}

// ---- mt19937_64, one generator per arena, state in HBM.  R: GameState.hpp:51, Engine.hpp:1304-1311
AG_DEV uint64_t mt_temper(uint64_t z) {
  z ^= (z >> 29) & 0x5555555555555555ULL; z ^= (z << 17) & 0x71D67FFFEDA60000ULL; z ^= (z << 37) & 0xFFF7EEE000000000ULL; z ^= (z >> 43);
  return z;
}
AG_DEV uint64_t mt_mix(uint64_t a, uint64_t b, uint64_t far) {
  uint64_t y = (a & 0xFFFFFFFF80000000ULL) | (b & 0x7FFFFFFFULL);
  return far ^ (y >> 1) ^ ((y & 1ULL) ? 0xB5026F5AA96619E9ULL : 0ULL);
}
template <int NS, bool AV> AG_DEV void mt_twist(AgCtx<NS, AV> &c) {  // three dependency phases, each data-parallel across the wave
  auto mt = g_mt(c);
  ag_mem_fence();
  for (int base = 0; base < 156; base += 64) {  // phase 1: reads old values only
    int lim = base + 64 < 156 ? base + 64 : 156;
#ifdef AGAR_CPU_EMU
    for (int i = base; i < lim; i++) mt[i] = mt_mix(mt[i], mt[i + 1], mt[i + 156]);
#else
    int i = base + AG_LANE; uint64_t v = 0;
    if (i < lim) v = mt_mix(mt[i], mt[i + 1], mt[i + 156]);
    ag_mem_fence();
    if (i < lim) mt[i] = v;
    ag_mem_fence();
#endif
  }
  for (int base = 156; base < 311; base += 64) {  // phase 2: far operand is a phase-1 result
    int lim = base + 64 < 311 ? base + 64 : 311;
#ifdef AGAR_CPU_EMU
    for (int i = base; i < lim; i++) mt[i] = mt_mix(mt[i], mt[i + 1], mt[i - 156]);
#else
    int i = base + AG_LANE; uint64_t v = 0;
    if (i < lim) v = mt_mix(mt[i], mt[i + 1], mt[i - 156]);
    ag_mem_fence();
    if (i < lim) mt[i] = v;
    ag_mem_fence();
#endif
  }
  AG_SERIAL { mt[311] = mt_mix(mt[311], mt[0], mt[155]); }
  ag_mem_fence();
}
// fill rb[0..n) (n <= 128, LDS) with the next n raw 64-bit outputs
template <int NS, bool AV> AG_DEV void mt_fill(AgCtx<NS, AV> &c, int n) {
  uint64_t *rb = (uint64_t *)(c.lds + L_NEW);
  int produced = 0;
  while (produced < n) {
    int idx = SR(c, AR_MTIDX);
    if (idx >= 312) { mt_twist(c); idx = 0; }
    int take = 312 - idx < n - produced ? 312 - idx : n - produced;
    auto mt = g_mt(c);
    AG_LANES(j, take) rb[produced + j] = mt_temper(mt[idx + j]);
    SW(c, AR_MTIDX, idx + take);
    produced += take;
  }
  ag_lds_order();
}
// std::uniform_real_distribution<float>(0,max) from one raw draw.  R: utils/random.hpp:6-20
AG_DEV float mt_to_float(uint64_t u, float maxv) {
  float s = (float)u;
  float r = ag_divf(s, 18446744073709551616.0f);
  if (r >= 1.0f) r = 0.99999994f;  // nextafter(1,0)
  float v = r * maxv;               // (hi - lo) with lo == 0
  return v + 0.0f;
}
// `count` random_location(radius) draws in sequence; sink(j, x, y).  R: Engine.hpp:143-148
template <int NS, bool AV, class SINK> AG_DEV void draw_locations(AgCtx<NS, AV> &c, int count, float radius, SINK sink) {
  float two_r = 2.0f * radius; float span = c.gs->g.W - two_r;
  const uint64_t *rb = (const uint64_t *)(c.lds + L_NEW);
  for (int done = 0; done < count; done += 64) {
    int b = count - done < 64 ? count - done : 64;
    mt_fill(c, 2 * b);
    AG_LANES(j, b) {
      float x = mt_to_float(rb[2 * j], span) + radius;
      float y = mt_to_float(rb[2 * j + 1], span) + radius;
      sink(done + j, x, y);
    }
    ag_lds_order();
  }
}

// ---- spawning.  R: Engine.hpp:418-424, 480-485, 426-475, 119-137 -----------------------------------
template <int NS, bool AV> AG_DEV void add_pellets(AgCtx<NS, AV> &c, int n) {
  if (n <= 0) return;
  int np = SR(c, AR_NPEL), idc = SR(c, AR_IDC);
  if (np + n > c.PC) { flag(c, 64u); n = c.PC - np; if (n <= 0) return; }
  float r = radius_of(c, AG_PELLET_MASS);
  auto gid = g_pid(c);
  float *stage = (float *)L_I(c, ag_cand_off(c));  // 64 (x,y) pairs; the candidate list is idle during regen
  float two_r = 2.0f * r; float span = c.gs->g.W - two_r;
  const uint64_t *rb = (const uint64_t *)(c.lds + L_NEW);
  for (int done = 0; done < n; done += 64) {  // random_location(r) x n in sequence.  R: Engine.hpp:143-148, 418-424
    int b = n - done < 64 ? n - done : 64;
    mt_fill(c, 2 * b);
    int base = np + done;
    AG_LANES(j, b) {
      stage[2 * j] = mt_to_float(rb[2 * j], span) + r;
      stage[2 * j + 1] = mt_to_float(rb[2 * j + 1], span) + r;
      gid[base + j] = idc + 1 + done + j;
    }
    ag_lds_order();
    AG_PEL_FOR(s, lane, i) { if (i >= base && i < base + b) { PELX(c, s, lane) = stage[2 * (i - base)]; PELY(c, s, lane) = stage[2 * (i - base) + 1]; PEL_MARK(c, s, lane); } }
    ag_lds_order();
  }
  SW(c, AR_NPEL, np + n); SW(c, AR_IDC, idc + n);
  c.pel_dirty = true;
  // (several players: the remembered "no pellet inside this cell" verdicts of simple_turns hold only while pellets disappear)
  if (c.P > 1) { AG_LANES(p, c.P) PLS(c, p)[PL_CAND_IDX] = -1; ag_lds_order(); }
}
template <int NS, bool AV> AG_DEV void add_viruses(AgCtx<NS, AV> &c, int n) {
  if (n <= 0) return;
  int nv = SR(c, AR_NVIR), idc = SR(c, AR_IDC), VC = c.gs->d.VC;
  if (nv + n > VC) { flag(c, 4u); n = VC - nv; if (n <= 0) return; }
  float r = radius_of(c, AG_VIRUS_MASS);
  auto vx = g_vx(c); auto vy = g_vy(c); auto vvx = g_vvx(c); auto vvy = g_vvy(c); auto vm = g_vm(c); auto vh = g_vh(c); auto vid = g_vid(c);
  draw_locations(c, n, r, [&](int j, float x, float y) {
    int k = nv + j; vx[k] = x; vy[k] = y; vvx[k] = 0.0f; vvy[k] = 0.0f; vm[k] = (int)AG_VIRUS_MASS; vh[k] = 0; vid[k] = idc + 1 + j; });
  SW(c, AR_NVIR, nv + n); SW(c, AR_IDC, idc + n);
  stage_viruses(c);
}
template <int NS, bool AV> AG_DEV void create_squared_pellets(AgCtx<NS, AV> &c) {
  float W = c.gs->g.W;
  float square = ag_divf(W, 2.0f);
  int pps = f2i(square);
  float cx = ag_divf(W, 2.0f), half = ag_divf(square, 2.0f);
  int idc = SR(c, AR_IDC);
  // every generated point lies inside the arena (centre +- W/4), so all 4*pps are kept, in order
  int total = 4 * pps;
  if (total > c.PC) { flag(c, 64u); total = c.PC; }
  auto gid = g_pid(c);
  AG_PEL_FOR(s, lane, k) {
    if (k < total) {
      int side = k / pps, i = k - side * pps; float t = (float)i * 1.0f; float x, y;
      if (side == 0) { x = (cx - half) + t; y = cx - half; }
      else if (side == 1) { x = cx + half; y = (cx - half) + t; }
      else if (side == 2) { x = (cx + half) - t; y = cx + half; }
      else { x = cx - half; y = (cx + half) - t; }
      PELX(c, s, lane) = x; PELY(c, s, lane) = y; PEL_MARK(c, s, lane); gid[k] = idc + 1 + k;
    }
  }
  SW(c, AR_NPEL, total); SW(c, AR_IDC, idc + total);
  c.pel_dirty = true;
  ag_lds_order();
}
// Engine::respawn on player slot p, operating on its LDS words (called outside tick_player).
template <int NS, bool AV> AG_DEV void respawn(AgCtx<NS, AV> &c, int p) {
  int *P = PLS(c, p);
  AG_SERIAL {  // Player::kill, R: core/Player.hpp:75-86
    P[PL_NCELLS] = 0; P[PL_MIN_MASS] = (int)AG_CELL_MIN_SIZE; P[PL_SPLIT_CD] = 0; P[PL_FEED_CD] = 0;
    P[PL_ANTI_TEAM] = f2u(1.0f); P[PL_ELAPSED] = 0; P[PL_LAST_DECAY] = 0; P[PL_NVTICKS] = 0;
  }
  unsigned pm = (unsigned)(c.gs->g.agent_mass > (int)AG_CELL_MIN_SIZE ? c.gs->g.agent_mass : (int)AG_CELL_MIN_SIZE);
  float r25 = radius_of(c, AG_CELL_MIN_SIZE), W = c.gs->g.W;
  Cells cs = cells_of(c, p);
  if (SR(c, AR_NPEL) > 0 && c.gs->g.squared) {
    ensure_pellets(c);
    float x, y; pel_get(c, 0, x, y);
    float t = 2.0f * r25; x += t; y += t;
    x = sminf(x, W - r25); y = sminf(y, W - r25);
    AG_SERIAL { cs.x[0] = x; cs.y[0] = y; }
  } else {
    draw_locations(c, 1, r25, [&](int, float x, float y) { cs.x[0] = x; cs.y[0] = y; });
  }
  int idc = SR(c, AR_IDC) + 1; SW(c, AR_IDC, idc);
  unsigned clock = (unsigned)SR(c, AR_CLOCK);
  AG_SERIAL {
    cs.vx[0] = 0; cs.vy[0] = 0; cs.sx[0] = 0; cs.sy[0] = 0; cs.m[0] = clamp_mass(pm); cs.id[0] = idc; cs.dl[0] = clock;
    // the radius / speed cache is filled right away, so that the first step after a (re)spawn can already be a quiet one
    cs.cmc[0] = clamp_mass(pm); cs.crad[0] = lut(g_lut_r(c), clamp_mass(pm)); cs.cms[0] = lut(g_lut_ms(c), clamp_mass(pm));
    P[PL_NCELLS] = 1;
  }
  ag_lds_order();
}

// ---- movement.  R: Engine.hpp:609-630, 695-698; core/types.hpp:176-223; Entities.hpp:161-164 ------
AG_DEV void boundary(float W, float &x, float &y, float r) {
  x = smaxf(0.0f, clampf(x, r, W - r));
  y = smaxf(0.0f, clampf(y, r, W - r));
}
AG_DEV void v_decelerate(float &dx, float &dy, float decel, float dt) {
  float xr = ag_divf(dx, vmag(dx, dy));
  float yr = ag_divf(dy, vmag(dx, dy));
  float ddx = xr * decel;
  if (fabsf(ddx * dt) <= fabsf(dx)) { float t = ddx * dt; dx -= t; } else dx = 0.0f;
  float ddy = yr * decel;
  if (fabsf(ddy * dt) <= fabsf(dy)) { float t = ddy * dt; dy -= t; } else dy = 0.0f;
}
AG_DEV float v_direction(float dx, float dy) {  // R: types.hpp:167-174
  float angle = ag_atanf(ag_divf(dx, dy));
  if (dx < 0) { if (dy > 0) angle = (float)((double)angle + 3.14159265358979323846); else angle = (float)((double)angle - 3.14159265358979323846); }
  return angle;
}

// pair-level pieces of the self-collision relaxation.  A visit of pair (a, b) reads both cells from LDS ONCE into
// lane-private registers (CellR), runs the reference's sequence of fp32 operations on them and writes back only what
// can change (position and velocity): LDS sees 16 reads + 8 writes per visit instead of a round trip per operand.
struct CellR { float x, y, vx, vy, sx, sy, r; unsigned m; };
AG_DEV CellR cellr_load(const Cells &s, int k) { CellR c; c.x = s.x[k]; c.y = s.y[k]; c.vx = s.vx[k]; c.vy = s.vy[k]; c.sx = s.sx[k]; c.sy = s.sy[k]; c.r = s.crad[k]; c.m = s.m[k]; return c; }
AG_DEV void cellr_store(const Cells &s, int k, const CellR &c) { s.x[k] = c.x; s.y[k] = c.y; s.vx[k] = c.vx; s.vy[k] = c.vy; }
AG_DEV void cell_move1(CellR &c, float dt) {
  float sx = c.vx + c.sx; float tx = sx * dt; c.x += tx;
  float sy = c.vy + c.sy; float ty = sy * dt; c.y += ty;
}
AG_DEV void avoid_static_overlap(CellR &A, CellR &B, float W) {  // R: Engine.hpp:701-749
  float dx = B.x - A.x, dy = B.y - A.y;
  float dist = vmag(dx, dy);
  float ra = A.r, rb = B.r;
  float target = ra + rb;
  if (dist > target) return;
  float den = fabsf(dx) + fabsf(dy);
  float xr = ag_divf(dx, den), yr = ag_divf(dy, den);
  float depth = target - dist;
  float a1 = 0.5f, a2 = 0.5f, b1 = 0.5f, b2 = 0.5f;
  if (A.x == ra || A.x == W - ra) { a1 = 1.0f; A.vx = 0; }
  if (A.y == ra || A.y == W - ra) { a2 = 1.0f; A.vy = 0; }
  if (B.x == rb || B.x == W - rb) { b1 = 1.0f; B.vx = 0; }
  if (B.y == rb || B.y == W - rb) { b2 = 1.0f; B.vy = 0; }
  float t;
  t = xr * depth; t = t * a1; A.x -= t;
  t = yr * depth; t = t * a2; A.y -= t;
  t = xr * depth; t = t * b1; B.x += t;
  t = yr * depth; t = t * b2; B.y += t;
  boundary(W, A.x, A.y, ra);
  boundary(W, B.x, B.y, rb);
}
AG_DEV void elastic(CellR &A, CellR &B, float dx, float dy, float dist) {  // R: Engine.hpp:893-938
  float nx = ag_divf(dx, dist), ny = ag_divf(dy, dist);
  float tx = -ny, ty = nx;
  float p1 = A.vx * nx, p2 = A.vy * ny; float dpN1 = p1 + p2;
  p1 = B.vx * nx; p2 = B.vy * ny; float dpN2 = p1 + p2;
  p1 = A.vx * tx; p2 = A.vy * ty; float dpT1 = p1 + p2;
  p1 = B.vx * tx; p2 = B.vy * ty; float dpT2 = p1 + p2;
  int m1 = (int)A.m, m2 = (int)B.m;
  float q1 = dpN1 * (float)(m1 - m2);
  float q2 = 2.0f * (float)m2; q2 = q2 * dpN2;
  float v1 = ag_divf(q1 + q2, (float)(m1 + m2));
  q1 = dpN2 * (float)(m2 - m1);
  q2 = 2.0f * (float)m1; q2 = q2 * dpN1;
  float v2 = ag_divf(q1 + q2, (float)(m1 + m2));
  if (A.m <= B.m) { float u = tx * dpT1, w = nx * v1; A.vx = u + w; u = ty * dpT1; w = ny * v1; A.vy = u + w; }
  if (A.m >= B.m) { float u = tx * dpT2, w = nx * v2; B.vx = u + w; u = ty * dpT2; w = ny * v2; B.vy = u + w; }
}
template <int NS, bool AV> AG_DEV bool cells_touch(const AgCtx<NS, AV> &c, const Cells &s, int a, int b) {
  return touches(s.x[a], s.y[a], cell_rad(c, s, a), s.x[b], s.y[b], cell_rad(c, s, b));
}
// avoid_static_overlap (small == true) or separate_cells (small == false) in one body: both start with the same
// distance / direction / depth computation (one sqrt, two divides), which lanes taking different branches would
// otherwise execute twice.  Same operations in the same order as the two functions above.
AG_DEV void resolve_overlap(CellR &A, CellR &B, bool small, float tx, float ty, float W) {
  float dx = B.x - A.x, dy = B.y - A.y;
  float dist = vmag(dx, dy);
  float ra = A.r, rb = B.r;
  float target = ra + rb;
  if (dist > target) return;
  float den = fabsf(dx) + fabsf(dy);
  float xr = ag_divf(dx, den), yr = ag_divf(dy, den);
  float depth = target - dist;
  float xd = xr * depth, yd = yr * depth;
  if (small) {  // R: Engine.hpp:701-749
    float a1 = 0.5f, a2 = 0.5f, b1 = 0.5f, b2 = 0.5f;
    if (A.x == ra || A.x == W - ra) { a1 = 1.0f; A.vx = 0; }
    if (A.y == ra || A.y == W - ra) { a2 = 1.0f; A.vy = 0; }
    if (B.x == rb || B.x == W - rb) { b1 = 1.0f; B.vx = 0; }
    if (B.y == rb || B.y == W - rb) { b2 = 1.0f; B.vy = 0; }
    float t;
    t = xd * a1; A.x -= t;
    t = yd * a2; A.y -= t;
    t = xd * b1; B.x += t;
    t = yd * b2; B.y += t;
    boundary(W, A.x, A.y, ra);
    boundary(W, B.x, B.y, rb);
  } else {      // R: Engine.hpp:803-848
    float diff_a = sqr_dist(tx, ty, A.x, A.y);
    float diff_b = sqr_dist(tx, ty, B.x, B.y);
    int s1 = A.m < B.m ? 1 : -1;
    int s2 = diff_a >= diff_b ? 1 : -1;
    int sg = (s1 == s2) ? s2 : 0;
    const bool ta = A.m < B.m;  // the lighter cell gives way
    float fs = (float)sg, px = ta ? A.x : B.x, py = ta ? A.y : B.y;
    float tx_ = xd * fs, ty_ = yd * fs;
    if (dx >= 0) px -= tx_; else px += tx_;
    if (dy >= 0) py -= ty_; else py += ty_;
    if (ta) { A.x = px; A.y = py; } else { B.x = px; B.y = py; }
  }
}
AG_DEV void prevent_overlap(CellR &A, CellR &B, float dt, float tx, float ty, float W) {  // R: Engine.hpp:857-888
  float dx = B.x - A.x, dy = B.y - A.y;
  float dist = vmag(dx, dy);
  float ra = A.r, rb = B.r;
  float target = ra + rb;
  if (dist > target) return;
  float u, t;
  u = A.vx + A.sx; t = u * dt; A.x -= t;
  u = A.vy + A.sy; t = u * dt; A.y -= t;
  u = B.vx + B.sx; t = u * dt; B.x -= t;
  u = B.vy + B.sy; t = u * dt; B.y -= t;
  elastic(A, B, dx, dy, dist);
  cell_move1(A, dt);
  cell_move1(B, dt);
  if (touches(A.x, A.y, ra, B.x, B.y, rb)) {
    int d = (int)(A.m - B.m);
    resolve_overlap(A, B, (d < 0 ? -d : d) <= 10, tx, ty, W);
  }
  boundary(W, A.x, A.y, ra);
  boundary(W, B.x, B.y, rb);
}
#ifndef AGAR_CPU_EMU
// ---- a pair visit on FOUR lanes --------------------------------------------------------------------------------------
// A level of the relaxation keeps at most n / 2 lanes busy with one pair each, and a visit of a touching pair is a
// dependent chain of ~500 instructions (8 IEEE divisions, 3 square roots, two cells x two components of everything).
// The chain, not the lane count, is what a tick waits for.  So a pair gets a QUAD of lanes: lane (c, k) owns component k
// (0 = x, 1 = y) of cell c (0 = A, the pair's first cell; 1 = B).  Everything per-component / per-cell (un-move, re-move,
// boundary clamps, products, the new velocity, the push) is done once per lane instead of four times per lane; sums over
// the two components (squared distances, dot products, |dx| + |dy|) are one DPP add with the neighbour lane, values of the
// other cell one DPP move.  Same fp32 operations on the same operands as the scalar form (prevent_overlap & co above:
// the host emulation keeps using those), only distributed: a + b is taken as b + a on the other lane (an exact identity in IEEE
// arithmetic; B - A is NOT taken as -(A - B): that would turn +0 into -0 when the components are equal).  ~135 instructions per lane instead of ~400.
AG_DEV float q_comp(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)); }   // quad_perm [1,0,3,2]: the other component
AG_DEV float q_cell(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)); }   // quad_perm [2,3,0,1]: the other cell
AG_DEV unsigned q_cellu(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); }
struct QuadK { bool cB, kY; float W, dt, tk; };   // lane role (cell B? component y?), arena width, dt, this lane's component of the player's target
AG_DEV float q_boundary(float p, float r, float W) { return smaxf(0.0f, clampf(p, r, W - r)); }
// avoid_static_overlap (small) / separate_cells (!small) on this lane's component p (velocity v) of its cell -- Engine.hpp:701-749, 803-848
AG_DEV void q_push(const QuadK &q, bool want, bool small, float r, float ro, unsigned m, unsigned mo, float &p, float &v) {
  const float po = q_cell(p);
  const float d = (q.cB ? p : po) - (q.cB ? po : p);        // B - A, this component (operands selected, not the sign flipped: x - x must stay +0)
  const float sq = d * d, dist = ag_sqrtf_lean(sq + q_comp(sq));
  const float target = r + ro;
  const bool go = want && !(dist > target);
  const float ad = fabsf(d), den = ad + q_comp(ad);
  const float xr = ag_divf(d, den);
  const float depth = target - dist;
  const float xd = xr * depth;
  // small: half the depth each, all of it for a cell that sits on a wall (which also loses that velocity component)
  const bool on = p == r || p == q.W - r;
  float t = xd * (on ? 1.0f : 0.5f);
  float ps = q.cB ? p + t : p - t;
  ps = q_boundary(ps, r, q.W);
  // large mass difference: the lighter cell gives way
  const float dd = fabsf(q.tk - p), ds = dd * dd, diff = ds + q_comp(ds), diffo = q_cell(diff);
  const float diffA = q.cB ? diffo : diff, diffB = q.cB ? diff : diffo;
  const unsigned mA = q.cB ? mo : m, mB = q.cB ? m : mo;
  const bool ta = mA < mB;
  const int s1 = ta ? 1 : -1, s2 = diffA >= diffB ? 1 : -1;
  const float fs = (float)((s1 == s2) ? s2 : 0);
  t = xd * fs;
  const float pl = (d >= 0) ? p - t : p + t;
  const bool moves = q.cB ? !ta : ta;
  p = go ? (small ? ps : (moves ? pl : p)) : p;
  v = (go && small && on) ? 0.0f : v;
}
// one visit: t0 = the pair touches (wave-level caller has tested it); stat = the final static sweep
AG_DEV void q_visit(const QuadK &q, bool t0, bool stat, float r, float ro, unsigned m, unsigned mo, float sv, float &p, float &v) {
  if (stat) { q_push(q, t0, true, r, ro, m, mo, p, v); return; }   // (wave-uniform except where sweeps 4 and 5 share a level)
  const float po = q_cell(p);
  const float d = (q.cB ? p : po) - (q.cB ? po : p);
  const float sq = d * d, dist = ag_sqrtf_lean(sq + q_comp(sq));
  const float target = r + ro;
  const bool go = t0 && !(dist > target);
  float u = v + sv, t = u * q.dt;
  float pn = p - t;                                           // take back this tick's move
  // elastic collision (Engine.hpp:893-938): the lighter cell (both when equal) gets a new velocity
  const float n = ag_divf(d, dist);
  const float nsw = q_comp(n), ek = q.kY ? nsw : -nsw;        // tangent: (-ny, nx)
  const float t1 = v * n, dpN = t1 + q_comp(t1), dpNo = q_cell(dpN);
  const float t2 = v * ek, dpT = t2 + q_comp(t2);
  const int mi = (int)m, mio = (int)mo;
  float q1 = dpN * (float)(mi - mio);
  float q2 = 2.0f * (float)mio; q2 = q2 * dpNo;
  const float vn = ag_divf(q1 + q2, (float)(mi + mio));
  const float uu = ek * dpT, ww = n * vn;
  float vv = (m <= mo) ? uu + ww : v;
  u = vv + sv; t = u * q.dt; pn = pn + t;                     // move again with the new velocity
  const float dk = fabsf(pn - q_cell(pn)), s2 = dk * dk;
  const float rr = target * target;
  const bool again = go && rr >= (s2 + q_comp(s2)) + 0.0f;
  if (ag_any(again)) {
    const int dm = (int)(m - mo);
    q_push(q, again, (dm < 0 ? -dm : dm) <= 10, r, ro, m, mo, pn, vv);
  }
  pn = q_boundary(pn, r, q.W);
  p = go ? pn : p; v = go ? vv : v;
}
#endif

// R: Engine.hpp:763-794.  The reference makes up to 5 sweeps over the pairs (a,b), a<b, in lexicographic order
// (prevent_overlap on touching pairs; stop after a sweep in which nothing touched) and, if the 5th still found an
// overlap, a 6th sweep of avoid_static_overlap.  Every visit touches only cells a and b, so two visits commute unless
// they share a cell.
// Schedule: visit (s, a, b) of sweep s runs at level  t = s * D + a + b,  D = min(n, 2n - 3), one pair per lane.  Any two
// visits that share a cell keep the reference's order (pairs of one sweep with equal a + b share no cell and every earlier
// pair sharing a cell has a smaller a + b; across sweeps, cell a's last visit of sweep s is (a, n-1) at s D + a + n - 1 <
// (s + 1) D + a + b; tests/test_sweep_schedule.py checks all of it exhaustively), so the result equals the sequential
// loops'.  Consecutive sweeps overlap: 5 D + 2n - 3 levels instead of 6 (2n - 3) -- 74 instead of 114 at n = 11 -- and a
// level never holds more than n / 2 pairs.  A sweep started before its predecessor is known to have found an overlap is
// harmless: if the predecessor finds none, no cell has moved and every later visit is a no-op (both prevent_overlap and
// avoid_static_overlap act on touching pairs only); the loop still ends at the first completed sweep without a hit.
// Issue priority of this wavefront among the (up to four) wavefronts of its SIMD, from the relaxation work in front of it.  A launch lasts
// as long as its slowest arena, and the slowest are not the ones with many cells (41 % of the mode-6 arenas have all 14) but the ones whose
// cells sit in one dense clump: every level touches, 1.1 M cycles of relaxation per step against a mean of 0.4 M (scripts/
// gpu_arena_spread.py: max / mean arena-step = 2.4, the same arenas step after step).  The wavefronts sharing their SIMD have slack, so the
// loaded ones go first: priority from the number of local levels with a touching pair.
#ifndef AG_PRIO_T3
#define AG_PRIO_T3 20
#define AG_PRIO_T2 14
#define AG_PRIO_T1 8
#endif
AG_DEV void ag_set_priority(int load) {
#ifndef AGAR_CPU_EMU
#ifndef AG_NO_PRIO
  if (load >= AG_PRIO_T3) __builtin_amdgcn_s_setprio(3); else if (load >= AG_PRIO_T2) __builtin_amdgcn_s_setprio(2); else if (load >= AG_PRIO_T1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
#else
  (void)load;
#endif
}
// Level skipping: a level whose pairs all fail the touch test is a no-op (both prevent_overlap and avoid_static_overlap act on touching
// pairs only), and 52 of the 72 levels of a mode-6 tick are such.  So the touch test of ALL pairs is made in one pass -- one lane per pair,
// the result OR-reduced into one bit per local level L = a + b ("some pair of level L touches at the current positions") -- and repeated
// only after a level that moved a cell.  The walk over the levels then goes straight from one level with a set bit to the next (a shift
// and a count-trailing-zeros on the 64-bit mask): the levels in between are never iterated.  (Measured on MI355X: iterating them with a
// cheap scalar test per level was 45 % SLOWER than the old loop although it executed fewer vector instructions -- every skipped level was
// a handful of taken branches between small blocks of a 150 KB kernel, each an instruction-fetch stall; the walk below has none.)
// Phases: sweep sN starts when sweep sN - 1 reaches its local level D + 1, so phase sN holds the levels LN = 1 .. D of sweep sN together
// with LO = LN + D of sweep sN - 1 (phase 5, the static sweep, runs on to LN = LL).
#if defined(AGAR_CPU_EMU) && defined(AGAR_STATS_LEVELS)
// measurement only (tests/emu, -DAGAR_STATS_LEVELS): how many visited levels could be taken two at a time
struct AgLevelStats { long calls, visited, merged_away, misspec, calls_dense, visited_dense, merged_dense, misspec_dense, by_n[33][3]; };
static AgLevelStats ag_level_stats;
extern "C" void agarcl_emu_level_stats(long *out) { memcpy(out, &ag_level_stats, sizeof(ag_level_stats)); }
extern "C" void agarcl_emu_level_stats_reset() { memset(&ag_level_stats, 0, sizeof(ag_level_stats)); }
#endif
template <int NS, bool AV> AG_DEV void self_collisions(AgCtx<NS, AV> &c, const Cells &s, int n, float tx, float ty) {
  // (move_player has just refreshed every cell's radius cache, so s.crad[] is valid for all n cells)
  const int NP = n * (n - 1) / 2;   // pairs (a, b), a < b, numbered row-major: k = a n - a (a + 1) / 2 + b - a - 1
  auto decode = [&](int k, int &a, int &b) { a = 0; int row = n - 1, rem = k; while (row > 0 && rem >= row) { rem -= row; a++; row--; } b = a + 1 + rem; };
#ifdef AGAR_CPU_EMU
  auto pair_of = [&](int k, int &a, int &b) { decode(k, a, b); };
#else
  // this lane's pairs of the first 128 (all of them up to 16 cells), decoded once per call -- NOT once per pass: arenas with 12+ cells,
  // whose second round decoded on the fly, ran 1.5x longer than the others, and a launch lasts as long as its slowest arena
  int a_l = 0, b_l = 0, a_l2 = 0, b_l2 = 0;
  // Up to 16 cells (NP <= 120, every local level below 32) the pair bookkeeping is branch-free: the rows are walked with a wave-uniform
  // counter (no per-lane loop), a lane without a pair gets the pair (0, 0) with an empty level bit, and the touch pass below is straight-line
  // code -- the relaxation of a dense clump is one dependent chain, and every taken branch in it is an instruction-fetch stall.
  const bool few = n <= 16;
  unsigned lb1 = 0u, lb2 = 0u;   // this lane's level bits 1 << (a + b) (0: no pair)
  if (few) {
    int st1 = 0, st2 = 0;
    for (int r = 0, st = 0, row = n - 1; r < n - 1; st += row, row--, r++) {
      if (AG_LANE >= st) { a_l = r; st1 = st; }
      if (AG_LANE + 64 >= st) { a_l2 = r; st2 = st; }
    }
    b_l = a_l + 1 + (AG_LANE - st1); b_l2 = a_l2 + 1 + (AG_LANE + 64 - st2);
    if (AG_LANE < NP) lb1 = 1u << (a_l + b_l); else { a_l = 0; b_l = 0; }
    if (AG_LANE + 64 < NP) lb2 = 1u << (a_l2 + b_l2); else { a_l2 = 0; b_l2 = 0; }
  } else {
    decode(AG_LANE, a_l, b_l);
    if (NP > 64) decode(AG_LANE + 64, a_l2, b_l2);
  }
  auto pair_of = [&](int k, int &a, int &b) { if (k < 64) { a = a_l; b = b_l; } else if (k < 128) { a = a_l2; b = b_l2; } else decode(k, a, b); };
#endif
  const int LL = 2 * n - 3;                  // local levels 1 .. LL (<= 61 at 32 cells)
  unsigned long long H = 0ull;               // bit L: some pair with a + b == L touches at the current positions
  auto level_hits = [&]() {
#ifndef AGAR_CPU_EMU
    if (few) {
      unsigned bits = touches(s.x[a_l], s.y[a_l], s.crad[a_l], s.x[b_l], s.y[b_l], s.crad[b_l]) ? lb1 : 0u;
      if (NP > 64) bits |= touches(s.x[a_l2], s.y[a_l2], s.crad[a_l2], s.x[b_l2], s.y[b_l2], s.crad[b_l2]) ? lb2 : 0u;
      H = wred_or(bits);
#ifdef AGAR_PROFILE_LEVELS
      c.tacc[15]++;
#endif
      return;
    }
#endif
    const unsigned lo = wave_or(NP, [&](int k) -> unsigned { int a, b; pair_of(k, a, b); const int L = a + b; return (L < 32 && touches(s.x[a], s.y[a], s.crad[a], s.x[b], s.y[b], s.crad[b])) ? 1u << L : 0u; });
    const unsigned hi = LL < 32 ? 0u : wave_or(NP, [&](int k) -> unsigned { int a, b; pair_of(k, a, b); const int L = a + b; return (L >= 32 && touches(s.x[a], s.y[a], s.crad[a], s.x[b], s.y[b], s.crad[b])) ? 1u << (L - 32) : 0u; });
    H = ((unsigned long long)hi << 32) | lo;
#if defined(AGAR_PROFILE_LEVELS) && !defined(AGAR_CPU_EMU)
    c.tacc[15]++;
#endif
  };
  level_hits();
  ag_set_priority((int)__builtin_popcountll(H));
  if (H == 0ull) return;   // no pair touches: the reference's first sweep is a no-op
  const float dt = c.gs->g.dt, W = c.gs->g.W;
  const int D = n < LL ? n : LL;
  auto levels_1_to = [](int k) -> unsigned long long { return ((1ull << (k + 1)) - 1ull) & ~1ull; };   // bits 1 .. k (k <= 62)
  unsigned touched = 0u;   // bit s: sweep s has visited a touching pair
  // the pairs of a local level L are a = a0 .. (L - 1) / 2, b = L - a: (a0 | count << 8) per level in a uniform block (one v_readlane per
  // look-up instead of a dozen scalar instructions in every step of the walk); word 0 = "no level"
  UBlock LD; ub_fill(LD, [&](int L) -> int { if (L < 1 || L > LL) return 0; const int a0 = L > n - 1 ? L - (n - 1) : 0; return a0 | ((((L - 1) >> 1) - a0 + 1) << 8); });
  for (int sN = 0; sN <= 5; sN++) {
    const int sO = sN - 1, plen = sN < 5 ? D : LL;
    const int eo = sN >= 1 ? LL - D : 0;   // the level of this phase at which the older sweep completes (0: it has no level here)
    const unsigned long long mN = levels_1_to(plen), mO = eo >= 1 ? levels_1_to(eo) : 0ull;
    bool done = false;
    for (int LN = 0;;) {
      // levels of this phase after LN at which something touches: the newer sweep's LN, the older sweep's LN + D
      const unsigned long long M = (H & mN) | ((H >> D) & mO);
      const unsigned long long R = M >> (LN + 1);
      const int Lhit = R ? LN + 1 + (int)__builtin_ctzll(R) : 1000;
      if (eo > LN && eo < Lhit) {   // the older sweep completes before anything else touches
        if (!((touched >> sO) & 1u)) { done = true; break; }
        LN = eo; continue;
      }
      if (Lhit > plen) break;
      LN = Lhit;
      const int LO = LN + D;
      const bool validO = LN <= eo;   // <=> sN >= 1 && LO <= LL  (eo = LL - D from the second phase on, 0 in the first)
      const int dO = ub_get(LD, validO ? LO : 0), dN = ub_get(LD, LN);
      const int a0O = dO & 255, wO = dO >> 8, a0N = dN & 255, wN = dN >> 8;
      bool he = false, ho = false, hn = false, hold = false;   // hn / hold: the newer / the older sweep has visited a touching pair at this level-time
#if defined(AGAR_CPU_EMU) && defined(AGAR_STATS_LEVELS)
      // could this level-time and the next one (LN + 1 / LO + 1) have been visited together?  Yes iff the pairs of both that touch at the
      // positions BEFORE this visit share no cell; a misspeculation is a pair of the next level-time that touches only after this one moved
      bool st_mergeable = false; unsigned long long st_batch_next = 0ull;   // (bit k: pair k of the next level-time touched before)
      auto st_pairs = [&](int LNx, int *pa, int *pb) -> int {   // pairs of level-time LNx of this phase: older sweep first
        int cnt = 0; const int LOx = LNx + D; const bool vO = sN >= 1 && LOx <= LL && LNx <= eo;
        if (vO) { const int a0 = LOx - (n - 1) > 0 ? LOx - (n - 1) : 0; for (int a = a0; a <= (LOx - 1) / 2; a++) { pa[cnt] = a; pb[cnt] = LOx - a; cnt++; } }
        if (LNx <= plen) { const int a0 = LNx - (n - 1) > 0 ? LNx - (n - 1) : 0; for (int a = a0; a <= (LNx - 1) / 2; a++) { pa[cnt] = a; pb[cnt] = LNx - a; cnt++; } }
        return cnt;
      };
      int st_na[64], st_nb[64], st_nn = 0;
      static thread_local bool st_skip = false; static thread_local int st_skip_LN = -1, st_skip_sN = -1;
      const bool st_was_merged = st_skip && st_skip_LN == LN && st_skip_sN == sN; st_skip = false;
      {
        int ca[64], cb[64]; const int cn = st_pairs(LN, ca, cb);
        unsigned used = 0u; bool ok = true;
        for (int k = 0; k < cn; k++) if (touches(s.x[ca[k]], s.y[ca[k]], s.crad[ca[k]], s.x[cb[k]], s.y[cb[k]], s.crad[cb[k]])) { const unsigned m = (1u << ca[k]) | (1u << cb[k]); if (used & m) ok = false; used |= m; }
        const bool next_in_phase = (LN + 1 <= plen) || (sN >= 1 && LN + 1 <= eo);
        if (next_in_phase && LN != eo) {   // (a sweep-end check sits between the two: not merged)
          st_nn = st_pairs(LN + 1, st_na, st_nb); bool any_next = false;
          for (int k = 0; k < st_nn; k++) if (touches(s.x[st_na[k]], s.y[st_na[k]], s.crad[st_na[k]], s.x[st_nb[k]], s.y[st_nb[k]], s.crad[st_nb[k]])) { const unsigned m = (1u << st_na[k]) | (1u << st_nb[k]); if (used & m) ok = false; used |= m; st_batch_next |= 1ull << k; any_next = true; }
          st_mergeable = ok && any_next && !st_was_merged;
        }
      }
#endif
#ifdef AGAR_CPU_EMU
      // lane j: a pair of the older sweep first, then of the newer one; only x, y, r are read before the pair is known to touch
      auto visit = [&](int j, int &sw) -> bool {
        int a, b;
        if (j < wO) { a = a0O + j; b = LO - a; sw = sO; } else { a = a0N + (j - wO); b = LN - a; sw = sN; }
        CellR A, B;
        A.x = s.x[a]; A.y = s.y[a]; A.r = s.crad[a]; B.x = s.x[b]; B.y = s.y[b]; B.r = s.crad[b];
        if (!touches(A.x, A.y, A.r, B.x, B.y, B.r)) return false;
        A.vx = s.vx[a]; A.vy = s.vy[a]; A.sx = s.sx[a]; A.sy = s.sy[a]; A.m = s.m[a];
        B.vx = s.vx[b]; B.vy = s.vy[b]; B.sx = s.sx[b]; B.sy = s.sy[b]; B.m = s.m[b];
        if (sw < 5) prevent_overlap(A, B, dt, tx, ty, W); else avoid_static_overlap(A, B, W);
        cellr_store(s, a, A); cellr_store(s, b, B);
        return true;
      };
      for (int j = 0; j < wO + wN; j++) { int sw = 0; if (visit(j, sw)) { if (sw & 1) ho = true; else he = true; } }
      hn = (sN & 1) ? ho : he; hold = (sN & 1) ? he : ho;   // he / ho are by sweep parity: the newer sweep's, the older sweep's
#else
      {  // four lanes per pair (q_visit): lane = 4 * pair + 2 * cell + component
        // Straight-line: a quad without a pair takes the pair (0, 0) with its touch bit forced off, and the visit is not guarded by "does any
        // pair of this level touch" -- the level was picked from H, so one does.  (A visit of a pair that does not touch changes nothing.)
        const int lane = AG_LANE, j = lane >> 2;
        // (selects between wave-uniform values, not branches: measured 423 -> 416 us on mode 6)
        const bool has = j < wO + wN, older = j < wO;
        const int offs = older ? a0O : a0N - wO, Lsel = older ? LO : LN, sw = older ? sO : sN;
        const int a = has ? j + offs : 0, b = has ? Lsel - a : 0;
        QuadK q; q.cB = (lane & 2) != 0; q.kY = (lane & 1) != 0; q.W = W; q.dt = dt; q.tk = q.kY ? ty : tx;
        const int self = q.cB ? b : a;
        float *pp = (q.kY ? s.y : s.x) + self, *vp = (q.kY ? s.vy : s.vx) + self;
        // everything a visit reads in ONE LDS round trip
        float p = *pp, v = *vp; const float r = s.crad[self], ro = q_cell(r), sv = (q.kY ? s.sy : s.sx)[self]; const unsigned m = s.m[self], mo = q_cellu(m);
        bool t0;
        { const float dk = fabsf(p - q_cell(p)), s2 = dk * dk; const float rs = r + ro, rr = rs * rs; const bool tt = rr >= (s2 + q_comp(s2)) + 0.0f; t0 = has & tt; }
        q_visit(q, t0, sw >= 5, r, ro, m, mo, sv, p, v);
        if (t0) { *pp = p; *vp = v; }
        // (one ballot: the older sweep's quads are the first wO, so its hits are the low 4 wO bits)
        const unsigned long long hit = __ballot(t0), mO = wO >= 16 ? ~0ull : (1ull << (4 * wO)) - 1ull;
        hold = (hit & mO) != 0ull; hn = (hit & ~mO) != 0ull;
      }
#endif
      ag_lds_order();
#if defined(AGAR_CPU_EMU) && defined(AGAR_STATS_LEVELS)
      {
        const bool dense = __builtin_popcountll(H) >= 14;
        ag_level_stats.visited++; if (dense) ag_level_stats.visited_dense++;
        if (n <= 32) ag_level_stats.by_n[n][0]++;
        if (st_was_merged) { ag_level_stats.merged_away++; if (dense) ag_level_stats.merged_dense++; if (n <= 32) ag_level_stats.by_n[n][1]++; }
        if (st_mergeable) {
          bool bad = false;
          for (int k = 0; k < st_nn; k++) if (!((st_batch_next >> k) & 1ull) && touches(s.x[st_na[k]], s.y[st_na[k]], s.crad[st_na[k]], s.x[st_nb[k]], s.y[st_nb[k]], s.crad[st_nb[k]])) bad = true;
          if (bad) { ag_level_stats.misspec++; if (dense) ag_level_stats.misspec_dense++; if (n <= 32) ag_level_stats.by_n[n][2]++; }
          else { st_skip = true; st_skip_LN = LN + 1; st_skip_sN = sN; }
        }
      }
#endif
#if defined(AGAR_PROFILE_LEVELS) && !defined(AGAR_CPU_EMU)
      c.tacc[13]++; if (hn || hold) c.tacc[14]++;   // diagnostic build: levels visited / with a touching pair
#endif
      (void)he; (void)ho;
      if (hn || hold) {
        touched |= (hn ? 1u << sN : 0u) | ((hold && sN >= 1) ? 1u << sO : 0u);
        level_hits();   // cells have moved: the touch bits are taken again
      }
      // a sweep that has completed without a single touching pair ends the relaxation
      if (LN == eo && !((touched >> sO) & 1u)) { done = true; break; }
    }
    if (done) break;
    if (plen == LL && !((touched >> sN) & 1u)) break;   // (the newer sweep completes inside its own phase only up to 3 cells, and in the last phase)
  }
}
// Kinematics of ONE cell for one tick (Engine::move_player's loop body, Engine.hpp:616-626).  Shared by the
// lane-parallel general path and the uniform-register quiet path so both execute the same fp32 sequence.
AG_DEV void move_one(float &x, float &y, float &vx, float &vy, float &svx, float &svy, float hi, float r, float tx, float ty, float dt, float W) {
  float d = tx - x; vx = 3.0f * d;
  d = ty - y; vy = 3.0f * d;
  if (vmag(vx, vy) > hi) {  // clamp_speed(0, hi): set_speed re-evaluates speed() after dx changed
    float f = ag_divf(hi, vmag(vx, vy)); vx *= f;
    float g = ag_divf(hi, vmag(vx, vy)); vy *= g;
  }
  float u = vx + svx; float t = u * dt; x += t;
  u = vy + svy; t = u * dt; y += t;
  // (0,0): 0/0 = NaN fails both magnitude tests, so the reference ends with (0,0) again
  if (!(svx == 0.0f && svy == 0.0f)) v_decelerate(svx, svy, AG_SPLIT_DECEL, dt);
  boundary(W, x, y, r);
}
// one cell's kinematics out of / into its player's LDS arrays (Engine::move_player's loop body); returns "its mass is at least half the tables"
template <int NS, bool AV> AG_DEV bool move_cell(const AgCtx<NS, AV> &c, const Cells &s, int i, float tx, float ty, float dt, float W) {
  float x = s.x[i], y = s.y[i], svx = s.sx[i], svy = s.sy[i]; unsigned m = s.m[i];
  if (s.cmc[i] != m) { s.cmc[i] = m; s.crad[i] = lut(g_lut_r(c), m); s.cms[i] = lut(g_lut_ms(c), m); }  // refresh the cache on mass change
  float hi = s.cms[i], r = s.crad[i], vx, vy;
  move_one(x, y, vx, vy, svx, svy, hi, r, tx, ty, dt, W);
  s.x[i] = x; s.y[i] = y; s.vx[i] = vx; s.vy[i] = vy; s.sx[i] = svx; s.sy[i] = svy;
  return m >= (unsigned)AG_LUT_SIZE / 2u;
}
// The kinematics of ALL players' cells in one pass, a lane per live cell.  Engine::tick moves player B after player A's whole turn, but B's
// move reads only B's cells and B's target, and nothing in A's turn writes those (A's turn changes A's cells, the shared pellets, viruses and
// foods) -- EXCEPT on the ticks on which bots choose their targets (every 10th, Engine.hpp:498-499): a shy / aggressive bot looks at the
// other players' cells as they are at that moment.  So: every tick that is not a bot tick.  A five-player arena (C1) pays the ~150
// instructions of move_one once instead of five times one after the other, each for a single active lane.
template <int NS, bool AV> AG_DEV void move_all_players(AgCtx<NS, AV> &c) {
  const float dt = c.gs->g.dt, W = c.gs->g.W;
  bool big = false;
#ifdef AGAR_CPU_EMU
  for (int p = 0; p < c.P; p++) {
    const int *PLp = PLS(c, p); const int n = PLp[PL_NCELLS]; Cells s = cells_of(c, p);
    for (int i = 0; i < n; i++) big = big | move_cell(c, s, i, u2f(PLp[PL_TX]), u2f(PLp[PL_TY]), dt, W);
  }
#else
  const int ncl = AG_LANE < c.P ? PLS(c, AG_LANE)[PL_NCELLS] : 0;   // (P <= 32 players: a lane each)
  const bool one_each = __ballot(AG_LANE < c.P && ncl != 1) == 0ull;
  for (int base = 0;; base += 64) {
    const int t = base + AG_LANE; int p = -1, i = 0, st = 0;
    if (one_each) { p = t < c.P ? t : -1; st = c.P; }   // (every player has exactly one cell: the flat list IS the player list)
    else for (int q = 0; q < c.P; q++) { const int nq = __builtin_amdgcn_readlane(ncl, q); if (t >= st && t < st + nq) { p = q; i = t - st; } st += nq; }
    if (p >= 0) { const int *PLp = PLS(c, p); Cells s = cells_of(c, p); big = big | move_cell(c, s, i, u2f(PLp[PL_TX]), u2f(PLp[PL_TY]), dt, W); }
    if (base + 64 >= st) break;
  }
#endif
  ag_lds_order();
  if (AG_RARE(ag_any(big))) c.lut_risk = true;
  c.moved_all = true;
}
template <int NS, bool AV> AG_DEV void move_player(AgCtx<NS, AV> &c, const Cells &s, int n) {
  float tx = PRF(c, PL_TX), ty = PRF(c, PL_TY), dt = c.gs->g.dt, W = c.gs->g.W;
  if (!c.moved_all) {
    bool big = false;
    AG_LANES(i, n) { big = big | move_cell(c, s, i, tx, ty, dt, W); }
    ag_lds_order();
    if (AG_RARE(ag_any(big))) c.lut_risk = true;
  }
  unsigned mn = n == 1 ? ag_uniu(s.m[0]) : wave_min(n, [&](int i) { return s.m[i]; });
  PW(c, PL_MIN_MASS, (int)mn);
  AG_T(c, 12);
#ifndef AG_ABL_SELFCOL
  if (n >= 2) self_collisions(c, s, n, tx, ty);
#endif
}

// ---- created-cell buffer ---------------------------------------------------------------------------
AG_DEV void put_created(const Cells &nw, int slot, float x, float y, float vx, float vy, float sx, float sy, unsigned m, int id, unsigned dl) {
  if (slot >= AG_CC) return;  // overflow is flagged by the caller
  nw.x[slot] = x; nw.y[slot] = y; nw.vx[slot] = vx; nw.vy[slot] = vy; nw.sx[slot] = sx; nw.sy[slot] = sy; nw.m[slot] = clamp_mass(m); nw.id[slot] = id; nw.dl[slot] = dl;
}
// Engine::cell_split for LDS cell k (lane-level).  R: Engine.hpp:1067-1093.  Caller checked mass >= 50.
template <int NS, bool AV> AG_DEV void do_cell_split(const AgCtx<NS, AV> &c, const Cells &s, const Cells &nw, int k, int slot, int id, float tx, float ty, unsigned dl) {
  unsigned m = s.m[k];
  unsigned split_mass = m / 2u, remaining = m - split_mass;
  s.m[k] = clamp_mass(remaining);
  float x = s.x[k], y = s.y[k], W = c.gs->g.W;
  float ddx = tx - x, ddy = ty - y;
  float ax = fabsf(ddx), ay = fabsf(ddy); float n2 = ax * ax, n2b = ay * ay; float nrm = ag_sqrtf(n2 + n2b);
  float dirx = ag_divf(ddx, nrm), diry = ag_divf(ddy, nrm);
  float r = radius_of(c, s.m[k]);
  float ox = dirx * r, oy = diry * r;
  float lx = x + ox, ly = y + oy;
  lx = smaxf(0.0f, clampf(lx, r, W - r));
  ly = smaxf(0.0f, clampf(ly, r, W - r));
  float ss = lut(g_lut_ss(c), split_mass);
  float vx = dirx * ss, vy = diry * ss;
  put_created(nw, slot, lx, ly, vx, vy, vx, vy, split_mass, id, dl);
  s.dl[k] = dl;
}

// ---- viruses.  R: Engine.hpp:1223-1252 (grid :1207-1221), disrupt :1263-1294 ---------------------------
template <int NS, bool AV> AG_DEV bool virus_collisions(AgCtx<NS, AV> &c, const Cells &s, int n, int create_limit, bool can_eat_virus) {
  int nv = SR(c, AR_NVIR);
  if (nv == 0) return false;
  unsigned maxm = n == 1 ? ag_uniu(s.m[0]) : wave_max(n, [&](int i) { return s.m[i]; });
  if (maxm < 111u) return false;  // virus mass >= 100 and can_eat needs mass > 1.1 * virus mass
  const Virs V = virs_of(c);   // x, y, radius, mass in LDS (stage_viruses)
  int vgw = c.gs->g.vgw, vgh = c.gs->g.vgh; unsigned VC = (unsigned)c.VC;
  // One pass, a lane per virus against all cells, proves the common case -- no cell reaches a virus it can eat.  Same predicate as the scan
  // below, so a miss here is a miss there; nothing has changed in between.
  if (!wave_any(nv, [&](int vi) {
        float vx = V.x[vi], vy = V.y[vi]; unsigned vmass = V.m[vi]; float vr = V.r[vi];
        int bx = f2i(vx) / AG_VIRUS_GRID, by = f2i(vy) / AG_VIRUS_GRID; bool h = bx >= 0 && bx < vgw && by >= 0 && by < vgh, any = false;
        for (int k = 0; k < n; k++) {
          unsigned m = s.m[k]; float x = s.x[k], y = s.y[k];
          int ddx = bx - f2i(x) / AG_VIRUS_GRID, ddy = by - f2i(y) / AG_VIRUS_GRID;
          any = any | (m >= 111u && ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1 && can_eat_mass(m, vmass) && collides(x, y, cell_rad(c, s, k), vx, vy, vr));
        }
        return h && any;
      })) return false;
  for (int k = 0; k < n; k++) {
    unsigned m = ag_uniu(s.m[k]);
    if (m < 111u) continue;
    float x = ag_unif(s.x[k]), y = ag_unif(s.y[k]);
    float r = radius_of(c, m);
    int gx = f2i(x) / AG_VIRUS_GRID, gy = f2i(y) / AG_VIRUS_GRID;
    unsigned key = wave_min(nv, [&](int vi) -> unsigned {
      float vx = V.x[vi], vy = V.y[vi]; unsigned vmass = V.m[vi];
      int bx = f2i(vx) / AG_VIRUS_GRID, by = f2i(vy) / AG_VIRUS_GRID;
      int ddx = bx - gx, ddy = by - gy;
      bool ok = ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1 && bx >= 0 && bx < vgw && by >= 0 && by < vgh;
      ok = ok && can_eat_mass(m, vmass) && collides(x, y, r, vx, vy, V.r[vi]);
      return ok ? (unsigned)((ddx + 1) * 3 + (ddy + 1)) * VC + (unsigned)vi : UINT_MAX;
    });
    if (key == UINT_MAX) continue;
    int vi = (int)(key % VC);
    if (can_eat_virus) {
      unsigned vmass = ag_uniu(V.m[vi]);
      AG_SERIAL { s.m[k] = clamp_mass(m + vmass); }
    } else {
      // disrupt
      unsigned total = m;
      unsigned nm = clamp_mass((unsigned)((float)m / 2.0f));
      nm = clamp_mass(nm + (total - nm) % AG_CELL_POP_SIZE);
      unsigned pop = total - nm;
      int num_new = (int)((pop + AG_CELL_POP_SIZE - 1u) / AG_CELL_POP_SIZE);
      if (create_limit < num_new) num_new = create_limit;
      float cvx = ag_unif(s.vx[k]), cvy = ag_unif(s.vy[k]);
      float theta = v_direction(cvx, cvy);
      float sp = lut(g_lut_ms(c), AG_CELL_POP_SIZE);
      float virx = ag_unif(V.x[vi]), viry = ag_unif(V.y[vi]);
      int idc = SR(c, AR_IDC), nc0 = c.ncreated;
      unsigned dl = (unsigned)SR(c, AR_CLOCK) + (unsigned)c.gs->g.recomb_ticks;
      if (nc0 + num_new > AG_CC) flag(c, 1u);
      Cells nw = created_of(c);
      AG_LANES(j, num_new) {
        float inc = (float)(2 * 3.14159265358979323846 * j / num_new);
        float dvel = theta + inc;
        float ang = theta + dvel;
        float svx = sp * ag_cosf(ang), svy = sp * ag_sinf(ang);
        unsigned rem = pop - AG_CELL_POP_SIZE * (unsigned)j;  // each earlier new cell took min(rem, 25)
        unsigned cmass = rem < AG_CELL_POP_SIZE ? rem : AG_CELL_POP_SIZE;
        put_created(nw, nc0 + j, virx, viry, cvx, cvy, svx, svy, cmass, idc + 1 + j, dl);
      }
      AG_SERIAL { s.m[k] = nm; s.dl[k] = dl; }
      SW(c, AR_IDC, idc + num_new); c.ncreated = nc0 + num_new;
    }
    int ne = SR(c, AR_NEVV);
    if (ne < AG_EVV_CAP) { AG_SERIAL { L_I(c, L_EVV)[ne] = vi; } } else { flag(c, 8u); AG_DBG8("virus events", ne, AG_EVV_CAP); }
    SW(c, AR_NEVV, ne + 1);
    ag_lds_order();
    return true;
  }
  return false;
}

// ---- pellets.  R: Engine.hpp:976-1000 (grid :962-974) ---------------------------------------------------
// The reference grows the eater (and therefore its radius) while it scans the buckets in a fixed
// order, so whether pellet j is eaten can depend on pellets eaten before it.  Fast path: no pellet
// inside the *current* radius => nothing is eaten.  Otherwise gather the candidate set that is closed
// under the maximal possible growth, order it like the reference's scan, and replay it on lane 0.
template <int NS, bool AV> AG_DEV int pellets_eat(AgCtx<NS, AV> &c, const Cells &s, int n) {
  int np = SR(c, AR_NPEL);
  if (np == 0) return 0;
  int eaten_total = 0;
  for (int k = 0; k < n; k++) {
    unsigned m = ag_uniu(s.m[k]);
    float x = ag_unif(s.x[k]), y = ag_unif(s.y[k]);
    int gx = f2i(x) / AG_PELLET_GRID, gy = f2i(y) / AG_PELLET_GRID;
    float r0 = ag_uniu(s.cmc[k]) == m ? ag_unif(s.crad[k]) : radius_of(c, m);
    float rr0 = r0 * r0;
    // sentinel-padded registers: pellets past n_pellets sit at 3e38 and can never be inside a radius
    auto hit = [&](float qx, float qy, float rr) -> bool {
      bool ok = rr >= sqr_dist(x, y, qx, qy);
      if constexpr (!AV) { int ddx = f2i(qx) / AG_PELLET_GRID - gx, ddy = f2i(qy) / AG_PELLET_GRID - gy; ok = ok && ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1; }
      return ok;
    };
    if (!pel_any(c, [&](float qx, float qy, int) { return hit(qx, qy, rr0); })) continue;  // hot path: ~3.5 VALU per 64 pellets
    pel_launder(c);
    int c0 = pel_count(c, [&](float qx, float qy, int) { return hit(qx, qy, rr0); });
    // closure of the candidate set under growth
    int K = c0; float rrK = rr0;
    for (;;) {
      if (m + (unsigned)K >= (unsigned)AG_LUT_SIZE) { flag(c, 32u); }
      float rk = radius_of(c, m + (unsigned)K); rrK = rk * rk;
      pel_launder(c);
      int cK = pel_count(c, [&](float qx, float qy, int) { return hit(qx, qy, rrK); });
      // (the radius grows with the mass, so cK >= K and the loop ends after at most n_pellets rounds -- unless the mass has wrapped
      // around 2^32, which the reference's unsigned arithmetic allows (negative decay factor, d2u_x86): then m + K can wrap back to a
      // tiny mass and the counts would oscillate for ever.  Such an arena is already flagged; the loop must still end.)
      if (cK <= K) break;
      K = cK;
    }
    const int EC = c.gs->d.EC, KC = c.gs->d.KC, EX = c.gs->d.EX;
    if (K > KC) { flag(c, 8u); AG_DBG8("cands", K, KC); }
    // candidate records (key, squared distance to the cell) in LDS, ascending index; key = (bucket visit rank) * capacity + index.  The cell
    // does not move while it eats -- only its radius grows -- so the distance the replay compares is the one computed here (the same fp32
    // operations on the same operands as sqr_dist in the replay would be), and the index is the key's low bits: 8 bytes per candidate
    const unsigned PC = (unsigned)c.PC;   // NS * 64: a power of two
    unsigned *cand = (unsigned *)L_I(c, ag_cand_off(c));
    pel_launder(c);
    int ncand = pel_compact(c, [&](float qx, float qy, int) { return hit(qx, qy, rrK); }, [&](float qx, float qy, int i, int rank) {
      if (rank < KC) {
        int ddx = f2i(qx) / AG_PELLET_GRID - gx, ddy = f2i(qy) / AG_PELLET_GRID - gy;
        cand[2 * rank] = (unsigned)((ddx + 1) * 3 + (ddy + 1)) * PC + (unsigned)i;
        cand[2 * rank + 1] = (unsigned)f2u(sqr_dist(x, y, qx, qy));
      }
    });
    if (ncand > KC) ncand = KC;
    ag_lds_order();
    int ne0 = SR(c, AR_NEVP);
    int *T = L_I(c, L_TMP); int *evp = L_I(c, ag_evp_off(c)); auto gsp = g_evp(c) + EC;
    auto lut_r = g_lut_r(c);
    AG_SERIAL {
      for (int a = 1; a < ncand; a++) {  // insertion sort of the 2-word records by key
        unsigned k0 = cand[2 * a], k1 = cand[2 * a + 1]; int b = a - 1;
        while (b >= 0 && cand[2 * b] > k0) { cand[2 * b + 2] = cand[2 * b]; cand[2 * b + 3] = cand[2 * b + 1]; b--; }
        cand[2 * b + 2] = k0; cand[2 * b + 3] = k1;
      }
      unsigned mc = m; int ne = ne0;
      for (int a = 0; a < ncand; a++) {
        float rc = lut(lut_r, mc); float rrc = rc * rc;
        if (rrc >= u2f((int)cand[2 * a + 1])) {
          // (the reference's pellets_to_remove is unbounded and holds one entry per EAT, not per pellet: a pellet under k overlapping cells is
          // eaten k times in a tick -- the stale-index quirk -- so a mass-1000 agent split into 13 cells in an 80 x 80 arena with 1300 pellets
          // produced 16 861 events in one tick (4 players: 37 821).  Beyond the LDS list they go to the arena's spill area in HBM, where one exists)
          if (ne < EC) evp[ne] = (int)(cand[2 * a] & (PC - 1u)); else if (ne - EC < EX) gsp[ne - EC] = (int)(cand[2 * a] & (PC - 1u));
          ne++; mc = clamp_mass(mc + AG_PELLET_MASS);
        }
      }
      s.m[k] = mc; T[0] = ne;
    }
    ag_lds_order();
    int ne1 = ag_uni(T[0]);
    if (ne1 > EC + EX) { flag(c, 8u); AG_DBG8("events", ne1, EC + EX); }
    AG_DBG8MAX(ne1);
    SW(c, AR_NEVP, ne1);
    eaten_total += ne1 - ne0;
  }
  return eaten_total;
}

// ---- foods.  R: Engine.hpp:1011-1025 (eat), 1027-1054 (emit), 632-687 (move / feed virus) ------------------
template <int NS, bool AV> AG_DEV int eat_food(AgCtx<NS, AV> &c, const Cells &s, int k) {
  int nf = SR(c, AR_NFOOD);
  if (nf == 0) return 0;
  unsigned m = ag_uniu(s.m[k]);
  if (m < AG_FOOD_MASS) return 0;
  float x = ag_unif(s.x[k]), y = ag_unif(s.y[k]);
  float r = radius_of(c, m), fr = radius_of(c, AG_FOOD_MASS);
  const Foods F = foods_of(c); auto fid = g_fid(c);
  auto eaten = [&](int i) -> bool { return can_eat_mass(m, AG_FOOD_MASS) && collides(x, y, r, F.x[i], F.y[i], fr); };
  int cnt = wave_count(nf, [&](int i) { return eaten(i); });
  if (cnt == 0) return 0;
  // order-preserving erase(remove_if(...)) in place: rank <= i, chunks ascending, and within a 64-chunk
  // all lanes load before any lane stores
  int kept = wave_compact(nf, [&](int i) { return !eaten(i); }, [&](int i, int rank) {
    float a = F.x[i], b = F.y[i], e = F.vx[i], f = F.vy[i]; int id = fid[i];
    F.x[rank] = a; F.y[rank] = b; F.vx[rank] = e; F.vy[rank] = f; fid[rank] = id;
  });
  SW(c, AR_NFOOD, kept); c.food_dirty = true;
  AG_SERIAL { s.m[k] = clamp_mass(m + (unsigned)(nf - kept) * AG_FOOD_MASS); }
  ag_mem_fence();   // (the ids moved in HBM; the next cell's erase reads them through other lanes)
  return nf - kept;
}
template <int NS, bool AV> AG_DEV void maybe_emit_food(AgCtx<NS, AV> &c, const Cells &s, int n) {
  int cd = PR(c, PL_FEED_CD);
  if (cd > 0) cd -= 1;
  if (PR(c, PL_ACTION) == 1 && cd == 0) {
    int nf = SR(c, AR_NFOOD), idc = SR(c, AR_IDC);
    float tx = PRF(c, PL_TX), ty = PRF(c, PL_TY);
    int FC = c.FC;
    const Foods F = foods_of(c); auto fid = g_fid(c);
    int made = wave_compact(n, [&](int i) { return s.m[i] >= AG_CELL_MIN_SIZE + AG_FOOD_MASS; }, [&](int i, int rank) {
      float x = s.x[i], y = s.y[i];
      float ddx = tx - x, ddy = ty - y;
      float ax = fabsf(ddx), ay = fabsf(ddy); float n2 = ax * ax, n2b = ay * ay; float nrm = ag_sqrtf(n2 + n2b);
      float dirx = ag_divf(ddx, nrm), diry = ag_divf(ddy, nrm);
      float r = radius_of(c, s.m[i]);
      float ox = dirx * r, oy = diry * r;
      int f = nf + rank;
      if (f < FC) { F.x[f] = x + ox; F.y[f] = y + oy; F.vx[f] = dirx * AG_FOOD_SPEED; F.vy[f] = diry * AG_FOOD_SPEED; fid[f] = idc + 1 + rank; }
      s.m[i] = clamp_mass(s.m[i] - AG_FOOD_MASS);
    });
    int nf2 = nf + made;
    if (nf2 > FC) { flag(c, 2u); nf2 = FC; }
    SW(c, AR_NFOOD, nf2); SW(c, AR_IDC, idc + made);
    if (made) c.food_dirty = true;
    cd = 10;
    ag_mem_fence();
  }
  PW(c, PL_FEED_CD, cd);
}
template <int NS, bool AV> AG_DEV void maybe_split(AgCtx<NS, AV> &c, const Cells &s, int n, int create_limit) {  // R: Engine.hpp:1056-1064, 1095-1107
  int cd = PR(c, PL_SPLIT_CD);
  if (cd > 0) cd -= 1;
  if (PR(c, PL_ACTION) == 2 && cd == 0) {
    if (create_limit != 0) {
      int idc = SR(c, AR_IDC), nc0 = c.ncreated;
      float tx = PRF(c, PL_TX), ty = PRF(c, PL_TY);
      unsigned dl = (unsigned)SR(c, AR_CLOCK) + (unsigned)c.gs->g.recomb_ticks;
      Cells nw = created_of(c);
      int made = wave_compact(n, [&](int i) { unsigned m = s.m[i]; return m >= AG_CELL_SPLIT_MINIMUM && m >= 2u * AG_CELL_MIN_SIZE; },
                              [&](int i, int rank) { if (create_limit < 0 || rank < create_limit) do_cell_split(c, s, nw, i, nc0 + rank, idc + 1 + rank, tx, ty, dl); });
      if (create_limit > 0 && made > create_limit) made = create_limit;
      if (nc0 + made > AG_CC) flag(c, 1u);
      SW(c, AR_IDC, idc + made); c.ncreated = nc0 + made;
      ag_lds_order();
    }
    cd = 30;
  }
  PW(c, PL_SPLIT_CD, cd);
}
template <int NS, bool AV> AG_DEV void move_foods(AgCtx<NS, AV> &c) {
  int nf = SR(c, AR_NFOOD);
  if (nf == 0) return;
  float dt = c.gs->g.dt, W = c.gs->g.W; float fr = radius_of(c, AG_FOOD_MASS);
  const Foods F = foods_of(c);   // the launch's working copy (LDS)
  bool moving = wave_any(nf, [&](int i) { return !(vmag(F.vx[i], F.vy[i]) == 0); });
  if (!moving) return;
  int nv = SR(c, AR_NVIR);
  const Virs V = virs_of(c);
  auto advance = [&](float &x, float &y, float &ux, float &uy) { v_decelerate(ux, uy, AG_FOOD_DECEL, dt); float t = ux * dt; x += t; t = uy * dt; y += t; boundary(W, x, y, fr); };
  // does any moving food reach a virus after its move?  (virus radii only grow by being fed)
  // (a lane per (food, virus) pair, everything out of LDS)
  bool hits = nv > 0 && wave_any(nf * nv, [&](int k) {
    int i = k / nv, v = k - i * nv;
    float x = F.x[i], y = F.y[i], ux = F.vx[i], uy = F.vy[i], wx = V.x[v], wy = V.y[v], wr = V.r[v];
    if (vmag(ux, uy) == 0) return false;
    advance(x, y, ux, uy);
    return collides(x, y, fr, wx, wy, wr);
  });
  c.food_dirty = true;
  if (!hits) {
    AG_LANES(i, nf) {
      float x = F.x[i], y = F.y[i], ux = F.vx[i], uy = F.vy[i];
      if (!(vmag(ux, uy) == 0)) { advance(x, y, ux, uy); F.x[i] = x; F.y[i] = y; F.vx[i] = ux; F.vy[i] = uy; }
    }
    ag_lds_order();
    return;
  }
  // a food reaches a virus (rare): the viruses are fed, grown and spawned in HBM -- which also holds their velocity, hit count and id --
  // and the LDS cache is read again afterwards; the foods stay in LDS, their ids in HBM
  auto vx = g_vx(c); auto vy = g_vy(c); auto vvx = g_vvx(c); auto vvy = g_vvy(c); auto vm = g_vm(c); auto vh = g_vh(c); auto vid = g_vid(c);
  auto fid = g_fid(c);
  int idc0 = SR(c, AR_IDC), VC = c.VC; float dt10 = c.gs->g.dt10;
  int *T = L_I(c, L_TMP);
  ag_mem_fence();
  AG_SERIAL {  // exact sequential replay incl. swap-pop and virus feeding.  R: Engine.hpp:632-687
    int n = nf, nvir = nv, idc = idc0, fl = 0;
    for (int i = 0; i < n;) {
      float x = F.x[i], y = F.y[i], ux = F.vx[i], uy = F.vy[i];
      if (vmag(ux, uy) == 0) { i++; continue; }
      float pvx = ux, pvy = uy;
      advance(x, y, ux, uy);
      F.x[i] = x; F.y[i] = y; F.vx[i] = ux; F.vy[i] = uy;
      bool hit = false;
      int nscan = nvir;
      for (int v = 0; v < nscan; v++) {
        if (collides(x, y, fr, vx[v], vy[v], radius_of(c, (unsigned)vm[v]))) {
          if (vh[v] >= AG_FOOD_HITS) {
            vh[v] = 0; vm[v] = (int)AG_VIRUS_MASS;
            float nx = vx[v], ny = vy[v];
            float t = pvx * dt10; nx += t; t = pvy * dt10; ny += t;
            boundary(W, nx, ny, radius_of(c, AG_VIRUS_MASS));
            idc++;
            if (nvir < VC) { vx[nvir] = nx; vy[nvir] = ny; vvx[nvir] = pvx; vvy[nvir] = pvy; vm[nvir] = (int)AG_VIRUS_MASS; vh[nvir] = 0; vid[nvir] = idc; nvir++; }
            else fl |= 4;
          } else { vh[v] += 1; vm[v] += (int)AG_FOOD_MASS; }
          hit = true; break;
        }
      }
      if (hit) {
        if (n > 1) {
          int j = n - 1;
          float a = F.x[j], b = F.y[j], e = F.vx[j], f = F.vy[j]; int id = fid[j];
          F.x[j] = F.x[i]; F.y[j] = F.y[i]; F.vx[j] = F.vx[i]; F.vy[j] = F.vy[i]; fid[j] = fid[i];
          F.x[i] = a; F.y[i] = b; F.vx[i] = e; F.vy[i] = f; fid[i] = id;
        }
        n--;
      } else i++;
    }
    T[0] = n; T[1] = nvir; T[2] = idc; T[3] = fl;
  }
  ag_lds_order();
  SW(c, AR_NFOOD, ag_uni(T[0])); SW(c, AR_NVIR, ag_uni(T[1])); SW(c, AR_IDC, ag_uni(T[2]));
  int fl = ag_uni(T[3]); if (fl) flag(c, (unsigned)fl);
  stage_viruses(c);   // (fences first)
}

// ---- recombine / decay.  R: Engine.hpp:1160-1179, 550-584; Entities.hpp:183-203 -------------------------
template <int NS, bool AV> AG_DEV int recombine_cells(AgCtx<NS, AV> &c, const Cells &s, int n) {
  if (n < 2) return n;
  unsigned clock = (unsigned)SR(c, AR_CLOCK);
  int nrec = wave_count(n, [&](int i) { return clock >= s.dl[i]; });
  if (nrec < 2) return n;
  bool any = wave_any(n * n, [&](int k) { int a = k / n, b = k - a * n; return a < b && clock >= s.dl[a] && clock >= s.dl[b] && cells_touch(c, s, a, b); });
  if (!any) return n;  // masses (radii) only grow once a first merge has happened
  int *T = L_I(c, L_TMP);
  AG_SERIAL {
    int m = n;
    for (int i = 0; i < m; i++) {
      if (!(clock >= s.dl[i])) continue;
      for (int j = i + 1; j < m;) {
        if (clock >= s.dl[j] && touches(s.x[i], s.y[i], radius_of(c, s.m[i]), s.x[j], s.y[j], radius_of(c, s.m[j]))) {
          s.m[i] = clamp_mass(s.m[i] + s.m[j]);
          int e = m - 1;  // swap(*it2, back()); pop_back()
          s.x[j] = s.x[e]; s.y[j] = s.y[e]; s.vx[j] = s.vx[e]; s.vy[j] = s.vy[e]; s.sx[j] = s.sx[e]; s.sy[j] = s.sy[e];
          s.m[j] = s.m[e]; s.id[j] = s.id[e]; s.dl[j] = s.dl[e]; s.cmc[j] = 0u;
          m--;
        } else j++;
      }
    }
    T[0] = m;
  }
  ag_lds_order();
  return ag_uni(T[0]);
}
template <int NS, bool AV> AG_DEV void decay(AgCtx<NS, AV> &c, const Cells &s, int p, int n) {
  int elapsed = PR(c, PL_ELAPSED);
  if (!(c.gs->g.mass_decay && elapsed % 60 == 0)) return;
  int nt = PR(c, PL_NVTICKS);
  if (nt > 0) {
    int *T = L_I(c, L_TMP); auto vt = g_vt(c, p);
    AG_SERIAL {
      int fall = elapsed - AG_ANTI_TEAM_TICKS, w = 0;
      for (int i = 0; i < nt; i++) if (!(vt[i] < fall)) vt[w++] = vt[i];
      T[0] = w;
    }
    ag_mem_fence();
    int w = ag_uni(T[0]);
    PW(c, PL_NVTICKS, w);
    if (w != 0) PW(c, PL_ANTI_TEAM, f2u(((const AG_GLOBAL float *)c.gs->lut_anti)[w - 1 < AG_ANTI_LUT ? w - 1 : AG_ANTI_LUT - 1]));
  }
  if (elapsed - PR(c, PL_LAST_DECAY) >= 60) {
    double rate = (double)PRF(c, PL_ANTI_TEAM);
    bool big = false;   // (a negative factor wraps the mass around 2^32, d2u_x86: the one way a cell grows that arena_tick's mass bound does not see)
    AG_LANES(i, n) {
      double nm = (double)s.m[i] * (1 - 0.002 * rate);
      unsigned um = d2u_x86(nm);
      s.m[i] = um > AG_CELL_MIN_SIZE ? um : AG_CELL_MIN_SIZE;
      big = big | (um >= (unsigned)AG_LUT_SIZE / 2u);
    }
    if (AG_RARE(ag_any(big))) c.lut_risk = true;
    PW(c, PL_LAST_DECAY, elapsed);
    ag_lds_order();
  }
}

#include "agar_multi.inl"

// ---- one player's tick.  R: Engine.hpp:495-542 --------------------------------------------------------------
template <int NS, bool AV> AG_DEV void tick_player(AgCtx<NS, AV> &c, int p) {
  ub_load(c.PB, PLS(c, p), PL_WORDS);
  int n = PR(c, PL_NCELLS);
  if (n == 0) return;  // dead players are skipped (Engine.hpp:216)
  Cells s = cells_of(c, p);
  PW(c, PL_ELAPSED, PR(c, PL_ELAPSED) + 1);
  if (c.P > 1 && SR(c, AR_TICKS) % 10 == 0 && !c.moved_all) bot_take_action(c, p);  // Engine.hpp:498-499  (moved_all on a bot tick: bot_tick_unordered has decided)
  AG_T(c, 2);
#ifndef AG_ABLATE_MOVE
  move_player(c, s, n);
#endif
  AG_T(c, 3);
  c.ncreated = 0;
  int create_limit = AG_PLAYER_CELL_LIMIT - n;
  bool can_eat_virus = n >= AG_PLAYER_CELL_LIMIT;
#ifdef AG_ABL_VIRUS
  if (false) {
#else
  if (virus_collisions(c, s, n, create_limit, can_eat_virus)) {
#endif
    int nt = PR(c, PL_NVTICKS), el = PR(c, PL_ELAPSED);
    if (nt < AG_VT_CAP) { auto vt = g_vt(c, p); AG_SERIAL { vt[nt] = el; } PW(c, PL_NVTICKS, nt + 1); ag_mem_fence(); } else flag(c, 16u);
    PW(c, PL_VIRUSES_EATEN, PR(c, PL_VIRUSES_EATEN) + 1);
  }
  AG_T(c, 4);
#ifndef AG_ABLATE_PELLETS
  int ate = pellets_eat(c, s, n);
#else
  int ate = 0;
#endif
  AG_T(c, 5);
  unsigned total = n == 1 ? ag_uniu(s.m[0]) : (unsigned)wave_sum(n, [&](int i) { return (int)s.m[i]; });
  c.mass_sum += total;
  if (ate) PW(c, PL_FOOD_EATEN, PR(c, PL_FOOD_EATEN) + ate);
  if ((unsigned)PR(c, PL_HIGHEST_MASS) < total) PW(c, PL_HIGHEST_MASS, (int)total);
  // per-cell: auto split (mass >= 22500) then eat ejected food.  R: Engine.hpp:520-525, 592-601
  bool big = total >= AG_MAX_MASS && wave_any(n, [&](int i) { return s.m[i] >= AG_MAX_MASS; });
  bool food_work = big;
#ifndef AG_ABL_FOODTEST
  if (!big && SR(c, AR_NFOOD) > 0) {
    // one pass over the foods (a lane each, out of LDS) against all cells proves the common case -- nobody touches any food;
    // without an auto-split no mass changes between here and a cell's own turn
    const Foods F = foods_of(c); const float fr = radius_of(c, AG_FOOD_MASS);
    food_work = wave_any(SR(c, AR_NFOOD), [&](int j) {
      float x = F.x[j], y = F.y[j]; bool h = false;
      for (int i = 0; i < n; i++) { unsigned m = s.m[i]; h = h | (m >= AG_FOOD_MASS && can_eat_mass(m, AG_FOOD_MASS) && collides(s.x[i], s.y[i], cell_rad(c, s, i), x, y, fr)); }
      return h;
    });
  }
#endif
  if (food_work) {
    float tx = PRF(c, PL_TX), ty = PRF(c, PL_TY);
    int fe_total = 0;
    for (int k = 0; k < n; k++) {
      if (big && ag_uniu(s.m[k]) >= AG_MAX_MASS) {
        if (n < AG_PLAYER_CELL_LIMIT) {
          int nc0 = c.ncreated, idc = SR(c, AR_IDC) + 1;
          unsigned dl = (unsigned)SR(c, AR_CLOCK) + (unsigned)c.gs->g.recomb_ticks;
          if (nc0 >= AG_CC) flag(c, 1u);
          Cells nw = created_of(c);
          AG_SERIAL { do_cell_split(c, s, nw, k, nc0, idc, tx, ty, dl); }
          SW(c, AR_IDC, idc); c.ncreated = nc0 + 1;
        } else { AG_SERIAL { s.m[k] = clamp_mass(AG_NEW_MASS_NO_SPLIT); } }
        ag_lds_order();
      }
      fe_total += eat_food(c, s, k);
    }
    if (fe_total) PW(c, PL_FOOD_EATEN, PR(c, PL_FOOD_EATEN) + fe_total);
  }
  AG_T(c, 6);
  create_limit -= c.ncreated;
#ifndef AG_ABL_EMITSPLIT
  maybe_emit_food(c, s, n);
  maybe_split(c, s, n, create_limit);
#endif
  // add created cells.  R: core/Player.hpp:195-201
  int ncr = c.ncreated;
  if (ncr > 0) {
    if (ncr > AG_CC) ncr = AG_CC;
    int room = AG_CC - n;
    if (ncr > room) { flag(c, 1u); ncr = room; }
    Cells nw = created_of(c);
    AG_LANES(j, ncr) {
      int k = n + j;
      s.x[k] = nw.x[j]; s.y[k] = nw.y[j]; s.vx[k] = nw.vx[j]; s.vy[k] = nw.vy[j]; s.sx[k] = nw.sx[j]; s.sy[k] = nw.sy[j];
      s.m[k] = nw.m[j]; s.id[k] = nw.id[j]; s.dl[k] = nw.dl[j]; s.cmc[k] = 0u;
    }
    n += ncr;
    ag_lds_order();
  }
  AG_T(c, 7);
#ifndef AG_ABL_RECOMB
  n = recombine_cells(c, s, n);
#endif
  PW(c, PL_NCELLS, n);
  decay(c, s, p, n);
  ub_store(c.PB, PLS(c, p), PL_WORDS);
  ag_lds_order();
  AG_T(c, 8);
}

// ---- the turns of all "simple" players at once (several players per arena, non-bot ticks) -----------------------------------------------------
// A player's turn (tick_player) is a chain of small steps -- load its words, virus test, pellet scan, totals, food test, eject / split
// cooldowns, recombine, decay, store -- each a few instructions for one to three active lanes behind an LDS round trip: ~3000 cycles per
// player and tick, one player after the other (scripts/gpu_phase_multi.py: 30 single-cell players = 365 k of 589 k cycles per 4-tick launch).
// For most players on most ticks every one of those steps is a no-op: ONE cell, below the mass a virus can be eaten at (or no virus), no pellet
// within its radius, no eject / split due, no ejected food anywhere, not a decay tick.  Such a turn only counts: elapsed + 1, the cooldowns - 1,
// the min / highest mass of the cell, the arena's mass sum.  It reads and writes nothing but the player's own words and cell and nothing another
// player's turn looks at (the kinematics have been done for everybody by move_all_players: this runs on non-bot ticks only), so all such turns
// are performed here, a lane per player, whatever their place in the iteration order; dead players (no turn at all) are ticked off too.
// Returns the mask, by player slot, of the players arena_tick need not visit.  Every predicate is the one tick_player's own steps test
// (pellets_eat's fast path, virus_collisions' mass bound, maybe_emit_food / maybe_split, decay), on the same values.
template <int NS, bool AV> AG_DEV unsigned simple_turns(AgCtx<NS, AV> &c) {
  const int P = c.P;
  if (SR(c, AR_NFOOD) != 0) return 0u;   // ejected food somewhere: the per-player food test decides (Engine.hpp:520-525)
  const int nv = SR(c, AR_NVIR), np = SR(c, AR_NPEL);
  const bool decays = c.gs->g.mass_decay != 0;
  // one look at every player's words and cell, a lane per player: the candidate, dead and ejecting players as three masks
  UBlock bx, by, br;
#ifdef AGAR_CPU_EMU
  unsigned codes[AG_MAX_PLAYERS];
#else
  unsigned mycode = 0u; int myx = 0, myy = 0, myr = 0;
#endif
  AG_LANES(p, P) {
    const int *PL = PLS(c, p); const int n = PL[PL_NCELLS];
    const Cells s = cells_of(c, p);
    const unsigned m = s.m[0];
    const int el = PL[PL_ELAPSED] + 1, fcd = PL[PL_FEED_CD], scd = PL[PL_SPLIT_CD], act = PL[PL_ACTION];
    const bool feed_due = act == 1 && (fcd > 0 ? fcd - 1 : fcd) == 0;
    // (the cell's cached radius must be the radius of its mass -- move_cell leaves it so; otherwise the player takes its ordinary turn)
    const bool ok = n == 1 && s.cmc[0] == m && m < AG_MAX_MASS && (nv == 0 || m < 111u) && !(decays && el % 60 == 0) && !feed_due && !(act == 2 && (scd > 0 ? scd - 1 : scd) == 0);
    // (the pellet test's verdict "nothing inside" is remembered with the cell it was made for -- x, y, mass, bit for bit; it holds for that very
    // cell for as long as pellets only disappear: add_pellets forgets all of them.  bench/main.cpp's ExampleBots never move: their test runs once
    // per regeneration instead of every tick -- 40 k of the 186 k cycles of a Tick/30 launch, scripts/gpu_phase_multi.py)
    const bool known = PL[PL_CAND_IDX] == (int)m && PL[PL_SAFE_X] == (int)f2u(s.x[0]) && PL[PL_SAFE_Y] == (int)f2u(s.y[0]);
    const unsigned code = (ok ? 1u : 0u) | (n == 0 ? 2u : 0u) | (n > 0 && feed_due ? 4u : 0u) | (known ? 8u : 0u);
#ifdef AGAR_CPU_EMU
    bx.w[p] = f2u(s.x[0]); by.w[p] = f2u(s.y[0]); br.w[p] = f2u(s.crad[0]); codes[p] = code;
#else
    myx = f2u(s.x[0]); myy = f2u(s.y[0]); myr = f2u(s.crad[0]); mycode = code;
#endif
  }
  unsigned cand, dead, ejectors, known;
#ifdef AGAR_CPU_EMU
  cand = dead = ejectors = known = 0u;
  for (int p = 0; p < P; p++) { const unsigned code = codes[p]; cand |= (code & 1u) << p; dead |= ((code >> 1) & 1u) << p; ejectors |= ((code >> 2) & 1u) << p; known |= ((code >> 3) & 1u) << p; }
#else
  bx.v = myx; by.v = myy; br.v = myr;
  cand = (unsigned)__ballot((mycode & 1u) != 0u); dead = (unsigned)__ballot((mycode & 2u) != 0u); ejectors = (unsigned)__ballot((mycode & 4u) != 0u);   // (players are lanes 0 .. P-1 <= 31)
  known = (unsigned)__ballot((mycode & 8u) != 0u);
#endif
  // ... unless somebody ejects food on this tick (maybe_emit_food: action feed with the cooldown run out): the food appears in the middle of the
  // tick and every player BEHIND the ejector in the iteration order tests it -- such a tick is played in order, player by player
  if (ejectors) return dead;
  if (np > 0 && (cand & ~known)) {
    // pellets_eat's fast path for each candidate's one cell -- nothing inside the current radius --, for ALL candidates in one sweep: every lane
    // tests its pellet slots against candidate after candidate and keeps a bit per candidate; ONE reduction at the end instead of a ballot and
    // a branch per player
    unsigned seen = 0u;
#ifdef AGAR_CPU_EMU
    for (unsigned todo = cand & ~known; todo; todo &= todo - 1u) {
#else
    unsigned mine = 0u;
    for (unsigned todo = cand & ~known; todo; todo &= todo - 1u) {
#endif
      const int p = __builtin_ctz(todo);
      const float x = u2f(ub_get(bx, p)), y = u2f(ub_get(by, p)), r0 = u2f(ub_get(br, p)); const float rr0 = r0 * r0;
      const int gx = f2i(x) / AG_PELLET_GRID, gy = f2i(y) / AG_PELLET_GRID;
      auto hits = [&](float qx, float qy) -> bool {
        bool ok = rr0 >= sqr_dist(x, y, qx, qy);
        if constexpr (!AV) { int ddx = f2i(qx) / AG_PELLET_GRID - gx, ddy = f2i(qy) / AG_PELLET_GRID - gy; ok = ok && ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1; }
        return ok;
      };
      (void)gx; (void)gy;
#ifdef AGAR_CPU_EMU
      if (pel_any(c, [&](float qx, float qy, int) { return hits(qx, qy); })) seen |= 1u << p;
    }
#else
      bool a = false;
      AG_PEL_FOR(s_, lane_, i_) { a = a | hits(PELX(c, s_, lane_), PELY(c, s_, lane_)); }
      mine |= a ? 1u << p : 0u;
    }
    seen = wred_or(mine);
#endif
    const unsigned fresh = cand & ~known & ~seen;   // verdicts made now: remembered with their cells
    AG_LANES(p, P) { if ((fresh >> p) & 1u) { int *PL = PLS(c, p); const Cells s = cells_of(c, p); PL[PL_SAFE_X] = (int)f2u(s.x[0]); PL[PL_SAFE_Y] = (int)f2u(s.y[0]); PL[PL_CAND_IDX] = (int)s.m[0]; } }
    cand &= ~seen;
  }
  if (cand) {
    AG_LANES(p, P) {
      if ((cand >> p) & 1u) {
        int *PL = PLS(c, p); const unsigned m = cells_of(c, p).m[0];
        PL[PL_ELAPSED] += 1; PL[PL_MIN_MASS] = (int)m;
        if ((unsigned)PL[PL_HIGHEST_MASS] < m) PL[PL_HIGHEST_MASS] = (int)m;
        const int fcd = PL[PL_FEED_CD], scd = PL[PL_SPLIT_CD];
        if (fcd > 0) PL[PL_FEED_CD] = fcd - 1;
        if (scd > 0) PL[PL_SPLIT_CD] = scd - 1;
      }
    }
    c.mass_sum += (unsigned)wave_sum(P, [&](int p) { return ((cand >> p) & 1u) ? (int)cells_of(c, p).m[0] : 0; });
    ag_lds_order();
  }
  return cand | dead;
}

// ---- end-of-tick bookkeeping.  R: Engine.hpp:1002-1009, 1253-1260, 150-200 ------------------------------------
template <int NS, bool AV> AG_DEV void remove_pellets(AgCtx<NS, AV> &c) {
  int ne = SR(c, AR_NEVP);
  if (ne == 0) return;
  int n = SR(c, AR_NPEL);
  const int *evp = L_I(c, ag_evp_off(c)); auto gid = g_pid(c);
  const int EC = c.gs->d.EC; int lim = ne < EC ? ne : EC;
  auto pop = [&](int idx) {   // wave-level replay of the stale-index swap-pop
    if (n > 1 && idx < n - 1) {  // std::swap(p[idx], p.back()): the half parked at the back is popped and never read again
      int b = n - 1;
      pel_move(c, idx, b);
      AG_SERIAL { gid[idx] = gid[b]; }
      ag_mem_fence();
    }
    if (n >= 1) n--;
  };
  for (int e = 0; e < lim; e++) pop(ag_uni(evp[e]));
  if (AG_RARE(ne > EC)) {   // the tick's further events, from the arena's spill area in HBM (dense arenas only): 64 at a time, a lane each
    const int EX = c.gs->d.EX, tot = ne < EC + EX ? ne : EC + EX;
    auto gsp = g_evp(c) + EC;
    ag_mem_fence();
    for (int base = EC; base < tot; base += 64) {
      const int cnt = tot - base < 64 ? tot - base : 64;
#ifdef AGAR_CPU_EMU
      for (int j = 0; j < cnt; j++) pop(gsp[base - EC + j]);
#else
      const int mine = AG_LANE < cnt ? gsp[base - EC + AG_LANE] : 0;
      for (int j = 0; j < cnt; j++) pop(__builtin_amdgcn_readlane(mine, j));
#endif
    }
  }
  // keep the padding invariant: everything at index >= n is a sentinel again
  int n0 = SR(c, AR_NPEL);
  AG_PEL_FOR(s, lane, i) { if (i >= n && i < n0) { PELX(c, s, lane) = AG_PEL_SENTINEL; PELY(c, s, lane) = AG_PEL_SENTINEL; PEL_MARK(c, s, lane); } }
  SW(c, AR_NPEL, n);
  c.pel_dirty = true;
}
template <int NS, bool AV> AG_DEV void remove_viruses(AgCtx<NS, AV> &c) {
  int ne = SR(c, AR_NEVV);
  if (ne == 0) return;
  int n0 = SR(c, AR_NVIR);
  int *T = L_I(c, L_TMP); const int *evv = L_I(c, L_EVV);
  auto vx = g_vx(c); auto vy = g_vy(c); auto vvx = g_vvx(c); auto vvy = g_vvy(c); auto vm = g_vm(c); auto vh = g_vh(c); auto vid = g_vid(c);
  AG_SERIAL {
    int n = n0; int lim = ne < AG_EVV_CAP ? ne : AG_EVV_CAP;
    for (int e = 0; e < lim; e++) {
      int idx = evv[e];
      if (n > 1 && idx < n - 1) {
        int b = n - 1;
        float a1 = vx[idx], a2 = vy[idx], a3 = vvx[idx], a4 = vvy[idx]; int a5 = vm[idx], a6 = vh[idx], a7 = vid[idx];
        vx[idx] = vx[b]; vy[idx] = vy[b]; vvx[idx] = vvx[b]; vvy[idx] = vvy[b]; vm[idx] = vm[b]; vh[idx] = vh[b]; vid[idx] = vid[b];
        vx[b] = a1; vy[b] = a2; vvx[b] = a3; vvy[b] = a4; vm[b] = a5; vh[b] = a6; vid[b] = a7;
      }
      if (n >= 1) n--;
    }
    T[0] = n;
  }
  ag_lds_order();
  SW(c, AR_NVIR, ag_uni(T[0]));
  stage_viruses(c);   // (fences first)
}
// sort(player.cells) by id (Engine.hpp:157); ids are unique so the result is the sorted order.
template <int NS, bool AV> AG_DEV void sort_cells_by_id(AgCtx<NS, AV> &c, int p) {
  int n = ag_uni(PLS(c, p)[PL_NCELLS]);
  if (n < 2) return;
  Cells s = cells_of(c, p);
  bool unsorted = wave_any(n - 1, [&](int i) { return s.id[i] > s.id[i + 1]; });
  if (!unsorted) return;
  Cells t = created_of(c);
  // Rank sort through the created-cell buffer.  Ids are unique in a game that started with reset(); after a snapshot
  // load cells keep their saved ids while the counter restarts (Engine.hpp:304-311), so new cells can DUPLICATE an id:
  // ties keep their order, as libstdc++'s std::sort does for the <= 16 cells a player can have (pure insertion sort).
  AG_LANES(i, n) {
    int id = s.id[i], r = 0;
    for (int j = 0; j < n; j++) r += (s.id[j] < id || (s.id[j] == id && j < i)) ? 1 : 0;
    t.x[r] = s.x[i]; t.y[r] = s.y[i]; t.vx[r] = s.vx[i]; t.vy[r] = s.vy[i]; t.sx[r] = s.sx[i]; t.sy[r] = s.sy[i];
    t.m[r] = s.m[i]; t.id[r] = id; t.dl[r] = s.dl[i];
  }
  ag_lds_order();
  AG_LANES(i, n) {
    s.x[i] = t.x[i]; s.y[i] = t.y[i]; s.vx[i] = t.vx[i]; s.vy[i] = t.vy[i]; s.sx[i] = t.sx[i]; s.sy[i] = t.sy[i];
    s.m[i] = t.m[i]; s.id[i] = t.id[i]; s.dl[i] = t.dl[i]; s.cmc[i] = 0u;
  }
  ag_lds_order();
}

// ---- quiet run: the dominant case in RL rollouts ------------------------------------------------------------------
// One player, one cell, no ejected food anywhere, no reachable virus, nothing to regenerate: then Engine::tick
// reduces to the cell's kinematics, counters, the periodic decay and -- now and then -- eating ONE pellet.
// Everything is wave-uniform, so a run of consecutive quiet ticks executes on registers only: state is read once,
// each tick is move_one + integer updates, and the result is committed once.
//   * Pellet-free disc (AR_SAFE, PL_SAFE_X/Y): a pellet scan at position p0 yields the distance dmin to the nearest
//     pellet, i.e. a disc around p0 with no pellet in it.  Pellets are static, so while the cell's centre stays within
//     S = dmin - radius - 0.01 of p0 nothing can be inside its radius: no scan, and a launch that stays in the disc
//     never reads the pellets from HBM at all.  (A random walk leaves the disc far later than its path length would
//     suggest.  Only with AV: otherwise buckets change visibility as the cell moves.)
//   * Tracked pellet (PL_CAND_*): the pass also names the nearest pellet itself and the distance dsec to the SECOND nearest.  The disc
//     then reaches to the second nearest -- S = dsec - radius(mass + 1) - 0.01: inside it no OTHER pellet lies within the radius the cell
//     has now or after one eat -- and the tracked pellet is tested by itself every tick (the pass's own fp32 distance).  When it comes
//     within the radius it is the only one, also for the grown radius: a plain eat WITHOUT a pass; the disc stands (it was measured for
//     the radius after this very eat).  A third of the passes were eats, and the larger disc is left far later: far fewer passes, and a
//     pass is the one thing in a quiet step that waits for HBM.
//   * A plain eat (exactly one pellet inside the radius, still exactly one inside the grown radius, not a regen
//     tick) is performed inline: mass + 1, swap-pop removal, event record -- identical to the general path's result.
// The run stops BEFORE the first tick that needs anything else (nothing of that tick has been written) and the
// general path takes over.
// quiet_ticks() is shared by two callers that differ only in where the pellets live (the `Pel` accessor):
// k_step's quiet_run (pellets in the wave's registers) and the lean kernel k_quiet (pellets streamed from HBM/L2).
// diagnostic builds (-DAGAR_PROFILE_REASONS): why a quiet run stopped (1 eject/split possible, 2 virus in reach, 3 virus regeneration,
// 4 generator exhausted at a regeneration tick, 5 anti-team bookkeeping, 6 several pellets in reach, 7 growth reaches a second pellet)
#ifdef AGAR_PROFILE_REASONS
#define AG_WHY(q, r) ((q).why = (r))
#else
#define AG_WHY(q, r) do { } while (0)
#endif
struct QState {  // per arena: wave-uniform in k_step, uniform over the arena's lane group in k_quiet
  unsigned m, m_move;  // mass; mass at the last tick's move (Player::min_mass bookkeeping)
  int action, nv, np, ticks, elapsed, fcd, scd, last_decay, nvt, food_eaten, hm, last_ev, last_ev2, done;  // last_ev / last_ev2: pellets eaten by the last tick (-1: none)
  int mtidx, idc;      // mt19937_64 read index and entity id counter (pellet regeneration)
  int passes;          // diagnostics: pellet passes that read the arena's pellet array from memory (PL_PASSES)
#ifdef AGAR_PROFILE_REASONS
  int why;
#endif
  float x, y, svx, svy, vx, vy, r, hi, tx, ty, slack, sx0, sy0;  // slack = S, (sx0, sy0) = centre of the pellet-free disc
  float cx, cy; int cidx;  // the tracked pellet (PL_CAND_*): the one pellet the disc does not exclude; cidx < 0: none
  double rate;
  bool pel_changed;
};
// One fused pass over an arena's pellets as seen from (x, y): squared distance to the nearest one, how many lie
// within rr and within rr1 (the radius after eating one), and the index of the first one within rr.
// dsec2: squared distance to the SECOND nearest pellet; near / nx / ny: the nearest pellet itself (lowest index among equally near ones; -1: none
// visible, or somebody is in reach -- then the caller has no use for it)
struct PelScan { float dmin2, dsec2; int cnt, cnt1, first; int near; float nx, ny; };
struct PelQuery { float x, y, rr, rr1; int gx, gy; };  // gx, gy: the cell's pellet bucket (used only without AV)
template <bool AV> AG_DEV bool pel_visible(const PelQuery &k, float qx, float qy) {  // R: Engine.hpp:976-990 (3x3 bucket walk)
  if constexpr (AV) return true;
  else { int ddx = f2i(qx) / AG_PELLET_GRID - k.gx, ddy = f2i(qy) / AG_PELLET_GRID - k.gy; return ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1; }
}
// lane-level accumulation of one pellet into the partial results (dmin as float bits: orders like the value for d2 >= 0)
template <bool AV> AG_DEV void pel_accumulate(const PelQuery &k, float qx, float qy, unsigned idx, unsigned &dmin, unsigned &dsec, int &c0, int &c1, unsigned &first, unsigned &ni, float &nx, float &ny) {
  if (!pel_visible<AV>(k, qx, qy)) return;
  float d2 = sqr_dist(k.x, k.y, qx, qy);
  unsigned b = (unsigned)f2u(d2);
  if (b < dmin) { ni = idx; nx = qx; ny = qy; }   // (callers feed ascending indices: the first of equally near pellets stays)
  unsigned hi = b > dmin ? b : dmin; dsec = hi < dsec ? hi : dsec; dmin = b < dmin ? b : dmin;  // two smallest, lane-private
  if (k.rr >= d2) { c0 += 1; first = idx < first ? idx : first; }
  if (k.rr1 >= d2) c1 += 1;
}

#ifndef AGAR_CPU_EMU
// nobody in reach: the nearest pellet itself (lowest index among equally near ones; index = slot * 64 + lane, so its low six bits name the
// lane that holds it) and the wave-wide second smallest distance (the owner offers its own second, every other lane its minimum)
AG_DEV void pel_reduce_near(unsigned lane_min, unsigned dmin, unsigned &dsec, unsigned &ni, float &nx, float &ny) {
  ni = wred_min(lane_min == dmin ? ni : 0xffffffffu);
  const int owner = (int)(ni & 63u);
  dsec = wred_min(AG_LANE == owner ? dsec : lane_min);
  nx = u2f(__builtin_amdgcn_readlane(f2u(nx), owner)); ny = u2f(__builtin_amdgcn_readlane(f2u(ny), owner));
}
// wave-wide combination of the lane-private partials of one pass (all 64 lanes active)
AG_DEV void pel_reduce(float rr, unsigned &dmin, unsigned &dsec, int &c0, int &c1, unsigned &first, unsigned &ni, float &nx, float &ny) {
  const unsigned lane_min = dmin;
  dmin = wred_min(dmin);
  if (!(rr >= u2f((int)dmin))) { pel_reduce_near(lane_min, dmin, dsec, ni, nx, ny); return; }
  ni = 0xffffffffu;
  {  // (uniform branch) only when somebody is in reach
    c0 = wred_add(c0); c1 = wred_add(c1); first = wred_min(first);
    // second smallest overall: the lane that owns the minimum offers its own second, every other lane its minimum
    // (two lanes holding the same minimal value would mean two pellets at that distance: c0 >= 2, no inline eat)
    const unsigned long long owners = __ballot(lane_min == dmin);
    const bool owner = AG_LANE == (int)__builtin_ctzll(owners);
    dsec = wred_min(owner ? dsec : lane_min);
  }
}
#endif
// Second, rare pass: the first one found one or two pellets in reach and exactly two within the radius after one eat, so TWO
// pellets may be eaten this tick.  It counts the pellets within the radius after two eats (closure under the growth) and finds the
// two pellets within rr1 that the reference's scan reaches first: key = (bucket visit rank: dx outer, dy inner, Engine.hpp:980-999)
// * capacity + index (capacity is a power of two).  Kept apart from the first pass so that the common path carries none of it.
struct PelScan2 { int cnt2; unsigned key1, key2; };
struct PelQuery2 { float x, y, rr1, rr2; int gx, gy; unsigned cap; };
template <bool AV> AG_DEV void pel_accumulate2(const PelQuery2 &k, float qx, float qy, unsigned idx, int &c2, unsigned &k1, unsigned &k2) {
  const int ddx = f2i(qx) / AG_PELLET_GRID - k.gx, ddy = f2i(qy) / AG_PELLET_GRID - k.gy;
  if (!AV && !(ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1)) return;
  const float d2 = sqr_dist(k.x, k.y, qx, qy);
  if (k.rr2 >= d2) c2 += 1;
  if (k.rr1 >= d2) {
    const unsigned key = (unsigned)((ddx + 1) * 3 + (ddy + 1)) * k.cap + idx;
    const unsigned kh = key > k1 ? key : k1; k2 = kh < k2 ? kh : k2; k1 = key < k1 ? key : k1;   // two smallest, lane-private
  }
}
#ifndef AGAR_CPU_EMU
AG_DEV void pel_reduce2(int &c2, unsigned &k1, unsigned &k2) {
  c2 = wred_add(c2);
  const unsigned lane_k1 = k1;
  k1 = wred_min(k1);
  k2 = wred_min(lane_k1 == k1 ? k2 : lane_k1);   // (keys are unique: one lane owns the minimum and offers its second)
}
#endif
// The loop alternates two phases so that the pellet pass sits at a point every lane of the wave reaches together
// (`pel.any` and `pel.scan` are wave-level calls; everything else is per-arena code):
//   A  each arena runs plain quiet ticks on its own until it is finished, has to stop, or has moved out of its
//      pellet-free disc (then the moved-but-uncommitted tick waits for a pass);
//   B  one wave-level pass serves every arena that waits, which then resolves its pending tick (eat / new disc).
// k_quiet advances several arenas per wavefront and lets the whole wave scan for whichever of them needs it.
template <bool AV, class PelT, class LutT, class MtT> AG_DEV void quiet_ticks(QState &q, const AgParams &g, LutT lut_r, LutT lut_ms, MtT mt, PelT &pel, int max_ticks, bool active = true) {
  const float dt = g.dt, W = g.W;
  const bool regen = g.regen != 0, decay = g.mass_decay != 0;
  const int tgt_p = g.target_pellets, tgt_v = g.target_viruses;
  float rr = q.r * q.r;
  if (!AV) q.slack = 0.0f;
  float s2 = q.slack * q.slack;
  q.done = 0; q.last_ev = -1; q.last_ev2 = -1; q.m_move = q.m; q.pel_changed = false;
  const float pel_r = g.pel_r; const float pel_span = W - 2.0f * pel_r;  // random_location(radius), Engine.hpp:143-148
  // countdowns instead of two integer modulos per tick: ticks until the next regeneration tick / decay check
  int to_regen = regen ? (120 - q.ticks % 120) % 120 : -1, to_decay = decay ? 59 - q.elapsed % 60 : -1;
  // the pending tick of phase B: where the cell moved to, what it would weigh after one pellet
  float nx = q.x, ny = q.y, nvx = 0.0f, nvy = 0.0f, nsx = q.svx, nsy = q.svy;
  unsigned nm = q.m; float rr1_pending = 0.0f, r1_pending = 0.0f;
  bool need = false; float dsec_pending = 0.0f;

  // everything of a tick after the pellets have been dealt with (ev: index of the eaten pellet or -1; ev2: a second one, eaten after it)
  // (tracked: ev is the tracked pellet, eaten without a pass -- the disc stands)
  auto finish_tick = [&](int ev, int ev2, bool rescanned, float nslack, bool tracked = false) {
    const bool regen_tick = to_regen == 0, decay_tick = to_decay == 0;
    q.m_move = q.m;
    q.x = nx; q.y = ny; q.vx = nvx; q.vy = nvy; q.svx = nsx; q.svy = nsy;
    if (rescanned) { q.slack = nslack; q.sx0 = nx; q.sy0 = ny; s2 = nslack * nslack; }
    q.elapsed += 1; q.done += 1; q.last_ev = ev; q.last_ev2 = ev2;
    if (AG_RARE(ev >= 0)) {  // Engine.hpp:991-994 (eat), :1002-1009 (swap-pop, one list entry after the other, stale indices)
      q.m = nm; q.food_eaten += 1;
      pel.swap_pop(ev, q.np); q.np -= 1;
      if (AG_RARE(ev2 >= 0)) { q.m = clamp_mass(nm + AG_PELLET_MASS); q.food_eaten += 1; pel.swap_pop(ev2, q.np); q.np -= 1; }
      q.r = lut(lut_r, q.m); q.hi = lut(lut_ms, q.m); rr = q.r * q.r;
      q.pel_changed = true;
      // one pellet eaten: it was the nearest one and the pass also saw the second nearest, which bounds the new pellet-free disc
      // around this very position -- no second pass on the next tick.  Two eaten: no bound is known, the next tick looks.
      q.cidx = -1;
      if (!tracked) {
        float sl = ag_sqrtf(dsec_pending) - q.r; sl = sl - 0.01f; sl = (ev2 < 0 && sl > 0.0f) ? sl : 0.0f;
        q.slack = sl; s2 = sl * sl;
      }
    }
    if ((unsigned)q.hm < q.m) q.hm = (int)q.m;
    if (q.fcd > 0) q.fcd -= 1; if (q.action == 1 && q.fcd == 0) q.fcd = 10;   // Engine.hpp:1046-1054 (nothing can be ejected: mass < 35)
    if (q.scd > 0) q.scd -= 1; if (q.action == 2 && q.scd == 0) q.scd = 30;   // Engine.hpp:1056-1064 (nothing can split: mass < 50)
    if (AG_RARE(decay_tick && q.elapsed - q.last_decay >= 60)) {             // Engine.hpp:575-584, Entities.hpp:199-203
      double dm = (double)q.m * (1 - 0.002 * q.rate); unsigned um = d2u_x86(dm);
      um = um > AG_CELL_MIN_SIZE ? um : AG_CELL_MIN_SIZE;
      q.last_decay = q.elapsed;
      if (um != q.m) { q.m = um; q.r = lut(lut_r, q.m); q.hi = lut(lut_ms, q.m); rr = q.r * q.r; }  // a smaller radius keeps the disc valid
    }
    q.ticks += 1;
    to_regen = to_regen == 0 ? 119 : to_regen - 1; to_decay = to_decay == 0 ? 59 : to_decay - 1;
    if (AG_RARE(regen_tick && tgt_p - q.np > 0)) {  // add_pellets(target - n): random_location(r) per pellet, Engine.hpp:418-424, 236-239
      const int n_new = tgt_p - q.np;
      // (a disc with a tracked pellet keeps every OTHER pellet out of the radius after one eat)
      const float reach = q.cidx >= 0 ? lut(lut_r, clamp_mass(q.m + AG_PELLET_MASS)) : q.r;
      for (int j = 0; j < n_new; j++) {
        float px = mt_to_float(mt_temper(mt[q.mtidx]), pel_span) + pel_r;
        float py = mt_to_float(mt_temper(mt[q.mtidx + 1]), pel_span) + pel_r;
        q.mtidx += 2; q.idc += 1;
        pel.append(q.np, px, py, q.idc);
        q.np += 1;
        // the pellet-free disc shrinks to exclude the new pellet (same margin as a pass leaves: sqrt(d2) - radius - 0.01); it is
        // NOT given up -- the new pellet falls into it about once in 10^4 times, and a lost disc costs a whole pass
        float ox = px - q.sx0, oy = py - q.sy0; float d2 = ox * ox, d2b = oy * oy; d2 = d2 + d2b;
        float sl = ag_sqrtf(d2) - reach; sl = sl - 0.01f; sl = sl > 0.0f ? sl : 0.0f;
        if (sl < q.slack) { q.slack = sl; s2 = sl * sl; }
      }
      q.pel_changed = true;
    }
  };

  for (;;) {
    // ---- phase A: one quiet tick per arena (all arenas of the wave stay on the same tick, so a pass never has to
    // wait for the others to finish their whole run) ----
    if (active) do {
      if (q.done >= max_ticks) { active = false; break; }
      if (q.m >= AG_CELL_MIN_SIZE + AG_FOOD_MASS && q.action != 0) { active = false; AG_WHY(q, 1); break; }  // eject needs >= 35, split >= 50
      if (q.m >= 111u && q.nv != 0) { active = false; AG_WHY(q, 2); break; }                                  // virus contact needs >= 111
      if (AG_RARE(q.m >= (unsigned)AG_LUT_SIZE)) { active = false; AG_WHY(q, 2); break; }                      // beyond the tables: the general path flags it
      // regeneration (Engine.hpp:236-239): viruses need the general path; pellets are topped up inline as long as the
      // generator's buffered outputs suffice (2 draws per pellet, one more pellet may be eaten this very tick)
      if (to_regen == 0 && (tgt_v - q.nv > 0 || q.mtidx + 2 * (tgt_p - q.np + 2) > 312)) { active = false; AG_WHY(q, tgt_v - q.nv > 0 ? 3 : 4); break; }   // (up to two pellets may be eaten this very tick)
      if (to_decay == 0 && q.nvt != 0) { active = false; AG_WHY(q, 5); break; }                               // anti-team bookkeeping
      nx = q.x; ny = q.y; nsx = q.svx; nsy = q.svy;
      move_one(nx, ny, nvx, nvy, nsx, nsy, q.hi, q.r, q.tx, q.ty, dt, W);
      if (q.np != 0) {
        float ox = nx - q.sx0, oy = ny - q.sy0; float o2 = ox * ox, o2b = oy * oy; o2 = o2 + o2b;
        if (AG_RARE(!(AV && o2 < s2))) {  // left the pellet-free disc (or none known): look at the pellets
          need = true;
          nm = clamp_mass(q.m + AG_PELLET_MASS);
          float r1 = lut(lut_r, nm); rr1_pending = r1 * r1; r1_pending = r1;
          break;
        }
        // inside the disc only the tracked pellet can be within the radius, now or after one eat
        if (q.cidx >= 0 && AG_RARE(rr >= sqr_dist(nx, ny, q.cx, q.cy))) {
          nm = clamp_mass(q.m + AG_PELLET_MASS);
          if (nm >= AG_CELL_MIN_SIZE + AG_FOOD_MASS && q.action != 0) { active = false; AG_WHY(q, 1); break; }   // (as after a pass: an eject / split may follow the eat)
          finish_tick(q.cidx, -1, false, 0.0f, true);
          break;
        }
      }
      finish_tick(-1, -1, false, 0.0f);
    } while (0);
    // ---- phase B: one pass for everybody who waits ----
    if (!pel.any(need || active)) break;
    // (the query is assembled here from the pending tick, so that nothing but the pending tick itself stays live across
    // the phases; the bucket indices matter only without AV)
    PelQuery k{nx, ny, rr, rr1_pending, AV ? 0 : f2i(nx) / AG_PELLET_GRID, AV ? 0 : f2i(ny) / AG_PELLET_GRID};
    PelScan sc = pel.template scan<AV>(need, k);
    // rare: exactly two pellets within the radius after one eat, at least one of them in reach now -- two may be eaten this tick
    const bool two = need && rr >= sc.dmin2 && sc.cnt1 == 2;
    PelScan2 sc2{0, 0xffffffffu, 0xffffffffu};
    if (AG_RARE(pel.any(two))) {
      const float r2 = lut(lut_r, clamp_mass(nm + AG_PELLET_MASS));
      PelQuery2 k2{nx, ny, rr1_pending, r2 * r2, f2i(nx) / AG_PELLET_GRID, f2i(ny) / AG_PELLET_GRID, pel.cap()};
      sc2 = pel.template scan2<AV>(two, k2);
    }
    if (need) {
      need = false; q.passes += pel.pass_cost();
      int ev = -1, ev2 = -1; float nslack = 0.0f;
      if (rr >= sc.dmin2) {  // somebody is inside the radius: one or two plain eats, or the general path's business
        const bool act1 = nm >= AG_CELL_MIN_SIZE + AG_FOOD_MASS && q.action != 0, act2 = nm + AG_PELLET_MASS >= AG_CELL_MIN_SIZE + AG_FOOD_MASS && q.action != 0;
        if (sc.cnt == 1 && sc.cnt1 == 1 && !act1) { ev = sc.first; dsec_pending = sc.dsec2; }
        else if (two && sc2.cnt2 == 2 && !act2) {
          // The reference eats while it scans and the radius grows with every eat (Engine.hpp:991-994): a pellet already inside now is
          // eaten whenever it is scanned; one that gets inside only through the first eat is eaten iff it is scanned after it.  No third
          // pellet lies within the radius after two eats, so nothing else can happen.
          const int iA = (int)(sc2.key1 & (pel.cap() - 1u)), iB = (int)(sc2.key2 & (pel.cap() - 1u));
          if (sc.cnt == 2 || iA == sc.first) { ev = iA; ev2 = iB; }   // both in reach, or the one in reach is scanned first
          else { ev = sc.first; dsec_pending = sc.dsec2; }              // the other one was passed before the radius grew: it waits for the next tick
        }
        else { active = false; AG_WHY(q, sc.cnt != 1 ? 6 : sc.cnt1 != 1 ? 7 : 1); }  // (nothing of this tick is committed)
      } else {
        // nobody in reach: the disc up to the second nearest pellet with the nearest one tracked, or -- should that leave no room -- the
        // plain disc up to the nearest
        float sl2 = ag_sqrtf(sc.dsec2) - r1_pending; sl2 = sl2 - 0.01f;
#ifdef AG_NO_TRACK   // (measurement switch: the plain disc only)
        sl2 = 0.0f;
#endif
        if (AV && sc.near >= 0 && sl2 > 0.0f) { nslack = sl2; q.cidx = sc.near; q.cx = sc.nx; q.cy = sc.ny; }
        else { float sl = ag_sqrtf(sc.dmin2) - q.r; sl = sl - 0.01f; nslack = sl > 0.0f ? sl : 0.0f; q.cidx = -1; }
      }
      if (active) finish_tick(ev, ev2, true, nslack);
    }
  }
}

// pellets in the wave's registers (k_step): one arena per wave, so every argument is wave-uniform
template <int NS, bool AV> struct RegPel {
  AgCtx<NS, AV> &c;
  AG_MEM bool any(bool p) const { return p; }
  AG_MEM int pass_cost() const { return 0; }  // register-resident: the load is counted once per launch in arena_store
  AG_MEM unsigned cap() const { return (unsigned)(NS * 64); }
  template <bool AV2> AG_MEM PelScan2 scan2(bool need, const PelQuery2 &k) {
    PelScan2 out{0, 0xffffffffu, 0xffffffffu};
    if (!need) return out;
    ensure_pellets(c); pel_launder(c);
    AG_PEL_FOR(s, lane, i) { pel_accumulate2<AV>(k, PELX(c, s, lane), PELY(c, s, lane), (unsigned)i, out.cnt2, out.key1, out.key2); }
#ifndef AGAR_CPU_EMU
    pel_reduce2(out.cnt2, out.key1, out.key2);
#endif
    return out;
  }
  template <bool AV2> AG_MEM PelScan scan(bool need, const PelQuery &k) {
    PelScan out{3.0e38f, 3.0e38f, 0, 0, -1, -1, 0.0f, 0.0f};
    if (!need) return out;
    ensure_pellets(c); pel_launder(c);
    unsigned dmin = 0x7f800000u, dsec = 0x7f800000u, first = 0xffffffffu, ni = 0xffffffffu; int c0 = 0, c1 = 0; float nx = 0.0f, ny = 0.0f;
    AG_PEL_FOR(s, lane, i) { pel_accumulate<AV>(k, PELX(c, s, lane), PELY(c, s, lane), (unsigned)i, dmin, dsec, c0, c1, first, ni, nx, ny); }
#ifndef AGAR_CPU_EMU
    pel_reduce(k.rr, dmin, dsec, c0, c1, first, ni, nx, ny);
#endif
    out.dmin2 = u2f((int)dmin); out.dsec2 = u2f((int)dsec); out.cnt = c0; out.cnt1 = c1; out.first = (int)first;
    out.near = (int)ni; out.nx = nx; out.ny = ny;
    return out;
  }
  AG_MEM void append(int idx, float x, float y, int id) {  // pellets.emplace_back
    ensure_pellets(c);
    AG_PEL_FOR(s, lane, i) { if (i == idx) { PELX(c, s, lane) = x; PELY(c, s, lane) = y; PEL_MARK(c, s, lane); } }
    auto gid = g_pid(c); AG_SERIAL { gid[idx] = id; }
    c.pel_dirty = true;
  }
  AG_MEM void swap_pop(int ev, int np) {
    ensure_pellets(c);   // the tracked pellet is eaten without a pass: on a launch that has not read the pellets yet the registers hold nothing to swap
    if (np > 1 && ev < np - 1) { pel_move(c, ev, np - 1); auto gid = g_pid(c); AG_SERIAL { gid[ev] = gid[np - 1]; } }
    int lastp = np - 1;
    AG_PEL_FOR(sl_, lane, i) { if (i == lastp) { PELX(c, sl_, lane) = AG_PEL_SENTINEL; PELY(c, sl_, lane) = AG_PEL_SENTINEL; PEL_MARK(c, sl_, lane); } }
    c.pel_dirty = true;
  }
};

// Returns the number of ticks performed (0 .. max_ticks).
template <int NS, bool AV> AG_DEV int quiet_run(AgCtx<NS, AV> &c, int max_ticks) {
  if (c.P != 1 || SR(c, AR_NFOOD) != 0) return 0;
  int *P = PLS(c, 0);
  ub_load(c.PB, P, PL_WORDS);
  if (PR(c, PL_NCELLS) != 1) return 0;
  Cells s = cells_of(c, 0);
  QState q;
  q.m = ag_uniu(s.m[0]);
  if (ag_uniu(s.cmc[0]) != q.m) return 0;  // radius / speed cache must be valid
  q.action = PR(c, PL_ACTION); q.nv = SR(c, AR_NVIR); q.np = SR(c, AR_NPEL);
  q.x = ag_unif(s.x[0]); q.y = ag_unif(s.y[0]); q.svx = ag_unif(s.sx[0]); q.svy = ag_unif(s.sy[0]);
  q.vx = ag_unif(s.vx[0]); q.vy = ag_unif(s.vy[0]); q.r = ag_unif(s.crad[0]); q.hi = ag_unif(s.cms[0]);
  q.tx = PRF(c, PL_TX); q.ty = PRF(c, PL_TY);
  q.ticks = SR(c, AR_TICKS); q.elapsed = PR(c, PL_ELAPSED); q.fcd = PR(c, PL_FEED_CD); q.scd = PR(c, PL_SPLIT_CD);
  q.last_decay = PR(c, PL_LAST_DECAY); q.nvt = PR(c, PL_NVTICKS); q.food_eaten = PR(c, PL_FOOD_EATEN); q.hm = PR(c, PL_HIGHEST_MASS);
  q.rate = (double)PRF(c, PL_ANTI_TEAM); q.slack = u2f(SR(c, AR_SAFE)); q.sx0 = PRF(c, PL_SAFE_X); q.sy0 = PRF(c, PL_SAFE_Y);
  q.mtidx = SR(c, AR_MTIDX); q.idc = SR(c, AR_IDC); q.passes = 0;
  q.cx = PRF(c, PL_CAND_X); q.cy = PRF(c, PL_CAND_Y); q.cidx = PR(c, PL_CAND_IDX);
  RegPel<NS, AV> pel{c};
  quiet_ticks<AV>(q, c.gs->g, g_lut_r(c), g_lut_ms(c), (const AG_GLOBAL uint64_t *)g_mt(c), pel, max_ticks);
  if (q.done == 0) return 0;
  int *evp = L_I(c, ag_evp_off(c));
  AG_SERIAL {
    s.x[0] = q.x; s.y[0] = q.y; s.vx[0] = q.vx; s.vy[0] = q.vy; s.sx[0] = q.svx; s.sy[0] = q.svy;
    s.m[0] = q.m; s.cmc[0] = q.m; s.crad[0] = q.r; s.cms[0] = q.hi;
    P[PL_ELAPSED] = q.elapsed; P[PL_MIN_MASS] = (int)q.m_move; P[PL_HIGHEST_MASS] = q.hm; P[PL_FEED_CD] = q.fcd; P[PL_SPLIT_CD] = q.scd;
    P[PL_FOOD_EATEN] = q.food_eaten; P[PL_LAST_DECAY] = q.last_decay; P[PL_SAFE_X] = f2u(q.sx0); P[PL_SAFE_Y] = f2u(q.sy0);
    P[PL_CAND_X] = f2u(q.cx); P[PL_CAND_Y] = f2u(q.cy); P[PL_CAND_IDX] = q.cidx;
    if (q.last_ev >= 0) evp[0] = q.last_ev;
    if (q.last_ev2 >= 0) evp[1] = q.last_ev2;
  }
  SW(c, AR_NEVP, (q.last_ev >= 0 ? 1 : 0) + (q.last_ev2 >= 0 ? 1 : 0)); SW(c, AR_NEVV, 0); SW(c, AR_NPEL, q.np);
  SW(c, AR_TICKS, q.ticks); SW(c, AR_CLOCK, SR(c, AR_CLOCK) + q.done); SW(c, AR_SAFE, f2u(q.slack));
  SW(c, AR_MTIDX, q.mtidx); SW(c, AR_IDC, q.idc);
  ag_lds_order();
  if (q.pel_changed) ag_mem_fence();
  return q.done;
}

// A bot tick (every 10th: Engine.hpp:498-499) without a bot that LOOKS at anybody: the four scripted kinds read the other players' cells as they
// are when their turn comes, which is what keeps such a tick in the reference's player order; an agent decides nothing and an ExampleBot
// (agario/bots/ExampleBot.hpp:45-51: no action, target = its own centroid) reads only itself.  With no live player of the four kinds the tick is an
// ordinary one -- kinematics for everybody, simple turns at once -- after the ExampleBots' targets have been set here, a lane per player
// (r05: bench/main.cpp's Tick/N populations and the multi-agent arenas paid a whole ordered round of turns on every bot tick: 44 % of a Tick/30
// launch for a tenth of its ticks).
template <int NS, bool AV> AG_DEV bool bot_tick_unordered(AgCtx<NS, AV> &c) {
  const int P = c.P;
  unsigned deciders = 0u, examples = 0u;
#ifdef AGAR_CPU_EMU
  for (int p = 0; p < P; p++) { const int *PL = PLS(c, p); const int kind = PL[PL_KIND]; const bool alive = PL[PL_NCELLS] > 0;
    if (alive && kind >= AG_KIND_HUNGRY && kind <= AG_KIND_AGGRESSIVE_SHY) deciders |= 1u << p;
    if (alive && kind == AG_KIND_EXAMPLE) examples |= 1u << p; }
#else
  { const int lane = AG_LANE; const int kind = lane < P ? PLS(c, lane)[PL_KIND] : 0; const bool alive = lane < P && PLS(c, lane)[PL_NCELLS] > 0;
    deciders = (unsigned)__ballot(alive && kind >= AG_KIND_HUNGRY && kind <= AG_KIND_AGGRESSIVE_SHY); examples = (unsigned)__ballot(alive && kind == AG_KIND_EXAMPLE); }
#endif
  if (deciders) return false;
  if (examples) {
    AG_LANES(p, P) {
      if ((examples >> p) & 1u) {   // Player::x / y (core/Player.hpp:102-126): sequential fp32 sums in cell order, as player_centroid
        int *PL = PLS(c, p); const int n = PL[PL_NCELLS]; const Cells s = cells_of(c, p);
        float sx = 0.0f, sy = 0.0f; unsigned tm = 0u;
        for (int i = 0; i < n; i++) { const unsigned m = s.m[i]; const float fm = (float)m; float t = s.x[i] * fm; sx += t; t = s.y[i] * fm; sy += t; tm += m; }
        PL[PL_ACTION] = 0; PL[PL_TX] = (int)f2u(ag_divf(sx, (float)tm)); PL[PL_TY] = (int)f2u(ag_divf(sy, (float)tm));
      }
    }
    ag_lds_order();
  }
  return true;
}

// ---- Engine::tick.  R: Engine.hpp:208-240 --------------------------------------------------------------------
template <int NS, bool AV> AG_DEV void arena_tick(AgCtx<NS, AV> &c) {
  ensure_pellets(c);
  SW(c, AR_SAFE, 0);  // the out-of-reach budget is only maintained by quiet_run
  if (c.P == 1) { AG_SERIAL { PLS(c, 0)[PL_CAND_IDX] = -1; } ag_lds_order(); }   // ... and the tracked pellet goes with the disc (no stale index may outlive it)
  SW(c, AR_NEVP, 0); SW(c, AR_NEVV, 0);
  c.lut_risk = false; c.mass_sum = 0u; c.moved_all = false;
#ifndef AG_NO_MOVE_ALL
  if (c.P > 1 && (SR(c, AR_TICKS) % 10 != 0 || bot_tick_unordered(c))) move_all_players(c);
#endif
  AG_T(c, 1);
  // (several players, kinematics done: the turns that are pure bookkeeping are performed for all such players at once -- simple_turns)
#ifdef AG_NO_SIMPLE_TURNS   // (measurement / bisection builds)
  const unsigned done_ = 0u;
#else
  const unsigned done_ = c.moved_all ? simple_turns(c) : 0u;
#endif
  // (nobody left -- bench/main.cpp's populations on nearly every tick --: the walk over the iteration order is skipped, not made P times in vain)
  if (c.P > 32 || (~done_ & (c.P >= 32 ? ~0u : (1u << c.P) - 1u)) != 0u)
    for (int k = 0; k < c.P; k++) { const int p_ = SR(c, AR_ORDER0 + k); if (!((done_ >> p_) & 1u)) tick_player(c, p_); }
  remove_pellets(c);
  remove_viruses(c);
  AG_T(c, 13);
#ifndef AG_ABL_SORT
  if (c.P > 1) {   // (a player with fewer than two cells has nothing to sort: one parallel look at the counts instead of a dependent LDS read per player)
    for (unsigned todo = wave_or(c.P, [&](int p) -> unsigned { return PLS(c, p)[PL_NCELLS] >= 2 ? 1u << p : 0u; }); todo; todo &= todo - 1u) sort_cells_by_id(c, __builtin_ctz(todo));
  } else
  for (int k = 0; k < c.P; k++) sort_cells_by_id(c, SR(c, AR_ORDER0 + k));
#endif
  AG_T(c, 14);
  // PrecisionCollisionDetection::solve: with one player every strip scan breaks on an own cell
  // (utils/collision_detection.hpp:51) => no eats.
  players_collision(c);
#ifndef AG_ABL_FOODMOVE
  move_foods(c);
#endif
  AG_T(c, 15);
  int ticks = SR(c, AR_TICKS);
  if (c.gs->g.regen && ticks % 120 == 0) {
    add_pellets(c, c.gs->g.target_pellets - SR(c, AR_NPEL));
    add_viruses(c, c.gs->g.target_viruses - SR(c, AR_NVIR));
  }
  SW(c, AR_TICKS, ticks + 1); SW(c, AR_CLOCK, SR(c, AR_CLOCK) + 1);
  // a mass beyond the radius / speed tables (2^19 entries; the reference computes them from the mass, without a limit): every
  // table look-up clamps, so the arena has left the reference -- say so (AGARCL_F_MASS_LUT_OVERFLOW) instead of diverging silently
  // (Two round trips to LDS per player, every tick, for a flag that is never raised: so the per-cell test runs only when a cell COULD have
  // reached the table size -- one had half of it when it moved (lut_risk), or all the mass of the arena plus everything a tick can add
  // (ejected foods eaten, a respawn per player) would.  Otherwise no cell can, and the loop would find nothing.)
  if (AG_RARE(c.lut_risk || c.mass_sum + 10u * (unsigned)c.FC + 4096u * (unsigned)c.P + 65536u >= (unsigned)AG_LUT_SIZE)) {
    for (int p = 0; p < c.P; p++) {
      const int n = ag_uni(PLS(c, p)[PL_NCELLS]);
      if (n == 0) continue;
      Cells s = cells_of(c, p);
      if (AG_RARE(wave_any(n, [&](int i) { return s.m[i] >= (unsigned)AG_LUT_SIZE; }))) flag(c, 32u);
    }
  }
  AG_T(c, 9);
}

// ---- BaseEnvironment.  R: environment/envs/BaseEnvironment.hpp:89-204 -----------------------------------------
template <int NS, bool AV> AG_DEV unsigned player_mass(const AgCtx<NS, AV> &c, int p) { int n = ag_uni(PLS(c, p)[PL_NCELLS]); Cells s = cells_of(c, p); unsigned t = 0; for (int i = 0; i < n; i++) t += ag_uniu(s.m[i]); return t; }
template <int NS, bool AV> AG_DEV void take_action(AgCtx<NS, AV> &c, int p, float dx, float dy, int action) {  // R: :162-176, Player.hpp:102-126
  int n = ag_uni(PLS(c, p)[PL_NCELLS]);
  if (n == 0) return;
  Cells s = cells_of(c, p); float sx = 0.0f, sy = 0.0f; unsigned tm = 0;
  for (int i = 0; i < n; i++) { unsigned m = ag_uniu(s.m[i]); float fm = (float)m; float t = ag_unif(s.x[i]) * fm; sx += t; t = ag_unif(s.y[i]) * fm; sy += t; tm += m; }
  float px = ag_divf(sx, (float)tm), py = ag_divf(sy, (float)tm);
  float ox = dx * 10.0f, oy = dy * 10.0f;
  int *P = PLS(c, p);
  AG_SERIAL { P[PL_ACTION] = action; P[PL_TX] = f2u(px + ox); P[PL_TY] = f2u(py + oy); }
  ag_lds_order();
}
template <int NS, bool AV> AG_DEV void respawn_dead(AgCtx<NS, AV> &c) { for (int k = 0; k < c.P; k++) { int p = SR(c, AR_ORDER0 + k); if (ag_uni(PLS(c, p)[PL_NCELLS]) == 0) respawn(c, p); } }

// k-th RL-controlled (non-bot) player in the engine's iteration order -- the order of BaseEnvironment::masses()
// (BaseEnvironment.hpp:125-138), i.e. of the rewards list; -1 if there are fewer.
template <int NS, bool AV> AG_DEV int agent_in_order(const AgCtx<NS, AV> &c, int k) {
  for (int j = 0; j < c.P; j++) { int p = SR(c, AR_ORDER0 + j); if (ag_uni(PLS(c, p)[PL_KIND]) != 0) continue; if (k == 0) return p; k--; }
  return -1;
}
// BaseEnvironment::step's result for agent i (BaseEnvironment.hpp:116-121): reward = mass or mass delta (+ c_death
// when respawned), done flag on agent 0; written as f64 / u8 / i32 and as the packed (reward, done) f32 pair.
// Lane-level: the caller selects the ONE lane that writes.
AG_DEV void emit_agent_result(const AgState *gs, int slot, int arena, int na, int i, unsigned m, unsigned before, int respawned, int done) {
  double r = (double)m;
  if (gs->g.reward_type) { float b = (float)before; float sub = b - (float)(respawned ? gs->g.c_death : 0); r -= (double)sub; }
  size_t o = (size_t)arena * na + i;
  auto rw = (AG_GLOBAL double *)gs->rewards; auto ms = (AG_GLOBAL int32_t *)gs->masses; auto dn = (AG_GLOBAL uint8_t *)gs->dones;
  auto pk = (AG_GLOBAL float *)(gs->packed + ((size_t)slot * gs->d.A * na + o) * 2);
  rw[o] = r; ms[o] = (int)m; dn[o] = (uint8_t)(i == 0 ? done : 0); pk[0] = (float)r; pk[1] = (i == 0 && done) ? 1.0f : 0.0f;
}

// q_done >= 0: the lean kernel (agar_quiet.inl) already did the prologue and the first q_done ticks of this step
// (single player; q_before = the agent's mass before the step).
// QUIET: try a quiet run in front of every general tick (quiet_run).  The host emulation and the general engine behind k_fused (256 registers)
// do; k_step does NOT: inlined there, the quiet path's ~45 live scalars and its pass sat on top of the general engine's working set in a
// 128-register budget -- 460 bytes of scratch per lane, 213 spilled registers, 20-29 of them saved in front of every tick whether or not a
// quiet run followed (12-20 KB of scratch written per arena-step).  Without it k_step needs 28 bytes.  Results are the same either way (a
// quiet tick is the general tick of a one-cell, food-free arena); an arena that turns quiet in the middle of a k_step step simply finishes
// that step on general ticks -- the front kernel takes its next one.  (As a real call the quiet run cost more than it saved: the callee
// saves 53 registers per call, and builds of it were not bit-stable across loop forms -- see DESIGN.md "tried and dropped".)
template <int NS, bool AV, bool QUIET = true> AG_DEV void env_step(AgCtx<NS, AV> &c, int ticks, bool with_env, int q_done = -1, int q_before = 0) {
  int na = c.gs->d.n_agents, mode = c.gs->g.mode;
  int *before = L_I(c, L_TMP) + 20;  // [n_agents <= 32] masses before the ticks, in rewards order (LDS, not private memory; words 20 .. 51 of the mailbox:
                                     // players_collision keeps its own tables at 52 .. 116, the scalar mailbox is words 0 .. 19)
  if (with_env && q_done >= 0) { AG_SERIAL { before[0] = q_before; } ag_lds_order(); }
  else if (with_env) {
    for (int i = 0; i < na; i++) {  // take_actions: agent i == player slot i (pids_[i]), BaseEnvironment.hpp:141-176
      size_t o = (size_t)c.arena * na + i;
      if (c.act) take_action(c, i, c.act_dxdy[2 * o], c.act_dxdy[2 * o + 1], c.act[o]);
    }
    SW(c, AR_RESPAWNED, 0);
    for (int i = 0; i < na; i++) {
      int p = agent_in_order(c, i); unsigned m = p >= 0 ? player_mass(c, p) : 0u;
      AG_SERIAL { before[i] = (int)m; }
      if (mode == 3 && m >= 23000u) SW(c, AR_DONE, 1);
    }
    ag_lds_order();
  }
  for (int t = q_done > 0 ? q_done : 0; t < ticks;) {
#ifndef AG_NO_QUIET
    if constexpr (QUIET) {
      if (c.P == 1 && SR(c, AR_NFOOD) == 0 && ag_uni(PLS(c, 0)[PL_NCELLS]) == 1) {
        t += quiet_run(c, ticks - t);
        if (t >= ticks) break;
      }
    }
#endif
    arena_tick(c); t++;
  }
  ag_set_priority(0);
  if (with_env) {
    if (c.gs->g.screen_respawn) {  // R: ScreenEnvironment.hpp:233-243 via BaseEnvironment.hpp:96-97: per agent, after the ticks
      for (int i = 0; i < na; i++) if (ag_uni(PLS(c, i)[PL_NCELLS]) == 0) { respawn(c, i); SW(c, AR_RESPAWNED, 1); }
    }
    if (mode == 0) respawn_dead(c);
    else if (mode > 6) {  // BaseEnvironment.hpp:103-114
      int done = SR(c, AR_DONE);
      for (int k = 0; k < c.P; k++) {
        bool dead = ag_uni(PLS(c, SR(c, AR_ORDER0 + k))[PL_NCELLS]) == 0;
        done = (dead || SR(c, AR_RESPAWNED)) ? 1 : 0;
        if (dead) { done = 1; break; }
      }
      SW(c, AR_DONE, done);
    }
    for (int i = 0; i < na; i++) {
      int p = agent_in_order(c, i); unsigned m = p >= 0 ? player_mass(c, p) : 0u;
      if (mode == 3 && m >= 23000u) SW(c, AR_DONE, 1);
      { unsigned b4 = (unsigned)ag_uni(before[i]); int rsp = SR(c, AR_RESPAWNED), dn = SR(c, AR_DONE); AG_SERIAL { emit_agent_result(c.gs, c.slot, c.arena, na, i, m, b4, rsp, dn); } }
    }
  }
}

template <int NS, bool AV> AG_DEV void env_reset(AgCtx<NS, AV> &c, int reset_ids) {  // R: BaseEnvironment.hpp:179-204, Engine.hpp:98-117
  if (reset_ids) SW(c, AR_IDC, 1);
  SW(c, AR_NPEL, 0); SW(c, AR_NVIR, 0); SW(c, AR_NFOOD, 0); SW(c, AR_TICKS, 0); SW(c, AR_FLAGS, 0);
  SW(c, AR_NEVP, 0); SW(c, AR_NEVV, 0); SW(c, AR_DONE, 0); SW(c, AR_RESPAWNED, 0);
  AG_PEL_FOR(s, lane, i) { PELX(c, s, lane) = AG_PEL_SENTINEL; PELY(c, s, lane) = AG_PEL_SENTINEL; }
  c.pel_loaded = true; c.pel_dirty = true; c.pel_all = true; SW(c, AR_SAFE, 0);
  if (c.gs->g.squared) create_squared_pellets(c); else add_pellets(c, c.gs->g.target_pellets);
  add_viruses(c, c.gs->g.target_viruses);
  int na = c.gs->d.n_agents, mode = c.gs->g.mode, nb = c.P - na - c.gs->g.example_bots;
  auto rnd = g_rnd(c);
  for (int i = 0; i < c.P; i++) {  // add_player (Engine.hpp:70-83): slot i gets pid next_pid++; agents first, then bots
    int pid = SR(c, AR_NEXT_PID); SW(c, AR_NEXT_PID, (pid + 1) & 0xFFFF);
    // BaseEnvironment::add_bots (mode 0, :374-399): Hungry, HungryShy, Aggressive, AggressiveShy by i % num_bots;
    // custom_add_bot (mode > 6, :401-425): one bot of type mode - 7
    int kind = 0;
    if (i >= na + nb) kind = AG_KIND_EXAMPLE;   // bench/main.cpp:21-24: ExampleBots added to the freshly reset engine
    else if (i >= na) { int b = mode > 6 ? mode - 7 : (i - na) % nb; kind = b == 0 ? AG_KIND_HUNGRY : b == 1 ? AG_KIND_HUNGRY_SHY : b == 2 ? AG_KIND_AGGRESSIVE : b == 3 ? AG_KIND_AGGRESSIVE_SHY : AG_KIND_HUNGRY; }
    int *P = PLS(c, i);
    AG_SERIAL {
      P[PL_PID] = pid; P[PL_KIND] = kind; P[PL_ACTION] = 0; P[PL_TX] = 0; P[PL_TY] = 0; P[PL_CAND_IDX] = -1;   // (AR_SAFE is zeroed above: no tracked pellet either)
      P[PL_FOOD_EATEN] = 0; P[PL_HIGHEST_MASS] = (int)AG_CELL_MIN_SIZE; P[PL_CELLS_EATEN] = 0; P[PL_VIRUSES_EATEN] = 0;
      if (kind == 0) (void)ag_rand_next(rnd);  // Player(pid, name) draws random_color(): core/Player.hpp:53, color.hpp:14-16
    }
    ag_lds_order();
    respawn(c, i);
  }
  ag_mem_fence();
  compute_player_order(c);
}
