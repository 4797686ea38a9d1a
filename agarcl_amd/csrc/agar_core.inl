// Wave-level Agar.io engine: ONE 64-lane wavefront simulates ONE arena.
//
// Programming discipline (what makes the code both a CDNA4 kernel and checkable on a host):
//   * code outside AG_LANES/AG_SERIAL is *wave-uniform* scalar code (same value in every lane;
//     the compiler keeps it in SGPRs where it can prove uniformity -- collectives end in
//     readfirstlane for that reason);
//   * AG_LANES(i, n) bodies are the data-parallel parts: lane-private temporaries only, all
//     communication through LDS/HBM;
//   * collectives (wave_sum / wave_min / wave_any / wave_compact) are the only cross-lane ops:
//     ballot + popcount prefix for ordered compaction, xor-shuffle trees for reductions;
//   * AG_SERIAL sections replay the reference's order-dependent semantics on lane 0.
// The reference semantics being reproduced are cited as "R:" (paths under /root/reference).
//
// Compiled for gfx950 by agar_engine.hip.  tests/emu/ compiles the same text with
// -DAGAR_CPU_EMU into a *test-only* host library (lanes become loops) so that kernel logic can be
// diffed against the oracle without a GPU; the product never loads that build.
#pragma once
#include "agar_types.h"
#include <limits.h>
#include <math.h>

#pragma clang fp contract(off)

#ifdef AGAR_CPU_EMU
#define AG_DEV static inline
#define AG_LANES(i, n) for (int i = 0; i < (n); ++i)
#define AG_SERIAL if (true)
#define AG_LANE0 true
AG_DEV int ag_uni(int v) { return v; }
AG_DEV unsigned ag_uniu(unsigned v) { return v; }
AG_DEV float ag_unif(float v) { return v; }
AG_DEV void ag_fence() {}
AG_DEV float ag_sqrtf(float x) { return sqrtf(x); }
AG_DEV float ag_divf(float a, float b) { return a / b; }
#else
#define AG_DEV __device__ __forceinline__
#define AG_LANES(i, n) for (int i = (int)threadIdx.x; i < (n); i += 64)
#define AG_SERIAL if (threadIdx.x == 0)
#define AG_LANE0 (threadIdx.x == 0)
AG_DEV int ag_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
AG_DEV unsigned ag_uniu(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
AG_DEV float ag_unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// make lane-0 / other-lane stores to LDS+HBM visible to the whole wave before it continues
AG_DEV void ag_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
// IEEE correctly rounded: built with -fhip-fp32-correctly-rounded-divide-sqrt (build.py); NOT __fsqrt_rn,
// which ROCm's headers map to the approximate __ocml_native_sqrt_f32.
AG_DEV float ag_sqrtf(float x) { return __builtin_sqrtf(x); }
AG_DEV float ag_divf(float a, float b) { return a / b; }
#endif

#include "agar_libm.inl"

// ---- collectives --------------------------------------------------------------------------------
#ifdef AGAR_CPU_EMU
template <class F> AG_DEV int wave_sum(int n, F f) { int s = 0; for (int i = 0; i < n; i++) s += f(i); return s; }
template <class F> AG_DEV unsigned wave_max(int n, F f) { unsigned s = 0; for (int i = 0; i < n; i++) { unsigned v = f(i); if (v > s) s = v; } return s; }
template <class F> AG_DEV unsigned wave_min(int n, F f) { unsigned s = UINT_MAX; for (int i = 0; i < n; i++) { unsigned v = f(i); if (v < s) s = v; } return s; }
template <class F> AG_DEV bool wave_any(int n, F f) { for (int i = 0; i < n; i++) if (f(i)) return true; return false; }
template <class P, class S> AG_DEV int wave_compact(int n, P pred, S sink) { int c = 0; for (int i = 0; i < n; i++) if (pred(i)) { sink(i, c); c++; } return c; }
#else
AG_DEV int wred_add(int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return __builtin_amdgcn_readfirstlane(v); }
AG_DEV unsigned wred_max(unsigned v) { for (int o = 32; o > 0; o >>= 1) { unsigned t = (unsigned)__shfl_xor((int)v, o, 64); v = t > v ? t : v; } return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
AG_DEV unsigned wred_min(unsigned v) { for (int o = 32; o > 0; o >>= 1) { unsigned t = (unsigned)__shfl_xor((int)v, o, 64); v = t < v ? t : v; } return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
template <class F> AG_DEV int wave_sum(int n, F f) { int s = 0; for (int i = (int)threadIdx.x; i < n; i += 64) s += f(i); return wred_add(s); }
template <class F> AG_DEV unsigned wave_max(int n, F f) { unsigned s = 0; for (int i = (int)threadIdx.x; i < n; i += 64) { unsigned v = f(i); s = v > s ? v : s; } return wred_max(s); }
template <class F> AG_DEV unsigned wave_min(int n, F f) { unsigned s = UINT_MAX; for (int i = (int)threadIdx.x; i < n; i += 64) { unsigned v = f(i); s = v < s ? v : s; } return wred_min(s); }
template <class F> AG_DEV bool wave_any(int n, F f) { bool a = false; for (int i = (int)threadIdx.x; i < n; i += 64) a = a | (bool)f(i); return __ballot(a) != 0ull; }
// ordered stream compaction: sink(i, rank) for every i with pred(i), rank = number of earlier hits
template <class P, class S> AG_DEV int wave_compact(int n, P pred, S sink) {
  int count = 0;
  const unsigned long long lt = (1ull << threadIdx.x) - 1ull;
  for (int base = 0; base < n; base += 64) {
    int i = base + (int)threadIdx.x;
    bool p = (i < n) && pred(i);
    unsigned long long m = __ballot(p);
    if (p) sink(i, count + __popcll(m & lt));
    count += __popcll(m);
  }
  return count;
}
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
AG_DEV unsigned clamp_mass(unsigned m) { return m > AG_CELL_MIN_SIZE ? m : AG_CELL_MIN_SIZE; }  // R: Entities.hpp:171-177

// per-arena context ------------------------------------------------------------------------------
struct AgCtx {
  AgDims d; AgParams g;
  int arena;
  // HBM slices of this arena
  float *gpx, *gpy; int32_t *gpid;
  float *vx, *vy, *vvx, *vvy; int32_t *vm, *vh, *vid;
  float *fx, *fy, *fvx, *fvy; int32_t *fid;
  uint64_t *mt; int32_t *gar; int32_t *gpl; int32_t *gvt;
  int32_t *gev_p, *gev_v;
  const float *lut_r, *lut_ms, *lut_ss, *lut_anti;
  // LDS
  float *px, *py;                                                    // [PC]
  float *cx, *cy, *cvx, *cvy, *csx, *csy; unsigned *cm; int *cid; unsigned *cdl;  // [P*CC]
  float *nx, *ny, *nvx, *nvy, *nsx, *nsy; unsigned *nm; int *nid; unsigned *ndl;  // [CC] created this tick
  int *S;        // [AR_WORDS] arena scalars
  int *PLS;      // [P*PL_WORDS] player scalars
  int *evp, *evv; unsigned *cand; uint64_t *rb; int *tmp;
  int *ncreated;
};

static inline
#ifndef AGAR_CPU_EMU
__host__ __device__
#endif
size_t ag_lds_bytes(const AgDims &d) {
  size_t w = 0;
  w += 2 * (size_t)d.PC;              // px py
  w += 9 * (size_t)d.P * d.CC;        // cells
  w += 9 * (size_t)d.CC;              // created
  w += AR_WORDS + (size_t)d.P * PL_WORDS;
  w += AG_EV_CAP + AG_EVV_CAP + AG_CAND_CAP;
  w += 2 * 128;                       // rb (128 x u64)
  w += 64;                            // tmp
  w += 4;                             // ncreated + pad
  return w * 4;
}

AG_DEV void ag_bind_lds(AgCtx &c, void *base) {
  // rb (u64) first for 8-byte alignment
  char *p = (char *)base;
  c.rb = (uint64_t *)p; p += 128 * 8;
  c.px = (float *)p; p += 4 * (size_t)c.d.PC; c.py = (float *)p; p += 4 * (size_t)c.d.PC;
  size_t nc = (size_t)c.d.P * c.d.CC;
  c.cx = (float *)p; p += 4 * nc; c.cy = (float *)p; p += 4 * nc; c.cvx = (float *)p; p += 4 * nc; c.cvy = (float *)p; p += 4 * nc;
  c.csx = (float *)p; p += 4 * nc; c.csy = (float *)p; p += 4 * nc; c.cm = (unsigned *)p; p += 4 * nc; c.cid = (int *)p; p += 4 * nc; c.cdl = (unsigned *)p; p += 4 * nc;
  size_t cc = (size_t)c.d.CC;
  c.nx = (float *)p; p += 4 * cc; c.ny = (float *)p; p += 4 * cc; c.nvx = (float *)p; p += 4 * cc; c.nvy = (float *)p; p += 4 * cc;
  c.nsx = (float *)p; p += 4 * cc; c.nsy = (float *)p; p += 4 * cc; c.nm = (unsigned *)p; p += 4 * cc; c.nid = (int *)p; p += 4 * cc; c.ndl = (unsigned *)p; p += 4 * cc;
  c.S = (int *)p; p += 4 * AR_WORDS; c.PLS = (int *)p; p += 4 * (size_t)c.d.P * PL_WORDS;
  c.evp = (int *)p; p += 4 * AG_EV_CAP; c.evv = (int *)p; p += 4 * AG_EVV_CAP; c.cand = (unsigned *)p; p += 4 * AG_CAND_CAP;
  c.tmp = (int *)p; p += 4 * 64; c.ncreated = (int *)p; p += 16;
}

AG_DEV void ag_bind_arena(AgCtx &c, const AgState &s, int a) {
  c.arena = a;
  size_t A = (size_t)a;
  c.gpx = s.pel_x + A * c.d.PC; c.gpy = s.pel_y + A * c.d.PC; c.gpid = s.pel_id + A * c.d.PC;
  c.vx = s.vir_x + A * c.d.VC; c.vy = s.vir_y + A * c.d.VC; c.vvx = s.vir_vx + A * c.d.VC; c.vvy = s.vir_vy + A * c.d.VC;
  c.vm = s.vir_mass + A * c.d.VC; c.vh = s.vir_hits + A * c.d.VC; c.vid = s.vir_id + A * c.d.VC;
  c.fx = s.food_x + A * c.d.FC; c.fy = s.food_y + A * c.d.FC; c.fvx = s.food_vx + A * c.d.FC; c.fvy = s.food_vy + A * c.d.FC; c.fid = s.food_id + A * c.d.FC;
  c.mt = s.mt + A * 312; c.gar = s.ar + A * AR_WORDS; c.gpl = s.pl + A * c.d.P * PL_WORDS; c.gvt = s.vticks + A * c.d.P * AG_VT_CAP;
  c.gev_p = s.ev_p + A * AG_EV_CAP; c.gev_v = s.ev_v + A * AG_EVV_CAP;
  c.lut_r = s.lut_r; c.lut_ms = s.lut_ms; c.lut_ss = s.lut_ss; c.lut_anti = s.lut_anti;
}

// uniform reads of the LDS scalar blocks
AG_DEV int SR(const AgCtx &c, int k) { return ag_uni(c.S[k]); }
AG_DEV int PR(const AgCtx &c, int p, int k) { return ag_uni(c.PLS[p * PL_WORDS + k]); }
AG_DEV float PRF(const AgCtx &c, int p, int k) { return u2f(PR(c, p, k)); }
AG_DEV void SW(AgCtx &c, int k, int v) { AG_SERIAL { c.S[k] = v; } }
AG_DEV void PW(AgCtx &c, int p, int k, int v) { AG_SERIAL { c.PLS[p * PL_WORDS + k] = v; } }
AG_DEV void flag(AgCtx &c, unsigned f) { AG_SERIAL { c.S[AR_FLAGS] |= (int)f; } }
AG_DEV float lut(const AgCtx &c, const float *t, unsigned m) { return t[m < (unsigned)AG_LUT_SIZE ? m : (unsigned)AG_LUT_SIZE - 1u]; }
AG_DEV float radius_of(const AgCtx &c, unsigned m) { return lut(c, c.lut_r, m); }

// ---- load / store arena state between HBM and LDS ------------------------------------------------
AG_DEV void arena_load(AgCtx &c, const AgState &s) {
  AG_LANES(i, AR_WORDS) c.S[i] = c.gar[i];
  AG_LANES(i, c.d.P * PL_WORDS) c.PLS[i] = c.gpl[i];
  ag_fence();
  int np = SR(c, AR_NPEL);
  AG_LANES(i, np) { c.px[i] = c.gpx[i]; c.py[i] = c.gpy[i]; }
  size_t cb = (size_t)c.arena * c.d.P * c.d.CC;
  for (int p = 0; p < c.d.P; p++) {
    int n = PR(c, p, PL_NCELLS); size_t o = cb + (size_t)p * c.d.CC; int l = p * c.d.CC;
    AG_LANES(i, n) {
      c.cx[l + i] = s.cell_x[o + i]; c.cy[l + i] = s.cell_y[o + i]; c.cvx[l + i] = s.cell_vx[o + i]; c.cvy[l + i] = s.cell_vy[o + i];
      c.csx[l + i] = s.cell_sx[o + i]; c.csy[l + i] = s.cell_sy[o + i]; c.cm[l + i] = s.cell_m[o + i]; c.cid[l + i] = s.cell_id[o + i]; c.cdl[l + i] = s.cell_dl[o + i];
    }
  }
  ag_fence();
}
AG_DEV void arena_store(AgCtx &c, const AgState &s) {
  ag_fence();
  int np = SR(c, AR_NPEL);
  AG_LANES(i, np) { c.gpx[i] = c.px[i]; c.gpy[i] = c.py[i]; }
  size_t cb = (size_t)c.arena * c.d.P * c.d.CC;
  int total_cells = 0;
  for (int p = 0; p < c.d.P; p++) {
    int n = PR(c, p, PL_NCELLS); size_t o = cb + (size_t)p * c.d.CC; int l = p * c.d.CC;
    total_cells += n;
    AG_LANES(i, n) {
      s.cell_x[o + i] = c.cx[l + i]; s.cell_y[o + i] = c.cy[l + i]; s.cell_vx[o + i] = c.cvx[l + i]; s.cell_vy[o + i] = c.cvy[l + i];
      s.cell_sx[o + i] = c.csx[l + i]; s.cell_sy[o + i] = c.csy[l + i]; s.cell_m[o + i] = c.cm[l + i]; s.cell_id[o + i] = c.cid[l + i]; s.cell_dl[o + i] = c.cdl[l + i];
    }
  }
  AG_LANES(i, AR_WORDS) c.gar[i] = c.S[i];
  AG_LANES(i, c.d.P * PL_WORDS) c.gpl[i] = c.PLS[i];
  int nevp = SR(c, AR_NEVP), nevv = SR(c, AR_NEVV);
  AG_LANES(i, nevp) c.gev_p[i] = c.evp[i];
  AG_LANES(i, nevv) c.gev_v[i] = c.evv[i];
  AG_SERIAL { int32_t *cn = s.counts + (size_t)c.arena * 4; cn[0] = np; cn[1] = c.S[AR_NVIR]; cn[2] = c.S[AR_NFOOD]; cn[3] = total_cells; }
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
AG_DEV void mt_twist(AgCtx &c) {  // three dependency phases, each data-parallel across the wave
  uint64_t *mt = c.mt;
  ag_fence();
  for (int base = 0; base < 156; base += 64) {  // phase 1: reads old values only
    int lim = base + 64 < 156 ? base + 64 : 156;
#ifdef AGAR_CPU_EMU
    for (int i = base; i < lim; i++) mt[i] = mt_mix(mt[i], mt[i + 1], mt[i + 156]);
#else
    int i = base + (int)threadIdx.x; uint64_t v = 0;
    if (i < lim) v = mt_mix(mt[i], mt[i + 1], mt[i + 156]);
    ag_fence();
    if (i < lim) mt[i] = v;
    ag_fence();
#endif
  }
  for (int base = 156; base < 311; base += 64) {  // phase 2: far operand is a phase-1 result
    int lim = base + 64 < 311 ? base + 64 : 311;
#ifdef AGAR_CPU_EMU
    for (int i = base; i < lim; i++) mt[i] = mt_mix(mt[i], mt[i + 1], mt[i - 156]);
#else
    int i = base + (int)threadIdx.x; uint64_t v = 0;
    if (i < lim) v = mt_mix(mt[i], mt[i + 1], mt[i - 156]);
    ag_fence();
    if (i < lim) mt[i] = v;
    ag_fence();
#endif
  }
  AG_SERIAL { mt[311] = mt_mix(mt[311], mt[0], mt[155]); }
  ag_fence();
}
// fill c.rb[0..n) (n <= 128) with the next n raw 64-bit outputs
AG_DEV void mt_fill(AgCtx &c, int n) {
  int produced = 0;
  while (produced < n) {
    int idx = SR(c, AR_MTIDX);
    if (idx >= 312) { mt_twist(c); idx = 0; }
    int take = 312 - idx < n - produced ? 312 - idx : n - produced;
    AG_LANES(j, take) c.rb[produced + j] = mt_temper(c.mt[idx + j]);
    SW(c, AR_MTIDX, idx + take);
    ag_fence();
    produced += take;
  }
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
template <class SINK> AG_DEV void draw_locations(AgCtx &c, int count, float radius, SINK sink) {
  float two_r = 2.0f * radius; float span = c.g.W - two_r;
  for (int done = 0; done < count; done += 64) {
    int b = count - done < 64 ? count - done : 64;
    mt_fill(c, 2 * b);
    AG_LANES(j, b) {
      float x = mt_to_float(c.rb[2 * j], span) + radius;
      float y = mt_to_float(c.rb[2 * j + 1], span) + radius;
      sink(done + j, x, y);
    }
    ag_fence();
  }
}

// ---- spawning.  R: Engine.hpp:418-424, 480-485, 426-475, 119-137 -----------------------------------
AG_DEV void add_pellets(AgCtx &c, int n) {
  if (n <= 0) return;
  int np = SR(c, AR_NPEL), idc = SR(c, AR_IDC);
  if (np + n > c.d.PC) { flag(c, 64u); n = c.d.PC - np; if (n <= 0) return; }
  float r = radius_of(c, AG_PELLET_MASS);
  draw_locations(c, n, r, [&](int j, float x, float y) { c.px[np + j] = x; c.py[np + j] = y; c.gpid[np + j] = idc + 1 + j; });
  SW(c, AR_NPEL, np + n); SW(c, AR_IDC, idc + n);
  ag_fence();
}
AG_DEV void add_viruses(AgCtx &c, int n) {
  if (n <= 0) return;
  int nv = SR(c, AR_NVIR), idc = SR(c, AR_IDC);
  if (nv + n > c.d.VC) { flag(c, 4u); n = c.d.VC - nv; if (n <= 0) return; }
  float r = radius_of(c, AG_VIRUS_MASS);
  draw_locations(c, n, r, [&](int j, float x, float y) {
    int k = nv + j; c.vx[k] = x; c.vy[k] = y; c.vvx[k] = 0.0f; c.vvy[k] = 0.0f; c.vm[k] = (int)AG_VIRUS_MASS; c.vh[k] = 0; c.vid[k] = idc + 1 + j; });
  SW(c, AR_NVIR, nv + n); SW(c, AR_IDC, idc + n);
  ag_fence();
}
AG_DEV void create_squared_pellets(AgCtx &c) {
  float W = c.g.W;
  float square = ag_divf(W, 2.0f);
  int pps = f2i(square);
  float cx = ag_divf(W, 2.0f), half = ag_divf(square, 2.0f);
  int idc = SR(c, AR_IDC);
  // every generated point lies inside the arena (centre +- W/4), so all 4*pps are kept, in order
  int total = 4 * pps;
  if (total > c.d.PC) { flag(c, 64u); total = c.d.PC; }
  AG_LANES(k, total) {
    int side = k / pps, i = k - side * pps; float t = (float)i * 1.0f; float x, y;
    if (side == 0) { x = (cx - half) + t; y = cx - half; }
    else if (side == 1) { x = cx + half; y = (cx - half) + t; }
    else if (side == 2) { x = (cx + half) - t; y = cx + half; }
    else { x = cx - half; y = (cx + half) - t; }
    c.px[k] = x; c.py[k] = y; c.gpid[k] = idc + 1 + k;
  }
  SW(c, AR_NPEL, total); SW(c, AR_IDC, idc + total);
  ag_fence();
}
AG_DEV void player_kill(AgCtx &c, int p) {  // R: core/Player.hpp:75-86
  AG_SERIAL {
    int *P = c.PLS + p * PL_WORDS;
    P[PL_NCELLS] = 0; P[PL_MIN_MASS] = (int)AG_CELL_MIN_SIZE; P[PL_SPLIT_CD] = 0; P[PL_FEED_CD] = 0;
    P[PL_ANTI_TEAM] = f2u(1.0f); P[PL_ELAPSED] = 0; P[PL_LAST_DECAY] = 0; P[PL_NVTICKS] = 0;
  }
  ag_fence();
}
AG_DEV void respawn(AgCtx &c, int p) {
  player_kill(c, p);
  unsigned pm = (unsigned)(c.g.agent_mass > (int)AG_CELL_MIN_SIZE ? c.g.agent_mass : (int)AG_CELL_MIN_SIZE);
  float r25 = radius_of(c, AG_CELL_MIN_SIZE);
  int l = p * c.d.CC;
  if (SR(c, AR_NPEL) > 0 && c.g.squared) {
    AG_SERIAL {
      float x = c.px[0], y = c.py[0]; float t = 2.0f * r25; x += t; y += t;
      x = sminf(x, c.g.W - r25); y = sminf(y, c.g.W - r25);
      c.cx[l] = x; c.cy[l] = y;
    }
  } else {
    draw_locations(c, 1, r25, [&](int, float x, float y) { c.cx[l] = x; c.cy[l] = y; });
  }
  AG_SERIAL {
    int idc = c.S[AR_IDC] + 1; c.S[AR_IDC] = idc;
    c.cvx[l] = 0; c.cvy[l] = 0; c.csx[l] = 0; c.csy[l] = 0; c.cm[l] = clamp_mass(pm); c.cid[l] = idc; c.cdl[l] = (unsigned)c.S[AR_CLOCK];
    c.PLS[p * PL_WORDS + PL_NCELLS] = 1;
  }
  ag_fence();
}

// ---- movement.  R: Engine.hpp:609-630, 695-698; core/types.hpp:176-223; Entities.hpp:161-164 ------
AG_DEV void boundary(const AgCtx &c, float &x, float &y, float r) {
  x = smaxf(0.0f, clampf(x, r, c.g.W - r));
  y = smaxf(0.0f, clampf(y, r, c.g.W - r));
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

// serial (lane 0) pieces of the self-collision relaxation; operate on LDS cells l+a, l+b
AG_DEV void cell_move1(AgCtx &c, int k, float dt) {
  float sx = c.cvx[k] + c.csx[k]; float tx = sx * dt; c.cx[k] += tx;
  float sy = c.cvy[k] + c.csy[k]; float ty = sy * dt; c.cy[k] += ty;
}
AG_DEV void avoid_static_overlap(AgCtx &c, int a, int b) {  // R: Engine.hpp:701-749
  float dx = c.cx[b] - c.cx[a], dy = c.cy[b] - c.cy[a];
  float dist = vmag(dx, dy);
  float ra = radius_of(c, c.cm[a]), rb = radius_of(c, c.cm[b]);
  float target = ra + rb;
  if (dist > target) return;
  float den = fabsf(dx) + fabsf(dy);
  float xr = ag_divf(dx, den), yr = ag_divf(dy, den);
  float depth = target - dist;
  float a1 = 0.5f, a2 = 0.5f, b1 = 0.5f, b2 = 0.5f; float W = c.g.W;
  if (c.cx[a] == ra || c.cx[a] == W - ra) { a1 = 1.0f; c.cvx[a] = 0; }
  if (c.cy[a] == ra || c.cy[a] == W - ra) { a2 = 1.0f; c.cvy[a] = 0; }
  if (c.cx[b] == rb || c.cx[b] == W - rb) { b1 = 1.0f; c.cvx[b] = 0; }
  if (c.cy[b] == rb || c.cy[b] == W - rb) { b2 = 1.0f; c.cvy[b] = 0; }
  float t;
  t = xr * depth; t = t * a1; c.cx[a] -= t;
  t = yr * depth; t = t * a2; c.cy[a] -= t;
  t = xr * depth; t = t * b1; c.cx[b] += t;
  t = yr * depth; t = t * b2; c.cy[b] += t;
  boundary(c, c.cx[a], c.cy[a], ra);
  boundary(c, c.cx[b], c.cy[b], rb);
}
AG_DEV void separate_cells(AgCtx &c, int a, int b, float tx, float ty) {  // R: Engine.hpp:803-848
  float dx = c.cx[b] - c.cx[a], dy = c.cy[b] - c.cy[a];
  float dist = vmag(dx, dy);
  float target = radius_of(c, c.cm[a]) + radius_of(c, c.cm[b]);
  if (dist > target) return;
  float den = fabsf(dx) + fabsf(dy);
  float xr = ag_divf(dx, den), yr = ag_divf(dy, den);
  float diff_a = sqr_dist(tx, ty, c.cx[a], c.cy[a]);
  float diff_b = sqr_dist(tx, ty, c.cx[b], c.cy[b]);
  float depth = target - dist;
  int s1 = c.cm[a] < c.cm[b] ? 1 : -1;
  int s2 = diff_a >= diff_b ? 1 : -1;
  int s = (s1 == s2) ? s2 : 0;
  int tc = c.cm[a] < c.cm[b] ? a : b;
  float fs = (float)s, t;
  if (dx >= 0) {
    t = xr * depth; t = t * fs; c.cx[tc] -= t;
    if (dy >= 0) { t = yr * depth; t = t * fs; c.cy[tc] -= t; } else { t = yr * depth; t = t * fs; c.cy[tc] += t; }
  } else {
    t = xr * depth; t = t * fs; c.cx[tc] += t;
    if (dy >= 0) { t = yr * depth; t = t * fs; c.cy[tc] -= t; } else { t = yr * depth; t = t * fs; c.cy[tc] += t; }
  }
}
AG_DEV void elastic(AgCtx &c, int a, int b, float dx, float dy, float dist) {  // R: Engine.hpp:893-938
  float nx = ag_divf(dx, dist), ny = ag_divf(dy, dist);
  float tx = -ny, ty = nx;
  float p1 = c.cvx[a] * nx, p2 = c.cvy[a] * ny; float dpN1 = p1 + p2;
  p1 = c.cvx[b] * nx; p2 = c.cvy[b] * ny; float dpN2 = p1 + p2;
  p1 = c.cvx[a] * tx; p2 = c.cvy[a] * ty; float dpT1 = p1 + p2;
  p1 = c.cvx[b] * tx; p2 = c.cvy[b] * ty; float dpT2 = p1 + p2;
  int m1 = (int)c.cm[a], m2 = (int)c.cm[b];
  float q1 = dpN1 * (float)(m1 - m2);
  float q2 = 2.0f * (float)m2; q2 = q2 * dpN2;
  float v1 = ag_divf(q1 + q2, (float)(m1 + m2));
  q1 = dpN2 * (float)(m2 - m1);
  q2 = 2.0f * (float)m1; q2 = q2 * dpN1;
  float v2 = ag_divf(q1 + q2, (float)(m1 + m2));
  if (c.cm[a] < c.cm[b]) {
    float u = tx * dpT1, w = nx * v1; c.cvx[a] = u + w; u = ty * dpT1; w = ny * v1; c.cvy[a] = u + w;
  } else if (c.cm[a] > c.cm[b]) {
    float u = tx * dpT2, w = nx * v2; c.cvx[b] = u + w; u = ty * dpT2; w = ny * v2; c.cvy[b] = u + w;
  } else {
    float u = tx * dpT1, w = nx * v1; c.cvx[a] = u + w; u = ty * dpT1; w = ny * v1; c.cvy[a] = u + w;
    u = tx * dpT2; w = nx * v2; c.cvx[b] = u + w; u = ty * dpT2; w = ny * v2; c.cvy[b] = u + w;
  }
}
AG_DEV bool cells_touch(const AgCtx &c, int a, int b) {
  return touches(c.cx[a], c.cy[a], radius_of(c, c.cm[a]), c.cx[b], c.cy[b], radius_of(c, c.cm[b]));
}
AG_DEV void prevent_overlap(AgCtx &c, int a, int b, float dt, float tx, float ty) {  // R: Engine.hpp:857-888
  float dx = c.cx[b] - c.cx[a], dy = c.cy[b] - c.cy[a];
  float dist = vmag(dx, dy);
  float target = radius_of(c, c.cm[a]) + radius_of(c, c.cm[b]);
  if (dist > target) return;
  float s, t;
  s = c.cvx[a] + c.csx[a]; t = s * dt; c.cx[a] -= t;
  s = c.cvy[a] + c.csy[a]; t = s * dt; c.cy[a] -= t;
  s = c.cvx[b] + c.csx[b]; t = s * dt; c.cx[b] -= t;
  s = c.cvy[b] + c.csy[b]; t = s * dt; c.cy[b] -= t;
  elastic(c, a, b, dx, dy, dist);
  cell_move1(c, a, dt);
  cell_move1(c, b, dt);
  if (cells_touch(c, a, b)) {
    int d = (int)(c.cm[a] - c.cm[b]);
    if ((d < 0 ? -d : d) <= 10) avoid_static_overlap(c, a, b);
    else separate_cells(c, a, b, tx, ty);
  }
  boundary(c, c.cx[a], c.cy[a], radius_of(c, c.cm[a]));
  boundary(c, c.cx[b], c.cy[b], radius_of(c, c.cm[b]));
}
AG_DEV void self_collisions(AgCtx &c, int p, int n) {  // R: Engine.hpp:763-794
  int l = p * c.d.CC;
  // wave-parallel any-touch test; when no pair touches the reference's first pass is a no-op
  bool any = wave_any(n * n, [&](int k) { int a = k / n, b = k - a * n; return a < b && cells_touch(c, l + a, l + b); });
  if (!any) return;
  float tx = PRF(c, p, PL_TX), ty = PRF(c, p, PL_TY), dt = c.g.dt;
  AG_SERIAL {
    bool overlap = false;
    for (int iter = 0; iter < 5; iter++) {
      overlap = false;
      for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++)
          if (cells_touch(c, l + a, l + b)) { overlap = true; prevent_overlap(c, l + a, l + b, dt, tx, ty); }
      if (!overlap) break;
    }
    if (overlap)
      for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++)
          if (cells_touch(c, l + a, l + b)) avoid_static_overlap(c, l + a, l + b);
  }
  ag_fence();
}
AG_DEV void move_player(AgCtx &c, int p, int n) {
  int l = p * c.d.CC;
  float tx = PRF(c, p, PL_TX), ty = PRF(c, p, PL_TY), dt = c.g.dt;
  AG_LANES(i, n) {
    int k = l + i;
    float x = c.cx[k], y = c.cy[k], svx = c.csx[k], svy = c.csy[k]; unsigned m = c.cm[k];
    float d = tx - x; float vx = 3.0f * d;
    d = ty - y; float vy = 3.0f * d;
    float hi = lut(c, c.lut_ms, m);
    if (vmag(vx, vy) > hi) {  // clamp_speed(0, hi): set_speed re-evaluates speed() after dx changed
      float f = ag_divf(hi, vmag(vx, vy)); vx *= f;
      float g = ag_divf(hi, vmag(vx, vy)); vy *= g;
    }
    float s = vx + svx; float t = s * dt; x += t;
    s = vy + svy; t = s * dt; y += t;
    v_decelerate(svx, svy, AG_SPLIT_DECEL, dt);
    boundary(c, x, y, radius_of(c, m));
    c.cx[k] = x; c.cy[k] = y; c.cvx[k] = vx; c.cvy[k] = vy; c.csx[k] = svx; c.csy[k] = svy;
  }
  ag_fence();
  unsigned mn = wave_min(n, [&](int i) { return c.cm[l + i]; });
  PW(c, p, PL_MIN_MASS, (int)mn);
  if (n >= 2) self_collisions(c, p, n);
}

// ---- created-cell buffer ---------------------------------------------------------------------------
AG_DEV void put_created(AgCtx &c, int slot, float x, float y, float vx, float vy, float sx, float sy, unsigned m, int id, unsigned dl) {
  if (slot >= c.d.CC) return;  // overflow is flagged by the caller
  c.nx[slot] = x; c.ny[slot] = y; c.nvx[slot] = vx; c.nvy[slot] = vy; c.nsx[slot] = sx; c.nsy[slot] = sy; c.nm[slot] = clamp_mass(m); c.nid[slot] = id; c.ndl[slot] = dl;
}
// Engine::cell_split for LDS cell k (lane-level).  R: Engine.hpp:1067-1093.  Caller checked mass >= 50.
AG_DEV void do_cell_split(AgCtx &c, int k, int slot, int id, float tx, float ty) {
  unsigned m = c.cm[k];
  unsigned split_mass = m / 2u, remaining = m - split_mass;
  c.cm[k] = clamp_mass(remaining);
  float x = c.cx[k], y = c.cy[k];
  float ddx = tx - x, ddy = ty - y;
  float ax = fabsf(ddx), ay = fabsf(ddy); float n2 = ax * ax, n2b = ay * ay; float nrm = ag_sqrtf(n2 + n2b);
  float dirx = ag_divf(ddx, nrm), diry = ag_divf(ddy, nrm);
  float r = radius_of(c, c.cm[k]);
  float ox = dirx * r, oy = diry * r;
  float lx = x + ox, ly = y + oy;
  lx = smaxf(0.0f, clampf(lx, r, c.g.W - r));
  ly = smaxf(0.0f, clampf(ly, r, c.g.W - r));
  float ss = lut(c, c.lut_ss, split_mass);
  float vx = dirx * ss, vy = diry * ss;
  unsigned dl = (unsigned)c.S[AR_CLOCK] + (unsigned)c.g.recomb_ticks;
  put_created(c, slot, lx, ly, vx, vy, vx, vy, split_mass, id, dl);
  c.cdl[k] = dl;
}

// ---- viruses.  R: Engine.hpp:1223-1252 (grid :1207-1221), disrupt :1263-1294 ---------------------------
AG_DEV bool virus_collisions(AgCtx &c, int p, int n, int create_limit, bool can_eat_virus) {
  int nv = SR(c, AR_NVIR);
  if (nv == 0) return false;
  int l = p * c.d.CC;
  unsigned maxm = wave_max(n, [&](int i) { return c.cm[l + i]; });
  if (maxm < 111u) return false;  // virus mass >= 100 and can_eat needs mass > 1.1 * virus mass
  for (int ci = 0; ci < n; ci++) {
    int k = l + ci;
    unsigned m = ag_uniu(c.cm[k]);
    if (m < 111u) continue;
    float x = ag_unif(c.cx[k]), y = ag_unif(c.cy[k]);
    float r = radius_of(c, m);
    int gx = f2i(x) / AG_VIRUS_GRID, gy = f2i(y) / AG_VIRUS_GRID;
    unsigned VC = (unsigned)c.d.VC;
    unsigned key = wave_min(nv, [&](int vi) -> unsigned {
      float vx = c.vx[vi], vy = c.vy[vi]; unsigned vmass = (unsigned)c.vm[vi];
      int bx = f2i(vx) / AG_VIRUS_GRID, by = f2i(vy) / AG_VIRUS_GRID;
      int ddx = bx - gx, ddy = by - gy;
      bool ok = ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1 && bx >= 0 && bx < c.g.vgw && by >= 0 && by < c.g.vgh;
      ok = ok && can_eat_mass(m, vmass) && collides(x, y, r, vx, vy, radius_of(c, vmass));
      return ok ? (unsigned)((ddx + 1) * 3 + (ddy + 1)) * VC + (unsigned)vi : UINT_MAX;
    });
    if (key == UINT_MAX) continue;
    int vi = (int)(key % VC);
    if (can_eat_virus) {
      AG_SERIAL { c.cm[k] = clamp_mass(m + (unsigned)c.vm[vi]); }
    } else {
      // disrupt
      unsigned total = m;
      unsigned nm = clamp_mass((unsigned)((float)m / 2.0f));
      nm = clamp_mass(nm + (total - nm) % AG_CELL_POP_SIZE);
      unsigned pop = total - nm;
      int num_new = (int)((pop + AG_CELL_POP_SIZE - 1u) / AG_CELL_POP_SIZE);
      if (create_limit < num_new) num_new = create_limit;
      float cvx = ag_unif(c.cvx[k]), cvy = ag_unif(c.cvy[k]);
      float theta = v_direction(cvx, cvy);
      float sp = lut(c, c.lut_ms, AG_CELL_POP_SIZE);
      float virx = ag_unif(c.vx[vi]), viry = ag_unif(c.vy[vi]);
      int idc = SR(c, AR_IDC), nc0 = ag_uni(*c.ncreated);
      unsigned dl = (unsigned)SR(c, AR_CLOCK) + (unsigned)c.g.recomb_ticks;
      if (nc0 + num_new > c.d.CC) flag(c, 1u);
      AG_LANES(j, num_new) {
        float inc = (float)(2 * 3.14159265358979323846 * j / num_new);
        float dvel = theta + inc;
        float ang = theta + dvel;
        float svx = sp * ag_cosf(ang), svy = sp * ag_sinf(ang);
        unsigned rem = pop - AG_CELL_POP_SIZE * (unsigned)j;  // each earlier new cell took min(rem, 25)
        unsigned cmass = rem < AG_CELL_POP_SIZE ? rem : AG_CELL_POP_SIZE;
        put_created(c, nc0 + j, virx, viry, cvx, cvy, svx, svy, cmass, idc + 1 + j, dl);
      }
      AG_SERIAL { c.cm[k] = nm; c.cdl[k] = dl; c.S[AR_IDC] = idc + num_new; *c.ncreated = nc0 + num_new; }
    }
    AG_SERIAL { int ne = c.S[AR_NEVV]; if (ne < AG_EVV_CAP) c.evv[ne] = vi; else c.S[AR_FLAGS] |= 8; c.S[AR_NEVV] = ne + 1; }
    ag_fence();
    return true;
  }
  return false;
}

// ---- pellets.  R: Engine.hpp:976-1000 (grid :962-974) ---------------------------------------------------
// The reference grows the eater (and therefore its radius) while it scans the buckets in a fixed
// order, so whether pellet j is eaten can depend on pellets eaten before it.  Fast path: no pellet
// inside the *current* radius => nothing is eaten.  Otherwise gather the candidate set that is closed
// under the maximal possible growth, order it like the reference's scan, and replay it on lane 0.
AG_DEV int pellets_eat(AgCtx &c, int p, int n) {
  int np = SR(c, AR_NPEL);
  if (np == 0) return 0;
  int l = p * c.d.CC, eaten_total = 0;
  bool all_vis = c.g.pgw <= 2 && c.g.pgh <= 2;  // every bucket is within +-1 of every other
  for (int ci = 0; ci < n; ci++) {
    int k = l + ci;
    unsigned m = ag_uniu(c.cm[k]);
    float x = ag_unif(c.cx[k]), y = ag_unif(c.cy[k]);
    int gx = f2i(x) / AG_PELLET_GRID, gy = f2i(y) / AG_PELLET_GRID;
    float r0 = radius_of(c, m); float rr0 = r0 * r0;
    auto hit = [&](int i, float rr) -> bool {
      float qx = c.px[i], qy = c.py[i];
      bool ok = rr >= sqr_dist(x, y, qx, qy);
      if (!all_vis) { int ddx = f2i(qx) / AG_PELLET_GRID - gx, ddy = f2i(qy) / AG_PELLET_GRID - gy; ok = ok && ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1; }
      return ok;
    };
    int c0 = wave_sum(np, [&](int i) { return hit(i, rr0) ? 1 : 0; });
    if (c0 == 0) continue;
    // closure of the candidate set under growth
    int K = c0; float rrK = rr0;
    for (;;) {
      if (m + (unsigned)K >= (unsigned)AG_LUT_SIZE) { flag(c, 32u); }
      float rk = radius_of(c, m + (unsigned)K); rrK = rk * rk;
      int cK = wave_sum(np, [&](int i) { return hit(i, rrK) ? 1 : 0; });
      if (cK == K) break;
      K = cK;
    }
    if (K > AG_CAND_CAP) { flag(c, 8u); }
    unsigned PC = (unsigned)c.d.PC;
    int ncand = wave_compact(np, [&](int i) { return hit(i, rrK); }, [&](int i, int rank) {
      if (rank < AG_CAND_CAP) {
        int ddx = f2i(c.px[i]) / AG_PELLET_GRID - gx, ddy = f2i(c.py[i]) / AG_PELLET_GRID - gy;
        c.cand[rank] = (unsigned)((ddx + 1) * 3 + (ddy + 1)) * PC + (unsigned)i;
      }
    });
    if (ncand > AG_CAND_CAP) ncand = AG_CAND_CAP;
    ag_fence();
    AG_SERIAL {
      for (int a = 1; a < ncand; a++) { unsigned v = c.cand[a]; int b = a - 1; while (b >= 0 && c.cand[b] > v) { c.cand[b + 1] = c.cand[b]; b--; } c.cand[b + 1] = v; }
      unsigned mc = m; int ne = c.S[AR_NEVP], e0 = ne;
      for (int a = 0; a < ncand; a++) {
        int i = (int)(c.cand[a] % PC);
        float rc = radius_of(c, mc); float rrc = rc * rc;
        if (rrc >= sqr_dist(x, y, c.px[i], c.py[i])) {
          if (ne < AG_EV_CAP) c.evp[ne] = i; else c.S[AR_FLAGS] |= 8;
          ne++; mc = clamp_mass(mc + AG_PELLET_MASS);
        }
      }
      c.cm[k] = mc; c.S[AR_NEVP] = ne; c.tmp[0] = ne - e0;
    }
    ag_fence();
    eaten_total += ag_uni(c.tmp[0]);
  }
  return eaten_total;
}

// ---- foods.  R: Engine.hpp:1011-1025 (eat), 1027-1054 (emit), 632-687 (move / feed virus) ------------------
AG_DEV int eat_food(AgCtx &c, int k) {
  int nf = SR(c, AR_NFOOD);
  if (nf == 0) return 0;
  unsigned m = ag_uniu(c.cm[k]);
  if (m < AG_FOOD_MASS) return 0;
  float x = ag_unif(c.cx[k]), y = ag_unif(c.cy[k]);
  float r = radius_of(c, m), fr = radius_of(c, AG_FOOD_MASS);
  auto eaten = [&](int i) -> bool { return can_eat_mass(m, AG_FOOD_MASS) && collides(x, y, r, c.fx[i], c.fy[i], fr); };
  int cnt = wave_sum(nf, [&](int i) { return eaten(i) ? 1 : 0; });
  if (cnt == 0) return 0;
  // order-preserving erase(remove_if(...)) in place
  int kept = wave_compact(nf, [&](int i) { return !eaten(i); }, [&](int i, int rank) {
    float a = c.fx[i], b = c.fy[i], e = c.fvx[i], f = c.fvy[i]; int id = c.fid[i];
    c.fx[rank] = a; c.fy[rank] = b; c.fvx[rank] = e; c.fvy[rank] = f; c.fid[rank] = id;
  });
  // NOTE: the predicate of a later 64-chunk reads entries no earlier chunk has overwritten
  // (rank <= i), and within a chunk all lanes load before any lane stores.
  AG_SERIAL { c.S[AR_NFOOD] = kept; c.cm[k] = clamp_mass(m + (unsigned)(nf - kept) * AG_FOOD_MASS); }
  ag_fence();
  return nf - kept;
}
AG_DEV void maybe_emit_food(AgCtx &c, int p, int n) {
  int cd = PR(c, p, PL_FEED_CD);
  if (cd > 0) cd -= 1;
  if (PR(c, p, PL_ACTION) == 1 && cd == 0) {
    int l = p * c.d.CC; int nf = SR(c, AR_NFOOD), idc = SR(c, AR_IDC);
    float tx = PRF(c, p, PL_TX), ty = PRF(c, p, PL_TY);
    int FC = c.d.FC;
    int made = wave_compact(n, [&](int i) { return c.cm[l + i] >= AG_CELL_MIN_SIZE + AG_FOOD_MASS; }, [&](int i, int rank) {
      int k = l + i; float x = c.cx[k], y = c.cy[k];
      float ddx = tx - x, ddy = ty - y;
      float ax = fabsf(ddx), ay = fabsf(ddy); float n2 = ax * ax, n2b = ay * ay; float nrm = ag_sqrtf(n2 + n2b);
      float dirx = ag_divf(ddx, nrm), diry = ag_divf(ddy, nrm);
      float r = radius_of(c, c.cm[k]);
      float ox = dirx * r, oy = diry * r;
      int f = nf + rank;
      if (f < FC) { c.fx[f] = x + ox; c.fy[f] = y + oy; c.fvx[f] = dirx * AG_FOOD_SPEED; c.fvy[f] = diry * AG_FOOD_SPEED; c.fid[f] = idc + 1 + rank; }
      c.cm[k] = clamp_mass(c.cm[k] - AG_FOOD_MASS);
    });
    int nf2 = nf + made;
    if (nf2 > FC) { flag(c, 2u); nf2 = FC; }
    SW(c, AR_NFOOD, nf2); SW(c, AR_IDC, idc + made);
    cd = 10;
  }
  PW(c, p, PL_FEED_CD, cd);
  ag_fence();
}
AG_DEV void maybe_split(AgCtx &c, int p, int n, int create_limit) {  // R: Engine.hpp:1056-1064, 1095-1107
  int cd = PR(c, p, PL_SPLIT_CD);
  if (cd > 0) cd -= 1;
  if (PR(c, p, PL_ACTION) == 2 && cd == 0) {
    if (create_limit != 0) {
      int l = p * c.d.CC; int idc = SR(c, AR_IDC), nc0 = ag_uni(*c.ncreated);
      float tx = PRF(c, p, PL_TX), ty = PRF(c, p, PL_TY);
      int made = wave_compact(n, [&](int i) { unsigned m = c.cm[l + i]; return m >= AG_CELL_SPLIT_MINIMUM && m >= 2u * AG_CELL_MIN_SIZE; },
                              [&](int i, int rank) { if (create_limit < 0 || rank < create_limit) do_cell_split(c, l + i, nc0 + rank, idc + 1 + rank, tx, ty); });
      if (create_limit > 0 && made > create_limit) made = create_limit;
      if (nc0 + made > c.d.CC) flag(c, 1u);
      AG_SERIAL { c.S[AR_IDC] = idc + made; *c.ncreated = nc0 + made; }
    }
    cd = 30;
  }
  PW(c, p, PL_SPLIT_CD, cd);
  ag_fence();
}
AG_DEV void move_foods(AgCtx &c) {
  int nf = SR(c, AR_NFOOD);
  if (nf == 0) return;
  float dt = c.g.dt; float fr = radius_of(c, AG_FOOD_MASS);
  bool moving = wave_any(nf, [&](int i) { return !(vmag(c.fvx[i], c.fvy[i]) == 0); });
  if (!moving) return;
  int nv = SR(c, AR_NVIR);
  auto advance = [&](int i, float &x, float &y, float &vx, float &vy) { v_decelerate(vx, vy, AG_FOOD_DECEL, dt); float t = vx * dt; x += t; t = vy * dt; y += t; boundary(c, x, y, fr); };
  // does any moving food reach a virus after its move?  (virus radii only grow by being fed)
  bool hits = nv > 0 && wave_any(nf, [&](int i) {
    float x = c.fx[i], y = c.fy[i], vx = c.fvx[i], vy = c.fvy[i];
    if (vmag(vx, vy) == 0) return false;
    advance(i, x, y, vx, vy);
    bool h = false;
    for (int v = 0; v < nv; v++) h = h || collides(x, y, fr, c.vx[v], c.vy[v], radius_of(c, (unsigned)c.vm[v]));
    return h;
  });
  if (!hits) {
    AG_LANES(i, nf) {
      float x = c.fx[i], y = c.fy[i], vx = c.fvx[i], vy = c.fvy[i];
      if (!(vmag(vx, vy) == 0)) { advance(i, x, y, vx, vy); c.fx[i] = x; c.fy[i] = y; c.fvx[i] = vx; c.fvy[i] = vy; }
    }
    ag_fence();
    return;
  }
  AG_SERIAL {  // exact sequential replay incl. swap-pop and virus feeding.  R: Engine.hpp:632-687
    int n = nf, nvir = nv, idc = c.S[AR_IDC];
    for (int i = 0; i < n;) {
      float x = c.fx[i], y = c.fy[i], vx = c.fvx[i], vy = c.fvy[i];
      if (vmag(vx, vy) == 0) { i++; continue; }
      float fvx = vx, fvy = vy;
      advance(i, x, y, vx, vy);
      c.fx[i] = x; c.fy[i] = y; c.fvx[i] = vx; c.fvy[i] = vy;
      bool hit = false;
      int nscan = nvir;
      for (int v = 0; v < nscan; v++) {
        if (collides(x, y, fr, c.vx[v], c.vy[v], radius_of(c, (unsigned)c.vm[v]))) {
          if (c.vh[v] >= AG_FOOD_HITS) {
            c.vh[v] = 0; c.vm[v] = (int)AG_VIRUS_MASS;
            float nx = c.vx[v], ny = c.vy[v];
            float t = fvx * c.g.dt10; nx += t; t = fvy * c.g.dt10; ny += t;
            boundary(c, nx, ny, radius_of(c, AG_VIRUS_MASS));
            idc++;
            if (nvir < c.d.VC) { c.vx[nvir] = nx; c.vy[nvir] = ny; c.vvx[nvir] = fvx; c.vvy[nvir] = fvy; c.vm[nvir] = (int)AG_VIRUS_MASS; c.vh[nvir] = 0; c.vid[nvir] = idc; nvir++; }
            else c.S[AR_FLAGS] |= 4;
          } else { c.vh[v] += 1; c.vm[v] += (int)AG_FOOD_MASS; }
          hit = true; break;
        }
      }
      if (hit) {
        if (n > 1) {
          int j = n - 1;
          float a = c.fx[j], b = c.fy[j], e = c.fvx[j], f = c.fvy[j]; int id = c.fid[j];
          c.fx[j] = c.fx[i]; c.fy[j] = c.fy[i]; c.fvx[j] = c.fvx[i]; c.fvy[j] = c.fvy[i]; c.fid[j] = c.fid[i];
          c.fx[i] = a; c.fy[i] = b; c.fvx[i] = e; c.fvy[i] = f; c.fid[i] = id;
        }
        n--;
      } else i++;
    }
    c.S[AR_NFOOD] = n; c.S[AR_NVIR] = nvir; c.S[AR_IDC] = idc;
  }
  ag_fence();
}

// ---- recombine / decay.  R: Engine.hpp:1160-1179, 550-584; Entities.hpp:183-203 -------------------------
AG_DEV void recombine_cells(AgCtx &c, int p, int n) {
  if (n < 2) return;
  int l = p * c.d.CC; unsigned clock = (unsigned)SR(c, AR_CLOCK);
  int nrec = wave_sum(n, [&](int i) { return clock >= c.cdl[l + i] ? 1 : 0; });
  if (nrec < 2) return;
  bool any = wave_any(n * n, [&](int k) { int a = k / n, b = k - a * n; return a < b && clock >= c.cdl[l + a] && clock >= c.cdl[l + b] && cells_touch(c, l + a, l + b); });
  if (!any) return;  // masses (radii) only grow once a first merge has happened
  AG_SERIAL {
    int m = n;
    for (int i = 0; i < m; i++) {
      if (!(clock >= c.cdl[l + i])) continue;
      for (int j = i + 1; j < m;) {
        if (clock >= c.cdl[l + j] && cells_touch(c, l + i, l + j)) {
          c.cm[l + i] = clamp_mass(c.cm[l + i] + c.cm[l + j]);
          int e = l + m - 1, q = l + j;  // swap(*it2, back()); pop_back()
          c.cx[q] = c.cx[e]; c.cy[q] = c.cy[e]; c.cvx[q] = c.cvx[e]; c.cvy[q] = c.cvy[e]; c.csx[q] = c.csx[e]; c.csy[q] = c.csy[e];
          c.cm[q] = c.cm[e]; c.cid[q] = c.cid[e]; c.cdl[q] = c.cdl[e];
          m--;
        } else j++;
      }
    }
    c.PLS[p * PL_WORDS + PL_NCELLS] = m;
  }
  ag_fence();
}
AG_DEV void decay(AgCtx &c, int p, int n) {
  int elapsed = PR(c, p, PL_ELAPSED);
  if (!(c.g.mass_decay && elapsed % 60 == 0)) return;
  int nt = PR(c, p, PL_NVTICKS);
  if (nt > 0) {
    AG_SERIAL {
      int *vt = c.gvt + p * AG_VT_CAP; int fall = elapsed - AG_ANTI_TEAM_TICKS, w = 0;
      for (int i = 0; i < nt; i++) if (!(vt[i] < fall)) vt[w++] = vt[i];
      c.PLS[p * PL_WORDS + PL_NVTICKS] = w;
      if (w != 0) c.PLS[p * PL_WORDS + PL_ANTI_TEAM] = f2u(c.lut_anti[w - 1 < AG_ANTI_LUT ? w - 1 : AG_ANTI_LUT - 1]);
    }
    ag_fence();
  }
  if (elapsed - PR(c, p, PL_LAST_DECAY) >= 60) {
    int l = p * c.d.CC; double rate = (double)PRF(c, p, PL_ANTI_TEAM);
    AG_LANES(i, n) {
      double nm = (double)c.cm[l + i] * (1 - 0.002 * rate);
      unsigned um = (unsigned)nm;
      c.cm[l + i] = um > AG_CELL_MIN_SIZE ? um : AG_CELL_MIN_SIZE;
    }
    PW(c, p, PL_LAST_DECAY, elapsed);
    ag_fence();
  }
}

// ---- one player's tick.  R: Engine.hpp:495-542 --------------------------------------------------------------
AG_DEV void tick_player(AgCtx &c, int p) {
  int n = PR(c, p, PL_NCELLS);
  int l = p * c.d.CC;
  PW(c, p, PL_ELAPSED, PR(c, p, PL_ELAPSED) + 1);
  // (bots' take_action every 10th tick: not on the HIP path yet -- agarcl_create rejects num_bots > 0)
  ag_fence();
  move_player(c, p, n);
  AG_SERIAL { *c.ncreated = 0; }
  ag_fence();
  int create_limit = AG_PLAYER_CELL_LIMIT - n;
  bool can_eat_virus = n >= AG_PLAYER_CELL_LIMIT;
  if (virus_collisions(c, p, n, create_limit, can_eat_virus)) {
    AG_SERIAL {
      int *P = c.PLS + p * PL_WORDS; int nt = P[PL_NVTICKS];
      if (nt < AG_VT_CAP) { c.gvt[p * AG_VT_CAP + nt] = P[PL_ELAPSED]; P[PL_NVTICKS] = nt + 1; } else c.S[AR_FLAGS] |= 16;
      P[PL_VIRUSES_EATEN] += 1;
    }
    ag_fence();
  }
  int ate = pellets_eat(c, p, n);
  unsigned total = (unsigned)wave_sum(n, [&](int i) { return (int)c.cm[l + i]; });
  AG_SERIAL {
    int *P = c.PLS + p * PL_WORDS;
    P[PL_FOOD_EATEN] += ate;
    if ((unsigned)P[PL_HIGHEST_MASS] < total) P[PL_HIGHEST_MASS] = (int)total;
  }
  // per-cell: auto split (mass >= 22500) then eat ejected food.  R: Engine.hpp:520-525, 592-601
  bool big = wave_any(n, [&](int i) { return c.cm[l + i] >= AG_MAX_MASS; });
  if (big || SR(c, AR_NFOOD) > 0) {
    float tx = PRF(c, p, PL_TX), ty = PRF(c, p, PL_TY);
    for (int ci = 0; ci < n; ci++) {
      int k = l + ci;
      if (big && ag_uniu(c.cm[k]) >= AG_MAX_MASS) {
        AG_SERIAL {
          if (n < AG_PLAYER_CELL_LIMIT) {
            if (c.cm[k] >= AG_CELL_SPLIT_MINIMUM) {
              int nc0 = *c.ncreated, idc = c.S[AR_IDC] + 1;
              if (nc0 >= c.d.CC) c.S[AR_FLAGS] |= 1;
              do_cell_split(c, k, nc0, idc, tx, ty);
              c.S[AR_IDC] = idc; *c.ncreated = nc0 + 1;
            }
          } else c.cm[k] = clamp_mass(AG_NEW_MASS_NO_SPLIT);
        }
        ag_fence();
      }
      int fe = eat_food(c, k);
      if (fe) { AG_SERIAL { c.PLS[p * PL_WORDS + PL_FOOD_EATEN] += fe; } }
    }
  }
  ag_fence();
  create_limit -= ag_uni(*c.ncreated);
  maybe_emit_food(c, p, n);
  maybe_split(c, p, n, create_limit);
  // add created cells.  R: core/Player.hpp:195-201
  int ncr = ag_uni(*c.ncreated);
  if (ncr > 0) {
    if (ncr > c.d.CC) ncr = c.d.CC;
    int room = c.d.CC - n;
    if (ncr > room) { flag(c, 1u); ncr = room; }
    AG_LANES(j, ncr) {
      int k = l + n + j;
      c.cx[k] = c.nx[j]; c.cy[k] = c.ny[j]; c.cvx[k] = c.nvx[j]; c.cvy[k] = c.nvy[j]; c.csx[k] = c.nsx[j]; c.csy[k] = c.nsy[j];
      c.cm[k] = c.nm[j]; c.cid[k] = c.nid[j]; c.cdl[k] = c.ndl[j];
    }
    n += ncr;
    PW(c, p, PL_NCELLS, n);
    ag_fence();
  }
  recombine_cells(c, p, n);
  n = PR(c, p, PL_NCELLS);
  decay(c, p, n);
}

// ---- end-of-tick bookkeeping.  R: Engine.hpp:1002-1009, 1253-1260, 150-200 ------------------------------------
AG_DEV void remove_pellets(AgCtx &c) {
  int ne = SR(c, AR_NEVP);
  if (ne == 0) return;
  AG_SERIAL {
    int n = c.S[AR_NPEL]; int lim = ne < AG_EV_CAP ? ne : AG_EV_CAP;
    for (int e = 0; e < lim; e++) {
      int idx = c.evp[e];
      if (n > 1 && idx < n - 1) {  // std::swap(p[idx], p.back()): the stale value parked at the back is popped
        int b = n - 1;
        float tx = c.px[idx], ty = c.py[idx]; int tid = c.gpid[idx];
        c.px[idx] = c.px[b]; c.py[idx] = c.py[b]; c.gpid[idx] = c.gpid[b];
        c.px[b] = tx; c.py[b] = ty; c.gpid[b] = tid;
      }
      if (n >= 1) n--;
    }
    c.S[AR_NPEL] = n;
  }
  ag_fence();
}
AG_DEV void remove_viruses(AgCtx &c) {
  int ne = SR(c, AR_NEVV);
  if (ne == 0) return;
  AG_SERIAL {
    int n = c.S[AR_NVIR]; int lim = ne < AG_EVV_CAP ? ne : AG_EVV_CAP;
    for (int e = 0; e < lim; e++) {
      int idx = c.evv[e];
      if (n > 1 && idx < n - 1) {
        int b = n - 1;
        float a1 = c.vx[idx], a2 = c.vy[idx], a3 = c.vvx[idx], a4 = c.vvy[idx]; int a5 = c.vm[idx], a6 = c.vh[idx], a7 = c.vid[idx];
        c.vx[idx] = c.vx[b]; c.vy[idx] = c.vy[b]; c.vvx[idx] = c.vvx[b]; c.vvy[idx] = c.vvy[b]; c.vm[idx] = c.vm[b]; c.vh[idx] = c.vh[b]; c.vid[idx] = c.vid[b];
        c.vx[b] = a1; c.vy[b] = a2; c.vvx[b] = a3; c.vvy[b] = a4; c.vm[b] = a5; c.vh[b] = a6; c.vid[b] = a7;
      }
      if (n >= 1) n--;
    }
    c.S[AR_NVIR] = n;
  }
  ag_fence();
}
// sort(player.cells) by id (Engine.hpp:157); ids are unique so the result is the sorted order.
AG_DEV void sort_cells_by_id(AgCtx &c, int p) {
  int n = PR(c, p, PL_NCELLS);
  if (n < 2) return;
  int l = p * c.d.CC;
  bool unsorted = wave_any(n - 1, [&](int i) { return c.cid[l + i] > c.cid[l + i + 1]; });
  if (!unsorted) return;
  AG_LANES(i, n) {  // rank sort through the created-cell buffer
    int id = c.cid[l + i], r = 0;
    for (int j = 0; j < n; j++) r += c.cid[l + j] < id ? 1 : 0;
    c.nx[r] = c.cx[l + i]; c.ny[r] = c.cy[l + i]; c.nvx[r] = c.cvx[l + i]; c.nvy[r] = c.cvy[l + i]; c.nsx[r] = c.csx[l + i]; c.nsy[r] = c.csy[l + i];
    c.nm[r] = c.cm[l + i]; c.nid[r] = id; c.ndl[r] = c.cdl[l + i];
  }
  ag_fence();
  AG_LANES(i, n) {
    c.cx[l + i] = c.nx[i]; c.cy[l + i] = c.ny[i]; c.cvx[l + i] = c.nvx[i]; c.cvy[l + i] = c.nvy[i]; c.csx[l + i] = c.nsx[i]; c.csy[l + i] = c.nsy[i];
    c.cm[l + i] = c.nm[i]; c.cid[l + i] = c.nid[i]; c.cdl[l + i] = c.ndl[i];
  }
  ag_fence();
}

// ---- Engine::tick.  R: Engine.hpp:208-240 --------------------------------------------------------------------
AG_DEV void arena_tick(AgCtx &c) {
  SW(c, AR_NEVP, 0); SW(c, AR_NEVV, 0);
  ag_fence();
  for (int k = 0; k < c.d.P; k++) {
    int p = SR(c, AR_ORDER0 + k);
    if (PR(c, p, PL_NCELLS) > 0) tick_player(c, p);
  }
  remove_pellets(c);
  remove_viruses(c);
  for (int k = 0; k < c.d.P; k++) sort_cells_by_id(c, SR(c, AR_ORDER0 + k));
  // PrecisionCollisionDetection::solve: with one player every strip scan breaks on an own cell
  // (utils/collision_detection.hpp:51) => no eats.  P > 1 is rejected at create time for now.
  move_foods(c);
  int ticks = SR(c, AR_TICKS);
  if (c.g.regen && ticks % 120 == 0) {
    add_pellets(c, c.g.target_pellets - SR(c, AR_NPEL));
    add_viruses(c, c.g.target_viruses - SR(c, AR_NVIR));
  }
  SW(c, AR_TICKS, ticks + 1); SW(c, AR_CLOCK, SR(c, AR_CLOCK) + 1);
  ag_fence();
}

// ---- BaseEnvironment.  R: environment/envs/BaseEnvironment.hpp:89-204 -----------------------------------------
AG_DEV unsigned player_mass(const AgCtx &c, int p) { int n = PR(c, p, PL_NCELLS), l = p * c.d.CC; unsigned t = 0; for (int i = 0; i < n; i++) t += ag_uniu(c.cm[l + i]); return t; }
AG_DEV void take_action(AgCtx &c, int p, float dx, float dy, int action) {  // R: :162-176, Player.hpp:102-126
  int n = PR(c, p, PL_NCELLS);
  if (n == 0) return;
  int l = p * c.d.CC; float sx = 0.0f, sy = 0.0f; unsigned tm = 0;
  for (int i = 0; i < n; i++) { unsigned m = ag_uniu(c.cm[l + i]); float fm = (float)m; float t = ag_unif(c.cx[l + i]) * fm; sx += t; t = ag_unif(c.cy[l + i]) * fm; sy += t; tm += m; }
  float px = ag_divf(sx, (float)tm), py = ag_divf(sy, (float)tm);
  float ox = dx * 10.0f, oy = dy * 10.0f;
  AG_SERIAL { int *P = c.PLS + p * PL_WORDS; P[PL_ACTION] = action; P[PL_TX] = f2u(px + ox); P[PL_TY] = f2u(py + oy); }
  ag_fence();
}
AG_DEV void respawn_dead(AgCtx &c) { for (int k = 0; k < c.d.P; k++) { int p = SR(c, AR_ORDER0 + k); if (PR(c, p, PL_NCELLS) == 0) respawn(c, p); } }

AG_DEV void env_step(AgCtx &c, const AgState &s, int ticks, bool with_env) {
  int na = c.d.n_agents;
  unsigned before[AG_MAX_PLAYERS];
  if (with_env) {
    for (int i = 0; i < na; i++) {
      size_t o = (size_t)c.arena * na + i;
      if (s.act) take_action(c, i, s.act_dxdy[2 * o], s.act_dxdy[2 * o + 1], s.act[o]);
    }
    SW(c, AR_RESPAWNED, 0);
    for (int i = 0; i < na; i++) { before[i] = player_mass(c, i); if (c.g.mode == 3 && before[i] >= 23000u) SW(c, AR_DONE, 1); }
  }
  for (int t = 0; t < ticks; t++) arena_tick(c);
  if (with_env) {
    if (c.g.mode == 0) respawn_dead(c);
    for (int i = 0; i < na; i++) {
      unsigned m = player_mass(c, i);
      if (c.g.mode == 3 && m >= 23000u) SW(c, AR_DONE, 1);
      double r = (double)m;
      if (c.g.reward_type) { float b = (float)before[i]; float sub = b - (float)(SR(c, AR_RESPAWNED) ? c.g.c_death : 0); r -= (double)sub; }
      size_t o = (size_t)c.arena * na + i;
      ag_fence();
      int done = SR(c, AR_DONE);
      AG_SERIAL { s.rewards[o] = r; s.masses[o] = (int)m; s.dones[o] = (uint8_t)(i == 0 ? done : 0); }
    }
  }
}

AG_DEV void env_reset(AgCtx &c, int reset_ids) {  // R: BaseEnvironment.hpp:179-204, Engine.hpp:98-117
  if (reset_ids) SW(c, AR_IDC, 1);
  SW(c, AR_NPEL, 0); SW(c, AR_NVIR, 0); SW(c, AR_NFOOD, 0); SW(c, AR_TICKS, 0); SW(c, AR_FLAGS, 0);
  SW(c, AR_NEVP, 0); SW(c, AR_NEVV, 0); SW(c, AR_DONE, 0); SW(c, AR_RESPAWNED, 0);
  ag_fence();
  if (c.g.squared) create_squared_pellets(c); else add_pellets(c, c.g.target_pellets);
  add_viruses(c, c.g.target_viruses);
  for (int i = 0; i < c.d.P; i++) {  // add_player (Engine.hpp:70-83): slot i gets pid next_pid++
    AG_SERIAL {
      int *P = c.PLS + i * PL_WORDS;
      int pid = c.S[AR_NEXT_PID]; c.S[AR_NEXT_PID] = (pid + 1) & 0xFFFF;
      P[PL_PID] = pid; P[PL_KIND] = 0; P[PL_ACTION] = 0; P[PL_TX] = 0; P[PL_TY] = 0;
      P[PL_FOOD_EATEN] = 0; P[PL_HIGHEST_MASS] = (int)AG_CELL_MIN_SIZE; P[PL_CELLS_EATEN] = 0; P[PL_VIRUSES_EATEN] = 0;
      c.S[AR_ORDER0 + i] = i;
    }
    ag_fence();
    respawn(c, i);
  }
}
