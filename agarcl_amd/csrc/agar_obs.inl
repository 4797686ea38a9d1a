// Grid observation kernel: GridObservation::add_frame of the reference
// (/root/reference/environment/envs/GridEnvironment.hpp:91-123, helpers :188-296) for every (arena, agent):
// an egocentric int32 tensor [C][G][G], C = 1 (out-of-bounds) + 2 pellets + 2 viruses + 1 own cells + 2 others.
//
// One 256-thread workgroup per (arena, agent).  Phase 1 streams the whole tensor out once (coalesced 4-byte
// stores: channel 0 = out-of-bounds mask computed per grid cell, the rest zero); phase 2 scatters the entities
// (a few hundred 4-byte writes / integer atomics per agent).  The tensor (512 KiB per agent at G = 128) is the
// HBM traffic of this kernel; entity state is read straight from the engine's SoA arrays.
//
// Order-dependent channel ops of the reference ("at least one": last writer in vector order wins) are kept exact:
// pellets all carry mass 1 (order-free), viruses / cells are few and are replayed in order by one thread.
#pragma once
#include "agar_types.h"
#include <limits.h>
#ifdef AGAR_CPU_EMU
#include <vector>
#endif

#ifdef AGAR_CPU_EMU
#define OBS_DEV static inline
#define OBS_FOR(k, n) for (int k = 0; k < (n); ++k)
#define OBS_BARRIER()
#define OBS_BARRIER_LDS()
#define OBS_THREAD0 true
#define OBS_ATOMIC_ADD(p, v) (*(p) += (v))
#else
#define OBS_DEV __device__ __forceinline__
#define OBS_FOR(k, n) for (int k = (int)threadIdx.x; k < (n); k += (int)blockDim.x)
#define OBS_BARRIER() __syncthreads()
// a barrier for data exchanged through LDS only: __syncthreads() also waits for every global store / atomic of the wave to be acknowledged
// (vmcnt counts stores on gfx9) -- a round trip per barrier behind a phase that scattered into the frame.  The one barrier that must be a
// full one is between undoing the previous frame's words and scattering the new ones (two threads may write the same word).
#define OBS_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define OBS_THREAD0 (threadIdx.x == 0)
#define OBS_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#endif

struct AgObsCfg { int G, cells, others, viruses, pellets; };
// Incremental clearing (agarcl_grid_obs on_device = 2): the frame buffer still holds this env's previous observation, so
// instead of streaming zeros over channels 1.. (512 KB per 128 x 128 frame, three quarters of the observation's time) the
// kernel zeroes exactly the words it wrote last time -- it keeps their offsets in a per-frame undo list -- and records the
// new ones.  undo == nullptr: off.  clear: the list of the previous call is valid for this buffer.
// sig: with the list, the out-of-bounds channel's signature of the previous call -- per frame G bytes "grid row i is inside the arena in x"
// followed by G bytes "column j is inside in y" (the mask is their outer product): channel 0 is 64 KB of the 128 x 128 frame and changes
// only when the view window moves across an arena wall, so a persistent tensor gets only the rows / columns whose byte changed.
struct AgObsUndo { int32_t *list; int32_t *count; int cap; int clear; uint8_t *sig; };
#define OBS_ECAP 1024  // viruses + cells of all players staged per frame (16 players x 32 cell slots = 512 cells at most; entities
                       // beyond the cap -- more than ~500 viruses -- are not drawn)
#define OBS_PELLET_WORDS 2   // words a pellet writes (channels "at least one" and "count")
#define OBS_ENTITY_WORDS 2   // words a staged entity writes at most (virus: last mass + sum; other cell: min + max; own cell: 1)
#define OBS_UNDO_CAP(PC) (OBS_PELLET_WORDS * (PC) + OBS_ENTITY_WORDS * OBS_ECAP)
// a frame whose recorded writes overflowed the list (cannot happen while the two constants above describe the kernel; kept as a net):
// the stored count is -1 and the next call zero-fills that frame instead of undoing it
#define OBS_UNDO_OVERFLOW (-1)

OBS_DEV int obs_channels(const AgObsCfg &o) { return 1 + o.cells + 2 * o.others + 2 * o.viruses + 2 * o.pellets; }

OBS_DEV float obs_smaxf(float a, float b) { return (a < b) ? b : a; }
OBS_DEV float obs_sminf(float a, float b) { return (b < a) ? b : a; }
OBS_DEV int obs_f2i(float f) { if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT_MIN; return (int)f; }

// mass-weighted centroid and total mass of player `p` (Player::x/y/mass, core/Player.hpp:102-126): sequential fp32
// sums in cell order, evaluated redundantly by every thread
OBS_DEV void obs_player(const AgState *gs, int arena, int p, float &px, float &py, unsigned &mass) {
  const int ag_ts_lg = gs->d.ts_lg;
  const int32_t *pl = AG_PL_PTR(gs, arena, p);
  const uint32_t *C = AG_CELLS_PTR(gs, arena, p);
  int n = pl[AG_TW(PL_NCELLS)]; float sx = 0.0f, sy = 0.0f; unsigned tm = 0;
  for (int i = 0; i < n; i++) {
    union { uint32_t u; float f; } x, y; x.u = C[AG_CELL_W(CF_X, i)]; y.u = C[AG_CELL_W(CF_Y, i)];
    unsigned m = C[AG_CELL_W(CF_M, i)]; float fm = (float)m;
    float t = x.f * fm; sx += t; t = y.f * fm; sy += t; tm += m;
  }
  px = sx / (float)tm; py = sy / (float)tm; mass = tm;
}

// zero_fill: also write the zeros of channels 1.. (the host emulation and odd grid sizes); on the GPU the bulk zero fill
// is a separate streaming kernel (k_grid_zero: plain 16-byte stores at the rate of a memset, 6.8 TB/s measured) and
// this function only writes channel 0 and scatters the entities.
OBS_DEV void grid_obs_agent(const AgState *gs, int arena, int agent, AgObsCfg o, int32_t *out, bool zero_fill = true, AgObsUndo un = AgObsUndo{nullptr, nullptr, 0, 0, nullptr}, int frame = 0) {
  const int G = o.G, GG = G * G, C = obs_channels(o);
#ifndef AGAR_CPU_EMU
  __shared__ int un_cnt;
  int32_t *ul = un.list ? un.list + (size_t)frame * un.cap : nullptr;
  // Everything this frame reads has an address known from (arena, agent, thread): the counts, the first 512 undo entries, the first 1024
  // pellets, the viruses, the agent's cell slots and the mask signature are requested HERE, together -- a frame used to make about ten
  // dependent round trips (count, then list; cell count, then cells; pellet count, then pellets; ...), and with two rounds of workgroups
  // per CU that chain was the kernel's time.  Slots behind a count hold valid memory (capacities), so nothing waits for a count to load.
  const int tid = (int)threadIdx.x, ag_ts_lg0 = gs->d.ts_lg;
  const bool undo = ul && un.clear;
  int pf_nprev = 0, pf_ul0 = 0, pf_ul1 = 0;
  if (undo) { pf_nprev = un.count[frame]; if (tid < un.cap) pf_ul0 = ul[tid]; if (tid + 256 < un.cap) pf_ul1 = ul[tid + 256]; }
  int pf_np, pf_nv, pf_nown;
  { const int ag_ts_lg = ag_ts_lg0; const int32_t *arw0 = AG_AR_PTR(gs, arena); pf_np = arw0[AG_TW(AR_NPEL)]; pf_nv = arw0[AG_TW(AR_NVIR)]; pf_nown = AG_PL_PTR(gs, arena, agent)[AG_TW(PL_NCELLS)]; }
  const int PCp = gs->d.PC;
  const float *pxy0 = gs->pel_xy + (size_t)arena * PCp * 2;
  float pf_px[4], pf_py[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { const int k = tid + 256 * j; pf_px[j] = 0.0f; pf_py[j] = 0.0f; if (o.pellets && k < PCp) { pf_px[j] = pxy0[2 * k]; pf_py[j] = pxy0[2 * k + 1]; } }
  float pf_vx = 0.0f, pf_vy = 0.0f; int pf_vm = 0;
  { const size_t vo0 = (size_t)arena * gs->d.VC; if (o.viruses && tid < gs->d.VC) { pf_vx = gs->vir_x[vo0 + tid]; pf_vy = gs->vir_y[vo0 + tid]; pf_vm = gs->vir_mass[vo0 + tid]; } }
  uint8_t pf_sx = 0, pf_sy = 0;
  if (undo && un.sig && tid < o.G) { const uint8_t *sg0 = un.sig + (size_t)frame * 2 * o.G; pf_sx = sg0[tid]; pf_sy = sg0[o.G + tid]; }
  __shared__ float own_x[AG_CC], own_y[AG_CC]; __shared__ unsigned own_m[AG_CC];
  if (tid < AG_CC) { const int ag_ts_lg = ag_ts_lg0; const uint32_t *Co = AG_CELLS_PTR(gs, arena, agent); union { uint32_t u; float f; } x, y; x.u = Co[AG_CELL_W(CF_X, tid)]; y.u = Co[AG_CELL_W(CF_Y, tid)]; own_x[tid] = x.f; own_y[tid] = y.f; own_m[tid] = Co[AG_CELL_W(CF_M, tid)]; }
  if (ul) {
    if (un.clear) {   // undo the previous observation's scattered writes
      const int n_prev = pf_nprev;   // (block-uniform)
      if (n_prev == OBS_UNDO_OVERFLOW) zero_fill = true;
      else { if (tid < n_prev) out[pf_ul0] = 0; if (tid + 256 < n_prev) out[pf_ul1] = 0; for (int k = tid + 512; k < n_prev; k += 256) out[ul[k]] = 0; }
    }
    if (threadIdx.x == 0) un_cnt = 0;
  }
  auto rec = [&](int off) { if (ul) { const int p_ = atomicAdd(&un_cnt, 1); if (p_ < un.cap) ul[p_] = off; } };
  OBS_BARRIER_LDS();   // the agent's cells are in LDS
  // (Player::x / y / mass: sequential fp32 sums in cell order, obs_player's arithmetic on the staged slots)
  float px, py; unsigned mass;
  { float sx = 0.0f, sy = 0.0f; unsigned tm = 0;
    for (int i = 0; i < pf_nown; i++) { const unsigned m = own_m[i]; const float fm = (float)m; float t = own_x[i] * fm; sx += t; t = own_y[i] * fm; sy += t; tm += m; }
    px = sx / (float)tm; py = sy / (float)tm; mass = tm; }
#else
  (void)un; (void)frame;
  auto rec = [&](int) {};
  float px, py; unsigned mass;
  obs_player(gs, arena, agent, px, py, mass);
#endif
  float view = obs_smaxf(obs_sminf((float)(2u * mass), 300.0f), 100.0f);  // clamp<float>(2*mass, 100, 300), :251-254
  float centering = (float)(G / 2.0);
  float W = gs->g.W;
  // phase 1: channel 0 = out-of-bounds mask (:235-248), all other channels 0.  The mask is separable -- a grid cell is
  // inside the arena iff its x is (a function of i only) and its y is (a function of j only) -- so the two IEEE
  // divisions per cell of the reference's formula are evaluated once per row / column, not once per cell.
#ifdef AGAR_CPU_EMU
  std::vector<uint8_t> inx_v((size_t)G), iny_v((size_t)G); uint8_t *inx = inx_v.data(), *iny = iny_v.data();
#else
  __shared__ uint8_t inx[1024], iny[1024];
#endif
  OBS_FOR(i, G) {
    float d = (float)i - centering; float t = d * view; t = t / (float)G;
    float lx = px + t, ly = py + t;
    inx[i] = (0 <= lx && lx < W) ? 1 : 0; iny[i] = (0 <= ly && ly < W) ? 1 : 0;
  }
  OBS_BARRIER_LDS();
  auto oob = [&](int k) -> int32_t { int i = k / G, j = k - i * G; return (inx[i] & iny[j]) ? 0 : -1; };
#ifndef AGAR_CPU_EMU
  // persistent tensor (the undo list of the previous call is valid): channel 0 still holds the previous mask.  Only the rows whose x byte and
  // the columns whose y byte changed are stored again -- none at all while the window stays clear of the walls or does not move
  bool mask_done = false;
  if (ul && un.sig) {
    __shared__ int n_rows, n_cols; __shared__ uint16_t ch_rows[1024], ch_cols[1024];
    uint8_t *sg = un.sig + (size_t)frame * 2 * G;
    const bool incremental = un.clear && !zero_fill;   // (block-uniform; zero_fill here means "the list overflowed": everything is rewritten)
    if (incremental) {
      if (threadIdx.x == 0) { n_rows = 0; n_cols = 0; }
      OBS_BARRIER_LDS();
      if (tid < G) {   // (G <= 256 rows / columns: the prefetched bytes; larger grids read theirs here)
        if (pf_sx != inx[tid]) ch_rows[atomicAdd(&n_rows, 1)] = (uint16_t)tid;
        if (pf_sy != iny[tid]) ch_cols[atomicAdd(&n_cols, 1)] = (uint16_t)tid;
      }
      for (int i = tid + 256; i < G; i += 256) {
        if (sg[i] != inx[i]) ch_rows[atomicAdd(&n_rows, 1)] = (uint16_t)i;
        if (sg[G + i] != iny[i]) ch_cols[atomicAdd(&n_cols, 1)] = (uint16_t)i;
      }
      OBS_BARRIER_LDS();
      const int nr = n_rows, nc = n_cols;
      OBS_FOR(k, nr * G) { const int i = ch_rows[k / G], j = k % G; out[i * G + j] = (inx[i] & iny[j]) ? 0 : -1; }
      OBS_FOR(k, nc * G) { const int j = ch_cols[k / G], i = k % G; out[i * G + j] = (inx[i] & iny[j]) ? 0 : -1; }
      mask_done = true;
    }
    OBS_FOR(i, G) { sg[i] = inx[i]; sg[G + i] = iny[i]; }
  }
  if (mask_done) { }
  else if ((GG & 3) == 0 && (((size_t)out) & 15) == 0) {
    // 16 bytes per lane per store (1 KiB per wave-instruction), streaming (non-temporal): the tensor is written once
    // and read by somebody else
    typedef int32_t v4 __attribute__((ext_vector_type(4)));
    v4 *o4 = (v4 *)out;
    OBS_FOR(q, GG / 4) { v4 v; v.x = oob(4 * q); v.y = oob(4 * q + 1); v.z = oob(4 * q + 2); v.w = oob(4 * q + 3); __builtin_nontemporal_store(v, &o4[q]); }
    const v4 z = {0, 0, 0, 0};
    if (zero_fill) OBS_FOR(q, (C - 1) * (GG / 4)) __builtin_nontemporal_store(z, &o4[GG / 4 + q]);
  } else
#endif
  {
    OBS_FOR(k, GG) out[k] = oob(k);
    if (zero_fill) OBS_FOR(k, (C - 1) * GG) out[GG + k] = 0;
  }
  OBS_BARRIER();
  // world -> grid (:258-268)
  auto w2g = [&](float ex, float ey, int &gx, int &gy) {
    float ddx = ex - px, ddy = ey - py;
    float t1 = (float)G * ddx; t1 = t1 / view; gx = obs_f2i(t1 + centering);
    float t2 = (float)G * ddy; t2 = t2 / view; gy = obs_f2i(t2 + centering);
    return 0 <= gx && gx < G && 0 <= gy && gy < G;
  };
  const int ag_ts_lg = gs->d.ts_lg;
  int ch = 0;
  if (o.pellets) {  // ch+1: "at least one" (= mass 1), ch+2: count
    int32_t *a1 = out + (size_t)(ch + 1) * GG, *a2 = out + (size_t)(ch + 2) * GG;
    // (r05: nine pellets in ten lie outside the window -- a comparison decides that before the two correctly rounded divisions of the reference's
    // formula: |d| > view (1/2 + 2/G) maps beyond the grid by more than a cell on either side, whatever the rounding)
    const float p_lim = view * (0.5f + 2.0f / (float)G) * 1.001f;
#ifdef AG_GRID_NOCULL   // (measurement builds)
    const bool p_cull = false;
#else
    const bool p_cull = true;
#endif
    auto pellet = [&](float qx, float qy) { if (p_cull && !(fabsf(qx - px) <= p_lim && fabsf(qy - py) <= p_lim)) return; int gx, gy; if (w2g(qx, qy, gx, gy)) { a1[gx * G + gy] = 1; OBS_ATOMIC_ADD(&a2[gx * G + gy], 1); rec((ch + 1) * GG + gx * G + gy); rec((ch + 2) * GG + gx * G + gy); } };
#ifndef AGAR_CPU_EMU
    const int np = pf_np;
#pragma unroll
    for (int j = 0; j < 4; j++) { if (tid + 256 * j < np) pellet(pf_px[j], pf_py[j]); }
    for (int k = tid + 1024; k < np; k += 256) pellet(pxy0[2 * k], pxy0[2 * k + 1]);
#else
    const float *pxy = gs->pel_xy + (size_t)arena * gs->d.PC * 2; int np = AG_AR_PTR(gs, arena)[AG_TW(AR_NPEL)];
    OBS_FOR(k, np) pellet(pxy[2 * k], pxy[2 * k + 1]);
#endif
    ch += 2;
  }
  // Viruses and cells: the reference applies them one after the other (GridEnvironment.hpp:222-229: "at least one"
  // keeps the LAST writer, the other channel ops are sum / min / max).  Here every entity is a thread: the entities are
  // staged in LDS (grid cell or -1, mass, kind), and for each grid cell the FIRST entity of a kind that maps to it writes
  // the combined value of all of them with a plain store -- no read-modify-write chain, same result.
#ifdef AGAR_CPU_EMU
  std::vector<int32_t> e_idx_v(OBS_ECAP), e_mass_v(OBS_ECAP), e_kind_v(OBS_ECAP); int32_t *e_idx = e_idx_v.data(), *e_mass = e_mass_v.data(), *e_kind = e_kind_v.data();
#else
  __shared__ int32_t e_idx[OBS_ECAP], e_mass[OBS_ECAP], e_kind[OBS_ECAP];
#endif
  const int32_t *arw = AG_AR_PTR(gs, arena);
#ifndef AGAR_CPU_EMU
  const int P = gs->d.P, nv = o.viruses ? pf_nv : 0;
#else
  const int P = gs->d.P, nv = o.viruses ? arw[AG_TW(AR_NVIR)] : 0;
#endif
  int c2 = o.pellets ? 2 : 0;
  int32_t *v1 = out + (size_t)(c2 + 1) * GG, *v2 = out + (size_t)(c2 + 2) * GG; if (o.viruses) c2 += 2;
  int32_t *cown = out + (size_t)(c2 + 1) * GG; if (o.cells) c2 += 1;
  int32_t *omin = out + (size_t)(c2 + 1) * GG, *omax = out + (size_t)(c2 + 2) * GG;
  int E = nv;
  {  // viruses: kind 0, in vector order
    size_t vo = (size_t)arena * gs->d.VC;
#ifndef AGAR_CPU_EMU
    if (tid < nv) { int gx, gy; bool in = w2g(pf_vx, pf_vy, gx, gy); e_idx[tid] = in ? gx * G + gy : -1; e_mass[tid] = pf_vm; e_kind[tid] = 0; }
    for (int k = tid + 256; k < nv; k += 256) { if (k < OBS_ECAP) { int gx, gy; bool in = w2g(gs->vir_x[vo + k], gs->vir_y[vo + k], gx, gy); e_idx[k] = in ? gx * G + gy : -1; e_mass[k] = gs->vir_mass[vo + k]; e_kind[k] = 0; } }
#else
    OBS_FOR(k, nv) { if (k < OBS_ECAP) { int gx, gy; bool in = w2g(gs->vir_x[vo + k], gs->vir_y[vo + k], gx, gy); e_idx[k] = in ? gx * G + gy : -1; e_mass[k] = gs->vir_mass[vo + k]; e_kind[k] = 0; } }
#endif
  }
  for (int k = -1; k < P; k++) {  // own cells (kind 1), then the other players in the engine's iteration order (kind 2)
    if (k < 0 ? !o.cells : !o.others) continue;
    const int p = k < 0 ? agent : arw[AG_TW(AR_ORDER0 + k)];
    if (k >= 0 && p == agent) continue;
#ifndef AGAR_CPU_EMU
    if (k < 0) {   // the agent's own cells are staged already
      const int n = pf_nown;
      if (tid < n && E + tid < OBS_ECAP) { int gx, gy; bool in = w2g(own_x[tid], own_y[tid], gx, gy); e_idx[E + tid] = in ? gx * G + gy : -1; e_mass[E + tid] = (int32_t)own_m[tid]; e_kind[E + tid] = 1; }
      E += n;
      continue;
    }
#endif
    const int32_t *pl = AG_PL_PTR(gs, arena, p);
    const uint32_t *Cc = AG_CELLS_PTR(gs, arena, p);
    const int n = pl[AG_TW(PL_NCELLS)];
    OBS_FOR(i, n) {
      if (E + i < OBS_ECAP) {
        union { uint32_t u; float f; } x, y; x.u = Cc[AG_CELL_W(CF_X, i)]; y.u = Cc[AG_CELL_W(CF_Y, i)];
        int gx, gy; bool in = w2g(x.f, y.f, gx, gy);
        e_idx[E + i] = in ? gx * G + gy : -1; e_mass[E + i] = (int32_t)Cc[AG_CELL_W(CF_M, i)]; e_kind[E + i] = k < 0 ? 1 : 2;
      }
    }
    E += n;
  }
  if (E > OBS_ECAP) E = OBS_ECAP;
  OBS_BARRIER_LDS();
  OBS_FOR(k, E) {
    const int idx = e_idx[k], kind = e_kind[k];
    if (idx < 0) continue;
    bool first = true; int last = k, sum = 0, mn = 0x7fffffff, mx = 0;
    // (branch-free and unrolled: the LDS reads of four entities are in flight together -- with a `continue` in it the loop was one dependent LDS
    // round trip per entity on the one wavefront that holds the entities)
#if !defined(AGAR_CPU_EMU) && !defined(AG_GRID_NOUNROLL)
#pragma unroll 4
#endif
    for (int j = 0; j < E; j++) {
      const bool same = e_idx[j] == idx && e_kind[j] == kind; const int m = e_mass[j];
      first = first && !(same && j < k);
      last = (same && j > last) ? j : last;
      sum += same ? m : 0; mn = (same && m < mn) ? m : mn; mx = (same && m > mx) ? m : mx;
    }
    if (!first) continue;
    if (kind == 0) { v1[idx] = e_mass[last]; v2[idx] = sum; rec((int)(v1 - out) + idx); rec((int)(v2 - out) + idx); }
    else if (kind == 1) { cown[idx] = sum; rec((int)(cown - out) + idx); }
    else { omin[idx] = mn; omax[idx] = mx; rec((int)(omin - out) + idx); rec((int)(omax - out) + idx); }
  }
#ifndef AGAR_CPU_EMU
  if (ul) { OBS_BARRIER_LDS(); if (threadIdx.x == 0) un.count[frame] = un_cnt <= un.cap ? un_cnt : OBS_UNDO_OVERFLOW; }
#endif
}
