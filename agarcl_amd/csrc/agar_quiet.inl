// Lean front kernel body: SIMT style, a group of QG lanes per arena (template parameter: 1 ... 32, chosen from the arena count),
// no LDS, nothing resident -> tiny register footprint, 64 / QG arenas per wavefront.
//
// In RL rollouts almost every env step of a single-player arena is "quiet" (see quiet_ticks in agar_core.inl):
// the whole step touches ~300 bytes of state and now and then scans the pellets.  The quiet tick itself is scalar
// work per arena; on a machine whose scalar unit has no fp32 it has to run on the vector ALU, where a wave-uniform
// formulation wastes 63 of 64 lanes and -- at 4096 arenas, 4 waves per SIMD -- is bound by VALU issue.  Here every
// lane of a group carries its arena's scalars redundantly (plain per-lane code, identical fp32 sequence), a pellet
// pass is a wave-level operation (all 64 lanes scan the asking arena's pellets), and only the group's first lane
// stores.  4096 arenas x 16 lanes = 1024 wavefronts = one per SIMD: latency-, not issue-bound; bigger batches take
// fewer lanes per arena so that the wavefront count stays there (agarcl_create).
//
// k_quiet runs first and does the env prologue (take_actions), as many quiet ticks as it can and -- if that was
// the whole step -- the env epilogue (rewards / dones / packed results).  What it could not finish is left to the
// general kernel k_step, which runs right after it on the same stream: qinfo[arena] = { ticks already done (or
// -1: nothing, not even the prologue), agent mass before the step }.  Arenas that are finished make k_step's wave
// exit on its first load.  Single-player envs only (P == 1); with several players k_quiet is never launched.
#pragma once
#ifndef AG_QG
#define AG_QG 16   // default lanes per arena (the host emulation's and the tests' choice; the kernels take QG as a parameter)
#endif

// Pellets streamed from HBM / L2.  A pass is a wave-level operation: for every lane group that asks for one, ALL 64
// lanes read that arena's pellets (NS x 512 B per wave-instruction, every load issued before the first use: one
// round trip per pass), accumulate lane-private partial results and combine them with DPP reductions.
// word w of one arena's block of a tile-transposed array
template <class T> struct TileWords { AG_GLOBAL T *p; int ag_ts_lg; AG_MEM AG_GLOBAL T &operator[](int w) const { return p[AG_TW(w)]; } };

template <int NS, int QG = AG_QG> struct GrpPel {
  AG_GLOBAL float *xy; AG_GLOBAL int32_t *id; int sub;
#ifdef AGAR_CPU_EMU
  AG_MEM bool lead() const { return true; }
  AG_MEM int pass_cost() const { return 1; }
  AG_MEM unsigned cap() const { return (unsigned)(NS * 64); }
  AG_MEM bool any(bool p) const { return p; }
  template <bool AV> AG_MEM PelScan2 scan2(bool need, const PelQuery2 &k) {
    PelScan2 out{0, 0xffffffffu, 0xffffffffu};
    if (!need) return out;
    for (int i = 0; i < NS * 64; i++) pel_accumulate2<AV>(k, xy[2 * i], xy[2 * i + 1], (unsigned)i, out.cnt2, out.key1, out.key2);
    return out;
  }
  template <bool AV> AG_MEM PelScan scan(bool need, const PelQuery &k) {
    PelScan out{3.0e38f, 3.0e38f, 0, 0, -1, -1, 0.0f, 0.0f};
    if (!need) return out;
    unsigned dmin = 0x7f800000u, dsec = 0x7f800000u, first = 0xffffffffu, ni = 0xffffffffu; int c0 = 0, c1 = 0; float nx = 0.0f, ny = 0.0f;
    for (int i = 0; i < NS * 64; i++) pel_accumulate<AV>(k, xy[2 * i], xy[2 * i + 1], (unsigned)i, dmin, dsec, c0, c1, first, ni, nx, ny);
    out.dmin2 = u2f((int)dmin); out.dsec2 = u2f((int)dsec); out.cnt = c0; out.cnt1 = c1; out.first = (int)first;
    if (!(k.rr >= out.dmin2)) { out.near = (int)ni; out.nx = nx; out.ny = ny; }
    return out;
  }
#else
  typedef float XY __attribute__((ext_vector_type(2)));
  AG_MEM bool lead() const { return sub == 0; }
  AG_MEM int pass_cost() const { return 1; }
  AG_MEM unsigned cap() const { return (unsigned)(NS * 64); }
  AG_MEM bool any(bool p) const { return __ballot(p) != 0ull; }
  // the rare second pass (agar_core.inl PelScan2), wave-level like scan(): noinline keeps its registers out of the hot loop's budget
  // (the query and the pellet pointer travel by value: a reference / `this` would be a generic pointer into the caller's scratch frame,
  // read with flat_* instructions)
  template <bool AV> AG_MEM PelScan2 scan2(bool need, const PelQuery2 &k) { return scan2_call<AV>(xy, need, k); }
  template <bool AV> static __attribute__((noinline)) AG_MEM_NOINLINE PelScan2 scan2_call(AG_GLOBAL float *xy, bool need, const PelQuery2 k) {
    PelScan2 out{0, 0xffffffffu, 0xffffffffu};
    unsigned long long todo = __ballot(need);
    if (todo) ag_mem_fence();
    const int lane = (int)threadIdx.x & 63;
    while (todo) {
      const int src = (int)__builtin_ctzll(todo);
      todo &= ~((QG == 64 ? ~0ull : ((1ull << (QG & 63)) - 1ull)) << src);
      PelQuery2 b;
      b.x = u2f(__builtin_amdgcn_readlane(f2u(k.x), src)); b.y = u2f(__builtin_amdgcn_readlane(f2u(k.y), src));
      b.rr1 = u2f(__builtin_amdgcn_readlane(f2u(k.rr1), src)); b.rr2 = u2f(__builtin_amdgcn_readlane(f2u(k.rr2), src));
      b.gx = __builtin_amdgcn_readlane(k.gx, src); b.gy = __builtin_amdgcn_readlane(k.gy, src); b.cap = k.cap;
      unsigned long long pa = (unsigned long long)(AG_GLOBAL void *)xy;
      unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)pa, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(pa >> 32), src);
      auto gp = (const AG_GLOBAL XY *)(((unsigned long long)hi << 32) | lo) + lane;
      int c2 = 0; unsigned k1 = 0xffffffffu, k2 = 0xffffffffu;
      for (int s = 0; s < NS; s++) { const XY p = gp[s * 64]; pel_accumulate2<AV>(b, p.x, p.y, (unsigned)(s * 64 + lane), c2, k1, k2); }
      pel_reduce2(c2, k1, k2);
      if ((lane & ~(QG - 1)) == src) { out.cnt2 = c2; out.key1 = k1; out.key2 = k2; }
    }
    return out;
  }
  template <bool AV> AG_MEM PelScan scan(bool need, const PelQuery &k) {
    PelScan out{3.0e38f, 3.0e38f, 0, 0, -1, -1, 0.0f, 0.0f};
    unsigned long long todo = __ballot(need);
    if (todo) ag_mem_fence();  // pellets this wave wrote earlier (swap-pop, regeneration) must be visible to the pass
    const int lane = (int)threadIdx.x & 63;
    while (todo) {
      const int src = (int)__builtin_ctzll(todo);  // first lane of the first group that still waits
      todo &= ~((QG == 64 ? ~0ull : ((1ull << (QG & 63)) - 1ull)) << src);
      PelQuery b;  // that group's query, broadcast to the wave
      b.x = u2f(__builtin_amdgcn_readlane(f2u(k.x), src)); b.y = u2f(__builtin_amdgcn_readlane(f2u(k.y), src));
      b.rr = u2f(__builtin_amdgcn_readlane(f2u(k.rr), src)); b.rr1 = u2f(__builtin_amdgcn_readlane(f2u(k.rr1), src));
      b.gx = __builtin_amdgcn_readlane(k.gx, src); b.gy = __builtin_amdgcn_readlane(k.gy, src);
      unsigned long long pa = (unsigned long long)(AG_GLOBAL void *)xy;
      unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)pa, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(pa >> 32), src);
      auto gp = (const AG_GLOBAL XY *)(((unsigned long long)hi << 32) | lo) + lane;
      XY p[NS];
      _Pragma("unroll") for (int s = 0; s < NS; s++) p[s] = gp[s * 64];
      // the usual outcome is "nothing in reach": the two smallest distances and the nearest pellet first, the counts only when something is
      unsigned dmin = 0x7f800000u, dsec = 0x7f800000u, first = 0xffffffffu, ni = 0xffffffffu; int c0 = 0, c1 = 0; float nx = 0.0f, ny = 0.0f;
      _Pragma("unroll") for (int s = 0; s < NS; s++) {
        if (pel_visible<AV>(b, p[s].x, p[s].y)) {
          const unsigned v = (unsigned)f2u(sqr_dist(b.x, b.y, p[s].x, p[s].y));
          if (v < dmin) { ni = (unsigned)(s * 64 + lane); nx = p[s].x; ny = p[s].y; }
          const unsigned hi = v > dmin ? v : dmin; dsec = hi < dsec ? hi : dsec; dmin = v < dmin ? v : dmin;
        }
      }
      const unsigned lane_min = dmin;
      dmin = wred_min(dmin);
      if (b.rr >= u2f((int)dmin)) {  // (uniform branch)
        dmin = 0x7f800000u; dsec = 0x7f800000u; ni = 0xffffffffu;
        _Pragma("unroll") for (int s = 0; s < NS; s++) pel_accumulate<AV>(b, p[s].x, p[s].y, (unsigned)(s * 64 + lane), dmin, dsec, c0, c1, first, ni, nx, ny);
        pel_reduce(b.rr, dmin, dsec, c0, c1, first, ni, nx, ny);
      } else pel_reduce_near(lane_min, dmin, dsec, ni, nx, ny);
      if ((lane & ~(QG - 1)) == src) { out.dmin2 = u2f((int)dmin); out.dsec2 = u2f((int)dsec); out.cnt = c0; out.cnt1 = c1; out.first = (int)first; out.near = (int)ni; out.nx = nx; out.ny = ny; }
    }
    return out;
  }
#endif
  AG_MEM void append(int idx, float x, float y, int pid) {  // pellets.emplace_back (the group's first lane writes)
    if (lead()) { xy[2 * idx] = x; xy[2 * idx + 1] = y; id[idx] = pid; }
  }
  AG_MEM void swap_pop(int ev, int np) {  // Engine.hpp:1002-1009 for one event (per-arena code; the next pass fences)
    if (lead()) {
      if (np > 1 && ev < np - 1) { xy[2 * ev] = xy[2 * (np - 1)]; xy[2 * ev + 1] = xy[2 * (np - 1) + 1]; id[ev] = id[np - 1]; }
      xy[2 * (np - 1)] = AG_PEL_SENTINEL; xy[2 * (np - 1) + 1] = AG_PEL_SENTINEL;
    }
  }
};

// `sub` = this lane's index within its arena's lane group (0 in the host emulation, which runs the same text as
// scalar code); `valid` = false for the padding groups of the last wavefront (they run along but never store).
// Returns the hand-over (per lane, uniform over the group): ticks done (-1: nothing, == ticks: step finished) and the
// agent's mass before the step.  parity >= 0: also count unfinished arenas in gs->qcount[parity] (two-kernel step).
struct QHandOver { int done, before; };
// TSLG: the env's AgDims::ts_lg as a compile-time constant (a run-time stride costs the front kernel ~190 bytes of scratch per lane)
template <int NS, bool AV, int QG, int TSLG> AG_DEV QHandOver quiet_arena(const AgHot hot, const AgState *gs, int arena, int sub, bool valid, const AG_GLOBAL float *act_dxdy, const AG_GLOBAL int32_t *act, int ticks, bool with_env, int slot, int parity = -1) {
  // tile-transposed arrays (agar_types.h): with tiles of 64, consecutive arenas -- consecutive lane groups -- are 4 bytes apart
  constexpr int ag_ts_lg = TSLG;
  const TileWords<int32_t> S{(AG_GLOBAL int32_t *)(hot.ar + AG_TILE_BASE(arena, AR_WORDS)), ag_ts_lg};
  const TileWords<int32_t> P{(AG_GLOBAL int32_t *)(hot.pl + AG_TILE_BASE(arena, PL_WORDS)), ag_ts_lg};    // (one player per arena here)
  auto C = (AG_GLOBAL uint32_t *)(hot.cells + AG_TILE_BASE(arena, CF_ALL * AG_CC));   // field f of cell 0 = C[AG_CELL_W(f, 0)]
  auto qi = (AG_GLOBAL int32_t *)(gs->qinfo + (size_t)arena * 2);
  GrpPel<NS, QG> pel{(AG_GLOBAL float *)(gs->pel_xy + (size_t)arena * 2 * (NS * 64)), (AG_GLOBAL int32_t *)(gs->pel_id + (size_t)arena * (NS * 64)), sub};
  const bool lead = pel.lead() && valid;
#if defined(AGAR_PROFILE) && !defined(AGAR_CPU_EMU)
  unsigned t0_ = (unsigned)__builtin_readcyclecounter(); unsigned long long w0_ = wall_clock64();
#endif
  QState q;
#ifdef AGAR_PROFILE_REASONS
  q.why = 0;
#endif
  q.m = C[AG_CELL_W(CF_M, 0)];
  int ncells = P[PL_NCELLS], nfood = S[AR_NFOOD]; unsigned cmc = C[AG_CELL_W(CF_CMC, 0)];
  // everything the step needs, requested up front: one round trip
  q.x = u2f((int)C[AG_CELL_W(CF_X, 0)]); q.y = u2f((int)C[AG_CELL_W(CF_Y, 0)]); q.vx = u2f((int)C[AG_CELL_W(CF_VX, 0)]); q.vy = u2f((int)C[AG_CELL_W(CF_VY, 0)]);
  q.svx = u2f((int)C[AG_CELL_W(CF_SX, 0)]); q.svy = u2f((int)C[AG_CELL_W(CF_SY, 0)]); q.r = u2f((int)C[AG_CELL_W(CF_CRAD, 0)]); q.hi = u2f((int)C[AG_CELL_W(CF_CMS, 0)]);
  q.action = P[PL_ACTION]; q.tx = u2f(P[PL_TX]); q.ty = u2f(P[PL_TY]);
  q.elapsed = P[PL_ELAPSED]; q.fcd = P[PL_FEED_CD]; q.scd = P[PL_SPLIT_CD]; q.last_decay = P[PL_LAST_DECAY]; q.nvt = P[PL_NVTICKS];
  q.food_eaten = P[PL_FOOD_EATEN]; q.hm = P[PL_HIGHEST_MASS]; q.rate = (double)u2f(P[PL_ANTI_TEAM]); q.sx0 = u2f(P[PL_SAFE_X]); q.sy0 = u2f(P[PL_SAFE_Y]); q.passes = P[PL_PASSES];
  q.cx = u2f(P[PL_CAND_X]); q.cy = u2f(P[PL_CAND_Y]); q.cidx = P[PL_CAND_IDX];
  q.nv = S[AR_NVIR]; q.np = S[AR_NPEL]; q.ticks = S[AR_TICKS]; q.slack = u2f(S[AR_SAFE]); q.mtidx = S[AR_MTIDX]; q.idc = S[AR_IDC];
  int clock = S[AR_CLOCK], done_flag = S[AR_DONE];
  float dx = 0.0f, dy = 0.0f; int action = 0;
  const bool acting = with_env && act;
  if (acting) { dx = act_dxdy[2 * (size_t)arena]; dy = act_dxdy[2 * (size_t)arena + 1]; action = act[arena]; }
  const bool ok = valid && ncells == 1 && nfood == 0 && cmc == q.m;  // (no early return: the tick loop below is wave-synchronous)
  const int mode = gs->g.mode;
  const unsigned before = q.m;
  if (acting) {  // take_action with one cell (BaseEnvironment.hpp:162-176, Player.hpp:102-126): mass-weighted centroid in fp32
    float fm = (float)q.m; float sx = 0.0f, sy = 0.0f; float t = q.x * fm; sx += t; t = q.y * fm; sy += t;
    float px = ag_divf(sx, fm), py = ag_divf(sy, fm);
    float ox = dx * 10.0f, oy = dy * 10.0f;
    q.action = action; q.tx = px + ox; q.ty = py + oy;
  }
  if (with_env && mode == 3 && q.m >= 23000u) done_flag = 1;
#if defined(AGAR_PROFILE) && !defined(AGAR_CPU_EMU)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned t1_ = (unsigned)__builtin_readcyclecounter();
#endif
  quiet_ticks<AV>(q, gs->g, (const AG_GLOBAL float *)gs->lut_r, (const AG_GLOBAL float *)gs->lut_ms, (const AG_GLOBAL uint64_t *)(gs->mt + (size_t)arena * 312), pel, ticks, ok);
#if defined(AGAR_PROFILE) && !defined(AGAR_CPU_EMU)
  unsigned t2_ = (unsigned)__builtin_readcyclecounter();
  if (lead && gs->prof) { gs->prof[(size_t)arena * 16 + 4] = w0_; gs->prof[(size_t)arena * 16 + 5] = wall_clock64(); gs->prof[(size_t)arena * 16 + 7] = (unsigned long long)(q.food_eaten); gs->prof[(size_t)arena * 16 + 0] += t1_ - t0_; gs->prof[(size_t)arena * 16 + 1] += t2_ - t1_; gs->prof[(size_t)arena * 16 + 2] += (unsigned long long)q.done; }
#endif
  const bool finished = q.done == ticks;
  if (finished && with_env && mode == 3 && q.m >= 23000u) done_flag = 1;
  if (lead && !ok) { qi[0] = -1; qi[1] = 0; }
#ifndef AGAR_CPU_EMU
  if (lead && !(ok && finished)) {
    // (HBM words behind descriptor pointers: cast to the global address space, so that these are global_*, not flat_*, instructions)
    if (parity >= 0) { const int at = __hip_atomic_fetch_add((AG_GLOBAL int32_t *)gs->qcount + parity, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ((AG_GLOBAL int32_t *)gs->qlist)[(size_t)parity * gs->d.A + at] = arena; }  // k_step's work list
    (void)__hip_atomic_fetch_add((AG_GLOBAL int32_t *)gs->qstat, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // statistics for the host's fused / two-kernel choice
#ifdef AGAR_PROFILE_REASONS
    (void)__hip_atomic_fetch_add((AG_GLOBAL int32_t *)gs->qstat + 4 + (ok ? (q.why & 7) : 0), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // [4] not a single-cell / food-free arena, [5..11] AG_WHY
#endif
  }
#endif
  {
#ifndef AGAR_CPU_EMU
  // The stores below go to the words the prologue loaded.  Left to itself the compiler keeps every one of those 64-bit addresses alive
  // across the whole tick loop -- 128 registers do not hold them: 176-192 bytes of scratch per lane, 50 MB of spill traffic a step at
  // 262 144 arenas.  An arena index the optimiser cannot see through makes it form the addresses again here: one register lives on.
  int arena_w = arena; asm volatile("" : "+v"(arena_w));
  const TileWords<int32_t> S{(AG_GLOBAL int32_t *)(hot.ar + AG_TILE_BASE(arena_w, AR_WORDS)), ag_ts_lg};
  const TileWords<int32_t> P{(AG_GLOBAL int32_t *)(hot.pl + AG_TILE_BASE(arena_w, PL_WORDS)), ag_ts_lg};
  auto C = (AG_GLOBAL uint32_t *)(hot.cells + AG_TILE_BASE(arena_w, CF_ALL * AG_CC));
  auto qi = (AG_GLOBAL int32_t *)(gs->qinfo + (size_t)arena_w * 2);
  arena = arena_w;
#endif
  if (lead && ok) {
    if (q.done > 0) {
      C[AG_CELL_W(CF_X, 0)] = (uint32_t)f2u(q.x); C[AG_CELL_W(CF_Y, 0)] = (uint32_t)f2u(q.y); C[AG_CELL_W(CF_VX, 0)] = (uint32_t)f2u(q.vx); C[AG_CELL_W(CF_VY, 0)] = (uint32_t)f2u(q.vy);
      C[AG_CELL_W(CF_SX, 0)] = (uint32_t)f2u(q.svx); C[AG_CELL_W(CF_SY, 0)] = (uint32_t)f2u(q.svy); C[AG_CELL_W(CF_M, 0)] = q.m;
      C[AG_CELL_W(CF_CMC, 0)] = q.m; C[AG_CELL_W(CF_CRAD, 0)] = (uint32_t)f2u(q.r); C[AG_CELL_W(CF_CMS, 0)] = (uint32_t)f2u(q.hi);
      P[PL_ELAPSED] = q.elapsed; P[PL_MIN_MASS] = (int)q.m_move; P[PL_HIGHEST_MASS] = q.hm; P[PL_FEED_CD] = q.fcd; P[PL_SPLIT_CD] = q.scd;
      P[PL_FOOD_EATEN] = q.food_eaten; P[PL_LAST_DECAY] = q.last_decay; P[PL_SAFE_X] = f2u(q.sx0); P[PL_SAFE_Y] = f2u(q.sy0); P[PL_PASSES] = q.passes;
      P[PL_CAND_X] = f2u(q.cx); P[PL_CAND_Y] = f2u(q.cy); P[PL_CAND_IDX] = q.cidx;
      S[AR_NEVP] = (q.last_ev >= 0 ? 1 : 0) + (q.last_ev2 >= 0 ? 1 : 0); S[AR_NEVV] = 0; S[AR_NPEL] = q.np; S[AR_TICKS] = q.ticks; S[AR_CLOCK] = clock + q.done; S[AR_SAFE] = f2u(q.slack); S[AR_MTIDX] = q.mtidx; S[AR_IDC] = q.idc;
      if (q.last_ev >= 0) { auto ge = (AG_GLOBAL int32_t *)(gs->ev_p + (size_t)arena * (size_t)(gs->d.EC + gs->d.EX)); ge[0] = q.last_ev; if (q.last_ev2 >= 0) ge[1] = q.last_ev2; }
      auto cn = (AG_GLOBAL int32_t *)(gs->counts + (size_t)arena * 4);
      cn[0] = q.np; cn[1] = q.nv; cn[2] = 0; cn[3] = 1;
    }
    if (acting) { P[PL_ACTION] = q.action; P[PL_TX] = f2u(q.tx); P[PL_TY] = f2u(q.ty); }
    if (with_env) { S[AR_RESPAWNED] = 0; S[AR_DONE] = done_flag; }
    // epilogue of BaseEnvironment::step for a live single player: no respawn in any mode
    if (finished && with_env) emit_agent_result(gs, slot, arena, 1, 0, q.m, before, 0, done_flag);
    qi[0] = q.done; qi[1] = (int)before;
  }
  }
  QHandOver h; h.done = ok ? q.done : -1; h.before = (int)before;
  return h;
}
