// Lean front kernel body: ONE wave per arena, no LDS, no pellet registers -> high occupancy.
//
// In RL rollouts almost every env step of a single-player arena is "quiet" (see quiet_ticks in agar_core.inl):
// the whole step touches ~300 bytes of state and now and then scans the pellets.  k_quiet runs first and does
// the env prologue (take_actions), as many quiet ticks as it can and -- if that was the whole step -- the env
// epilogue (rewards / dones / packed results).  What it could not finish is left to the general kernel k_step,
// which runs right after it on the same stream: qinfo[arena] = { ticks already done (or -1: nothing, not even the
// prologue), agent mass before the step }.  Arenas that are finished make k_step's wave exit on its first load.
// Single-player envs only (P == 1); with several players k_quiet is never launched.
#pragma once

// lane k <- src[k * stride] (k < n): one gather load
#ifdef AGAR_CPU_EMU
template <class PT> AG_DEV void ub_load_strided(UBlock &b, PT src, int n, int stride) { for (int i = 0; i < 32; i++) b.w[i] = i < n ? (int)src[(size_t)i * stride] : 0; }
template <class PT> AG_DEV void ub_store_strided(const UBlock &b, PT dst, int n, int stride) { for (int i = 0; i < n; i++) dst[(size_t)i * stride] = (uint32_t)b.w[i]; }
#else
template <class PT> AG_DEV void ub_load_strided(UBlock &b, PT src, int n, int stride) { int l = (int)threadIdx.x; b.v = l < n ? (int)src[(size_t)l * stride] : 0; }
template <class PT> AG_DEV void ub_store_strided(const UBlock &b, PT dst, int n, int stride) { int l = (int)threadIdx.x; if (l < n) dst[(size_t)l * stride] = (uint32_t)b.v; }
#endif

template <int NS, bool AV> AG_DEV void quiet_arena(const AgState *gs, int arena, const AG_GLOBAL float *act_dxdy, const AG_GLOBAL int32_t *act, int ticks, bool with_env, int slot) {
  auto gar = (AG_GLOBAL int32_t *)(gs->ar + (size_t)arena * AR_WORDS);
  auto gpl = (AG_GLOBAL int32_t *)(gs->pl + (size_t)arena * PL_WORDS);
  auto gcell = (AG_GLOBAL uint32_t *)(gs->cells + (size_t)arena * (CF_ALL * AG_CC));
  auto qi = (AG_GLOBAL int32_t *)(gs->qinfo + (size_t)arena * 2);
  UBlock S, PB, CB;  // arena words, player words, the 12 words of cell 0: three independent loads, one round trip
  ub_load(S, gar, AR_WORDS);
  ub_load(PB, gpl, PL_WORDS);
  ub_load_strided(CB, gcell, CF_ALL, AG_CC);
  float dx = 0.0f, dy = 0.0f; int action = 0;
  if (with_env && act) { dx = act_dxdy[2 * (size_t)arena]; dy = act_dxdy[2 * (size_t)arena + 1]; action = act[arena]; }
  QState q;
  q.m = (unsigned)ub_get(CB, CF_M);
  if (ub_get(PB, PL_NCELLS) != 1 || ub_get(S, AR_NFOOD) != 0 || (unsigned)ub_get(CB, CF_CMC) != q.m) { AG_SERIAL { qi[0] = -1; qi[1] = 0; } return; }
  q.x = u2f(ub_get(CB, CF_X)); q.y = u2f(ub_get(CB, CF_Y));
  q.action = ub_get(PB, PL_ACTION); q.tx = u2f(ub_get(PB, PL_TX)); q.ty = u2f(ub_get(PB, PL_TY));
  int done_flag = ub_get(S, AR_DONE), mode = gs->g.mode;
  unsigned before = q.m;
  if (with_env) {
    if (act) {  // take_action with one cell (BaseEnvironment.hpp:162-176, Player.hpp:102-126): mass-weighted centroid in fp32
      float fm = (float)q.m; float sx = 0.0f, sy = 0.0f; float t = q.x * fm; sx += t; t = q.y * fm; sy += t;
      float px = ag_divf(sx, fm), py = ag_divf(sy, fm);
      float ox = dx * 10.0f, oy = dy * 10.0f;
      q.action = action; q.tx = px + ox; q.ty = py + oy;
      ub_set(PB, PL_ACTION, action); ub_set(PB, PL_TX, f2u(q.tx)); ub_set(PB, PL_TY, f2u(q.ty));
    }
    ub_set(S, AR_RESPAWNED, 0);
    if (mode == 3 && q.m >= 23000u) { done_flag = 1; ub_set(S, AR_DONE, 1); }
  }
  q.nv = ub_get(S, AR_NVIR); q.np = ub_get(S, AR_NPEL);
  q.vx = u2f(ub_get(CB, CF_VX)); q.vy = u2f(ub_get(CB, CF_VY)); q.svx = u2f(ub_get(CB, CF_SX)); q.svy = u2f(ub_get(CB, CF_SY));
  q.r = u2f(ub_get(CB, CF_CRAD)); q.hi = u2f(ub_get(CB, CF_CMS));
  q.ticks = ub_get(S, AR_TICKS); q.elapsed = ub_get(PB, PL_ELAPSED); q.fcd = ub_get(PB, PL_FEED_CD); q.scd = ub_get(PB, PL_SPLIT_CD);
  q.last_decay = ub_get(PB, PL_LAST_DECAY); q.nvt = ub_get(PB, PL_NVTICKS); q.food_eaten = ub_get(PB, PL_FOOD_EATEN); q.hm = ub_get(PB, PL_HIGHEST_MASS);
  q.rate = (double)u2f(ub_get(PB, PL_ANTI_TEAM)); q.slack = u2f(ub_get(S, AR_SAFE));
  MemPel<NS> pel{(AG_GLOBAL float *)(gs->pel_xy + (size_t)arena * 2 * (NS * 64)), (AG_GLOBAL int32_t *)(gs->pel_id + (size_t)arena * (NS * 64))};
  quiet_ticks<AV>(q, gs->g, (const AG_GLOBAL float *)gs->lut_r, (const AG_GLOBAL float *)gs->lut_ms, pel, ticks);
  if (q.done > 0) {
    ub_set(CB, CF_X, f2u(q.x)); ub_set(CB, CF_Y, f2u(q.y)); ub_set(CB, CF_VX, f2u(q.vx)); ub_set(CB, CF_VY, f2u(q.vy));
    ub_set(CB, CF_SX, f2u(q.svx)); ub_set(CB, CF_SY, f2u(q.svy)); ub_set(CB, CF_M, (int)q.m);
    ub_set(CB, CF_CMC, (int)q.m); ub_set(CB, CF_CRAD, f2u(q.r)); ub_set(CB, CF_CMS, f2u(q.hi));
    ub_store_strided(CB, gcell, CF_ALL, AG_CC);
    ub_set(PB, PL_ELAPSED, q.elapsed); ub_set(PB, PL_MIN_MASS, (int)q.m_move); ub_set(PB, PL_HIGHEST_MASS, q.hm);
    ub_set(PB, PL_FEED_CD, q.fcd); ub_set(PB, PL_SPLIT_CD, q.scd); ub_set(PB, PL_FOOD_EATEN, q.food_eaten); ub_set(PB, PL_LAST_DECAY, q.last_decay);
    ub_set(S, AR_NEVP, q.last_ev >= 0 ? 1 : 0); ub_set(S, AR_NEVV, 0); ub_set(S, AR_NPEL, q.np);
    ub_set(S, AR_TICKS, q.ticks); ub_set(S, AR_CLOCK, ub_get(S, AR_CLOCK) + q.done); ub_set(S, AR_SAFE, f2u(q.slack));
    if (q.last_ev >= 0) { auto ge = (AG_GLOBAL int32_t *)(gs->ev_p + (size_t)arena * AG_EV_CAP); AG_SERIAL { ge[0] = q.last_ev; } }
    auto cn = (AG_GLOBAL int32_t *)(gs->counts + (size_t)arena * 4);
    int nv = q.nv, np = q.np;
    AG_SERIAL { cn[0] = np; cn[1] = nv; cn[2] = 0; cn[3] = 1; }
  }
  bool finished = q.done == ticks;
  if (finished && with_env) {  // epilogue of BaseEnvironment::step for a live single player: no respawn in any mode
    if (mode == 3 && q.m >= 23000u) { done_flag = 1; ub_set(S, AR_DONE, 1); }
    emit_agent_result(gs, slot, arena, 1, 0, q.m, before, 0, done_flag);
  }
  if (q.done > 0 || with_env) { ub_store(PB, gpl, PL_WORDS); ub_store(S, gar, AR_WORDS); }
  int qd = q.done;
  AG_SERIAL { qi[0] = qd; qi[1] = (int)before; }
}
