// "ram" observation: a flat fp32 vector per (arena, agent) -- the agent's own cells and its K nearest pellets / viruses / other players'
// cells, relative to the agent's centre.  BASELINE configs[0] names a ram observation, but the reference has none to match: "ram" is
// accepted by gym_agario/AgarioEnv.py:52 and rejected at :211, agario-ram-v0 is never registered, environment/test/ram-env-test.hpp is
// empty ("the build's own flat fp32 dump ... excluded from parity and included only in timing").  So the layout is this
// repository's own (include/agarcl_batch.h: agarcl_ram_obs), checked on the GPU against the host restatement oracle/ram_oracle.py:
//
//   [0..3]                         px, py (Player::x / y: mass-weighted centre, sequential fp32 sums in cell order), total mass, cell count
//   [4 ..)  KC x (dx, dy, mass)    own cells in cell order
//   then    KP x (dx, dy)          the KP pellets nearest to (px, py), nearest first (ties: lower index)
//   then    KV x (dx, dy, mass)    the KV nearest viruses
//   then    KO x (dx, dy, mass)    the KO nearest cells of the other players (players in slot order, cells in cell order)
// dx = x - px, dy = y - py in fp32; rows beyond the number of entities are zero.  One wavefront per (arena, agent); selection = K rounds of
// "smallest (squared distance, index) key not selected yet": a lane-private scan of the lane's entities + two wave minima.
#pragma once

struct AgRamCfg { int KC, KP, KV, KO; };
#ifdef AGAR_CPU_EMU
static inline
#else
__host__ __device__ inline
#endif
int ag_ram_dim(const AgRamCfg &o) { return 4 + 3 * o.KC + 2 * o.KP + 3 * o.KV + 3 * o.KO; }

#ifndef AGAR_CPU_EMU
// the register form of ram_nearest (below) for n <= 64 J entities
template <int J, class PosT, class EmitT> AG_DEV void ram_nearest_regs(int n, int rounds, float px, float py, PosT pos, EmitT emit) {
  unsigned d[J];
#pragma unroll
  for (int j = 0; j < J; j++) {
    const int i = AG_LANE + 64 * j; float x = 0.0f, y = 0.0f; d[j] = 0xffffffffu;
    if (i < n && pos(i, x, y)) { const float dx = x - px, dy = y - py; const float a = dx * dx, b = dy * dy; d[j] = (unsigned)f2u(a + b); }
  }
  int mine = -1, count = 0;
  for (int r = 0; r < rounds; r++) {
    unsigned bd = 0xffffffffu; int bj = 0;
#pragma unroll
    for (int j = 0; j < J; j++) { if (d[j] < bd) { bd = d[j]; bj = j; } }   // (ascending j = ascending index within the lane: ties keep the lower one)
    const unsigned md = wred_min(bd);
    if (md == 0xffffffffu) break;
    const unsigned bi = (unsigned)(AG_LANE + 64 * bj);
    // the winner's index: one lane holds the minimum almost always (then its index is a v_readlane away); equal keys in several lanes take
    // the second reduction
    const unsigned long long tie = __builtin_amdgcn_ballot_w64(bd == md);
    const unsigned mi = (tie & (tie - 1ull)) == 0ull ? (unsigned)__builtin_amdgcn_readlane((int)bi, (int)__builtin_ctzll(tie)) : wred_min(bd == md ? bi : 0xffffffffu);
    if (bd == md && bi == mi) {
#pragma unroll
      for (int j = 0; j < J; j++) { if (j == bj) d[j] = 0xffffffffu; }
    }
    if (AG_LANE == r) mine = (int)mi;
    count = r + 1;
  }
  if (AG_LANE < count) { float x, y; (void)pos(mine, x, y); emit(AG_LANE, mine, x - px, y - py); }
}
// K nearest of n entities: pos(i, x, y) gives entity i's position (false: no such entity); emit(rank, i, dx, dy) writes a selected one
template <class PosT, class EmitT> AG_DEV void ram_nearest(int n, int K, float px, float py, PosT pos, EmitT emit) {
  const int rounds = K < n ? K : n;
#ifndef AG_RAM_RESCAN
  // Up to 1024 entities and 64 rows: every lane reads its (at most 16) entities ONCE and keeps their distance keys in registers; a round is
  // a register scan + two wave minima, the winner's key is retired in place, and the selected entities are fetched and written after the
  // last round, a lane per row.  The general form below reads all entities again in every round -- a dependent trip to L2 per round, and
  // another for the winner's position: ~40 of them in a row were this kernel's 64 us.  Same order: ascending (distance bits, index), every
  // entity at most once; a hole or a key of all ones is never selected, as below.
  if (n <= 1024 && rounds <= 64) {
    if (n <= 64) ram_nearest_regs<1>(n, rounds, px, py, pos, emit);
    else if (n <= 256) ram_nearest_regs<4>(n, rounds, px, py, pos, emit);
    else if (n <= 512) ram_nearest_regs<8>(n, rounds, px, py, pos, emit);
    else ram_nearest_regs<16>(n, rounds, px, py, pos, emit);
    return;
  }
#endif
  unsigned last_d = 0u, last_i = 0u; bool first = true;
  for (int r = 0; r < rounds; r++) {
    unsigned best_d = 0xffffffffu, best_i = 0xffffffffu;
    for (int i = AG_LANE; i < n; i += 64) {
      float x, y; if (!pos(i, x, y)) continue;   // (a hole of a fixed-capacity table: never a candidate, whatever the centre is)
      const float dx = x - px, dy = y - py; const float a = dx * dx, b = dy * dy;
      const unsigned d = (unsigned)f2u(a + b);   // (>= 0, or NaN: its bits order above every distance)
      const bool above = first || d > last_d || (d == last_d && (unsigned)i > last_i);
      if (above && (d < best_d || (d == best_d && (unsigned)i < best_i))) { best_d = d; best_i = (unsigned)i; }
    }
    const unsigned md = wred_min(best_d);
    const unsigned mi = wred_min(best_d == md ? best_i : 0xffffffffu);
    if (mi == 0xffffffffu) break;
    last_d = md; last_i = mi; first = false;
    AG_SERIAL { float x, y; (void)pos((int)mi, x, y); emit(r, (int)mi, x - px, y - py); }
  }
}

AG_DEV void ram_obs_agent(const AgState *gs, int arena, int agent, AgRamCfg o, float *out_) {
  const int P = gs->d.P, ag_ts_lg = gs->d.ts_lg, dim = ag_ram_dim(o);
  AG_GLOBAL float *out = (AG_GLOBAL float *)out_ + ((size_t)arena * gs->d.n_agents + agent) * dim;
  const AG_GLOBAL int32_t *ar = (const AG_GLOBAL int32_t *)AG_AR_PTR(gs, arena);
  const AG_GLOBAL int32_t *pl = (const AG_GLOBAL int32_t *)AG_PL_PTR(gs, arena, agent);   // agent i == player slot i
  const AG_GLOBAL uint32_t *C = (const AG_GLOBAL uint32_t *)AG_CELLS_PTR(gs, arena, agent);
  AG_LANES(i, dim) out[i] = 0.0f;
  // the counts and the agent's cell slots (lane i = slot i, whatever the cell count) are requested before the zero fill is fenced: one
  // round trip for "stores acknowledged" and these loads together, instead of fence, count, cells, counts one after the other
  const int n = pl[AG_TW(PL_NCELLS)], np = ar[AG_TW(AR_NPEL)], nv = ar[AG_TW(AR_NVIR)];
  float own_x = 0.0f, own_y = 0.0f; unsigned own_m = 0u;
  if (AG_LANE < AG_CC) { own_x = u2f((int)C[AG_CELL_W(CF_X, AG_LANE)]); own_y = u2f((int)C[AG_CELL_W(CF_Y, AG_LANE)]); own_m = C[AG_CELL_W(CF_M, AG_LANE)]; }
  ag_mem_fence();
  float sx = 0.0f, sy = 0.0f; unsigned tm = 0;
  for (int i = 0; i < n; i++) {   // (sequential fp32 sums in cell order, every lane the same: the slots are broadcast from registers)
    const unsigned m = (unsigned)__builtin_amdgcn_readlane((int)own_m, i); const float fm = (float)m;
    float t = u2f(__builtin_amdgcn_readlane(f2u(own_x), i)) * fm; sx += t; t = u2f(__builtin_amdgcn_readlane(f2u(own_y), i)) * fm; sy += t; tm += m;
  }
  // a dead agent (the observation of the terminal step, before any respawn): its record stays all-zero with cell count 0 -- Player::x() would
  // give 0 / 0 = NaN, every offset of every row would be NaN and the row order meaningless, and the vector goes straight into a learner
  if (n == 0) return;
  const float px = ag_divf(sx, (float)tm), py = ag_divf(sy, (float)tm);
  AG_SERIAL { out[0] = px; out[1] = py; out[2] = (float)tm; out[3] = (float)n; }
  AG_GLOBAL float *oc = out + 4, *op = oc + 3 * o.KC, *ov = op + 2 * o.KP, *oo = ov + 3 * o.KV;
  if (AG_LANE < (n < o.KC ? n : o.KC)) { const int i = AG_LANE; oc[3 * i] = own_x - px; oc[3 * i + 1] = own_y - py; oc[3 * i + 2] = (float)own_m; }
  const AG_GLOBAL float *pxy = (const AG_GLOBAL float *)(gs->pel_xy + (size_t)arena * gs->d.PC * 2);
  ram_nearest(np, o.KP, px, py, [&](int i, float &x, float &y) { x = pxy[2 * i]; y = pxy[2 * i + 1]; return true; },
              [&](int r, int, float dx, float dy) { op[2 * r] = dx; op[2 * r + 1] = dy; });
  const AG_GLOBAL float *vx = (const AG_GLOBAL float *)(gs->vir_x + (size_t)arena * gs->d.VC), *vy = (const AG_GLOBAL float *)(gs->vir_y + (size_t)arena * gs->d.VC);
  const AG_GLOBAL int32_t *vm = (const AG_GLOBAL int32_t *)(gs->vir_mass + (size_t)arena * gs->d.VC);
  ram_nearest(nv, o.KV, px, py, [&](int i, float &x, float &y) { x = vx[i]; y = vy[i]; return true; },
              [&](int r, int i, float dx, float dy) { ov[3 * r] = dx; ov[3 * r + 1] = dy; ov[3 * r + 2] = (float)vm[i]; });
  // other players' cells: entity index = slot * AG_CC + cell (slots in ascending order, the agent's own slot skipped)
  if (P > 1 && o.KO > 0) {
    const AG_GLOBAL uint32_t *C0 = (const AG_GLOBAL uint32_t *)AG_CELLS_PTR(gs, arena, 0);
    const AG_GLOBAL int32_t *pl0 = (const AG_GLOBAL int32_t *)AG_PL_PTR(gs, arena, 0);
    auto cell = [&](int e, float &x, float &y, float &m) -> bool {   // entity e = (slot, cell); false for the agent's own slot and empty cell slots
      const int s = e / AG_CC, i = e - s * AG_CC;
      const bool live = s != agent && i < pl0[AG_TW(s * PL_WORDS + PL_NCELLS)];
      const AG_GLOBAL uint32_t *Cs = C0 + AG_TW(s * (CF_ALL * AG_CC));
      x = live ? u2f((int)Cs[AG_CELL_W(CF_X, i)]) : 0.0f; y = live ? u2f((int)Cs[AG_CELL_W(CF_Y, i)]) : 0.0f; m = live ? (float)Cs[AG_CELL_W(CF_M, i)] : 0.0f;
      return live;
    };
    int live_total = 0; for (int s = 0; s < P; s++) if (s != agent) live_total += pl0[AG_TW(s * PL_WORDS + PL_NCELLS)];
    ram_nearest(P * AG_CC, o.KO < live_total ? o.KO : live_total, px, py, [&](int e, float &x, float &y) { float m; return cell(e, x, y, m); },
                [&](int r, int e, float dx, float dy) { float x, y, m; cell(e, x, y, m); oo[3 * r] = dx; oo[3 * r + 1] = dy; oo[3 * r + 2] = m; });
  }
}
#endif
