// Several players per arena: the engine's player iteration order, cell-eats-cell and the scripted bots.
// Everything here is rare-path or inherently sequential in the reference, so it runs as lane-0 serial code over a
// per-arena HBM scratch area (allocated only when an arena has more than one player); the wave-parallel part is the
// pre-test that proves the common "nobody can eat anybody" case and skips the sequential replay.
//
// Third-party behaviour the reference's results depend on and that is restated here for the device (GCC 11.4
// libstdc++ / glibc 2.35; the product's own restatements -- the oracle has separate ones):
//   * std::unordered_map iteration order (_Hashtable: insert at bucket begin, _Prime_rehash_policy growth);
//   * std::sort (introsort: median-of-3 quicksort down to 16, heapsort fallback, final insertion sort);
//   * glibc rand() (TYPE_3 additive feedback), consumed by Player colours and the bots' fallbacks.
#pragma once

// ---- libstdc++ _Hashtable order emulation over caller-provided arrays ---------------------------------------------
struct AgHMap { int bucket_count, next_resize, n, head; AG_GLOBAL int32_t *key, *next, *before; };
AG_DEV int ag_next_bkt(int n, int *next_resize) {  // _Prime_rehash_policy::_M_next_bkt
  const int primes[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 103, 109, 113,
                        127, 137, 139, 149, 157, 167, 179, 193, 199, 211, 227, 241, 257, 277, 293, 313, 337, 359, 383, 409, 439, 467,
                        503, 541, 577, 619, 661, 709, 761, 823, 887, 953, 1031, 1109};
  const int fast[] = {2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13};
  int b = 1109;
  if (n < 14) b = fast[n];
  else for (int i = 0; i < (int)(sizeof(primes) / sizeof(int)); i++) if (primes[i] >= n) { b = primes[i]; break; }
  *next_resize = b;
  return b;
}
AG_DEV void ag_hm_rehash(AgHMap &m, int nb) {  // _M_rehash_aux(n, true_type)
  int p = m.head;
  for (int i = 0; i < nb; i++) m.before[i] = -2;
  m.bucket_count = nb; m.head = -1;
  int bbegin = 0;
  while (p != -1) {
    int nx = m.next[p];
    int b = (int)((unsigned)m.key[p] % (unsigned)nb);
    if (m.before[b] == -2) {
      m.next[p] = m.head; m.head = p; m.before[b] = -1;
      if (m.next[p] != -1) m.before[bbegin] = p;
      bbegin = b;
    } else {
      int bf = m.before[b];
      if (bf == -1) { m.next[p] = m.head; m.head = p; } else { m.next[p] = m.next[bf]; m.next[bf] = p; }
    }
    p = nx;
  }
}
AG_DEV void ag_hm_insert(AgHMap &m, int key) {  // _M_insert_unique_node; key assumed absent
  if (m.n + 1 > m.next_resize) {  // _M_need_rehash(bucket_count, element_count, 1)
    int lhs = m.n + 1, floor11 = m.next_resize ? 0 : 11;
    int min_bkts = lhs > floor11 ? lhs : floor11;  // max_load_factor 1.0
    if (min_bkts >= m.bucket_count) {
      int want = min_bkts + 1, grow = m.bucket_count * 2;
      ag_hm_rehash(m, ag_next_bkt(want > grow ? want : grow, &m.next_resize));
    } else m.next_resize = m.bucket_count;
  }
  int node = m.n++;
  m.key[node] = key;
  int b = (int)((unsigned)key % (unsigned)m.bucket_count);
  if (m.before[b] != -2) {  // _M_insert_bucket_begin
    int bf = m.before[b];
    if (bf == -1) { m.next[node] = m.head; m.head = node; } else { m.next[node] = m.next[bf]; m.next[bf] = node; }
  } else {
    m.next[node] = m.head; m.head = node;
    if (m.next[node] != -1) m.before[(unsigned)m.key[m.next[node]] % (unsigned)m.bucket_count] = node;
    m.before[b] = -1;
  }
}

// ---- libstdc++ std::sort on (float key, int payload) pairs held in two parallel arrays ---------------------------------
struct AgSortArr { AG_GLOBAL int32_t *k; AG_GLOBAL int32_t *v; AG_GLOBAL int32_t *stk; };  // k holds float bits; stk: 120-word work stack
AG_DEV float ag_sk(const AgSortArr &a, int i) { return u2f(a.k[i]); }
AG_DEV void ag_sswap(const AgSortArr &a, int i, int j) { int tk = a.k[i], tv = a.v[i]; a.k[i] = a.k[j]; a.v[i] = a.v[j]; a.k[j] = tk; a.v[j] = tv; }
AG_DEV void ag_unguarded_linear_insert(const AgSortArr &a, int last) {
  int vk = a.k[last], vv = a.v[last]; float fv = u2f(vk); int next = last - 1;
  while (fv < ag_sk(a, next)) { a.k[last] = a.k[next]; a.v[last] = a.v[next]; last = next; --next; }
  a.k[last] = vk; a.v[last] = vv;
}
AG_DEV void ag_insertion_sort(const AgSortArr &a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (ag_sk(a, i) < ag_sk(a, first)) {
      int vk = a.k[i], vv = a.v[i];
      for (int j = i; j > first; j--) { a.k[j] = a.k[j - 1]; a.v[j] = a.v[j - 1]; }
      a.k[first] = vk; a.v[first] = vv;
    } else ag_unguarded_linear_insert(a, i);
  }
}
AG_DEV void ag_adjust_heap(const AgSortArr &a, int first, int hole, int len, int vk, int vv) {
  const int top = hole; int child = hole; float fv = u2f(vk);
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (ag_sk(a, first + child) < ag_sk(a, first + child - 1)) child--;
    a.k[first + hole] = a.k[first + child]; a.v[first + hole] = a.v[first + child]; hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); a.k[first + hole] = a.k[first + child - 1]; a.v[first + hole] = a.v[first + child - 1]; hole = child - 1; }
  int parent = (hole - 1) / 2;
  while (hole > top && ag_sk(a, first + parent) < fv) { a.k[first + hole] = a.k[first + parent]; a.v[first + hole] = a.v[first + parent]; hole = parent; parent = (hole - 1) / 2; }
  a.k[first + hole] = vk; a.v[first + hole] = vv;
}
AG_DEV void ag_heapsort(const AgSortArr &a, int first, int last) {  // partial_sort(first, last, last)
  int len = last - first;
  if (len >= 2) { int parent = (len - 2) / 2; for (;;) { ag_adjust_heap(a, first, parent, len, a.k[first + parent], a.v[first + parent]); if (parent == 0) break; parent--; } }
  while (last - first > 1) { --last; int vk = a.k[last], vv = a.v[last]; a.k[last] = a.k[first]; a.v[last] = a.v[first]; ag_adjust_heap(a, first, 0, last - first, vk, vv); }
}
AG_DEV void ag_std_sort(const AgSortArr &a, int first, int last) {
  if (first == last) return;
  int n = last - first, lg = 0; while ((1 << (lg + 1)) <= n) lg++;
  // __introsort_loop with an explicit stack (recursion on the right part, iteration on the left)
  AG_GLOBAL int32_t *stk_f = a.stk, *stk_l = a.stk + 40, *stk_d = a.stk + 80; int sp = 0;
  stk_f[0] = first; stk_l[0] = last; stk_d[0] = 2 * lg; sp = 1;
  while (sp > 0) {
    --sp; int f = stk_f[sp], l = stk_l[sp], depth = stk_d[sp];
    while (l - f > 16) {
      if (depth == 0) { ag_heapsort(a, f, l); break; }
      --depth;
      int mid = f + (l - f) / 2, x = f + 1, y = mid, z = l - 1;  // __move_median_to_first(f, f+1, mid, l-1)
      if (ag_sk(a, x) < ag_sk(a, y)) { if (ag_sk(a, y) < ag_sk(a, z)) ag_sswap(a, f, y); else if (ag_sk(a, x) < ag_sk(a, z)) ag_sswap(a, f, z); else ag_sswap(a, f, x); }
      else if (ag_sk(a, x) < ag_sk(a, z)) ag_sswap(a, f, x); else if (ag_sk(a, y) < ag_sk(a, z)) ag_sswap(a, f, z); else ag_sswap(a, f, y);
      int lo = f + 1, hi = l;  // __unguarded_partition(f+1, l, f)
      for (;;) {
        while (ag_sk(a, lo) < ag_sk(a, f)) ++lo;
        --hi;
        while (ag_sk(a, f) < ag_sk(a, hi)) --hi;
        if (!(lo < hi)) break;
        ag_sswap(a, lo, hi); ++lo;
      }
      if (sp < 40) { stk_f[sp] = lo; stk_l[sp] = l; stk_d[sp] = depth; sp++; }  // right part later (same order of effects: disjoint ranges)
      l = lo;
    }
  }
  if (last - first > 16) { ag_insertion_sort(a, first, first + 16); for (int i = first + 16; i != last; ++i) ag_unguarded_linear_insert(a, i); }
  else ag_insertion_sort(a, first, last);
}

// ---- glibc rand().  State: 34-word ring + position, per arena in HBM -------------------------------------------------------
AG_DEV int ag_rand_next(AG_GLOBAL int32_t *st) {
  int p = st[34];
  unsigned v = (unsigned)st[(p + 3) % 34] + (unsigned)st[(p + 31) % 34];
  st[p] = (int)v; st[34] = (p + 1) % 34;
  return (int)(v >> 1);
}

// ---- scratch layout (int32 words) -------------------------------------------------------------------------------------------
#define AGM_T (AG_MAX_PLAYERS * AG_CC)  // max cells per arena
#define AGM_RES 1024
#define AGM_BKT 1200
enum { AGM_GP = 0, AGM_GI = AGM_GP + AGM_T, AGM_GM = AGM_GI + AGM_T, AGM_GID = AGM_GM + AGM_T, AGM_ROW = AGM_GID + AGM_T, AGM_IK = AGM_ROW + AGM_T,
       AGM_IV = AGM_IK + AGM_T, AGM_HAS = AGM_IV + AGM_T, AGM_RS = AGM_HAS + AGM_T, AGM_RC = AGM_RS + 104, AGM_FILL = AGM_RC + 104,
       AGM_RQ = AGM_FILL + 104, AGM_RG = AGM_RQ + AGM_RES, AGM_HK = AGM_RG + AGM_RES, AGM_HN = AGM_HK + AGM_T, AGM_HB = AGM_HN + AGM_T,
       AGM_GX = AGM_HB + AGM_BKT, AGM_GY = AGM_GX + AGM_T, AGM_STK = AGM_GY + AGM_T, AGM_WORDS = AGM_STK + 128 };

template <int NS, bool AV> AG_DEV AG_GLOBAL int32_t *g_scratch(const AgCtx<NS, AV> &c) { return (AG_GLOBAL int32_t *)(c.gs->scratch + (size_t)c.arena * AGM_WORDS); }
template <int NS, bool AV> AG_DEV AG_GLOBAL int32_t *g_rnd(const AgCtx<NS, AV> &c) { return (AG_GLOBAL int32_t *)(c.gs->rnd + (size_t)c.arena * 35); }

// Player iteration order after inserting this episode's pids into the (persistent) players map.
// R: GameState.hpp:44-46,61-67 (clear() keeps the bucket array), Engine.hpp:70-83.
template <int NS, bool AV> AG_DEV void compute_player_order(AgCtx<NS, AV> &c) {
  int P = c.P;
  if (P <= 1) { if (P == 1) SW(c, AR_ORDER0, 0); return; }   // (P == 0: bench/main.cpp's Tick/0 -- an engine without players; the map is untouched)
  auto scr = g_scratch(c);
  int bc = SR(c, AR_HM_BUCKETS), nr = SR(c, AR_HM_RESIZE);
  int *T = L_I(c, L_TMP);
  AG_SERIAL {
    AgHMap m; m.bucket_count = bc > 0 ? bc : 1; m.next_resize = nr; m.n = 0; m.head = -1;
    m.key = scr + AGM_HK; m.next = scr + AGM_HN; m.before = scr + AGM_HB;
    for (int i = 0; i < m.bucket_count; i++) m.before[i] = -2;
    for (int i = 0; i < P; i++) ag_hm_insert(m, PLS(c, i)[PL_PID]);  // node index == player slot
    int k = 0;
    for (int p = m.head; p != -1; p = m.next[p]) T[k++] = p;
    T[32] = m.bucket_count; T[33] = m.next_resize;
  }
  ag_mem_fence();
  for (int k = 0; k < P; k++) SW(c, AR_ORDER0 + k, ag_uni(T[k]));
  SW(c, AR_HM_BUCKETS, ag_uni(T[32])); SW(c, AR_HM_RESIZE, ag_uni(T[33]));
}

AG_DEV bool cell_can_eat_cell(unsigned a, unsigned b) { return a > 25u && can_eat_mass(a, b); }  // R: Entities.hpp:148-151

// Engine::players_collision for P > 1.  R: Engine.hpp:150-200, utils/collision_detection.hpp:10-64.
template <int NS, bool AV> AG_DEV void players_collision(AgCtx<NS, AV> &c) {
  int P = c.P;
  if (P <= 1) return;
  // flat cell list in player order (cells already id-sorted): offsets per order position
  // [P+1] prefix of cell counts and the slot of each order position, in LDS: lanes index them with lane-varying
  // subscripts (the register-resident uniform block can only be read with a wave-uniform index)
  int *off = L_I(c, L_TMP) + 52, *ordl = L_I(c, L_TMP) + 85;   // (33 and 32 words: up to AG_MAX_PLAYERS = 32 players; words 20 .. 51 belong to env_step)
  auto build_tables = [&]() -> int {
    int acc = 0; for (int k = 0; k < P; k++) { int slot = SR(c, AR_ORDER0 + k); AG_SERIAL { off[k] = acc; ordl[k] = slot; } acc += ag_uni(PLS(c, slot)[PL_NCELLS]); } AG_SERIAL { off[P] = acc; }
    ag_lds_order();
    return acc;
  };
  auto locate = [&](int g, int &p, int &i) { int k = 0; while (g >= off[k + 1]) k++; p = ordl[k]; i = g - off[k]; };
  // wave-parallel necessary condition: solve() can only report (collides && can_eat) pairs of different players
  bool any; int T;
  int chk_found = -1; (void)chk_found;
#ifndef AGAR_CPU_EMU
  {
    // Lane k < P holds order position k (its player slot and cell count, one parallel LDS read); every lane then finds its own cell of the flat
    // list by walking the P counts with v_readlane -- no LDS round trip per player.  Up to 64 cells in the arena (the usual case: bench/main.cpp's
    // Tick/30 has 30) lane g HOLDS cell g -- loaded and its radius looked up once -- and the eaten candidate b is broadcast with v_readlane: T
    // iterations of some twenty instructions.  The generic form walks T * T (eater, eaten) pairs with two searches through LDS tables, eight
    // LDS reads and two table look-ups EACH, behind a table build of one dependent LDS round trip per player: at 30 players that pre-test alone
    // was 44 % of the tick (plcol/foods 395 k of 907 k cycles per 4-tick launch, scripts/gpu_phase_multi.py).
    const int lane = AG_LANE;
    const int myslot = __builtin_amdgcn_ds_bpermute((AR_ORDER0 + (lane < P ? lane : 0)) << 2, c.S.v);     // word AR_ORDER0 + lane of the arena block
    const int mycnt = lane < P ? PLS(c, myslot)[PL_NCELLS] : 0;
    int acc = 0, pa = 0, ia = 0;
    if (__ballot(lane < P && mycnt != 1) == 0ull) { pa = lane < P ? myslot : 0; acc = P; }   // (every player has exactly one cell: lane k holds the cell of order position k)
    else for (int k = 0; k < P; k++) {
      const int nk = __builtin_amdgcn_readlane(mycnt, k), sk = __builtin_amdgcn_readlane(myslot, k);
      const bool mine = lane >= acc && lane < acc + nk;
      pa = mine ? sk : pa; ia = mine ? lane - acc : ia;
      acc += nk;
    }
    T = acc;
    if (T == 0) return;
    if (T <= 64) {
      const bool act = lane < T;
      const Cells A = cells_of(c, pa);
      const unsigned ma = act ? A.m[ia] : 0u; const float xa = act ? A.x[ia] : 0.0f, ya = act ? A.y[ia] : 0.0f; const float ra = act ? cell_rad(c, A, ia) : 0.0f;   // (the cell's cached table entry when it is valid: no trip to the table in L2)
      // can_eat(a, b) = a > 25 && (double)a > (double)b * 1.1 (Entities.hpp:148-151, Ball.hpp:45-47): the victim's side of it once per lane
      const double eat_thr = (double)ma * 1.1; const bool eater = act && ma > 25u; const double dma = (double)ma;
      const int tlo = (int)(unsigned)(__builtin_bit_cast(unsigned long long, eat_thr) & 0xffffffffull), thi = (int)(unsigned)(__builtin_bit_cast(unsigned long long, eat_thr) >> 32);
      bool hit = false;
      // (nobody can eat anybody while the heaviest cell is not 1.1 x the lightest one -- bench/main.cpp's ExampleBots, all of about the same mass)
      const unsigned mmax = wred_max(act ? ma : 0u), mmin = wred_min(act ? ma : 0xffffffffu);
      if (cell_can_eat_cell(mmax, mmin))
      for (int b = 0; b < T; b++) {
        const int pb = __builtin_amdgcn_readlane(pa, b);
        const float xb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xa), b)), yb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ya), b));
        const float rb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ra), b));
        const unsigned long long tb = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(thi, b) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readlane(tlo, b);
        hit = hit | (eater && pa != pb && dma > __builtin_bit_cast(double, tb) && collides(xa, ya, ra, xb, yb, rb));
      }
      any = ag_any(hit);
      if (!any) return;
      // The strip scan itself, a lane per eater (r05).  The test above is necessary, not sufficient: solve() finds a (collides, can eat) pair only
      // where its strips lead it -- the scan of a strip starts at a binary search of the y keys for the eater's x - r and ends at the first cell of
      // the eater's own player (utils/collision_detection.hpp:33-60) --, so two players can overlap tick after tick without an eat, and every such
      // tick went through the lane-0 replay below over HBM scratch: ~70 k cycles per tick, 250-300 k of the 330-400 k cycles of the slowest arenas
      // of a C1 launch (scripts/gpu_arena_spread_c1.py) -- the arenas the whole launch waits for.  Here every lane walks the strips of ITS cell over
      // tables in LDS and records what the scan would; an eat (rare) is then applied from these records.  The replay remains for what the tables do
      // not cover: more than 64 cells in the arena, a strip of more than 16, more than four victims of one eater, more than 13 eaters (and the host build).
      // Strips of up to 16 cells: std::sort is the plain insertion sort there, i.e. the stable order by the y key (fill order = flat order).
#ifndef AG_PLCOL_REPLAY_ONLY   // (differential builds: the replay alone, scripts/gpu_plcol_diff.py)
      {
        const float W_ = c.gs->g.W;
        float t_ = xa / W_; t_ = t_ * 100.0f;
        const int row_ = act ? f2i(t_) : -1; const bool inrow = act && row_ >= 0 && row_ <= 101; const int rowk = inrow ? row_ : -1;
        int cnt = 0, rank = 0;
        for (int b = 0; b < T; b++) {
          const int rb = __builtin_amdgcn_readlane(rowk, b); const float yb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ya), b));
          const bool same = inrow && rb == rowk;
          cnt += same ? 1 : 0; rank += (same && (yb < ya || (yb == ya && b < lane))) ? 1 : 0;
        }
        if (!ag_any(cnt > 16)) {
          // tables: strip counts [104] and members [102][16] (flat indices, bytes), the players' slots [64] -- in the candidate records of the pellet
          // replay (>= 2 KB, free between two pellets_eat) --; x, y, radius, mass by flat index in the created-cells block (free outside a turn)
          unsigned char *sc = (unsigned char *)(c.lds + ag_cand_off(c)), *se = sc + 104, *gpb = se + 102 * 16;
          float *gx = (float *)(c.lds + L_NEW), *gy = gx + 64, *gr = gy + 64; unsigned *gmm = (unsigned *)(gr + 64);
          if (lane < 26) ((int *)sc)[lane] = 0;
          ag_lds_order();
          if (inrow) { se[rowk * 16 + rank] = (unsigned char)lane; sc[rowk] = (unsigned char)cnt; }
          if (act) { gx[lane] = xa; gy[lane] = ya; gr[lane] = ra; gmm[lane] = ma; gpb[lane] = (unsigned char)pa; }
          ag_lds_order();
          // every lane records what the scan of ITS cell would (up to four victims, in the scan's order: strips ascending, then along the strip)
          int nh = 0, v0 = 0, v1 = 0, v2 = 0, v3 = 0;
          if (eater) {   // (cell_can_eat_cell needs the eater's mass > 25)
            const float left = xa - ra, right = xa + ra;
            float t = left / W_; t = t * 100.0f; int top = f2i(t);
            t = right / W_; t = t * 100.0f; int bottom = f2i(t);
            if (top < 0) top = 0;
            if (bottom > 101) bottom = 101;
            for (int i = top; i <= bottom; i++) {
              const int l = sc[i];
              if (l == 0) continue;
              const unsigned char *el = se + i * 16; int start = 0;
              for (int j = 4; j >= 0; j--) if (start + (1 << j) < l && gy[el[start + (1 << j)]] < left) start += (1 << j);
              for (int j = start; j < l; j++) {
                const int g2 = el[j];
                if ((int)gpb[g2] == pa) break;
                if (collides(xa, ya, ra, gx[g2], gy[g2], gr[g2]) && cell_can_eat_cell(ma, gmm[g2])) {
                  v0 = nh == 0 ? g2 : v0; v1 = nh == 1 ? g2 : v1; v2 = nh == 2 ? g2 : v2; v3 = nh == 3 ? g2 : v3; nh++;
                }
              }
            }
          }
          const unsigned long long hitm = __ballot(nh > 0);
          ag_lds_order();
#ifdef AG_PLCOL_CHECK   // (diagnostic build: the replay runs anyway and its result count is compared -- flag 0x4000 on a difference)
          chk_found = hitm != 0ull ? 1 : 0;
#else
          if (hitm == 0ull) return;
          const int ne_ = __popcll(hitm);
          if (!ag_any(nh > 4) && ne_ <= 13) {
            // An eat does happen (rare): applied from the lanes' records instead of the replay's HBM tables (~90 k cycles of dependent loads for
            // one eat -- after the scan above, the arenas a C1 launch waited for).  Engine.hpp:168-194: the results map (an unordered_map keyed by
            // the eater's flat index, filled in ascending order) is walked in ITS iteration order, each eater's victims in the order recorded;
            // eater and victim are looked up by the ids of the snapshot (lower_bound over cells that may have been erased meanwhile: kept as is).
            const int ida = act ? A.id[ia] : 0;
            int *Tm = L_I(c, L_TMP);
            if (ne_ > 1) {
              auto scr = g_scratch(c);
              AG_SERIAL {
                AgHMap rm; rm.bucket_count = 1; rm.next_resize = 0; rm.n = 0; rm.head = -1; rm.key = scr + AGM_HK; rm.next = scr + AGM_HN; rm.before = scr + AGM_HB;
                rm.before[0] = -2;
                for (unsigned long long m = hitm; m; m &= m - 1ull) ag_hm_insert(rm, (int)__builtin_ctzll(m));
                int k = 0;
                for (int nd = rm.head; nd != -1; nd = rm.next[nd]) Tm[k++] = rm.key[nd];
              }
              ag_mem_fence();
            }
            for (int k = 0; k < ne_; k++) {
              const int id = ne_ == 1 ? (int)__builtin_ctzll(hitm) : ag_uni(Tm[k]);
              const int nhid = __builtin_amdgcn_readlane(nh, id), pe = __builtin_amdgcn_readlane(pa, id), gide = __builtin_amdgcn_readlane(ida, id);
              for (int e = 0; e < nhid; e++) {
                const int v = e == 0 ? __builtin_amdgcn_readlane(v0, id) : e == 1 ? __builtin_amdgcn_readlane(v1, id) : e == 2 ? __builtin_amdgcn_readlane(v2, id) : __builtin_amdgcn_readlane(v3, id);
                const int pv = __builtin_amdgcn_readlane(pa, v), gidv = __builtin_amdgcn_readlane(ida, v); const unsigned gmv = (unsigned)__builtin_amdgcn_readlane((int)ma, v);
                AG_SERIAL {
                  Cells E = cells_of(c, pe), V = cells_of(c, pv);
                  int *PE = PLS(c, pe), *PV = PLS(c, pv);
                  int ne = PE[PL_NCELLS], it_ = 0;
                  while (it_ < ne && E.id[it_] < gide) it_++;  // lower_bound by id
                  if (it_ != ne) { E.m[it_] = clamp_mass(E.m[it_] + gmv); PE[PL_CELLS_EATEN] += 1; }
                  int nv = PV[PL_NCELLS], ei = 0;
                  while (ei < nv && V.id[ei] < gidv) ei++;
                  if (ei != nv) {  // vector::erase
                    for (int j = ei; j + 1 < nv; j++) { V.x[j] = V.x[j + 1]; V.y[j] = V.y[j + 1]; V.vx[j] = V.vx[j + 1]; V.vy[j] = V.vy[j + 1]; V.sx[j] = V.sx[j + 1]; V.sy[j] = V.sy[j + 1];
                      V.m[j] = V.m[j + 1]; V.id[j] = V.id[j + 1]; V.dl[j] = V.dl[j + 1]; V.cmc[j] = 0u; }
                    PV[PL_NCELLS] = nv - 1;
                  }
                }
                ag_lds_order();
              }
            }
            return;
          }
#endif
        }
      }
#endif
      (void)build_tables();      // the rare path below reads the LDS tables
    } else {
      (void)build_tables();
      any = true;                // (more than 64 cells: the generic pre-test below decides)
    }
  }
  if (T > 64)
#else
  T = build_tables();
  if (T == 0) return;
#endif
  any = wave_any(T * T, [&](int q) {
    int a = q / T, b = q - a * T, pa, ia, pb, ib; locate(a, pa, ia); locate(b, pb, ib);
    if (pa == pb) return false;
    Cells A = cells_of(c, pa), B = cells_of(c, pb);
    return cell_can_eat_cell(A.m[ia], B.m[ib]) && collides(A.x[ia], A.y[ia], radius_of(c, A.m[ia]), B.x[ib], B.y[ib], radius_of(c, B.m[ib]));
  });
  if (!any) return;
  auto scr = g_scratch(c); float W = c.gs->g.W; auto lut_r = g_lut_r(c);
  // snapshot the gallery (cells_per_player holds COPIES: masses / ids as of now)
  AG_LANES(g, T) {
    int p, i; locate(g, p, i); Cells s = cells_of(c, p);
    scr[AGM_GP + g] = p; scr[AGM_GI + g] = i; scr[AGM_GM + g] = (int)s.m[i]; scr[AGM_GID + g] = s.id[i];
    scr[AGM_GX + g] = f2u(s.x[i]); scr[AGM_GY + g] = f2u(s.y[i]);
    float t = s.x[i] / W; t = t * 100.0f; scr[AGM_ROW + g] = f2i(t);
    scr[AGM_HAS + g] = 0;
  }
  ag_mem_fence();
  int *Tm = L_I(c, L_TMP);
  AG_SERIAL {
    AG_GLOBAL int32_t *gp = scr + AGM_GP, *gm = scr + AGM_GM, *gid = scr + AGM_GID, *row = scr + AGM_ROW, *has = scr + AGM_HAS;
    AG_GLOBAL int32_t *rs = scr + AGM_RS, *rc = scr + AGM_RC, *fill = scr + AGM_FILL, *rq = scr + AGM_RQ, *rg = scr + AGM_RG, *gxs = scr + AGM_GX, *gys = scr + AGM_GY;
    AgSortArr it; it.k = scr + AGM_IK; it.v = scr + AGM_IV; it.stk = scr + AGM_STK;
    for (int r = 0; r < 104; r++) { rc[r] = 0; fill[r] = 0; }
    for (int i = 0; i < T; i++) if (row[i] >= 0 && row[i] <= 101) rc[row[i]]++;
    rs[0] = 0; for (int r = 0; r < 102; r++) rs[r + 1] = rs[r] + rc[r];
    for (int i = 0; i < T; i++) if (row[i] >= 0 && row[i] <= 101) { int r = row[i]; it.k[rs[r] + fill[r]] = gys[i]; it.v[rs[r] + fill[r]] = i; fill[r]++; }
    for (int r = 0; r < 102; r++) ag_std_sort(it, rs[r], rs[r] + rc[r]);
    AgHMap rm; rm.bucket_count = 1; rm.next_resize = 0; rm.n = 0; rm.head = -1; rm.key = scr + AGM_HK; rm.next = scr + AGM_HN; rm.before = scr + AGM_HB;
    rm.before[0] = -2;
    int nres = 0, flags = 0;
    for (int id = 0; id < T; id++) {
      float qx = u2f(gxs[id]), qy = u2f(gys[id]); unsigned qm = (unsigned)gm[id]; float qr = lut(lut_r, qm);
      float left = qx - qr, right = qx + qr;
      float t = left / W; t = t * 100.0f; int top = f2i(t);
      t = right / W; t = t * 100.0f; int bottom = f2i(t);
      for (int i = top; i <= bottom; i++) {
        if (i < 0 || i > 101 || rc[i] == 0) continue;
        int l = rc[i], base = rs[i], start = 0;
        for (int j = 10; j >= 0; j--) if (start + (1 << j) < l && u2f(it.k[base + start + (1 << j)]) < left) start += (1 << j);
        for (int j = start; j < l; j++) {
          int gi2 = it.v[base + j];
          if (gp[id] == gp[gi2]) break;
          unsigned om = (unsigned)gm[gi2];
          if (collides(qx, qy, qr, u2f(gxs[gi2]), u2f(gys[gi2]), lut(lut_r, om)) && cell_can_eat_cell(qm, om)) {
            if (!has[id]) { has[id] = 1; ag_hm_insert(rm, id); }
            if (nres < AGM_RES) { rq[nres] = id; rg[nres] = gi2; nres++; } else flags |= 8;
          }
        }
      }
    }
    // apply in the results map's iteration order.  R: Engine.hpp:168-194
    for (int nd = rm.head; nd != -1; nd = rm.next[nd]) {
      int id = rm.key[nd];
      for (int e = 0; e < nres; e++) {
        if (rq[e] != id) continue;
        int v = rg[e];
        int pe = gp[id], pv = gp[v];
        Cells E = cells_of(c, pe), V = cells_of(c, pv);
        int *PE = PLS(c, pe), *PV = PLS(c, pv);
        int ne = PE[PL_NCELLS], it_ = 0;
        while (it_ < ne && E.id[it_] < gid[id]) it_++;  // lower_bound by id
        if (it_ != ne) { E.m[it_] = clamp_mass(E.m[it_] + (unsigned)gm[v]); PE[PL_CELLS_EATEN] += 1; }
        int nv = PV[PL_NCELLS], ei = 0;
        while (ei < nv && V.id[ei] < gid[v]) ei++;
        if (ei != nv) {  // vector::erase
          for (int j = ei; j + 1 < nv; j++) { V.x[j] = V.x[j + 1]; V.y[j] = V.y[j + 1]; V.vx[j] = V.vx[j + 1]; V.vy[j] = V.vy[j + 1]; V.sx[j] = V.sx[j + 1]; V.sy[j] = V.sy[j + 1];
            V.m[j] = V.m[j + 1]; V.id[j] = V.id[j + 1]; V.dl[j] = V.dl[j + 1]; V.cmc[j] = 0u; }
          PV[PL_NCELLS] = nv - 1;
        }
      }
    }
    Tm[0] = flags;
#ifdef AG_PLCOL_CHECK
    Tm[1] = nres;
#endif
  }
  ag_mem_fence();
  int fl = ag_uni(Tm[0]); if (fl) flag(c, (unsigned)fl);
#ifdef AG_PLCOL_CHECK
  if (chk_found >= 0 && (ag_uni(Tm[1]) > 0) != (chk_found != 0)) flag(c, 0x4000u);
#endif
}

// ---- bots.  R: agario/bots/Bot.hpp, HungryBot.hpp, HungryShyBot.hpp, AggressiveBot.hpp, AggressiveShyBot.hpp -------------------
enum { AG_KIND_AGENT = 0, AG_KIND_HUNGRY = 1, AG_KIND_HUNGRY_SHY = 2, AG_KIND_AGGRESSIVE = 3, AG_KIND_AGGRESSIVE_SHY = 4,
       AG_KIND_EXAMPLE = 5 };   // agario/bots/ExampleBot.hpp:45-51: action none, target = its own location

// uniform centroid / mass of player slot p (Player::x/y/mass, core/Player.hpp:102-126)
template <int NS, bool AV> AG_DEV void player_centroid(const AgCtx<NS, AV> &c, int p, float &px, float &py, unsigned &mass) {
  int n = ag_uni(PLS(c, p)[PL_NCELLS]); Cells s = cells_of(c, p); float sx = 0.0f, sy = 0.0f; unsigned tm = 0;
  for (int i = 0; i < n; i++) { unsigned m = ag_uniu(s.m[i]); float fm = (float)m; float t = ag_unif(s.x[i]) * fm; sx += t; t = ag_unif(s.y[i]) * fm; sy += t; tm += m; }
  px = ag_divf(sx, (float)tm); py = ag_divf(sy, (float)tm); mass = tm;
}
AG_DEV float dist_to(float ax, float ay, float bx, float by) { float dx = fabsf(bx - ax), dy = fabsf(by - ay); float p = dx * dx, q = dy * dy; return ag_sqrtf(p + q); }

// Bot::nearest_pellet (Bot.hpp:90-127): first pellet (lowest index) attaining the minimum distance among dist > 0.01
template <int NS, bool AV> AG_DEV void nearest_pellet(AgCtx<NS, AV> &c, float sx, float sy, float &ox, float &oy) {
  int np = SR(c, AR_NPEL);
  if (np == 0) {
    auto st = g_rnd(c); int *T = L_I(c, L_TMP); int Wi = f2i(c.gs->g.W);
    AG_SERIAL { T[0] = ag_rand_next(st) % Wi; T[1] = ag_rand_next(st) % Wi; }
    ag_mem_fence();
    ox = (float)ag_uni(T[0]); oy = (float)ag_uni(T[1]); return;
  }
  // ONE pass (r05: there were two, each with the correctly rounded square root per pellet): every lane keeps its nearest pellet and that one's
  // index (its indices ascend, the strict < keeps the first), the wave takes the least distance and then the least index among the lanes that
  // attain it.  (double)d > 0.01 of the reference is d > 0.01f exactly: the float nearest to 0.01 lies below it.
  unsigned best = UINT_MAX, bi = UINT_MAX;
  {
    unsigned lane_best = UINT_MAX, lane_bi = UINT_MAX;
    AG_PEL_FOR(s, lane, i) { if (i < np) { float d = dist_to(PELX(c, s, lane), PELY(c, s, lane), sx, sy); if (d > 0.01f) { unsigned b = (unsigned)f2u(d); if (b < lane_best) { lane_best = b; lane_bi = (unsigned)i; } } } }
#ifdef AGAR_CPU_EMU
    best = lane_best; bi = lane_bi;
#else
    best = wred_min(lane_best); bi = wred_min(lane_best == best ? lane_bi : UINT_MAX);
#endif
  }
  if (best == UINT_MAX) { ox = 0.0f; oy = 0.0f; return; }  // default-constructed Location, min_distance stays max
  pel_get(c, (int)bi, ox, oy);
}
#ifdef AGAR_CPU_EMU
// the host loops of AG_PEL_FOR visit every (slot, lane): the per-"lane" minima above are already global minima
#endif

template <int NS, bool AV> AG_DEV bool bot_shy_check(AgCtx<NS, AV> &c, int p, float sx, float sy, float &tx, float &ty) {
  // R: HungryShyBot.hpp:26-40.  `mass()` there is the value-initialised typedef agario::mass (== 0): an unqualified
  // name in a template with a dependent base -- so the test is "other player alive".
  for (int k = 0; k < c.P; k++) {
    int o = SR(c, AR_ORDER0 + k);
    if (o == p) continue;
    float ox, oy; unsigned om; player_centroid(c, o, ox, oy, om);
    float d = dist_to(sx, sy, ox, oy);
    if (d < 25.0f && om > 0u) { float dx = ox - sx, dy = oy - sy; tx = sx - dx; ty = sy - dy; return true; }
  }
  return false;
}
template <int NS, bool AV> AG_DEV bool bot_aggressive_check(AgCtx<NS, AV> &c, int p, float sx, float sy, float &tx, float &ty) {
  // R: AggressiveBot.hpp:30-52, Bot.hpp:52-88
  Cells me = cells_of(c, p); int n = ag_uni(PLS(c, p)[PL_NCELLS]);
  unsigned lm = 0; for (int i = 0; i < n; i++) { unsigned m = ag_uniu(me.m[i]); if (i == 0 || m > lm) lm = m; }
  for (int k = 0; k < c.P; k++) {
    int o = SR(c, AR_ORDER0 + k);
    if (o == p) continue;
    float ox, oy; unsigned om; player_centroid(c, o, ox, oy, om);
    float d = dist_to(sx, sy, ox, oy);
    if (d <= 20.0f) {
      Cells oc = cells_of(c, o); int no = ag_uni(PLS(c, o)[PL_NCELLS]);
      unsigned edible = 0; float ax = 0.0f, ay = 0.0f;
      for (int i = 0; i < no; i++) { unsigned m = ag_uniu(oc.m[i]); if (cell_can_eat_cell(lm, m)) { float fm = (float)m; float qx = ag_unif(oc.x[i]) * fm, qy = ag_unif(oc.y[i]) * fm; ax += qx; ay += qy; edible += m; } }
      if (edible > 0) {
        float fm = (float)edible; float qx = ag_divf(ax, fm), qy = ag_divf(ay, fm);
        float dsx = qx - sx, dsy = qy - sy; float ex = dsx * 3.0f, ey = dsy * 3.0f;
        tx = sx + ex; ty = sy + ey; return true;
      }
    }
  }
  return false;
}
#ifndef AGAR_CPU_EMU
// The two checks above for all other players at once, a lane per order position (r05): each lane sums ITS player's centroid (and, for the
// aggressive kinds, the part of it the bot's largest cell can eat) in cell order -- the same fp32 sums --, and the first position that
// satisfies the test is taken with a ballot.  The serial forms walk the players one after the other with three dependent LDS reads per cell,
// twice for the aggressive-shy kind: 8.5 k cycles per decision, 42 k of the 125 k cycles of a C1 arena-step that holds a bot tick
// (scripts/gpu_arena_spread_c1.py).
template <int NS, bool AV> AG_DEV bool bot_checks_lanes(AgCtx<NS, AV> &c, int p, float sx, float sy, bool shy, bool aggressive, float &tx, float &ty) {
  const int lane = AG_LANE, P = c.P;
  const int o = __builtin_amdgcn_ds_bpermute((AR_ORDER0 + (lane < P ? lane : 0)) << 2, c.S.v);   // word AR_ORDER0 + lane of the arena block
  const bool on = lane < P && o != p;
  const int no = on ? PLS(c, o)[PL_NCELLS] : 0;
  const Cells oc = cells_of(c, on ? o : p);
  unsigned lm = 0u;
  if (aggressive) { const int n = ag_uni(PLS(c, p)[PL_NCELLS]); const Cells me = cells_of(c, p); lm = wave_max(n, [&](int i) { return me.m[i]; }); }
  float cx = 0.0f, cy = 0.0f, ax = 0.0f, ay = 0.0f; unsigned tm = 0u, edible = 0u;
  const int nmax = (int)wred_max((unsigned)no);
  for (int i = 0; i < nmax; i++) {
    if (i < no) {
      const unsigned m = oc.m[i]; const float fm = (float)m; const float qx = oc.x[i] * fm, qy = oc.y[i] * fm;
      cx += qx; cy += qy; tm += m;
      if (aggressive && cell_can_eat_cell(lm, m)) { ax += qx; ay += qy; edible += m; }
    }
  }
  const float ox = ag_divf(cx, (float)tm), oy = ag_divf(cy, (float)tm);
  const float d = dist_to(sx, sy, ox, oy);
  if (shy) {
    const unsigned long long hm = __ballot(on && d < 25.0f && tm > 0u);
    if (hm) {
      const int k = (int)__builtin_ctzll(hm);
      const float hx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ox), k)), hy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(oy), k));
      const float dx = hx - sx, dy = hy - sy; tx = sx - dx; ty = sy - dy; return true;
    }
  }
  if (aggressive) {
    const unsigned long long hm = __ballot(on && d <= 20.0f && edible > 0u);
    if (hm) {
      const int k = (int)__builtin_ctzll(hm);
      const float fm = (float)edible; const float qx = ag_divf(ax, fm), qy = ag_divf(ay, fm);
      const float hx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx), k)), hy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy), k));
      const float dsx = hx - sx, dsy = hy - sy; const float ex = dsx * 3.0f, ey = dsy * 3.0f;
      tx = sx + ex; ty = sy + ey; return true;
    }
  }
  return false;
}
#endif
// Player::take_action override of the bot in slot p (called every 10th tick).  Operates on the register copy c.PB.
template <int NS, bool AV> AG_DEV void bot_take_action(AgCtx<NS, AV> &c, int p) {
  int kind = PR(c, PL_KIND);
  if (kind == AG_KIND_AGENT) return;
  float sx, sy; unsigned sm; player_centroid(c, p, sx, sy, sm);
  float tx = PRF(c, PL_TX), ty = PRF(c, PL_TY); int action = PR(c, PL_ACTION);
  bool done = false;
  if (kind == AG_KIND_EXAMPLE) { PW(c, PL_ACTION, 0); PW(c, PL_TX, f2u(sx)); PW(c, PL_TY, f2u(sy)); return; }   // R: ExampleBot.hpp:45-51
#ifndef AGAR_CPU_EMU
  if (kind == AG_KIND_HUNGRY_SHY) { action = 0; done = bot_checks_lanes(c, p, sx, sy, true, false, tx, ty); }
  else if (kind == AG_KIND_AGGRESSIVE) done = bot_checks_lanes(c, p, sx, sy, false, true, tx, ty);
  else if (kind == AG_KIND_AGGRESSIVE_SHY) done = bot_checks_lanes(c, p, sx, sy, true, true, tx, ty);
#else
  if (kind == AG_KIND_HUNGRY_SHY) { action = 0; done = bot_shy_check(c, p, sx, sy, tx, ty); }
  else if (kind == AG_KIND_AGGRESSIVE) done = bot_aggressive_check(c, p, sx, sy, tx, ty);
  else if (kind == AG_KIND_AGGRESSIVE_SHY) { done = bot_shy_check(c, p, sx, sy, tx, ty); if (!done) done = bot_aggressive_check(c, p, sx, sy, tx, ty); }
#endif
  if (!done) { action = 0; nearest_pellet(c, sx, sy, tx, ty); }
  PW(c, PL_ACTION, action); PW(c, PL_TX, f2u(tx)); PW(c, PL_TY, f2u(ty));
}
