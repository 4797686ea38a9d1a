/* TEST INFRASTRUCTURE ONLY -- see agar_oracle.h for scope and parity status (PINNED against
 * oracle/_ref/libagar_ref.so and tests/golden/).
 *
 * Plain-C sequential restatement of the reference engine.  "R:" comments cite the reference
 * (paths relative to /root/reference).  fp32 expressions are written one operation per statement
 * where the reference's numWrapper<float> arithmetic (agario/core/num_wrapper.hpp) fixes the
 * evaluation order; build with -ffp-contract=off (oracle/Makefile).
 *
 * Third-party behaviour the reference relies on and that is restated here (GCC 11.4 libstdc++,
 * glibc 2.35 -- the toolchain that built oracle/_ref):
 *   - std::mt19937_64 + std::uniform_real_distribution<float>  (one 64-bit draw per float)
 *   - std::unordered_map iteration order (_Hashtable insert-at-bucket-begin + prime rehash policy)
 *   - std::sort (introsort, threshold 16)
 *   - glibc rand()/srand() (TYPE_3 additive feedback generator)
 */
#include "agar_oracle.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* R: agario/core/settings.hpp:5-50, agario/core/Entities.hpp:9-18 */
#define CELL_MIN_SIZE 25u
#define CELL_MAX_SPEED 300
#define CELL_SPLIT_MINIMUM 50u
#define SPLIT_DECELERATION 80.0f
#define FOOD_SPEED 100.0f
#define FOOD_DECEL 80.0f
#define CELL_EAT_MARGIN 1.1
#define CELL_POP_REDUCTION 2.0f
#define CELL_POP_SIZE 25u
#define PLAYER_CELL_LIMIT 14
#define PLAYER_RATE 0.002
#define NUMBER_OF_FOOD_HITS 7
#define MAX_MASS_IN_THE_GAME 22500u
#define NEW_MASS_IF_NO_SPLIT 22000u
#define ANTI_TEAM_ACTIVATION_TIME 60
#define PELLET_MASS 1u
#define FOOD_MASS 10u
#define VIRUS_INITIAL_MASS 100u
#define CELL_EAT_REQUIREMENT 25u
#define SHY_RADIUS 25.0f
#define AGGRESSIVE_RADIUS 20.0f

enum { KIND_AGENT = 0, KIND_HUNGRY = 1, KIND_HUNGRY_SHY = 2, KIND_AGGRESSIVE = 3, KIND_AGGRESSIVE_SHY = 4,
       KIND_EXAMPLE = 5 }; /* R: agario/bots/ExampleBot.hpp:45-51 */

typedef struct { float x, y; int id; } OPellet;
typedef struct { float x, y, vx, vy; unsigned mass; int hits; int id; } OVirus;
typedef struct { float x, y, vx, vy; int id; } OFood;
typedef struct { float x, y, vx, vy, svx, svy; unsigned mass; int id; long long deadline; } OCell;

typedef struct {
  int pid, kind, is_bot;
  OCell *cells; int n_cells, cap_cells;
  int action; float tx, ty;
  unsigned long split_cd, feed_cd;
  int *vticks; int n_vticks, cap_vticks;
  float anti_team; int elapsed, last_decay;
  int food_eaten; unsigned highest_mass; int cells_eaten, viruses_eaten; unsigned min_mass_cell;
} OPlayer;

/* libstdc++ _Hashtable order emulation (unique integer keys, identity hash) */
typedef struct {
  int bucket_count, next_resize, n, head, cap;
  int *key, *next;      /* per node */
  int *before; int cap_b; /* per bucket: -2 empty, -1 before_begin, else node index */
} HMap;

struct OArena {
  /* config (R: GameState.hpp:15-39, BaseEnvironment.hpp:36-67) */
  int num_agents, ticks_per_step, num_bots, reward_type, c_death, mode;
  int example_bots; /* R: bench/main.cpp:21-24 -- ExampleBots added to the freshly reset engine */
  float W, H; size_t target_pellets, target_viruses; int pellet_regen;
  int mass_decay, squared, agent_mass, regen; /* R: Engine.hpp:362-416 */
  long long recomb_ticks, clock; /* virtual steady clock in ticks */
  /* state */
  OPellet *pellets; int n_pellets, cap_pellets;
  OVirus *viruses; int n_viruses, cap_viruses;
  OFood *foods; int n_foods, cap_foods;
  OPlayer *players; int n_players, cap_players; /* node index == players[] index */
  HMap pmap;
  uint64_t mt[313]; int32_t rnd[35];
  unsigned long ticks; int next_pid; int id_counter; int main_agent_pid;
  int *pids; uint8_t *dones; int respawned_flag;
  int screen_hook; /* ScreenEnvironment::_partial_observation: a dead agent is respawned right after the ticks (E5) */
  /* event log of last tick */
  int *ev_p; int n_ev_p, cap_ev_p; int *ev_v; int n_ev_v, cap_ev_v;
};

#define GROW(ptr, n, cap, type) do { if ((n) >= (cap)) { (cap) = (cap) ? (cap) * 2 : 16; (ptr) = (type *)realloc((ptr), sizeof(type) * (size_t)(cap)); } } while (0)

/* ---- C++ std::min/max/clamp on floats, NaN behaviour included (R: core/utils.hpp:19-21) ------- */
static inline float smaxf(float a, float b) { return (a < b) ? b : a; }
static inline float sminf(float a, float b) { return (b < a) ? b : a; }
static inline float clampf(float x, float lo, float hi) { return smaxf(sminf(x, hi), lo); }
/* static_cast<int>(float) as compiled for x86-64 (cvttss2si: NaN/out of range -> INT_MIN) */
static inline int f2i(float f) { if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT_MIN; return (int)f; }

/* ---- mt19937_64 (libstdc++ <random>) ---------------------------------------------------------- */
void ora_mt_seed(uint64_t *mt, uint64_t seed) {
  mt[0] = seed;
  for (int i = 1; i < 312; i++) mt[i] = 6364136223846793005ULL * (mt[i - 1] ^ (mt[i - 1] >> 62)) + (uint64_t)i;
  mt[312] = 312;
}
uint64_t ora_mt_next(uint64_t *mt) {
  if (mt[312] >= 312) {
    const uint64_t UM = 0xFFFFFFFF80000000ULL, LM = 0x7FFFFFFFULL, A = 0xB5026F5AA96619E9ULL;
    for (int i = 0; i < 312; i++) {
      uint64_t y = (mt[i] & UM) | (mt[(i + 1) % 312] & LM);
      mt[i] = mt[(i + 156) % 312] ^ (y >> 1) ^ ((y & 1) ? A : 0);
    }
    mt[312] = 0;
  }
  uint64_t z = mt[mt[312]++];
  z ^= (z >> 29) & 0x5555555555555555ULL;
  z ^= (z << 17) & 0x71D67FFFEDA60000ULL;
  z ^= (z << 37) & 0xFFF7EEE000000000ULL;
  z ^= (z >> 43);
  return z;
}
/* std::uniform_real_distribution<float>(lo,hi)(mt19937_64): generate_canonical<float,24> takes one
 * 64-bit draw: float(u) / 2^64f, clamped below 1, then * (hi-lo) + lo.  R: utils/random.hpp:6-20 */
float ora_uniform_float(uint64_t *mt, float lo, float hi) {
  float s = (float)ora_mt_next(mt);
  float r = s / 18446744073709551616.0f;
  if (r >= 1.0f) r = nextafterf(1.0f, 0.0f);
  float range = hi - lo;
  float v = r * range;
  return v + lo;
}

/* ---- glibc rand()/srand(), TYPE_3 ------------------------------------------------------------- */
void ora_rand_seed(int32_t *st, unsigned seed) {
  int32_t r[344];
  if (seed == 0) seed = 1;
  r[0] = (int32_t)seed;
  for (int i = 1; i < 31; i++) {
    long long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
    long long w = 16807 * lo - 2836 * hi;
    if (w < 0) w += 2147483647;
    r[i] = (int32_t)w;
  }
  for (int i = 31; i < 34; i++) r[i] = r[i - 31];
  for (int i = 34; i < 344; i++) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
  /* keep the last 34 values as a ring */
  for (int i = 0; i < 34; i++) st[i] = r[310 + i];
  st[34] = 0; /* ring position of the oldest (i-34) entry */
}
int ora_rand_next(int32_t *st) {
  /* ring holds o[k-34..k-1]; new = o[k-31] + o[k-3] */
  int p = st[34];
  uint32_t v = (uint32_t)st[(p + 3) % 34] + (uint32_t)st[(p + 31) % 34];
  st[p] = (int32_t)v;
  st[34] = (p + 1) % 34;
  return (int)(v >> 1);
}

/* ---- libstdc++ unordered_map order ------------------------------------------------------------ */
static const int PRIMES[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97,
  103, 109, 113, 127, 137, 139, 149, 157, 167, 179, 193, 199, 211, 227, 241, 257, 277, 293, 313, 337, 359, 383, 409,
  439, 467, 503, 541, 577, 619, 661, 709, 761, 823, 887, 953, 1031, 1109, 1193, 1289, 1381, 1493, 1613, 1741, 1879,
  2029, 2179, 2357, 2549, 2753, 2971, 3209, 3469, 3739, 4027, 4349, 4703, 5087, 5503, 5953, 6427, 6949, 7517, 8123,
  8783, 9497, 10273, 11113, 12011, 12983, 14033, 15173, 16411};
static int next_bkt(int n, int *next_resize) { /* _Prime_rehash_policy::_M_next_bkt */
  static const unsigned char fast[] = {2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13};
  int b;
  if (n < 14) b = fast[n];
  else { b = 16411; for (size_t i = 0; i < sizeof(PRIMES) / sizeof(int); i++) if (PRIMES[i] >= n) { b = PRIMES[i]; break; } }
  *next_resize = b; /* floor(b * max_load_factor 1.0) */
  return b;
}
static void hm_init(HMap *m) { memset(m, 0, sizeof(*m)); m->bucket_count = 1; m->next_resize = 0; m->head = -1; }
static void hm_free(HMap *m) { free(m->key); free(m->next); free(m->before); }
static void hm_set_buckets(HMap *m, int nb) {
  if (nb > m->cap_b) { m->cap_b = nb; m->before = (int *)realloc(m->before, sizeof(int) * (size_t)nb); }
  for (int i = 0; i < nb; i++) m->before[i] = -2;
  m->bucket_count = nb;
}
static void hm_clear(HMap *m) { /* keeps bucket array size + rehash policy, like unordered_map::clear() */
  m->n = 0; m->head = -1;
  if (m->before) for (int i = 0; i < m->bucket_count; i++) m->before[i] = -2;
}
static inline int *hm_nextp(HMap *m, int before) { return before == -1 ? &m->head : &m->next[before]; }
static void hm_rehash(HMap *m, int nb) { /* _M_rehash_aux(n, true_type) */
  int p = m->head;
  hm_set_buckets(m, nb);
  m->head = -1;
  int bbegin = 0;
  while (p != -1) {
    int nx = m->next[p];
    int b = (int)((unsigned)m->key[p] % (unsigned)nb);
    if (m->before[b] == -2) {
      m->next[p] = m->head; m->head = p; m->before[b] = -1;
      if (m->next[p] != -1) m->before[bbegin] = p;
      bbegin = b;
    } else {
      int *bn = hm_nextp(m, m->before[b]);
      m->next[p] = *bn; *bn = p;
    }
    p = nx;
  }
}
static int hm_insert(HMap *m, int key) { /* _M_insert_unique_node; key assumed absent */
  if (m->n >= m->cap) { m->cap = m->cap ? m->cap * 2 : 16; m->key = (int *)realloc(m->key, sizeof(int) * (size_t)m->cap); m->next = (int *)realloc(m->next, sizeof(int) * (size_t)m->cap); }
  if (!m->before) hm_set_buckets(m, m->bucket_count);
  /* _M_need_rehash(bucket_count, element_count, 1) */
  if (m->n + 1 > m->next_resize) {
    int lhs = m->n + 1, floor11 = m->next_resize ? 0 : 11;
    double min_bkts = (double)(lhs > floor11 ? lhs : floor11) / 1.0;
    if (min_bkts >= (double)m->bucket_count) {
      int want = (int)floor(min_bkts) + 1, grow = m->bucket_count * 2;
      int nb = next_bkt(want > grow ? want : grow, &m->next_resize);
      hm_rehash(m, nb);
    } else {
      m->next_resize = (int)floor((double)m->bucket_count * 1.0);
    }
  }
  int node = m->n++;
  m->key[node] = key;
  int b = (int)((unsigned)key % (unsigned)m->bucket_count);
  if (m->before[b] != -2) { /* _M_insert_bucket_begin */
    int *bn = hm_nextp(m, m->before[b]);
    m->next[node] = *bn; *bn = node;
  } else {
    m->next[node] = m->head; m->head = node;
    if (m->next[node] != -1) m->before[(unsigned)m->key[m->next[node]] % (unsigned)m->bucket_count] = node;
    m->before[b] = -1;
  }
  return node;
}
int ora_hash_order(const int *keys, int n, int *order_out, int *bucket_count_io, int *next_resize_io) {
  HMap m; hm_init(&m);
  m.bucket_count = *bucket_count_io; m.next_resize = *next_resize_io;
  for (int i = 0; i < n; i++) hm_insert(&m, keys[i]);
  int k = 0;
  for (int p = m.head; p != -1; p = m.next[p]) order_out[k++] = m.key[p];
  *bucket_count_io = m.bucket_count; *next_resize_io = m.next_resize;
  hm_free(&m);
  return k;
}

/* ---- libstdc++ std::sort (bits/stl_algo.h: __introsort_loop / __final_insertion_sort) ---------- */
typedef struct { float k; int v; } SItem;
#define SLESS(a, b) ((a).k < (b).k)
static void s_swap(SItem *a, SItem *b) { SItem t = *a; *a = *b; *b = t; }
static void s_unguarded_linear_insert(SItem *last) {
  SItem val = *last; SItem *next = last - 1;
  while (SLESS(val, *next)) { *last = *next; last = next; --next; }
  *last = val;
}
static void s_insertion_sort(SItem *first, SItem *last) {
  if (first == last) return;
  for (SItem *i = first + 1; i != last; ++i) {
    if (SLESS(*i, *first)) { SItem val = *i; memmove(first + 1, first, (size_t)(i - first) * sizeof(SItem)); *first = val; }
    else s_unguarded_linear_insert(i);
  }
}
static void s_adjust_heap(SItem *first, long hole, long len, SItem value) {
  const long top = hole; long child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (SLESS(first[child], first[child - 1])) child--;
    first[hole] = first[child]; hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); first[hole] = first[child - 1]; hole = child - 1; }
  long parent = (hole - 1) / 2; /* __push_heap */
  while (hole > top && SLESS(first[parent], value)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
  first[hole] = value;
}
static void s_heapsort(SItem *first, SItem *last) { /* partial_sort(first,last,last) */
  long len = last - first;
  if (len >= 2) { long parent = (len - 2) / 2; for (;;) { SItem v = first[parent]; s_adjust_heap(first, parent, len, v); if (parent == 0) break; parent--; } }
  while (last - first > 1) { --last; SItem v = *last; *last = *first; s_adjust_heap(first, 0, last - first, v); }
}
static void s_introsort_loop(SItem *first, SItem *last, long depth) {
  while (last - first > 16) {
    if (depth == 0) { s_heapsort(first, last); return; }
    --depth;
    SItem *mid = first + (last - first) / 2, *a = first + 1, *b = mid, *c = last - 1; /* __move_median_to_first */
    if (SLESS(*a, *b)) { if (SLESS(*b, *c)) s_swap(first, b); else if (SLESS(*a, *c)) s_swap(first, c); else s_swap(first, a); }
    else if (SLESS(*a, *c)) s_swap(first, a); else if (SLESS(*b, *c)) s_swap(first, c); else s_swap(first, b);
    SItem *lo = first + 1, *hi = last; /* __unguarded_partition(first+1, last, first) */
    for (;;) {
      while (SLESS(*lo, *first)) ++lo;
      --hi;
      while (SLESS(*first, *hi)) --hi;
      if (!(lo < hi)) break;
      s_swap(lo, hi); ++lo;
    }
    s_introsort_loop(lo, last, depth);
    last = lo;
  }
}
static void s_sort(SItem *first, SItem *last) {
  if (first == last) return;
  long n = last - first, lg = 0; while ((1L << (lg + 1)) <= n) lg++;
  s_introsort_loop(first, last, 2 * lg);
  if (last - first > 16) { s_insertion_sort(first, first + 16); for (SItem *i = first + 16; i != last; ++i) s_unguarded_linear_insert(i); }
  else s_insertion_sort(first, last);
}
void ora_std_sort_by_float(float *keys, int *payload, int n) {
  SItem *it = (SItem *)malloc(sizeof(SItem) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) { it[i].k = keys[i]; it[i].v = payload[i]; }
  s_sort(it, it + n);
  for (int i = 0; i < n; i++) { keys[i] = it[i].k; payload[i] = it[i].v; }
  free(it);
}

/* ---- mass -> radius / speeds (double islands).  R: core/utils.hpp:8-11, Engine.hpp:1296-1302 --- */
static inline float radius_of(unsigned mass) { double area = (double)mass / 1.0; return (float)sqrt(area / M_PI); }
static inline float max_speed_of(unsigned mass) { return (float)((double)CELL_MAX_SPEED / pow((double)mass, 0.439)); }
static inline float split_speed_of(unsigned mass) {
  double v = 3.0 * pow((double)max_speed_of(mass), 1.2);
  double c = (v < 130.0) ? v : 130.0; c = (c < 20.0) ? 20.0 : c; /* clamp<double> = max(min(x,hi),lo) */
  return (float)c;
}

/* ---- Ball predicates.  R: core/Ball.hpp:31-47 (pow(r,2) resolves to float powf and g++ -O3 folds
 * powf(r,2.0f) to r*r; the squared distance is |dx|*|dx| + |dy|*|dy| in fp32, types.hpp:87-91) ---- */
static inline float sqr_dist(float ax, float ay, float bx, float by) {
  float dx = fabsf(ax - bx), dy = fabsf(ay - by);
  float a = dx * dx, b = dy * dy;
  return a + b;
}
static inline int collides(float ax, float ay, float ar, float bx, float by, float br) {
  float r = smaxf(ar, br);
  float rr = r * r;
  return rr >= sqr_dist(ax, ay, bx, by);
}
static inline int touches(float ax, float ay, float ar, float bx, float by, float br) {
  float r = ar + br;
  float rr = r * r;
  float d = sqr_dist(ax, ay, bx, by) + 0.0f;
  return rr >= d;
}
static inline int can_eat_mass(unsigned a, unsigned b) { return (double)a > (double)b * CELL_EAT_MARGIN; }
static inline int cell_can_eat_cell(unsigned a, unsigned b) { return a > CELL_EAT_REQUIREMENT && can_eat_mass(a, b); } /* R: Entities.hpp:148-151 */

/* R: core/Entities.hpp:171-179 */
static inline void cell_set_mass(OCell *c, unsigned m) { c->mass = m > CELL_MIN_SIZE ? m : CELL_MIN_SIZE; }
static inline void cell_inc_mass(OCell *c, unsigned inc) { cell_set_mass(c, c->mass + inc); }
static inline float cell_radius(const OCell *c) { return radius_of(c->mass); }

/* ---- Player helpers.  R: core/Player.hpp:75-126 ------------------------------------------------ */
static unsigned player_mass(const OPlayer *p) { unsigned t = 0; for (int i = 0; i < p->n_cells; i++) t += p->cells[i].mass; return t; }
static float player_x(const OPlayer *p) {
  float s = 0; for (int i = 0; i < p->n_cells; i++) { float m = (float)p->cells[i].mass; float t = p->cells[i].x * m; s += t; }
  return s / (float)player_mass(p);
}
static float player_y(const OPlayer *p) {
  float s = 0; for (int i = 0; i < p->n_cells; i++) { float m = (float)p->cells[i].mass; float t = p->cells[i].y * m; s += t; }
  return s / (float)player_mass(p);
}
static void player_kill(OPlayer *p) {
  p->n_cells = 0; p->min_mass_cell = CELL_MIN_SIZE;
  p->split_cd = 0; p->feed_cd = 0; p->anti_team = 1.0f; p->elapsed = 0; p->last_decay = 0; p->n_vticks = 0;
}
static OCell *player_push_cell(OPlayer *p) { GROW(p->cells, p->n_cells, p->cap_cells, OCell); return &p->cells[p->n_cells++]; }

/* Cell ctor: id = ++global_id (Ball.hpp:18), timer = now, not yet latched (Entities.hpp:122-128) */
static void make_cell(OArena *a, OCell *c, float x, float y, float vx, float vy, unsigned mass) {
  c->x = x; c->y = y; c->vx = vx; c->vy = vy; c->svx = 0; c->svy = 0;
  c->id = ++a->id_counter; cell_set_mass(c, mass); c->deadline = a->clock;
}
static inline int cell_can_recombine(const OArena *a, const OCell *c) { return a->clock >= c->deadline; }
static inline void cell_reset_timer(const OArena *a, OCell *c) { c->deadline = a->clock + a->recomb_ticks; }

/* ---- Engine ----------------------------------------------------------------------------------- */
static float rnd_dist(OArena *a, float max) { return ora_uniform_float(a->mt, 0.0f, max); } /* R: Engine.hpp:1304-1311 */

static void random_location(OArena *a, float radius, float *ox, float *oy) { /* R: Engine.hpp:143-148 */
  float two_r = 2.0f * radius;
  float x = rnd_dist(a, a->W - two_r) + radius;
  float y = rnd_dist(a, a->H - two_r) + radius;
  *ox = x; *oy = y;
}

static void add_pellet_at(OArena *a, float x, float y) {
  GROW(a->pellets, a->n_pellets, a->cap_pellets, OPellet);
  OPellet *p = &a->pellets[a->n_pellets++]; p->x = x; p->y = y; p->id = ++a->id_counter;
}
static void add_pellets(OArena *a, int n) { /* R: Engine.hpp:418-424 */
  float r = radius_of(PELLET_MASS);
  for (int i = 0; i < n; i++) { float x, y; random_location(a, r, &x, &y); add_pellet_at(a, x, y); }
}
static void add_virus_full(OArena *a, float x, float y, float vx, float vy) {
  GROW(a->viruses, a->n_viruses, a->cap_viruses, OVirus);
  OVirus *v = &a->viruses[a->n_viruses++];
  v->x = x; v->y = y; v->vx = vx; v->vy = vy; v->mass = VIRUS_INITIAL_MASS; v->hits = 0; v->id = ++a->id_counter;
}
static void add_viruses(OArena *a, int n) { /* R: Engine.hpp:480-485 */
  float r = radius_of(VIRUS_INITIAL_MASS);
  for (int i = 0; i < n; i++) { float x, y; random_location(a, r, &x, &y); add_virus_full(a, x, y, 0.0f, 0.0f); }
}
static void create_squared_pellets(OArena *a) { /* R: Engine.hpp:426-475 */
  float square = sminf(a->H, a->W) / 2.0f;
  float spacing = 1.0f;
  int pps = f2i(square / spacing);
  float cx = a->W / 2.0f, cy = a->H / 2.0f, half = square / 2.0f;
  for (int i = 0; i < pps; i++) { float t = (float)i * spacing; float x = (cx - half) + t, y = cy - half;
    if (x >= 0 && x <= a->W && y >= 0 && y <= a->H) add_pellet_at(a, x, y); }
  for (int i = 0; i < pps; i++) { float t = (float)i * spacing; float x = cx + half, y = (cy - half) + t;
    if (x >= 0 && x <= a->W && y >= 0 && y <= a->H) add_pellet_at(a, x, y); }
  for (int i = 0; i < pps; i++) { float t = (float)i * spacing; float x = (cx + half) - t, y = cy + half;
    if (x >= 0 && x <= a->W && y >= 0 && y <= a->H) add_pellet_at(a, x, y); }
  for (int i = 0; i < pps; i++) { float t = (float)i * spacing; float x = cx - half, y = (cy + half) - t;
    if (x >= 0 && x <= a->W && y >= 0 && y <= a->H) add_pellet_at(a, x, y); }
}

static void set_mode(OArena *a, int mode) { /* R: Engine.hpp:367-416 */
  switch (mode) {
    case 0: case 4: a->mass_decay = 1; a->squared = 0; a->regen = 1; a->agent_mass = 25; break;
    case 1: a->mass_decay = 0; a->squared = 1; a->regen = 0; a->agent_mass = 25; break;
    case 2: a->mass_decay = 1; a->squared = 1; a->regen = 0; a->agent_mass = 25; break;
    case 3: a->mass_decay = 0; a->squared = 0; a->regen = 1; a->agent_mass = 25; break;
    case 5: set_mode(a, 2); a->agent_mass = 1000; break;
    case 6: set_mode(a, 4); a->agent_mass = 1000; break;
    case 7: case 8: case 9: case 10: set_mode(a, 4); break;
    default: break;
  }
}

static void respawn(OArena *a, OPlayer *p) { /* R: Engine.hpp:119-137 */
  player_kill(p);
  unsigned pm = (unsigned)(a->agent_mass > (int)CELL_MIN_SIZE ? a->agent_mass : (int)CELL_MIN_SIZE);
  float r25 = radius_of(CELL_MIN_SIZE);
  float x, y;
  if (a->n_pellets > 0 && a->squared) {
    x = a->pellets[0].x; y = a->pellets[0].y;
    float t = 2.0f * r25;
    x += t; y += t;
    x = sminf(x, a->W - r25);
    y = sminf(y, a->H - r25);
  } else {
    random_location(a, r25, &x, &y);
  }
  OCell *c = player_push_cell(p);
  make_cell(a, c, x, y, 0.0f, 0.0f, pm);
}

static OPlayer *find_player(OArena *a, int pid) { for (int i = 0; i < a->n_players; i++) if (a->players[i].pid == pid) return &a->players[i]; return NULL; }

static int add_player(OArena *a, int kind) { /* R: Engine.hpp:70-83 */
  int pid = a->next_pid; a->next_pid = (a->next_pid + 1) & 0xFFFF;
  GROW(a->players, a->n_players, a->cap_players, OPlayer);
  OPlayer *p = &a->players[a->n_players];
  memset(p, 0, sizeof(*p));
  p->pid = pid; p->kind = kind; p->is_bot = kind != KIND_AGENT;
  p->action = 0; p->tx = 0; p->ty = 0; p->anti_team = 1.0f; p->highest_mass = CELL_MIN_SIZE; /* R: Player.hpp:25-51 */
  if (kind == KIND_AGENT) (void)ora_rand_next(a->rnd); /* random_color(), R: Player.hpp:53, color.hpp:14-16 */
  int node = hm_insert(&a->pmap, pid);
  (void)node; /* node index == players[] index by construction */
  a->n_players++;
  respawn(a, p);
  return pid;
}

static void clear_state(OArena *a) { /* R: GameState.hpp:61-67 */
  for (int i = 0; i < a->n_players; i++) { free(a->players[i].cells); free(a->players[i].vticks); }
  a->n_players = 0; hm_clear(&a->pmap);
  a->n_pellets = 0; a->n_foods = 0; a->n_viruses = 0; a->ticks = 0;
}

static void engine_reset(OArena *a) { /* R: Engine.hpp:98-117 */
  clear_state(a);
  if (a->squared) create_squared_pellets(a); else add_pellets(a, (int)a->target_pellets);
  add_viruses(a, (int)a->target_viruses);
}

/* R: Engine.hpp:695-698 */
static inline void boundary(const OArena *a, float *x, float *y, float r) {
  *x = smaxf(0.0f, clampf(*x, r, a->W - r));
  *y = smaxf(0.0f, clampf(*y, r, a->H - r));
}

/* Velocity helpers.  R: core/types.hpp:160-231 */
static inline float vmag(float dx, float dy) { float a = dx * dx, b = dy * dy; return sqrtf(a + b); }
static void v_decelerate(float *dx, float *dy, float decel, float dt) {
  float xr = *dx / vmag(*dx, *dy);
  float yr = *dy / vmag(*dx, *dy);
  float ddx = xr * decel;
  if (fabsf(ddx * dt) <= fabsf(*dx)) { float t = ddx * dt; *dx -= t; } else *dx = 0;
  float ddy = yr * decel;
  if (fabsf(ddy * dt) <= fabsf(*dy)) { float t = ddy * dt; *dy -= t; } else *dy = 0;
}
static void v_clamp_speed_hi(float *dx, float *dy, float high) { /* clamp_speed(0, high) */
  if (vmag(*dx, *dy) > high) {
    float f = high / vmag(*dx, *dy); *dx *= f;   /* set_speed: speed() is re-evaluated after dx changed */
    float g = high / vmag(*dx, *dy); *dy *= g;
  } else if (vmag(*dx, *dy) < 0.0f) { /* unreachable: low == 0 */ }
}
static float v_direction(float dx, float dy) { /* R: types.hpp:167-174 */
  float angle = atanf(dx / dy);
  if (dx < 0) { if (dy > 0) angle = (float)((double)angle + M_PI); else angle = (float)((double)angle - M_PI); }
  return angle;
}
static inline void cell_move(OCell *c, float dt) { /* R: Entities.hpp:161-164 */
  float sx = c->vx + c->svx; float tx = sx * dt; c->x += tx;
  float sy = c->vy + c->svy; float ty = sy * dt; c->y += ty;
}

/* R: Engine.hpp:701-749 */
static void avoid_static_overlap(OArena *a, OCell *ca, OCell *cb) {
  float dx = cb->x - ca->x, dy = cb->y - ca->y;
  float dist = vmag(dx, dy);
  float ra = cell_radius(ca), rb = cell_radius(cb);
  float target = ra + rb;
  if (dist > target) return;
  float den = fabsf(dx) + fabsf(dy);
  float xr = dx / den, yr = dy / den;
  float depth = target - dist;
  float a1 = 0.5f, a2 = 0.5f, b1 = 0.5f, b2 = 0.5f;
  if (ca->x == ra || ca->x == a->W - ra) { a1 = 1.0f; ca->vx = 0; }
  if (ca->y == ra || ca->y == a->H - ra) { a2 = 1.0f; ca->vy = 0; }
  if (cb->x == rb || cb->x == a->W - rb) { b1 = 1.0f; cb->vx = 0; }
  if (cb->y == rb || cb->y == a->H - rb) { b2 = 1.0f; cb->vy = 0; }
  float t;
  t = xr * depth; t = t * a1; ca->x -= t;
  t = yr * depth; t = t * a2; ca->y -= t;
  t = xr * depth; t = t * b1; cb->x += t;
  t = yr * depth; t = t * b2; cb->y += t;
  boundary(a, &ca->x, &ca->y, ra);
  boundary(a, &cb->x, &cb->y, rb);
}

/* R: Engine.hpp:803-848 */
static void separate_cells(OCell *ca, OCell *cb, float tx, float ty) {
  float dx = cb->x - ca->x, dy = cb->y - ca->y;
  float dist = vmag(dx, dy);
  float target = cell_radius(ca) + cell_radius(cb);
  if (dist > target) return;
  float den = fabsf(dx) + fabsf(dy);
  float xr = dx / den, yr = dy / den;
  float diff_a = sqr_dist(tx, ty, ca->x, ca->y);
  float diff_b = sqr_dist(tx, ty, cb->x, cb->y);
  float depth = target - dist;
  int s1 = ca->mass < cb->mass ? 1 : -1;
  int s2 = diff_a >= diff_b ? 1 : -1;
  int s = (s1 == s2) ? s2 : 0;
  OCell *tc = ca->mass < cb->mass ? ca : cb;
  float fs = (float)s, t;
  if (dx >= 0) {
    t = xr * depth; t = t * fs; tc->x -= t;
    if (dy >= 0) { t = yr * depth; t = t * fs; tc->y -= t; } else { t = yr * depth; t = t * fs; tc->y += t; }
  } else {
    t = xr * depth; t = t * fs; tc->x += t;
    if (dy >= 0) { t = yr * depth; t = t * fs; tc->y -= t; } else { t = yr * depth; t = t * fs; tc->y += t; }
  }
}

/* R: Engine.hpp:893-938 */
static void elastic(OCell *ca, OCell *cb, float dx, float dy, float dist) {
  float nx = dx / dist, ny = dy / dist;
  float tx = -ny, ty = nx;
  float p1 = ca->vx * nx, p2 = ca->vy * ny; float dpNorm1 = p1 + p2;
  p1 = cb->vx * nx; p2 = cb->vy * ny; float dpNorm2 = p1 + p2;
  p1 = ca->vx * tx; p2 = ca->vy * ty; float dpTan1 = p1 + p2;
  p1 = cb->vx * tx; p2 = cb->vy * ty; float dpTan2 = p1 + p2;
  int m1 = (int)ca->mass, m2 = (int)cb->mass;
  float q1 = dpNorm1 * (float)(m1 - m2);
  float q2 = 2.0f * (float)m2; q2 = q2 * dpNorm2;
  float v1 = (q1 + q2) / (float)(m1 + m2);
  q1 = dpNorm2 * (float)(m2 - m1);
  q2 = 2.0f * (float)m1; q2 = q2 * dpNorm1;
  float v2 = (q1 + q2) / (float)(m1 + m2);
  if (ca->mass < cb->mass) {
    float u = tx * dpTan1, w = nx * v1; ca->vx = u + w; u = ty * dpTan1; w = ny * v1; ca->vy = u + w;
  } else if (ca->mass > cb->mass) {
    float u = tx * dpTan2, w = nx * v2; cb->vx = u + w; u = ty * dpTan2; w = ny * v2; cb->vy = u + w;
  } else {
    float u = tx * dpTan1, w = nx * v1; ca->vx = u + w; u = ty * dpTan1; w = ny * v1; ca->vy = u + w;
    u = tx * dpTan2; w = nx * v2; cb->vx = u + w; u = ty * dpTan2; w = ny * v2; cb->vy = u + w;
  }
}

/* R: Engine.hpp:857-888 */
static void prevent_overlap(OArena *a, OCell *ca, OCell *cb, float dt, float tx, float ty) {
  float dx = cb->x - ca->x, dy = cb->y - ca->y;
  float dist = vmag(dx, dy);
  float target = cell_radius(ca) + cell_radius(cb);
  if (dist > target) return;
  float s, t;
  s = ca->vx + ca->svx; t = s * dt; ca->x -= t;
  s = ca->vy + ca->svy; t = s * dt; ca->y -= t;
  s = cb->vx + cb->svx; t = s * dt; cb->x -= t;
  s = cb->vy + cb->svy; t = s * dt; cb->y -= t;
  elastic(ca, cb, dx, dy, dist);
  cell_move(ca, dt);
  cell_move(cb, dt);
  if (touches(ca->x, ca->y, cell_radius(ca), cb->x, cb->y, cell_radius(cb))) {
    int d = (int)(ca->mass - cb->mass);
    if (abs(d) <= 10) avoid_static_overlap(a, ca, cb);
    else separate_cells(ca, cb, tx, ty);
  }
  boundary(a, &ca->x, &ca->y, cell_radius(ca));
  boundary(a, &cb->x, &cb->y, cell_radius(cb));
}

/* R: Engine.hpp:763-794 */
static void self_collisions(OArena *a, OPlayer *p, float dt) {
  int overlap = 0;
  for (int iter = 0; iter < 5; iter++) {
    overlap = 0;
    for (int ia = 0; ia < p->n_cells; ia++)
      for (int ib = ia + 1; ib < p->n_cells; ib++) {
        OCell *ca = &p->cells[ia], *cb = &p->cells[ib];
        if (touches(ca->x, ca->y, cell_radius(ca), cb->x, cb->y, cell_radius(cb))) { overlap = 1; prevent_overlap(a, ca, cb, dt, p->tx, p->ty); }
      }
    if (!overlap) break;
  }
  if (overlap)
    for (int ia = 0; ia < p->n_cells; ia++)
      for (int ib = ia + 1; ib < p->n_cells; ib++) {
        OCell *ca = &p->cells[ia], *cb = &p->cells[ib];
        if (touches(ca->x, ca->y, cell_radius(ca), cb->x, cb->y, cell_radius(cb))) avoid_static_overlap(a, ca, cb);
      }
}

/* R: Engine.hpp:609-630 */
static void move_player(OArena *a, OPlayer *p, float dt) {
  unsigned smallest = UINT_MAX;
  for (int i = 0; i < p->n_cells; i++) {
    OCell *c = &p->cells[i];
    float d = p->tx - c->x; c->vx = 3.0f * d;
    d = p->ty - c->y; c->vy = 3.0f * d;
    if (c->mass < smallest) smallest = c->mass;
    v_clamp_speed_hi(&c->vx, &c->vy, max_speed_of(c->mass));
    cell_move(c, dt);
    v_decelerate(&c->svx, &c->svy, SPLIT_DECELERATION, dt);
    boundary(a, &c->x, &c->y, cell_radius(c));
  }
  p->min_mass_cell = smallest;
  self_collisions(a, p, dt);
}

typedef struct { OCell *v; int n, cap; } CellVec;
static OCell *cv_push(CellVec *cv) { GROW(cv->v, cv->n, cv->cap, OCell); return &cv->v[cv->n++]; }

/* R: Engine.hpp:1263-1294 */
static void disrupt(OArena *a, OCell *cell, const OVirus *virus, CellVec *created, int create_limit) {
  unsigned total = cell->mass;
  cell_set_mass(cell, (unsigned)((float)cell->mass / CELL_POP_REDUCTION));
  cell_inc_mass(cell, (total - cell->mass) % CELL_POP_SIZE);
  unsigned pop_mass = total - cell->mass;
  int num_new = (int)((pop_mass + CELL_POP_SIZE - 1) / CELL_POP_SIZE);
  if (create_limit < num_new) num_new = create_limit;
  unsigned remaining = pop_mass;
  float theta = v_direction(cell->vx, cell->vy);
  for (int c = 0; c < num_new; c++) {
    float inc = (float)(2 * M_PI * c / num_new);
    float dvel = v_direction(cell->vx, cell->vy) + inc;
    float ang = theta + dvel;
    float sp = max_speed_of(CELL_POP_SIZE);
    float vx = sp * cosf(ang), vy = sp * sinf(ang);
    unsigned nm = remaining < CELL_POP_SIZE ? remaining : CELL_POP_SIZE;
    OCell *nc = cv_push(created);
    make_cell(a, nc, virus->x, virus->y, cell->vx, cell->vy, nm);
    nc->svx = vx; nc->svy = vy;
    cell_reset_timer(a, nc);
    remaining -= nm;
  }
  cell_reset_timer(a, cell);
}

/* R: Engine.hpp:1223-1252 (+ grid build :1207-1221) */
static int virus_collisions(OArena *a, OPlayer *p, CellVec *created, int create_limit, int can_eat_virus) {
  const int gs = 25;
  int gw = f2i((a->W + (float)gs - 1.0f) / (float)gs), gh = f2i((a->H + (float)gs - 1.0f) / (float)gs);
  for (int ci = 0; ci < p->n_cells; ci++) {
    OCell *cell = &p->cells[ci];
    int gx = f2i(cell->x) / gs, gy = f2i(cell->y) / gs;
    for (int dx = -1; dx <= 1; dx++)
      for (int dy = -1; dy <= 1; dy++) {
        int nx = gx + dx, ny = gy + dy;
        if (!(nx >= 0 && nx < gw && ny >= 0 && ny < gh)) continue;
        for (int vi = 0; vi < a->n_viruses; vi++) { /* bucket members in ascending index order */
          OVirus *v = &a->viruses[vi];
          if (f2i(v->x) / gs != nx || f2i(v->y) / gs != ny) continue;
          if (can_eat_mass(cell->mass, v->mass) && collides(cell->x, cell->y, cell_radius(cell), v->x, v->y, radius_of(v->mass))) {
            if (can_eat_virus) cell_inc_mass(cell, v->mass);
            else disrupt(a, cell, v, created, create_limit);
            GROW(a->ev_v, a->n_ev_v, a->cap_ev_v, int); a->ev_v[a->n_ev_v++] = vi;
            return 1;
          }
        }
      }
  }
  return 0;
}

/* R: Engine.hpp:976-1000 (+ grid build :962-974) */
static void pellets_eat(OArena *a, OPlayer *p) {
  const int gs = 510;
  int gw = f2i((a->W + (float)gs - 1.0f) / (float)gs), gh = f2i((a->H + (float)gs - 1.0f) / (float)gs);
  for (int ci = 0; ci < p->n_cells; ci++) {
    OCell *cell = &p->cells[ci];
    int gx = f2i(cell->x) / gs, gy = f2i(cell->y) / gs;
    for (int dx = -1; dx <= 1; dx++)
      for (int dy = -1; dy <= 1; dy++) {
        int nx = gx + dx, ny = gy + dy;
        if (!(nx >= 0 && nx < gw && ny >= 0 && ny < gh)) continue;
        for (int pi = 0; pi < a->n_pellets; pi++) {
          OPellet *pl = &a->pellets[pi];
          if (f2i(pl->x) / gs != nx || f2i(pl->y) / gs != ny) continue;
          if (can_eat_mass(cell->mass, PELLET_MASS) && collides(cell->x, cell->y, cell_radius(cell), pl->x, pl->y, radius_of(PELLET_MASS))) {
            GROW(a->ev_p, a->n_ev_p, a->cap_ev_p, int); a->ev_p[a->n_ev_p++] = pi;
            cell_inc_mass(cell, PELLET_MASS);
          }
        }
      }
  }
}

/* R: Engine.hpp:1067-1093 */
static int cell_split(OArena *a, OCell *cell, CellVec *created, float tx, float ty) {
  if (cell->mass < CELL_SPLIT_MINIMUM || cell->mass < 2 * CELL_MIN_SIZE) return 0;
  unsigned split_mass = cell->mass / 2;
  unsigned remaining = cell->mass - split_mass;
  cell_set_mass(cell, remaining);
  float ddx = tx - cell->x, ddy = ty - cell->y;
  float ax = fabsf(ddx), ay = fabsf(ddy); float n2 = ax * ax; float n2b = ay * ay; float nrm = sqrtf(n2 + n2b);
  float dirx = ddx / nrm, diry = ddy / nrm;
  float r = cell_radius(cell);
  float ox = dirx * r, oy = diry * r;
  float lx = cell->x + ox, ly = cell->y + oy;
  lx = smaxf(0.0f, clampf(lx, r, a->W - r));
  ly = smaxf(0.0f, clampf(ly, r, a->H - r));
  float ss = split_speed_of(split_mass);
  float vx = dirx * ss, vy = diry * ss;
  OCell *nc = cv_push(created);
  make_cell(a, nc, lx, ly, vx, vy, split_mass);
  nc->svx = vx; nc->svy = vy;
  cell_reset_timer(a, cell);
  cell_reset_timer(a, nc);
  return 1;
}

/* R: Engine.hpp:1011-1025 */
static int eat_food(OArena *a, OCell *cell) {
  if (cell->mass < FOOD_MASS) return 0;
  int prev = a->n_foods, w = 0;
  float cr = cell_radius(cell), fr = radius_of(FOOD_MASS);
  for (int i = 0; i < a->n_foods; i++) {
    OFood *f = &a->foods[i];
    int eaten = can_eat_mass(cell->mass, FOOD_MASS) && collides(cell->x, cell->y, cr, f->x, f->y, fr);
    if (!eaten) a->foods[w++] = *f;
  }
  a->n_foods = w;
  int num = prev - w;
  cell_inc_mass(cell, (unsigned)num * FOOD_MASS);
  return num;
}

/* R: Engine.hpp:1027-1054 */
static void maybe_emit_food(OArena *a, OPlayer *p) {
  if (p->feed_cd > 0) p->feed_cd -= 1;
  if (p->action == 1 && p->feed_cd == 0) {
    for (int i = 0; i < p->n_cells; i++) {
      OCell *cell = &p->cells[i];
      if (cell->mass < CELL_MIN_SIZE + FOOD_MASS) continue;
      float ddx = p->tx - cell->x, ddy = p->ty - cell->y;
      float ax = fabsf(ddx), ay = fabsf(ddy); float n2 = ax * ax; float n2b = ay * ay; float nrm = sqrtf(n2 + n2b);
      float dirx = ddx / nrm, diry = ddy / nrm;
      float r = cell_radius(cell);
      float ox = dirx * r, oy = diry * r;
      GROW(a->foods, a->n_foods, a->cap_foods, OFood);
      OFood *f = &a->foods[a->n_foods++];
      f->x = cell->x + ox; f->y = cell->y + oy;
      f->vx = dirx * FOOD_SPEED; f->vy = diry * FOOD_SPEED;
      f->id = ++a->id_counter;
      cell_inc_mass(cell, (unsigned)(-(int)FOOD_MASS));
    }
    p->feed_cd = 10;
  }
}

/* R: Engine.hpp:1056-1064, 1095-1107 */
static void maybe_split(OArena *a, OPlayer *p, CellVec *created, int create_limit) {
  if (p->split_cd > 0) p->split_cd -= 1;
  if (p->action == 2 && p->split_cd == 0) {
    if (create_limit != 0) {
      int num = 0;
      for (int i = 0; i < p->n_cells; i++)
        if (cell_split(a, &p->cells[i], created, p->tx, p->ty)) { if (++num == create_limit) break; }
    }
    p->split_cd = 30;
  }
}

/* R: Engine.hpp:1160-1179 */
static void recombine_cells(OArena *a, OPlayer *p) {
  for (int i = 0; i < p->n_cells; i++) {
    if (!cell_can_recombine(a, &p->cells[i])) continue;
    OCell *cell = &p->cells[i];
    for (int j = i + 1; j < p->n_cells;) {
      OCell *other = &p->cells[j];
      if (cell_can_recombine(a, other) && touches(cell->x, cell->y, cell_radius(cell), other->x, other->y, cell_radius(other))) {
        cell_inc_mass(cell, other->mass);
        OCell t = *other; *other = p->cells[p->n_cells - 1]; p->cells[p->n_cells - 1] = t;
        p->n_cells--;
      } else j++;
    }
  }
}

/* ---- bots.  R: agario/bots/Bot.hpp, HungryBot.hpp, HungryShyBot.hpp, AggressiveBot.hpp,
 * AggressiveShyBot.hpp ------------------------------------------------------------------------- */
static float dist_to(float ax, float ay, float bx, float by) { /* a.distance_to(b) = (b-a).norm() */
  float dx = fabsf(bx - ax), dy = fabsf(by - ay); float p = dx * dx, q = dy * dy; return sqrtf(p + q);
}
static void nearest_pellet(OArena *a, OPlayer *self, float *ox, float *oy) { /* R: Bot.hpp:90-127 */
  if (a->n_pellets == 0) {
    int rx = ora_rand_next(a->rnd) % f2i(a->W);
    int ry = ora_rand_next(a->rnd) % f2i(a->H);
    *ox = (float)rx; *oy = (float)ry; return;
  }
  float tx = 0, ty = 0, mind = FLT_MAX;
  float sx = player_x(self), sy = player_y(self);
  for (int i = 0; i < a->n_pellets; i++) {
    float d = dist_to(a->pellets[i].x, a->pellets[i].y, sx, sy);
    if (d < mind && (double)d > 0.01) { tx = a->pellets[i].x; ty = a->pellets[i].y; mind = d; }
  }
  if ((double)mind < 0.01) {
    int rx = ora_rand_next(a->rnd) % f2i(a->W);
    float nx = tx + (float)rx;
    int ry = ora_rand_next(a->rnd) % f2i(a->H);
    float ny = ty + (float)ry;
    tx += nx; ty += ny;
  }
  *ox = tx; *oy = ty;
}
static const OCell *largest_cell(const OPlayer *p) { int l = 0; for (int i = 0; i < p->n_cells; i++) if (i == 0 || p->cells[i].mass > p->cells[l].mass) l = i; return &p->cells[l]; }
static unsigned edible_mass(const OPlayer *other, const OCell *lc) { unsigned m = 0; for (int i = 0; i < other->n_cells; i++) if (cell_can_eat_cell(lc->mass, other->cells[i].mass)) m += other->cells[i].mass; return m; }
static void target_player(OPlayer *self, const OPlayer *other, const OCell *lc) { /* R: Bot.hpp:52-63 */
  unsigned mass = 0; float tx = 0, ty = 0;
  for (int i = 0; i < other->n_cells; i++) {
    const OCell *c = &other->cells[i];
    if (cell_can_eat_cell(lc->mass, c->mass)) { float m = (float)c->mass; float px = c->x * m, py = c->y * m; tx += px; ty += py; mass += c->mass; }
  }
  float fm = (float)mass;
  float sx = player_x(self), sy = player_y(self);
  float qx = tx / fm, qy = ty / fm;
  float dsx = qx - sx, dsy = qy - sy;
  float ex = dsx * 3.0f, ey = dsy * 3.0f;
  /* this->location() is evaluated again for the sum (same value) */
  self->tx = sx + ex; self->ty = sy + ey;
}
static int shy_check(OArena *a, OPlayer *self) { /* R: HungryShyBot.hpp:26-40 ; `mass()` there is the
   value-initialised typedef agario::mass (== 0), not Player::mass(): unqualified name in a template
   with a dependent base. */
  for (int n = a->pmap.head; n != -1; n = a->pmap.next[n]) {
    OPlayer *o = &a->players[n];
    if (o->pid == self->pid) continue;
    float sx = player_x(self), sy = player_y(self), ox = player_x(o), oy = player_y(o);
    float d = dist_to(sx, sy, ox, oy);
    if (d < SHY_RADIUS && player_mass(o) > 0u) {
      float dx = ox - sx, dy = oy - sy;
      self->tx = sx - dx; self->ty = sy - dy;
      return 1;
    }
  }
  return 0;
}
static int aggressive_check(OArena *a, OPlayer *self) { /* R: AggressiveBot.hpp:30-52 */
  const OCell *lc = largest_cell(self);
  for (int n = a->pmap.head; n != -1; n = a->pmap.next[n]) {
    OPlayer *o = &a->players[n];
    if (o->pid == self->pid) continue;
    float d = dist_to(player_x(self), player_y(self), player_x(o), player_y(o));
    if (d <= AGGRESSIVE_RADIUS) {
      if (edible_mass(o, lc) > 0) { target_player(self, o, lc); return 1; }
    }
  }
  return 0;
}
static void bot_take_action(OArena *a, OPlayer *p) {
  switch (p->kind) {
    case KIND_HUNGRY: p->action = 0; nearest_pellet(a, p, &p->tx, &p->ty); break;
    case KIND_HUNGRY_SHY: p->action = 0; if (!shy_check(a, p)) nearest_pellet(a, p, &p->tx, &p->ty); break;
    case KIND_AGGRESSIVE: if (!aggressive_check(a, p)) { p->action = 0; nearest_pellet(a, p, &p->tx, &p->ty); } break;
    case KIND_AGGRESSIVE_SHY: if (!shy_check(a, p) && !aggressive_check(a, p)) { p->action = 0; nearest_pellet(a, p, &p->tx, &p->ty); } break;
    case KIND_EXAMPLE: p->action = 0; p->tx = player_x(p); p->ty = player_y(p); break; /* R: ExampleBot.hpp:45-51: none, target = location() */
    default: break;
  }
}

/* R: Engine.hpp:495-542 */
static void tick_player(OArena *a, OPlayer *p, float dt) {
  p->elapsed += 1;
  if (a->ticks % 10 == 0) bot_take_action(a, p);
  move_player(a, p, dt);
  int prev_cells = p->n_cells;
  CellVec created = {0, 0, 0};
  int create_limit = PLAYER_CELL_LIMIT - prev_cells;
  int can_eat_virus = p->n_cells >= PLAYER_CELL_LIMIT;
  if (virus_collisions(a, p, &created, create_limit, can_eat_virus)) {
    GROW(p->vticks, p->n_vticks, p->cap_vticks, int); p->vticks[p->n_vticks++] = p->elapsed;
    p->viruses_eaten++;
  }
  int before = a->n_ev_p;
  pellets_eat(a, p);
  p->food_eaten += a->n_ev_p - before;
  { unsigned m = player_mass(p); if (p->highest_mass < m) p->highest_mass = m; }
  for (int i = 0; i < p->n_cells; i++) {
    OCell *cell = &p->cells[i];
    if (cell->mass >= MAX_MASS_IN_THE_GAME) { /* R: Engine.hpp:592-601 */
      if (p->n_cells < PLAYER_CELL_LIMIT) cell_split(a, cell, &created, p->tx, p->ty);
      else cell_set_mass(cell, NEW_MASS_IF_NO_SPLIT);
    }
    p->food_eaten += eat_food(a, cell);
  }
  create_limit -= created.n;
  maybe_emit_food(a, p);
  maybe_split(a, p, &created, create_limit);
  for (int i = 0; i < created.n; i++) { OCell *c = player_push_cell(p); *c = created.v[i]; }
  free(created.v);
  recombine_cells(a, p);
  if (a->mass_decay && p->elapsed % 60 == 0) {
    /* R: Engine.hpp:550-568 */
    int fall_off = p->elapsed - 60 * ANTI_TEAM_ACTIVATION_TIME, w = 0;
    for (int i = 0; i < p->n_vticks; i++) if (!(p->vticks[i] < fall_off)) p->vticks[w++] = p->vticks[i];
    p->n_vticks = w;
    if (w != 0) p->anti_team = (float)pow(1.1, (double)(unsigned long)(w - 1));
    /* R: Engine.hpp:575-584, Entities.hpp:199-203 */
    if (p->elapsed - p->last_decay >= 60) {
      for (int i = 0; i < p->n_cells; i++) {
        double nm = (double)p->cells[i].mass * (1 - PLAYER_RATE * (double)p->anti_team);
        unsigned um = (unsigned)nm;
        p->cells[i].mass = um > CELL_MIN_SIZE ? um : CELL_MIN_SIZE;
      }
      p->last_decay = p->elapsed;
    }
  }
}

/* R: Engine.hpp:1002-1009 / 1253-1260 */
static void remove_pellets(OArena *a) {
  for (int k = 0; k < a->n_ev_p; k++) {
    int idx = a->ev_p[k];
    size_t sz = (size_t)a->n_pellets;
    if ((size_t)idx < sz - 1 && sz > 1) { OPellet t = a->pellets[idx]; a->pellets[idx] = a->pellets[sz - 1]; a->pellets[sz - 1] = t; }
    if (sz >= 1) a->n_pellets--;
  }
}
static void remove_viruses(OArena *a) {
  for (int k = 0; k < a->n_ev_v; k++) {
    int idx = a->ev_v[k];
    size_t sz = (size_t)a->n_viruses;
    if ((size_t)idx < sz - 1 && sz > 1) { OVirus t = a->viruses[idx]; a->viruses[idx] = a->viruses[sz - 1]; a->viruses[sz - 1] = t; }
    if (sz >= 1) a->n_viruses--;
  }
}

static int cmp_cell_id(const void *x, const void *y) { int a = ((const OCell *)x)->id, b = ((const OCell *)y)->id; return (a > b) - (a < b); }

/* R: Engine.hpp:150-200 + utils/collision_detection.hpp:10-64 */
typedef struct { int pid; OCell c; } GCell;
static int lower_bound_id(const OCell *cells, int n, int id) { int lo = 0, hi = n; while (lo < hi) { int mid = lo + (hi - lo) / 2; if (cells[mid].id < id) lo = mid + 1; else hi = mid; } return lo; }
static void players_collision(OArena *a) {
  int total = 0;
  for (int n = a->pmap.head; n != -1; n = a->pmap.next[n]) total += a->players[n].n_cells;
  GCell *g = (GCell *)malloc(sizeof(GCell) * (size_t)(total > 0 ? total : 1));
  int k = 0;
  for (int n = a->pmap.head; n != -1; n = a->pmap.next[n]) {
    OPlayer *p = &a->players[n];
    qsort(p->cells, (size_t)p->n_cells, sizeof(OCell), cmp_cell_id); /* ids unique -> order is the sorted order */
    for (int i = 0; i < p->n_cells; i++) { g[k].pid = p->pid; g[k].c = p->cells[i]; k++; }
  }
  if (a->n_players <= 1 || total == 0) { free(g); return; } /* one player: every scan breaks on an own cell */
  /* strips: row = int(x / W * 100) */
  int *row = (int *)malloc(sizeof(int) * (size_t)total);
  SItem *items = (SItem *)malloc(sizeof(SItem) * (size_t)total);
  int *row_start = (int *)calloc(103, sizeof(int)), *row_cnt = (int *)calloc(102, sizeof(int));
  for (int i = 0; i < total; i++) { float t = g[i].c.x / a->W; t = t * 100.0f; row[i] = f2i(t); if (row[i] >= 0 && row[i] <= 101) row_cnt[row[i]]++; }
  for (int r = 0; r < 102; r++) row_start[r + 1] = row_start[r] + row_cnt[r];
  int *fill = (int *)calloc(102, sizeof(int));
  for (int i = 0; i < total; i++) if (row[i] >= 0 && row[i] <= 101) { int r = row[i]; items[row_start[r] + fill[r]].k = g[i].c.y; items[row_start[r] + fill[r]].v = i; fill[r]++; }
  for (int r = 0; r < 102; r++) s_sort(items + row_start[r], items + row_start[r] + row_cnt[r]);
  /* results: unordered_map<int, vector<...>> keyed by query index */
  HMap rm; hm_init(&rm);
  int *res_q = NULL, *res_g = NULL; int n_res = 0, cap_res = 0, cap_res2 = 0; /* (query, gallery) hits in push order */
  int *has = (int *)calloc((size_t)total, sizeof(int));
  for (int id = 0; id < total; id++) {
    const OCell *q = &g[id].c;
    float qr = cell_radius(q);
    float left = q->x - qr, right = q->x + qr;
    float t = left / a->W; t = t * 100.0f; int top = f2i(t);
    t = right / a->W; t = t * 100.0f; int bottom = f2i(t);
    for (int i = top; i <= bottom; i++) {
      if (i < 0 || i > 101 || row_cnt[i] == 0) continue;
      int l = row_cnt[i]; SItem *v = items + row_start[i];
      int start = 0;
      for (int j = 10; j >= 0; j--) if (start + (1 << j) < l && v[start + (1 << j)].k < left) start += (1 << j);
      for (int j = start; j < l; j++) {
        int gi = v[j].v;
        if (g[id].pid == g[gi].pid) break;
        const OCell *o = &g[gi].c;
        if (collides(q->x, q->y, qr, o->x, o->y, cell_radius(o)) && cell_can_eat_cell(q->mass, o->mass)) {
          if (!has[id]) { has[id] = 1; hm_insert(&rm, id); }
          GROW(res_q, n_res, cap_res, int); GROW(res_g, n_res, cap_res2, int);
          res_q[n_res] = id; res_g[n_res] = gi; n_res++;
        }
      }
    }
  }
  for (int n = rm.head; n != -1; n = rm.next[n]) {
    int id = rm.key[n];
    for (int e = 0; e < n_res; e++) {
      if (res_q[e] != id) continue;
      const GCell *victim = &g[res_g[e]];
      OPlayer *eaten = find_player(a, victim->pid);
      OPlayer *pl = find_player(a, g[id].pid);
      int it = lower_bound_id(pl->cells, pl->n_cells, g[id].c.id);
      if (it != pl->n_cells) { cell_inc_mass(&pl->cells[it], victim->c.mass); pl->cells_eaten++; }
      int ei = lower_bound_id(eaten->cells, eaten->n_cells, victim->c.id);
      if (ei != eaten->n_cells) { memmove(&eaten->cells[ei], &eaten->cells[ei + 1], sizeof(OCell) * (size_t)(eaten->n_cells - ei - 1)); eaten->n_cells--; }
    }
  }
  hm_free(&rm); free(res_q); free(res_g); free(has); free(fill); free(row_cnt); free(row_start); free(items); free(row); free(g);
}

/* R: Engine.hpp:632-687 */
static void move_foods(OArena *a, float dt, float dt10) {
  for (int i = 0; i < a->n_foods;) {
    OFood *f = &a->foods[i];
    if (vmag(f->vx, f->vy) == 0) { i++; continue; }
    float fvx = f->vx, fvy = f->vy;
    v_decelerate(&f->vx, &f->vy, FOOD_DECEL, dt);
    { float t = f->vx * dt; f->x += t; t = f->vy * dt; f->y += t; }
    float fr = radius_of(FOOD_MASS);
    boundary(a, &f->x, &f->y, fr);
    int hit = 0;
    int nv = a->n_viruses;
    for (int vi = 0; vi < nv; vi++) {
      OVirus *v = &a->viruses[vi];
      if (collides(f->x, f->y, fr, v->x, v->y, radius_of(v->mass))) {
        if (v->hits >= NUMBER_OF_FOOD_HITS) {
          v->hits = 0; v->mass = VIRUS_INITIAL_MASS;
          float nx = v->x, ny = v->y;
          { float t = fvx * dt10; nx += t; t = fvy * dt10; ny += t; }
          boundary(a, &nx, &ny, radius_of(VIRUS_INITIAL_MASS));
          add_virus_full(a, nx, ny, fvx, fvy);
        } else { v->hits += 1; v->mass += FOOD_MASS; }
        hit = 1; break;
      }
    }
    if (hit) {
      if (a->n_foods > 1) { OFood t = a->foods[i]; a->foods[i] = a->foods[a->n_foods - 1]; a->foods[a->n_foods - 1] = t; }
      a->n_foods--;
    } else i++;
  }
}

void ora_tick(OArena *a, double dt_d) { /* R: Engine.hpp:208-240 */
  float dt = (float)dt_d, dt10 = (float)(dt_d * 10);
  a->n_ev_p = 0; a->n_ev_v = 0;
  for (int n = a->pmap.head; n != -1; n = a->pmap.next[n]) {
    OPlayer *p = &a->players[n];
    if (p->n_cells > 0) tick_player(a, p, dt);
  }
  remove_pellets(a);
  remove_viruses(a);
  players_collision(a);
  move_foods(a, dt, dt10);
  if (a->regen && a->ticks % 120 == 0) {
    add_pellets(a, (int)(a->target_pellets - (size_t)a->n_pellets));
    add_viruses(a, (int)(a->target_viruses - (size_t)a->n_viruses));
  }
  a->ticks++;
  a->clock++;
}

/* ---- BaseEnvironment ------------------------------------------------------------------------- */
static void env_reset(OArena *a) { /* R: BaseEnvironment.hpp:179-204, 374-425 */
  engine_reset(a);
  for (int i = 0; i < a->num_agents; i++) { int pid = add_player(a, KIND_AGENT); a->main_agent_pid = pid; a->pids[i] = pid; a->dones[i] = 0; }
  if (a->mode == 0) {
    for (int i = 0; i < a->num_bots; i++) {
      int k = i % a->num_bots;
      add_player(a, k == 0 ? KIND_HUNGRY : k == 1 ? KIND_HUNGRY_SHY : k == 2 ? KIND_AGGRESSIVE : k == 3 ? KIND_AGGRESSIVE_SHY : KIND_HUNGRY);
    }
  } else if (a->mode > 6) {
    int k = a->mode - 7;
    add_player(a, k == 0 ? KIND_HUNGRY : k == 1 ? KIND_HUNGRY_SHY : k == 2 ? KIND_AGGRESSIVE : k == 3 ? KIND_AGGRESSIVE_SHY : KIND_HUNGRY);
  }
  for (int i = 0; i < a->example_bots; i++) add_player(a, KIND_EXAMPLE); /* R: bench/main.cpp:21-24,31-35 */
}

OArena *ora_create_ex(int num_agents, int ticks_per_step, int arena_size, int pellet_regen, int num_pellets,
                      int num_viruses, int num_bots, int reward_type, int c_death, int mode, int recomb_ticks, int example_bots);
OArena *ora_create(int num_agents, int ticks_per_step, int arena_size, int pellet_regen, int num_pellets,
                   int num_viruses, int num_bots, int reward_type, int c_death, int mode, int recomb_ticks) {
  return ora_create_ex(num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets, num_viruses, num_bots, reward_type, c_death, mode, recomb_ticks, 0);
}
OArena *ora_create_ex(int num_agents, int ticks_per_step, int arena_size, int pellet_regen, int num_pellets,
                   int num_viruses, int num_bots, int reward_type, int c_death, int mode, int recomb_ticks, int example_bots) {
  OArena *a = (OArena *)calloc(1, sizeof(OArena));
  a->num_agents = num_agents; a->ticks_per_step = ticks_per_step; a->num_bots = num_bots; a->example_bots = example_bots;
  a->reward_type = reward_type != 0; a->c_death = c_death; a->mode = mode;
  a->W = (float)arena_size; a->H = (float)arena_size;
  a->target_pellets = (size_t)num_pellets; a->target_viruses = (size_t)num_viruses; a->pellet_regen = pellet_regen;
  a->mass_decay = 1; a->squared = 0; a->agent_mass = 25; a->regen = 1;
  set_mode(a, mode);
  a->recomb_ticks = recomb_ticks; a->clock = 0; a->id_counter = 1; a->main_agent_pid = -1;
  a->pids = (int *)calloc((size_t)(num_agents > 0 ? num_agents : 1), sizeof(int));
  a->dones = (uint8_t *)calloc((size_t)(num_agents > 0 ? num_agents : 1), 1);
  hm_init(&a->pmap);
  ora_mt_seed(a->mt, 5489u); ora_rand_seed(a->rnd, 1);
  env_reset(a); /* the reference ctor resets once (unseeded there; contents differ, counters agree) */
  return a;
}
void ora_destroy(OArena *a) {
  if (!a) return;
  clear_state(a); hm_free(&a->pmap);
  free(a->pellets); free(a->viruses); free(a->foods); free(a->players); free(a->pids); free(a->dones); free(a->ev_p); free(a->ev_v); free(a);
}
void ora_seed(OArena *a, unsigned s) { ora_mt_seed(a->mt, (uint64_t)s); ora_rand_seed(a->rnd, s); }
void ora_reset(OArena *a, int reset_ids) { if (reset_ids) a->id_counter = 1; env_reset(a); }

int ora_take_action(OArena *a, int pid, float dx, float dy, int action) { /* R: BaseEnvironment.hpp:162-176 */
  OPlayer *p = find_player(a, pid);
  if (!p) return -1;
  if (p->n_cells == 0) return 0;
  float ox = dx * 10.0f, oy = dy * 10.0f;
  float tx = player_x(p) + ox, ty = player_y(p) + oy;
  p->action = action; p->tx = tx; p->ty = ty;
  return 0;
}
int ora_take_actions(OArena *a, const float *dxdy, const int *act, int n) {
  if (n != a->num_agents) return -1;
  for (int i = 0; i < n; i++) if (ora_take_action(a, a->pids[i], dxdy[2 * i], dxdy[2 * i + 1], act[i]) != 0) return -1;
  return 0;
}
void ora_respawn_dead(OArena *a) { for (int n = a->pmap.head; n != -1; n = a->pmap.next[n]) if (a->players[n].n_cells == 0) respawn(a, &a->players[n]); }

static int env_masses(OArena *a, double *out) { /* R: BaseEnvironment.hpp:125-138 */
  int k = 0;
  for (int n = a->pmap.head; n != -1; n = a->pmap.next[n]) {
    OPlayer *p = &a->players[n];
    if (p->is_bot) continue;
    unsigned m = player_mass(p);
    out[k++] = (double)m;
    if (a->mode == 3 && m >= 23000u) a->dones[0] = 1;
  }
  return k;
}
int ora_step(OArena *a, double *rewards_out) { /* R: BaseEnvironment.hpp:89-122 */
  double *before = (double *)malloc(sizeof(double) * (size_t)(a->n_players + 1));
  a->respawned_flag = 0;
  env_masses(a, before); /* masses<float>: exact for masses < 2^24 */
  for (int t = 0; t < a->ticks_per_step; t++) ora_tick(a, 1.0 / 30.0);
  if (a->screen_hook) { /* R: environment/envs/ScreenEnvironment.hpp:233-243, called per agent at BaseEnvironment.hpp:96-97 */
    for (int i = 0; i < a->num_agents; i++) {
      OPlayer *p = find_player(a, a->pids[i]);
      if (p && p->n_cells == 0) { respawn(a, p); a->respawned_flag = 1; }
    }
  }
  if (a->mode == 0) ora_respawn_dead(a);
  else if (a->mode > 6) {
    for (int n = a->pmap.head; n != -1; n = a->pmap.next[n]) {
      int dead = a->players[n].n_cells == 0;
      a->dones[0] = (uint8_t)(dead | a->respawned_flag);
      if (dead) { a->dones[0] = 1; break; }
    }
  }
  int k = env_masses(a, rewards_out);
  if (a->reward_type) for (int i = 0; i < a->num_agents && i < k; i++) {
    float b = (float)before[i]; float sub = b - (float)(a->respawned_flag ? a->c_death : 0);
    rewards_out[i] -= (double)sub;
  }
  free(before);
  return k;
}
void ora_set_screen_hook(OArena *a, int on) { a->screen_hook = on != 0; }
void ora_dones(OArena *a, uint8_t *out) { for (int i = 0; i < a->num_agents; i++) out[i] = a->dones[i]; }
int ora_pids(OArena *a, int *out) { for (int i = 0; i < a->num_agents; i++) out[i] = a->pids[i]; return a->num_agents; }
int ora_set_player(OArena *a, int pid, float tx, float ty, int action) { OPlayer *p = find_player(a, pid); if (!p) return -1; p->tx = tx; p->ty = ty; p->action = action; return 0; }
long long ora_ticks(OArena *a) { return (long long)a->ticks; }
int ora_player_masses(OArena *a, int *pids, int *masses) {
  int k = 0;
  for (int n = a->pmap.head; n != -1; n = a->pmap.next[n]) { pids[k] = a->players[n].pid; masses[k] = (int)player_mass(&a->players[n]); k++; }
  return k;
}
int ora_last_events(OArena *a, int *pellet_idx, int cap_p, int *virus_idx, int cap_v, int *n_virus) {
  for (int i = 0; i < a->n_ev_p && i < cap_p; i++) pellet_idx[i] = a->ev_p[i];
  for (int i = 0; i < a->n_ev_v && i < cap_v; i++) virus_idx[i] = a->ev_v[i];
  *n_virus = a->n_ev_v;
  return a->n_ev_p;
}

/* ---- blob ------------------------------------------------------------------------------------- */
static inline uint32_t fu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float uf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int ora_dump(OArena *a, uint32_t *buf, int cap) {
  int need = 8 + 3 * a->n_pellets + 7 * a->n_viruses + 5 * a->n_foods;
  for (int i = 0; i < a->n_players; i++) need += 17 + a->players[i].n_vticks + 9 * a->players[i].n_cells;
  if (need > cap) return -need;
  uint32_t *o = buf;
  *o++ = 0x31524741u; *o++ = (uint32_t)a->ticks; *o++ = (uint32_t)a->id_counter; *o++ = (uint32_t)a->next_pid;
  *o++ = (uint32_t)a->n_pellets; *o++ = (uint32_t)a->n_viruses; *o++ = (uint32_t)a->n_foods; *o++ = (uint32_t)a->n_players;
  for (int i = 0; i < a->n_pellets; i++) *o++ = fu(a->pellets[i].x);
  for (int i = 0; i < a->n_pellets; i++) *o++ = fu(a->pellets[i].y);
  for (int i = 0; i < a->n_pellets; i++) *o++ = (uint32_t)a->pellets[i].id;
  for (int i = 0; i < a->n_viruses; i++) *o++ = fu(a->viruses[i].x);
  for (int i = 0; i < a->n_viruses; i++) *o++ = fu(a->viruses[i].y);
  for (int i = 0; i < a->n_viruses; i++) *o++ = fu(a->viruses[i].vx);
  for (int i = 0; i < a->n_viruses; i++) *o++ = fu(a->viruses[i].vy);
  for (int i = 0; i < a->n_viruses; i++) *o++ = a->viruses[i].mass;
  for (int i = 0; i < a->n_viruses; i++) *o++ = (uint32_t)a->viruses[i].hits;
  for (int i = 0; i < a->n_viruses; i++) *o++ = (uint32_t)a->viruses[i].id;
  for (int i = 0; i < a->n_foods; i++) *o++ = fu(a->foods[i].x);
  for (int i = 0; i < a->n_foods; i++) *o++ = fu(a->foods[i].y);
  for (int i = 0; i < a->n_foods; i++) *o++ = fu(a->foods[i].vx);
  for (int i = 0; i < a->n_foods; i++) *o++ = fu(a->foods[i].vy);
  for (int i = 0; i < a->n_foods; i++) *o++ = (uint32_t)a->foods[i].id;
  for (int n = a->pmap.head; n != -1; n = a->pmap.next[n]) {
    OPlayer *p = &a->players[n];
    *o++ = (uint32_t)p->pid; *o++ = (uint32_t)p->is_bot; *o++ = (uint32_t)p->n_cells; *o++ = (uint32_t)p->action;
    *o++ = fu(p->tx); *o++ = fu(p->ty); *o++ = (uint32_t)p->split_cd; *o++ = (uint32_t)p->feed_cd;
    *o++ = (uint32_t)p->elapsed; *o++ = (uint32_t)p->last_decay; *o++ = fu(p->anti_team);
    *o++ = (uint32_t)p->food_eaten; *o++ = p->highest_mass; *o++ = (uint32_t)p->cells_eaten; *o++ = (uint32_t)p->viruses_eaten;
    *o++ = p->min_mass_cell; *o++ = (uint32_t)p->n_vticks;
    for (int i = 0; i < p->n_vticks; i++) *o++ = (uint32_t)p->vticks[i];
    for (int i = 0; i < p->n_cells; i++) {
      OCell *c = &p->cells[i];
      *o++ = fu(c->x); *o++ = fu(c->y); *o++ = fu(c->vx); *o++ = fu(c->vy); *o++ = fu(c->svx); *o++ = fu(c->svy);
      *o++ = c->mass; *o++ = (uint32_t)c->id;
      long long rem = c->deadline - a->clock; *o++ = (uint32_t)(rem > 0 ? rem : 0);
    }
  }
  return (int)(o - buf);
}
int ora_load(OArena *a, const uint32_t *b, int words) {
  if (words < 8 || b[0] != 0x31524741u) return -1;
  uint32_t np = b[4], nv = b[5], nf = b[6], npl = b[7];
  if ((int)npl != a->n_players) return -2;
  const uint32_t *p = b + 8;
  a->ticks = b[1];
  a->n_pellets = 0;
  for (uint32_t i = 0; i < np; i++) { add_pellet_at(a, uf(p[i]), uf(p[np + i])); a->pellets[i].id = (int)p[2 * np + i]; }
  p += 3 * np;
  a->n_viruses = 0;
  for (uint32_t i = 0; i < nv; i++) {
    add_virus_full(a, uf(p[i]), uf(p[nv + i]), uf(p[2 * nv + i]), uf(p[3 * nv + i]));
    a->viruses[i].mass = p[4 * nv + i]; a->viruses[i].hits = (int)p[5 * nv + i]; a->viruses[i].id = (int)p[6 * nv + i];
  }
  p += 7 * nv;
  a->n_foods = 0;
  for (uint32_t i = 0; i < nf; i++) {
    GROW(a->foods, a->n_foods, a->cap_foods, OFood);
    OFood *f = &a->foods[a->n_foods++];
    f->x = uf(p[i]); f->y = uf(p[nf + i]); f->vx = uf(p[2 * nf + i]); f->vy = uf(p[3 * nf + i]); f->id = (int)p[4 * nf + i];
  }
  p += 5 * nf;
  for (int n = a->pmap.head; n != -1; n = a->pmap.next[n]) {
    OPlayer *pl = &a->players[n];
    if ((int)p[0] != pl->pid) return -3;
    uint32_t nc = p[2];
    pl->action = (int)p[3]; pl->tx = uf(p[4]); pl->ty = uf(p[5]); pl->split_cd = p[6]; pl->feed_cd = p[7];
    pl->elapsed = (int)p[8]; pl->last_decay = (int)p[9]; pl->anti_team = uf(p[10]);
    pl->food_eaten = (int)p[11]; pl->highest_mass = p[12]; pl->cells_eaten = (int)p[13]; pl->viruses_eaten = (int)p[14];
    pl->min_mass_cell = p[15];
    uint32_t nt = p[16];
    pl->n_vticks = 0;
    for (uint32_t i = 0; i < nt; i++) { GROW(pl->vticks, pl->n_vticks, pl->cap_vticks, int); pl->vticks[pl->n_vticks++] = (int)p[17 + i]; }
    p += 17 + nt;
    pl->n_cells = 0;
    for (uint32_t i = 0; i < nc; i++, p += 9) {
      OCell *c = player_push_cell(pl);
      c->x = uf(p[0]); c->y = uf(p[1]); c->vx = uf(p[2]); c->vy = uf(p[3]); c->svx = uf(p[4]); c->svy = uf(p[5]);
      cell_set_mass(c, p[6]); c->id = (int)p[7]; c->deadline = a->clock + (long long)p[8];
    }
  }
  a->id_counter = (int)b[2]; a->next_pid = (int)b[3];
  if (p - b != words) return -4;
  return 0;
}

/* ---- grid observation.  R: environment/envs/GridEnvironment.hpp:91-123, 188-296 ---------------- */
int ora_grid_obs(OArena *a, int agent_index, int G, int observe_cells, int observe_others, int observe_viruses,
                 int observe_pellets, int32_t *out) {
  OPlayer *pl = find_player(a, a->pids[agent_index]);
  int C = 1 + observe_cells + 2 * observe_others + 2 * observe_viruses + 2 * observe_pellets;
  memset(out, 0, sizeof(int32_t) * (size_t)C * (size_t)G * (size_t)G);
  float view = clampf((float)(2u * player_mass(pl)), 100.0f, 300.0f);
  float centering = (float)(G / 2.0);
  float px = player_x(pl), py = player_y(pl);
  int ch = 0;
  for (int i = 0; i < G; i++)
    for (int j = 0; j < G; j++) {
      float xd = (float)i - centering, yd = (float)j - centering;
      float dx = xd * view; dx = dx / (float)G;
      float dy = yd * view; dy = dy / (float)G;
      float lx = px + dx, ly = py + dy;
      int inb = 0 <= lx && lx < a->W && 0 <= ly && ly < a->H;
      out[(size_t)ch * G * G + (size_t)i * G + j] = inb ? 0 : -1;
    }
#define W2G(ex, ey, gx, gy) do { float ddx = (ex) - px, ddy = (ey) - py; float t1 = (float)G * ddx; t1 = t1 / view; gx = f2i(t1 + centering); \
    float t2 = (float)G * ddy; t2 = t2 / view; gy = f2i(t2 + centering); } while (0)
#define INSIDE(gx, gy) (0 <= (gx) && (gx) < G && 0 <= (gy) && (gy) < G)
  if (observe_pellets) {
    ch++;
    for (int i = 0; i < a->n_pellets; i++) { int gx, gy; W2G(a->pellets[i].x, a->pellets[i].y, gx, gy); if (INSIDE(gx, gy)) out[(size_t)ch * G * G + (size_t)gx * G + gy] = 1; }
    ch++;
    for (int i = 0; i < a->n_pellets; i++) { int gx, gy; W2G(a->pellets[i].x, a->pellets[i].y, gx, gy); if (INSIDE(gx, gy)) out[(size_t)ch * G * G + (size_t)gx * G + gy] += 1; }
  }
  if (observe_viruses) {
    ch++;
    for (int i = 0; i < a->n_viruses; i++) { int gx, gy; W2G(a->viruses[i].x, a->viruses[i].y, gx, gy); if (INSIDE(gx, gy)) out[(size_t)ch * G * G + (size_t)gx * G + gy] = (int)a->viruses[i].mass; }
    ch++;
    for (int i = 0; i < a->n_viruses; i++) { int gx, gy; W2G(a->viruses[i].x, a->viruses[i].y, gx, gy); if (INSIDE(gx, gy)) out[(size_t)ch * G * G + (size_t)gx * G + gy] += (int)a->viruses[i].mass; }
  }
  if (observe_cells) {
    ch++;
    for (int i = 0; i < pl->n_cells; i++) { int gx, gy; W2G(pl->cells[i].x, pl->cells[i].y, gx, gy); if (INSIDE(gx, gy)) out[(size_t)ch * G * G + (size_t)gx * G + gy] += (int)pl->cells[i].mass; }
  }
  if (observe_others) {
    ch++;
    for (int n = a->pmap.head; n != -1; n = a->pmap.next[n]) {
      OPlayer *o = &a->players[n];
      if (o->pid == pl->pid) continue;
      for (int i = 0; i < o->n_cells; i++) { int gx, gy; W2G(o->cells[i].x, o->cells[i].y, gx, gy); if (INSIDE(gx, gy)) { int32_t *d = &out[(size_t)ch * G * G + (size_t)gx * G + gy]; int m = (int)o->cells[i].mass; *d = (*d == 0) ? m : (*d < m ? *d : m); } }
      for (int i = 0; i < o->n_cells; i++) { int gx, gy; W2G(o->cells[i].x, o->cells[i].y, gx, gy); if (INSIDE(gx, gy)) { int32_t *d = &out[(size_t)(ch + 1) * G * G + (size_t)gx * G + gy]; int m = (int)o->cells[i].mass; *d = (*d > m ? *d : m); } }
    }
  }
#undef W2G
#undef INSIDE
  return C;
}

long long ora_run_random(OArena *a, long long ticks, double dt, unsigned policy_seed, int allow_actions) {
  uint64_t s = 0x9E3779B97F4A7C15ull * ((uint64_t)policy_seed + 1);
  for (long long t = 0; t < ticks; t++) {
    if (t % 4 == 0) {
      for (int i = 0; i < a->num_agents; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; float dx = (float)((double)(s >> 40) / 8388608.0 - 1.0);
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; float dy = (float)((double)(s >> 40) / 8388608.0 - 1.0);
        int act = 0;
        if (allow_actions) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; act = (int)(s % 3); }
        ora_take_action(a, a->pids[i], dx, dy, act);
      }
    }
    ora_tick(a, dt);
    if (t % 4 == 3) ora_respawn_dead(a);
  }
  return ticks;
}
