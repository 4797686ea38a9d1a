// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or executed from the product path.
//
// Thin extern "C" driver over the *unmodified* reference engine, compiled from the sources where
// they lie under /root/reference (see oracle/Makefile; output goes to oracle/_ref/ only).
//   agario::Engine<false>                       /root/reference/agario/engine/Engine.hpp
//   agario::env::BaseEnvironment<false>         /root/reference/environment/envs/BaseEnvironment.hpp
//
// What this file adds (and nothing else):
//   * a flat "state blob" dump/load so that the reference, the C restatement (oracle/agar_oracle.c)
//     and the HIP engine can be compared word for word (layout: oracle/BLOB_FORMAT.md);
//   * a VIRTUAL std::chrono::steady_clock::now().  The reference's recombine timer is wall-clock
//     (agario/core/Entities.hpp:127,183-193, settings.hpp:13), which makes a tick-level comparison
//     impossible.  We do not touch or re-declare any reference header: we simply provide the
//     definition of the libstdc++ symbol std::chrono::steady_clock::now() in this shared object
//     (linked -Bsymbolic) so that the reference code, compiled as-is, reads a clock that advances
//     by exactly one tick per Engine::tick() call (it follows the engine's own tick counter).
//     10 s == `recomb_ticks` ticks.
//
// No reference source is copied here.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <mutex>
#include <sstream>
#include <vector>

// ---- virtual steady clock ---------------------------------------------------------------------
namespace {
struct VClock {
  // virtual tick counter = offset + the engine's own state.ticks (which Engine::tick increments
  // once at its very end, agario/engine/Engine.hpp:238, and reset() zeroes, GameState.hpp:61-67)
  const unsigned long *engine_ticks = nullptr;
  long long offset = 0;
  long long recomb = 300;  // ticks that make up RECOMBINE_TIMER_SEC (10 s)
  long long tick() const { return offset + (engine_ticks ? (long long)*engine_ticks : 0); }
  long long ns_at(long long t) const { return (long long)(((__int128)t * 10000000000LL) / recomb); }
  long long ns() const { return ns_at(tick()); }
};
thread_local VClock *g_clock = nullptr;
thread_local VClock g_default_clock;
}  // namespace

namespace std { namespace chrono { inline namespace _V2 {
steady_clock::time_point steady_clock::now() noexcept {
  const VClock *c = g_clock ? g_clock : &g_default_clock;
  return time_point(duration(c->ns()));
}
}}}

#include <agario/engine/Engine.hpp>
#include <agario/bots/ExampleBot.hpp>
#include <environment/envs/BaseEnvironment.hpp>

using Engine = agario::Engine<false>;
using Player = agario::Player<false>;
using Cell = agario::Cell<false>;
using Pellet = agario::Pellet<false>;
using Virus = agario::Virus<false>;
using Food = agario::Food<false>;
using Base = agario::env::BaseEnvironment<false>;

namespace {

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

struct RefEnv : public Base {
  VClock clock;
  using Base::Base;
  // ScreenEnvironment's respawn hook (environment/envs/ScreenEnvironment.hpp:233-243).  ScreenEnvironment itself needs OpenGL and
  // cannot be compiled here; the hook is the three statements below, which this harness restates (that much is NOT the
  // reference's own object code).  Everything around it -- BaseEnvironment::step calling _partial_observation per agent after the
  // ticks, is_main_player_respawned feeding the c_death reward term and the mode > 6 done flag (BaseEnvironment.hpp:89-122),
  // Engine::respawn -- is the unmodified reference, so the oracle's E5 path is pinned against it through this switch.
  bool screen_hook = false;
  void _partial_observation(int agent_index, int tick_index) override {
    Base::_partial_observation(agent_index, tick_index);
    if (!screen_hook) return;
    auto &player = this->engine_.player(this->pids_[agent_index]);
    if (player.dead()) { this->engine_.respawn(player); this->is_main_player_respawned = true; }
  }
  // bench/main.cpp:21-24,31-35: `example_bots` ExampleBots (the reference's own class, agario/bots/ExampleBot.hpp) join the freshly reset engine
  int example_bots = 0;
  void add_example_bots() { for (int i = 0; i < example_bots; i++) this->engine_.template add_player<agario::bot::ExampleBot<false>>(); }
  Engine &eng() { return this->engine_; }
  std::vector<agario::pid> &pids() { return this->pids_; }
  void set_done(int i, bool v) { this->dones_[i] = v; }
};

struct Silence {  // the reference constructors print to std::cout
  std::streambuf *old;
  std::ostringstream sink;
  Silence() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~Silence() { std::cout.rdbuf(old); }
};

struct Use {  // select this env's virtual clock for the duration of a call
  VClock *prev;
  explicit Use(RefEnv *e) : prev(g_clock) { g_clock = &e->clock; }
  ~Use() { g_clock = prev; }
};

int remaining_ticks(const VClock &c, Cell &cell) {
  // smallest k >= 0 such that the clock at tick+k is >= the cell's timer
  long long timer = cell._recombine_timer.time_since_epoch().count();
  if (cell.can_recombine()) return 0;  // also latches exactly like the engine would
  for (int k = 0; k <= (int)c.recomb + 1; k++)
    if (c.ns_at(c.tick() + k) >= timer) return k;
  return (int)c.recomb + 1;
}

}  // namespace

extern "C" {

void *ref_env_create_ex(int num_agents, int ticks_per_step, int arena_size, int pellet_regen, int num_pellets, int num_viruses, int num_bots,
                        int reward_type, int c_death, int mode, int recomb_ticks, int example_bots);
void *ref_env_create(int num_agents, int ticks_per_step, int arena_size, int pellet_regen,
                     int num_pellets, int num_viruses, int num_bots, int reward_type, int c_death,
                     int mode, int recomb_ticks) {
  return ref_env_create_ex(num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets, num_viruses, num_bots, reward_type, c_death, mode, recomb_ticks, 0);
}
void *ref_env_create_ex(int num_agents, int ticks_per_step, int arena_size, int pellet_regen, int num_pellets, int num_viruses, int num_bots,
                        int reward_type, int c_death, int mode, int recomb_ticks, int example_bots) {
  static std::mutex ctor_mutex;  // the std::cout redirection below is process-global
  std::lock_guard<std::mutex> lock(ctor_mutex);
  Silence s;
  try {
    VClock tmp; tmp.recomb = recomb_ticks; VClock *prev = g_clock; g_clock = &tmp;
    auto *e = new RefEnv(num_agents, ticks_per_step, arena_size, pellet_regen != 0, num_pellets,
                         num_viruses, num_bots, reward_type != 0, c_death, mode, false);
    e->clock.recomb = recomb_ticks;
    e->clock.engine_ticks = &e->eng().state.ticks;
    e->example_bots = example_bots;
    e->add_example_bots();   // (the constructor has reset once: BaseEnvironment.hpp:66)
    g_clock = prev;
    return e;
  } catch (const std::exception &ex) {
    std::cerr << "ref_env_create: " << ex.what() << std::endl;
    return nullptr;
  }
}

void ref_env_destroy(void *h) { delete (RefEnv *)h; }
void ref_env_set_screen_hook(void *h, int on) { ((RefEnv *)h)->screen_hook = on != 0; }

void ref_env_seed(void *h, unsigned s) { ((RefEnv *)h)->seed((int)s); }

// reset_ids != 0 : restart the process-global entity id counter (agario/core/Ball.hpp:15,97) so that
// this arena's first entity gets id 2, as in a fresh process.
void ref_env_reset(void *h, int reset_ids) {
  auto *e = (RefEnv *)h; Use u(e);
  if (reset_ids) agario::Ball::global_id = 1;
  e->clock.offset += (long long)e->eng().state.ticks;  // keep the virtual clock monotonic across reset
  e->reset();
  e->add_example_bots();
}

int ref_env_take_actions(void *h, const float *dxdy, const int *act, int n) {
  auto *e = (RefEnv *)h; Use u(e);
  std::vector<agario::env::Action> a;
  for (int i = 0; i < n; i++) a.emplace_back(dxdy[2 * i], dxdy[2 * i + 1], (agario::action)act[i]);
  try { e->take_actions(a); } catch (const std::exception &) { return -1; }
  return 0;
}

// BaseEnvironment::step() (environment/envs/BaseEnvironment.hpp:89-122)
int ref_env_step(void *h, double *rewards_out) {
  auto *e = (RefEnv *)h; Use u(e);
  auto r = e->step();
  for (size_t i = 0; i < r.size(); i++) rewards_out[i] = r[i];
  return (int)r.size();
}

void ref_env_dones(void *h, uint8_t *out) {
  auto *e = (RefEnv *)h;
  auto d = e->dones();
  for (size_t i = 0; i < d.size(); i++) out[i] = d[i] ? 1 : 0;
}

int ref_env_pids(void *h, int *out) {
  auto *e = (RefEnv *)h;
  int n = 0;
  for (auto p : e->pids()) out[n++] = p;
  return n;
}

// ---- engine-level driving ---------------------------------------------------------------------
void ref_tick(void *h, double dt) {
  auto *e = (RefEnv *)h; Use u(e);
  e->eng().tick(agario::time_delta(dt));
}

int ref_set_player(void *h, int pid, float tx, float ty, int action) {
  auto *e = (RefEnv *)h;
  try {
    auto &p = e->eng().player((agario::pid)pid);
    p.target = agario::Location(tx, ty);
    p.action = (agario::action)action;
  } catch (const std::exception &) { return -1; }
  return 0;
}

// BaseEnvironment::take_action for one pid (target relative to the mass-weighted centroid)
int ref_take_action(void *h, int pid, float dx, float dy, int action) {
  auto *e = (RefEnv *)h; Use u(e);
  try { e->take_action((agario::pid)pid, dx, dy, action); } catch (const std::exception &) { return -1; }
  return 0;
}

void ref_respawn_dead(void *h) {
  auto *e = (RefEnv *)h; Use u(e);
  e->repsawn_all_players();
}

long long ref_ticks(void *h) { return (long long)((RefEnv *)h)->eng().ticks(); }

// masses of all players in map-iteration order (dead => 0)
int ref_player_masses(void *h, int *pids, int *masses) {
  auto *e = (RefEnv *)h;
  int n = 0;
  for (auto &pr : e->eng().state.players) { pids[n] = pr.first; masses[n] = (int)pr.second->mass(); n++; }
  return n;
}

// ---- state blob (oracle/BLOB_FORMAT.md) --------------------------------------------------------
int ref_dump(void *h, uint32_t *buf, int cap) {
  auto *e = (RefEnv *)h; Use u(e);
  auto &st = e->eng().state;
  std::vector<uint32_t> o;
  o.push_back(0x31524741u);
  o.push_back((uint32_t)st.ticks);
  o.push_back((uint32_t)agario::Ball::global_id);
  o.push_back((uint32_t)st.next_pid);
  o.push_back((uint32_t)st.pellets.size());
  o.push_back((uint32_t)st.viruses.size());
  o.push_back((uint32_t)st.foods.size());
  o.push_back((uint32_t)st.players.size());
  for (auto &p : st.pellets) o.push_back(f2u(p.x));
  for (auto &p : st.pellets) o.push_back(f2u(p.y));
  for (auto &p : st.pellets) o.push_back((uint32_t)p.id);
  for (auto &v : st.viruses) o.push_back(f2u(v.x));
  for (auto &v : st.viruses) o.push_back(f2u(v.y));
  for (auto &v : st.viruses) o.push_back(f2u(v.velocity.dx));
  for (auto &v : st.viruses) o.push_back(f2u(v.velocity.dy));
  for (auto &v : st.viruses) o.push_back((uint32_t)v.mass());
  for (auto &v : st.viruses) o.push_back((uint32_t)v.get_num_food_hits());
  for (auto &v : st.viruses) o.push_back((uint32_t)v.id);
  for (auto &f : st.foods) o.push_back(f2u(f.x));
  for (auto &f : st.foods) o.push_back(f2u(f.y));
  for (auto &f : st.foods) o.push_back(f2u(f.velocity.dx));
  for (auto &f : st.foods) o.push_back(f2u(f.velocity.dy));
  for (auto &f : st.foods) o.push_back((uint32_t)f.id);
  for (auto &pr : st.players) {
    Player &p = *pr.second;
    o.push_back((uint32_t)p.pid());
    o.push_back(p.is_bot ? 1u : 0u);
    o.push_back((uint32_t)p.cells.size());
    o.push_back((uint32_t)p.action);
    o.push_back(f2u(p.target.x));
    o.push_back(f2u(p.target.y));
    o.push_back((uint32_t)p.split_cooldown);
    o.push_back((uint32_t)p.feed_cooldown);
    o.push_back((uint32_t)p.elapsed_ticks);
    o.push_back((uint32_t)p.last_decay_tick);
    o.push_back(f2u(p.anti_team_decay));
    o.push_back((uint32_t)p.food_eaten);
    o.push_back((uint32_t)p.highest_mass);
    o.push_back((uint32_t)p.cells_eaten);
    o.push_back((uint32_t)p.viruses_eaten);
    o.push_back((uint32_t)p.get_min_mass_cell());
    o.push_back((uint32_t)p.virus_eaten_ticks.size());
    for (int t : p.virus_eaten_ticks) o.push_back((uint32_t)t);
    for (auto &c : p.cells) {
      o.push_back(f2u(c.x)); o.push_back(f2u(c.y));
      o.push_back(f2u(c.velocity.dx)); o.push_back(f2u(c.velocity.dy));
      o.push_back(f2u(c.splitting_velocity.dx)); o.push_back(f2u(c.splitting_velocity.dy));
      o.push_back((uint32_t)c.mass());
      o.push_back((uint32_t)c.id);
      o.push_back((uint32_t)remaining_ticks(e->clock, c));
    }
  }
  if ((int)o.size() > cap) return -(int)o.size();
  std::memcpy(buf, o.data(), o.size() * 4);
  return (int)o.size();
}

// Overwrite the arena with the contents of a blob.  The players must already exist with the same
// pids in the same iteration order (their dynamic type -- bot or agent -- is kept).
int ref_load(void *h, const uint32_t *b, int words) {
  auto *e = (RefEnv *)h; Use u(e);
  auto &st = e->eng().state;
  if (words < 8 || b[0] != 0x31524741u) return -1;
  const uint32_t *p = b + 8;
  uint32_t np = b[4], nv = b[5], nf = b[6], npl = b[7];
  if (npl != st.players.size()) return -2;
  e->clock.offset += (long long)st.ticks - (long long)b[1];  // clock stays continuous
  st.ticks = b[1];
  st.pellets.clear();
  for (uint32_t i = 0; i < np; i++) {
    st.pellets.emplace_back(agario::Location(u2f(p[i]), u2f(p[np + i])));
    st.pellets.back().id = (int)p[2 * np + i];
  }
  p += 3 * np;
  st.viruses.clear();
  for (uint32_t i = 0; i < nv; i++) {
    Virus v(agario::Location(u2f(p[i]), u2f(p[nv + i])), agario::Velocity(agario::distance(u2f(p[2 * nv + i])), agario::distance(u2f(p[3 * nv + i]))));
    v.set_mass(p[4 * nv + i]);
    v.set_num_food_hits((int)p[5 * nv + i]);
    v.id = (int)p[6 * nv + i];
    st.viruses.emplace_back(std::move(v));
  }
  p += 7 * nv;
  st.foods.clear();
  for (uint32_t i = 0; i < nf; i++) {
    Food f(agario::Location(u2f(p[i]), u2f(p[nf + i])), agario::Velocity(agario::distance(u2f(p[2 * nf + i])), agario::distance(u2f(p[3 * nf + i]))));
    f.id = (int)p[4 * nf + i];
    st.foods.emplace_back(std::move(f));
  }
  p += 5 * nf;
  for (auto &pr : st.players) {
    Player &pl = *pr.second;
    if (p[0] != pl.pid()) return -3;
    uint32_t nc = p[2];
    pl.action = (agario::action)p[3];
    pl.target = agario::Location(u2f(p[4]), u2f(p[5]));
    pl.split_cooldown = p[6];
    pl.feed_cooldown = p[7];
    pl.elapsed_ticks = (int)p[8];
    pl.last_decay_tick = (int)p[9];
    pl.anti_team_decay = u2f(p[10]);
    pl.food_eaten = (int)p[11];
    pl.highest_mass = p[12];
    pl.cells_eaten = (int)p[13];
    pl.viruses_eaten = (int)p[14];
    pl.set_min_mass_cell(p[15]);
    uint32_t nt = p[16];
    pl.virus_eaten_ticks.clear();
    for (uint32_t i = 0; i < nt; i++) pl.virus_eaten_ticks.push_back((int)p[17 + i]);
    p += 17 + nt;
    pl.cells.clear();
    for (uint32_t i = 0; i < nc; i++, p += 9) {
      pl.add_cell(agario::Location(u2f(p[0]), u2f(p[1])), (agario::mass)p[6]);
      Cell &c = pl.cells.back();
      c.velocity = agario::Velocity(agario::distance(u2f(p[2])), agario::distance(u2f(p[3])));
      c.splitting_velocity = agario::Velocity(agario::distance(u2f(p[4])), agario::distance(u2f(p[5])));
      c.id = (int)p[7];
      // remaining ticks until recombine-eligible -> absolute virtual-clock deadline
      c.reset_recombine_timer();
      long long k = (long long)p[8];
      long long ns = (k == 0) ? e->clock.ns() : e->clock.ns_at(e->clock.tick() + k - e->clock.recomb) + 10000000000LL;
      c._recombine_timer = agario::real_time(std::chrono::nanoseconds(ns));
    }
  }
  agario::Ball::global_id = (int)b[2];
  st.next_pid = (agario::pid)b[3];
  if (p - b != words) return -4;
  return 0;
}

// ---- JSON snapshots: the reference's own save_env_state / load_env_state ---------
// (environment/envs/BaseEnvironment.hpp:213-343, agario/engine/Engine.hpp:247-348)
int ref_env_save_json(void *h, const char *path) {
  auto *e = (RefEnv *)h; Use u(e);
  try { e->save_env_state(path); return 0; } catch (const std::exception &ex) { std::cerr << "ref_env_save_json: " << ex.what() << std::endl; return -1; }
}
// reset_ids != 0: the entity id counter restarts at 1 before the load (fresh-process behaviour).  Note that the
// reference's reset() is a no-op ever after (is_loading_env_state stays set, BaseEnvironment.hpp:180-181).
int ref_env_load_json(void *h, const char *path, int reset_ids) {
  auto *e = (RefEnv *)h; Use u(e);
  static std::mutex m; std::lock_guard<std::mutex> lock(m);
  Silence s;  // load_env_state prints the agents' pids / names
  try {
    if (reset_ids) agario::Ball::global_id = 1;
    e->clock.offset += (long long)e->eng().state.ticks;  // Engine::load_env_state zeroes state.ticks: keep the clock monotonic
    e->load_env_state(path);
    return 0;
  } catch (const std::exception &ex) { std::cerr << "ref_env_load_json: " << ex.what() << std::endl; return -1; }
}

void ref_set_global_id(int v) { agario::Ball::global_id = v; }
int ref_get_global_id() { return agario::Ball::global_id; }

// ---- timing helper for bench.py's cpu_baseline leg ---------------------------------------------
// Runs `ticks` engine ticks with a fixed pseudo-random policy on the agent (pid list order), the
// policy being the same counter-based generator the product bench uses.  Returns ticks executed.
long long ref_run_random(void *h, long long ticks, double dt, unsigned policy_seed, int allow_actions) {
  auto *e = (RefEnv *)h; Use u(e);
  auto &eng = e->eng();
  uint64_t s = 0x9E3779B97F4A7C15ull * (policy_seed + 1);
  auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (long long t = 0; t < ticks; t++) {
    if (t % 4 == 0) {
      for (auto pid : e->pids()) {
        float dx = (float)((next() >> 40) / 8388608.0 - 1.0);
        float dy = (float)((next() >> 40) / 8388608.0 - 1.0);
        int a = allow_actions ? (int)(next() % 3) : 0;
        e->take_action(pid, dx, dy, a);
      }
    }
    eng.tick(agario::time_delta(dt));
    if (t % 4 == 3) e->repsawn_all_players();
  }
  return ticks;
}

}  // extern "C"
