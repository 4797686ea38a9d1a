/* TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-arena, sequential restatement of the reference engine's per-tick path
 *   agario::Engine<false>::tick                /root/reference/agario/engine/Engine.hpp:208-240
 *   agario::env::BaseEnvironment<false>::step  /root/reference/environment/envs/BaseEnvironment.hpp:89-122
 * Every function in agar_oracle.c cites the reference file:line it follows.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks this restatement word for word
 * (integers exact, fp32 bit-exact) against oracle/_ref/libagar_ref.so -- the unmodified reference
 * compiled from /root/reference by oracle/Makefile -- and against the golden blobs committed under
 * tests/golden/ that were generated from that same reference build (tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (agarcl_amd/) never does.
 */
#ifndef AGAR_ORACLE_H
#define AGAR_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct OArena OArena;

/* = BaseEnvironment ctor (BaseEnvironment.hpp:36-67): builds the engine and performs one reset()
 * (which consumes pids 0..num_agents+num_bots-1 exactly like the reference ctor does). */
OArena *ora_create(int num_agents, int ticks_per_step, int arena_size, int pellet_regen, int num_pellets,
                   int num_viruses, int num_bots, int reward_type, int c_death, int mode, int recomb_ticks);
/* ... plus `example_bots` ExampleBots (agario/bots/ExampleBot.hpp:45-51) added after the agents and the mode's bots at every reset, as
 * bench/main.cpp:21-24,31-35 adds them to a freshly reset engine; num_agents may then be 0 */
OArena *ora_create_ex(int num_agents, int ticks_per_step, int arena_size, int pellet_regen, int num_pellets,
                      int num_viruses, int num_bots, int reward_type, int c_death, int mode, int recomb_ticks, int example_bots);
void ora_destroy(OArena *a);
void ora_seed(OArena *a, unsigned s);                 /* BaseEnvironment.hpp:211, Engine.hpp:242-245 */
void ora_reset(OArena *a, int reset_ids);             /* BaseEnvironment.hpp:179-204 */
int ora_take_actions(OArena *a, const float *dxdy, const int *act, int n); /* :141-176 */
int ora_step(OArena *a, double *rewards_out);         /* :89-122 */
void ora_dones(OArena *a, uint8_t *out);              /* :206 */
int ora_pids(OArena *a, int *out);

void ora_tick(OArena *a, double dt);                  /* Engine.hpp:208-240 */
int ora_set_player(OArena *a, int pid, float tx, float ty, int action);
int ora_take_action(OArena *a, int pid, float dx, float dy, int action);
void ora_respawn_dead(OArena *a);                     /* BaseEnvironment.hpp:73-81 */
long long ora_ticks(OArena *a);
int ora_player_masses(OArena *a, int *pids, int *masses);

int ora_dump(OArena *a, uint32_t *buf, int cap);      /* oracle/BLOB_FORMAT.md */
int ora_load(OArena *a, const uint32_t *blob, int words);

/* eat-event log of the most recent ora_tick: pellet indices appended to pellets_to_remove
 * (Engine.hpp:992) in order, then virus indices (Engine.hpp:1243). */
int ora_last_events(OArena *a, int *pellet_idx, int cap_p, int *virus_idx, int cap_v, int *n_virus);

/* GridObservation::add_frame (environment/envs/GridEnvironment.hpp:91-123) for agent `agent_index`,
 * frame 0, into out[C][G][G] (int32).  Returns the channel count. */
int ora_grid_obs(OArena *a, int agent_index, int grid_size, int observe_cells, int observe_others,
                 int observe_viruses, int observe_pellets, int32_t *out);

/* same counter-based random policy as ref_run_random in ref_harness.cpp (CPU baseline timing) */
long long ora_run_random(OArena *a, long long ticks, double dt, unsigned policy_seed, int allow_actions);

/* helpers exposed for unit tests of the libstdc++/glibc emulations */
void ora_mt_seed(uint64_t *mt /*[313]*/, uint64_t seed);
uint64_t ora_mt_next(uint64_t *mt /*[313]*/);
float ora_uniform_float(uint64_t *mt, float lo, float hi);
void ora_rand_seed(int32_t *st /*[35]*/, unsigned seed);
int ora_rand_next(int32_t *st);
/* iteration order of a fresh-or-reused libstdc++ unordered_map<unsigned short,...> after inserting
 * the given keys in order; `bucket_count_io`/`next_resize_io` carry the rehash-policy state across
 * clear() calls (pass 1 and 0 for a brand-new map). */
/* E5: the respawn hook of ScreenEnvironment::_partial_observation (ScreenEnvironment.hpp:233-243); parity UNPINNED
 * (that class needs OpenGL and cannot be built here) */
void ora_set_screen_hook(OArena *a, int on);
int ora_hash_order(const int *keys, int n, int *order_out, int *bucket_count_io, int *next_resize_io);
/* libstdc++ std::sort (introsort) on (key=float, payload=int) pairs, comparator key< */
void ora_std_sort_by_float(float *keys, int *payload, int n);

#ifdef __cplusplus
}
#endif
#endif
