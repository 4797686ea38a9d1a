/* agarcl_batch.h -- C ABI of the MI355X-native batched Agar.io engine (libagarcl_hip.so).
 *
 * Drop-in boundary.  The reference exposes ONE arena per object through the pybind11 module `agarcl`
 * (/root/reference/environment/bindings.cpp:94-376).  Nothing in the reference calls a C ABI today;
 * these entry points are what a maintainer binds instead of the C++ classes (INTEGRATION.md shows
 * the pybind11 stub).  Each function cites the reference interface it replaces.  All arenas of one
 * `agarcl_env` are stepped in lock-step by one kernel launch; the reference's single-arena classes
 * are the num_arenas == 1 case.
 *
 * Conventions: every call returns 0 on success or a negative AGARCL_E_* code and never throws;
 * agarcl_last_error() returns a thread-local message (the reference throws EngineException /
 * EnvironmentException, Engine.hpp:25-27, BaseEnvironment.hpp:19-21 -> Python RuntimeError).
 * Not re-entrant per env; distinct envs may be used from distinct threads.
 * Pointers named *_dev are device (HBM) pointers, *_host are host pointers.
 */
#ifndef AGARCL_BATCH_H
#define AGARCL_BATCH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AGARCL_OK 0
#define AGARCL_E_INVALID (-1)     /* bad argument (e.g. action count mismatch, BaseEnvironment.hpp:142-144) */
#define AGARCL_E_MODE (-2)        /* invalid mode number (Engine.hpp:413-414) */
#define AGARCL_E_UNSUPPORTED (-3) /* configuration the HIP path does not implement yet (fails loudly) */
#define AGARCL_E_HIP (-4)         /* HIP runtime error / no device */
#define AGARCL_E_NOMEM (-5)
#define AGARCL_E_CAPACITY (-6)    /* blob does not fit the arena capacities */

/* per-arena sticky flag bits (agarcl_get_flags): a fixed-capacity SoA array overflowed -- the arena
 * has diverged from the unbounded std::vector semantics of the reference and must be reset. */
#define AGARCL_F_CELLS_OVERFLOW 1u
#define AGARCL_F_FOODS_OVERFLOW 2u
#define AGARCL_F_VIRUSES_OVERFLOW 4u
#define AGARCL_F_EVENTS_OVERFLOW 8u   /* more eat events in one tick than the arena holds: 256 (+ a spill area of 64 per pellet slot in dense arenas) */
#define AGARCL_F_VTICKS_OVERFLOW 16u
#define AGARCL_F_MASS_LUT_OVERFLOW 32u
#define AGARCL_F_PELLETS_OVERFLOW 64u

/* Mirrors the positional constructor arguments of GridEnvironment / BaseEnvironment
 * (bindings.cpp:102; BaseEnvironment.hpp:36-67) plus the engine time step and SoA capacities. */
typedef struct agarcl_config {
  int32_t num_agents;     /* RL-controlled players per arena */
  int32_t ticks_per_step; /* Engine::tick calls per step() (BaseEnvironment.hpp:93-94) */
  int32_t arena_size;     /* square arena (BaseEnvironment.hpp:54) */
  int32_t pellet_regen;
  int32_t num_pellets;
  int32_t num_viruses;
  int32_t num_bots;
  int32_t reward_type;    /* 0: reward = mass, 1: reward = mass difference (BaseEnvironment.hpp:116-121) */
  int32_t c_death;
  int32_t mode_number;    /* Engine.hpp:367-416 */
  double dt;              /* seconds per tick; 0 -> DEFAULT_DT = 1/30 (BaseEnvironment.hpp:14) */
  int32_t cap_cells;      /* cells per player, 0 -> 32 (reference: unbounded vector, nominal limit 14) */
  int32_t cap_viruses;    /* 0 -> num_viruses + 64; 16 when num_viruses == 0 (such an arena never grows a virus; a state LOADED into it may bring up to 16) */
  int32_t cap_foods;      /* 0 -> max(128, 16 per player), less up to 32 where that puts an arena's LDS block under 10 240 bytes = 16 arenas per compute unit
                           *      (agent + 1 bot: 107) (ejected foods live in LDS during a launch: 16 bytes each) */
  /* ScreenEnvironment semantics (environment/envs/ScreenEnvironment.hpp:233-243): a dead agent is respawned right
   * after the ticks of a step in EVERY mode, and that step's rewards get + c_death (BaseEnvironment.hpp:116-120) */
  int32_t screen_respawn;
  /* replaces: bench/main.cpp:15-24 `engine.add_player<ExampleBot>()` x N (agario/bots/ExampleBot.hpp:45-51: action none, target = own
   * location): N ExampleBots join every arena after the agents and the mode's bots at every reset.  With them num_agents may be 0 (an
   * engine without a Player, as the reference's Tick benchmark builds it; drive it with agarcl_tick).  Players per arena <= 32. */
  int32_t example_bots;
  int32_t reserved[3];
} agarcl_config;

typedef struct agarcl_env agarcl_env;

const char *agarcl_last_error(void);
int agarcl_device_count(void);

/* replaces: GridEnvironment(...) ctor, bindings.cpp:102 (one object per arena).  Like the reference
 * ctor (BaseEnvironment.hpp:66) it performs one reset() (seed 5489-like default; call
 * agarcl_seed + agarcl_reset for reproducible episodes). */
int agarcl_create(const agarcl_config *cfg, int32_t num_arenas, int32_t device, agarcl_env **out);
/* replaces: object destruction / close(), bindings.cpp:134 */
int agarcl_destroy(agarcl_env *env);
/* adopt an existing HIP stream (hipStream_t) for all launches and copies; NULL = the legacy default
 * stream.  Until called, the env uses a private non-blocking stream. */
int agarcl_set_stream(agarcl_env *env, void *hip_stream);
int agarcl_sync(agarcl_env *env);
/* Timing on the env's own stream (measurement only; nothing in the reference corresponds): two HIP events created without the
 * system-scope fence of an ordinary event record.  which: 0 = start, 1 = stop.  agarcl_timer_elapsed_ms waits for the stop mark. */
int agarcl_timer_mark(agarcl_env *env, int32_t which);
int agarcl_timer_elapsed_ms(agarcl_env *env, float *ms);

/* replaces: seed(int), bindings.cpp:103 -> Engine::seed, Engine.hpp:242-245.  seeds_host[num_arenas];
 * NULL -> arena i gets base_seed + i. */
int agarcl_seed(agarcl_env *env, const uint32_t *seeds_host, uint32_t base_seed);
/* replaces: reset(), bindings.cpp:130 -> BaseEnvironment::reset, BaseEnvironment.hpp:179-204.
 * mask_host[num_arenas] (nullable = all).  reset_ids != 0 restarts the arena's entity-id counter
 * (the reference's is process-global and never restarts, core/Ball.hpp:15). */
int agarcl_reset(agarcl_env *env, const uint8_t *mask_host, int32_t reset_ids);
/* The same with a device-resident mask u8[num_arenas] (e.g. agarcl_dones_dev of a single-agent env): a pure
 * stream-ordered launch, no copy and no synchronisation -- what a vectorised RL loop calls every step to restart
 * finished arenas.  The mask is read when the launch executes. */
int agarcl_reset_device(agarcl_env *env, const uint8_t *mask_dev, int32_t reset_ids);

/* replaces: take_actions(list[(dx,dy,a)]), bindings.cpp:117-119 -> BaseEnvironment.hpp:141-176.
 * dxdy[num_arenas][num_agents][2] f32, act[num_arenas][num_agents] i32 (0 none, 1 feed, 2 split:
 * core/types.hpp:59-61).  on_device != 0: the pointers are HBM pointers and no copy is made -- they
 * must stay valid until the next agarcl_step has been enqueued. */
int agarcl_set_actions(agarcl_env *env, const float *dxdy, const int32_t *act, int32_t on_device);
/* replaces: step() -> list[float], bindings.cpp:132 -> BaseEnvironment::step, BaseEnvironment.hpp:89-122.
 * ticks <= 0 -> ticks_per_step.  Asynchronous on the env's stream. */
int agarcl_step(agarcl_env *env, int32_t ticks);
/* agarcl_set_actions(on_device = 1) + agarcl_step in one call: what a vectorised loop whose policy output already lives in
 * HBM does every step (take_actions + step, bindings.cpp:117-119,132).  One host call per env step instead of two. */
int agarcl_step_actions(agarcl_env *env, const float *dxdy_dev, const int32_t *act_dev, int32_t ticks);
/* engine-level tick without the env's action/reward logic (Engine::tick, Engine.hpp:208-240);
 * targets/actions as last set via agarcl_set_targets. */
int agarcl_tick(agarcl_env *env, int32_t ticks);
/* absolute targets (Player::target/action, core/Player.hpp:26-27) for every player slot:
 * txy[num_arenas][players][2], act[num_arenas][players]; host pointers. */
int agarcl_set_targets(agarcl_env *env, const float *txy_host, const int32_t *act_host);
/* BaseEnvironment::repsawn_all_players (BaseEnvironment.hpp:73-81) */
int agarcl_respawn_dead(agarcl_env *env);

/* results of the last step, HBM-resident: rewards f64[num_arenas][num_agents] (vector<double>,
 * BaseEnvironment.hpp:31,116-121), dones u8[...] (BaseEnvironment.hpp:206), masses i32[...]
 * (capacity flags: agarcl_get_flags / agarcl_poll_flags) */
const double *agarcl_rewards_dev(agarcl_env *env);
const uint8_t *agarcl_dones_dev(agarcl_env *env);
const int32_t *agarcl_masses_dev(agarcl_env *env);
/* (reward, done) as f32 pairs [num_arenas][num_agents][2] in a ring of AGARCL_PACKED_SLOTS contiguous buffers: the
 * k-th agarcl_step of an env writes slot k % 64, so an asynchronous gather (RCCL) of the last 8 / 16 / 32 steps' results -- one
 * contiguous block -- can overlap the next block of steps.  agarcl_last_slot = slot of the last step. */
#define AGARCL_PACKED_SLOTS 64
const float *agarcl_packed_dev(agarcl_env *env, int32_t slot);
int agarcl_last_slot(agarcl_env *env);
/* synchronising host copies */
int agarcl_get_rewards(agarcl_env *env, double *out_host);
int agarcl_get_dones(agarcl_env *env, uint8_t *out_host);
int agarcl_get_masses(agarcl_env *env, int32_t *out_host);
int agarcl_get_flags(agarcl_env *env, uint32_t *out_host);
/* Cheap watch on the flags: every kernel ORs the flags it raises into one device word, which the engine fetches
 * asynchronously (pinned memory, never waited for) every few steps.  Returns through *or_of_flags the OR of all
 * flags seen so far (0 = none); never blocks; may lag the device by up to ~64 steps.  agarcl_get_flags is the exact,
 * synchronising query.  Every reset restarts the watch: after agarcl_reset / agarcl_reset_device it reports the OR over the arenas that
 * are still flagged -- none after an unmasked reset, and none once a masked reset has covered every flagged arena. */
int agarcl_poll_flags(agarcl_env *env, uint32_t *or_of_flags);
/* live entity counts of the last step: i32[num_arenas][4] = pellets, viruses, foods, cells(all players) */
int agarcl_get_counts(agarcl_env *env, int32_t *out_host);
/* eat events of the LAST tick executed, per arena (pellets_to_remove / viruses_to_remove order,
 * Engine.hpp:992,1243): n_events i32[num_arenas][2], pellet_idx i32[num_arenas][cap] */
int agarcl_get_events(agarcl_env *env, int32_t *n_events_host, int32_t *pellet_idx_host, int32_t cap,
                      int32_t *virus_idx_host, int32_t cap_v);

/* replaces: configure_observation(dict) + observation_shape() + get_state() of GridEnvironment,
 * bindings.cpp:104-116,133 -> GridObservation::add_frame, GridEnvironment.hpp:91-123 (one frame: the state
 * after the last step).  Writes i32[num_arenas][num_agents][C][G][G], C = 1 + cells + 2*others + 2*viruses +
 * 2*pellets, into `out` (HBM pointer if on_device != 0, else a host buffer); returns C through *channels.
 * out == NULL only queries C.
 * on_device == 2: an HBM buffer that still holds THIS env's previous observation of the same configuration, unmodified (a
 * persistent observation tensor that is rewritten every step): only the words scattered into it last time are cleared instead of
 * streaming zeros over the whole tensor -- same contents as on_device == 1.  The first such call, and any call with another buffer
 * or configuration, clears everything.  The buffer is owned by that sequence of calls: a plain (on_device == 1) call of this env into
 * the same buffer voids the record and the next on_device == 2 call clears everything again; anything ELSE that writes into the buffer
 * between two on_device == 2 calls (another env, the caller) leaves words the incremental clear does not know about.
 * Which one a consumer gets: on_device == 1 is the stateless default (full clear, 2.1 GB of stores at 4096 x 8 x 128 x 128: ~395 us);
 * on_device == 2 is the opt-in fast path for a tensor that is rewritten every step (~115 us). */
int agarcl_grid_obs(agarcl_env *env, int32_t grid_size, int32_t observe_cells, int32_t observe_others,
                    int32_t observe_viruses, int32_t observe_pellets, int32_t *out, int32_t on_device,
                    int32_t *channels);

/* replaces: ScreenEnvironment::get_state() (bindings.cpp:157-168) = one frame of Renderer::render_screen read back with
 * glReadPixels (agario/rendering/renderer.hpp:163-185, FrameBufferObject.hpp:105), restated as rasterisation rules
 * (agarcl_amd/csrc/agar_screen.inl; rule-level parity only: see that file).  Writes u8[num_arenas][num_agents][height]
 * [width][3], rows bottom-up, i.e. per agent exactly the bytes ScreenObservation exposes as uint8 [1][W][H][3]
 * (environment/envs/ScreenEnvironment.hpp:24-128).  agent_view != 0: the 4-channel frame of
 * Renderer::multi_channel_render_screen (renderer.hpp:128-155) after ScreenObservation::post_processing_frame_data
 * (ScreenEnvironment.hpp:48-88), u8[...][height][width][4].  `out` is an HBM pointer if on_device != 0, else a host buffer. */
int agarcl_screen_obs(agarcl_env *env, int32_t width, int32_t height, int32_t agent_view, uint8_t *out, int32_t on_device);

/* "ram" observation (BASELINE configs[0]).  The reference has NO counterpart to replace: obs_type "ram" passes the check at
 * gym_agario/AgarioEnv.py:52 and is rejected at :211, agario-ram-v0 is never registered (gym_agario/__init__.py:9-23) and
 * environment/test/ram-env-test.hpp is empty -- so the layout is this library's own.  Writes f32[num_arenas][num_agents][D],
 * D = 4 + 3 k_cells + 2 k_pellets + 3 k_viruses + 3 k_others, returned through *dim (out == NULL only queries D):
 *   px, py, total mass, cell count | k_cells x (dx, dy, mass) own cells in cell order | k_pellets x (dx, dy) nearest pellets, nearest first
 *   (ties: lower index) | k_viruses x (dx, dy, mass) nearest viruses | k_others x (dx, dy, mass) nearest cells of the other players;
 * dx = x - px, dy = y - py; absent rows are zero.  A dead agent (no cells: the terminal step, before any respawn) gets an all-zero record
 * with cell count 0 -- never NaN.  `out` is an HBM pointer if on_device != 0, else a host buffer. */
int agarcl_ram_obs(agarcl_env *env, int32_t k_cells, int32_t k_pellets, int32_t k_viruses, int32_t k_others, float *out, int32_t on_device, int32_t *dim);

/* replaces: GoBiggerEnvironment::get_state() (bindings.cpp:28-47,353) -> GoBiggerObservation::add_frame
 * (environment/envs/GoBiggerEnvironment.hpp:446-541), as padded tensors for EVERY player of every arena (row k of the
 * player axis = the k-th player in the engine's map iteration order; P = agarcl_players_per_arena):
 *   hdr   i32[num_arenas][P][8]          pid, committed, n_virus, n_food, n_spore, n_clone, score (total mass), player slot
 *   food  f32[num_arenas][P][cap_food][4]   x - px, y - py, radius, mass      (pellets, in vector order)
 *   virus f32[num_arenas][P][cap_virus][4]  x - px, y - py, radius, mass
 *   spore f32[num_arenas][P][cap_spore][4]  x - px, y - py, radius, mass      (ejected foods; velocity (0,0), owner = pid)
 *   clone f32[num_arenas][P][cap_clone][7]  x - px, y - py, radius, mass, vx, vy, direction   (the player's own cells)
 * (px, py) = the player's mass-weighted centre.  An entity is listed iff it falls inside the player's egocentric grid of
 * grid_size cells (view = clamp(2 * mass, 100, 300)); counts are the true numbers, rows beyond a capacity are dropped,
 * unused rows are zero.  committed = 0: nothing was inside (e.g. a dead player) -- the reference then keeps that
 * player's previous state.  Pointers are HBM pointers if on_device != 0, else host buffers. */
int agarcl_gobigger_obs(agarcl_env *env, int32_t grid_size, int32_t cap_food, int32_t cap_virus, int32_t cap_spore, int32_t cap_clone,
                        int32_t *hdr, float *food, float *virus, float *spore, float *clone, int32_t on_device);

/* full-state exchange for parity tests and snapshots (layout: oracle/BLOB_FORMAT.md); synchronising */
int agarcl_dump_arena(agarcl_env *env, int32_t arena, uint32_t *buf_host, int32_t cap_words);
int agarcl_load_arena(agarcl_env *env, int32_t arena, const uint32_t *blob_host, int32_t words);

/* ---- snapshots.  The reference's wire format is JSON written by BaseEnvironment::save_env_state
 * (environment/envs/BaseEnvironment.hpp:213-310) and read by Engine::load_env_state (agario/engine/Engine.hpp:247-348)
 * + BaseEnvironment::load_env_state (:312-343).  The JSON text is produced / parsed on the host side above this ABI
 * (agarcl_amd/snapshot.py); these entry points move the state in and out of HBM. ----------------------------------- */
/* Loading a snapshot replaces the arena's player set (fresh pids 0..P-1 in file order, Engine.hpp:266-283): the blob
 * lists the players in the NEW map iteration order; kinds[k] (0 agent, 1 Hungry, 2 HungryShy, 3 Aggressive,
 * 4 AggressiveShy: the bot classes of agario/bots) describes the k-th of them; hm_* is the state of the players map
 * after the inserts (libstdc++ bucket count / next resize, kept across clear()).  Non-bot players become agents
 * 0,1,.. in map order (BaseEnvironment.hpp:324-335). */
int agarcl_adopt_arena(agarcl_env *env, int32_t arena, const uint32_t *blob_host, int32_t words, const int32_t *kinds,
                       int32_t hm_buckets, int32_t hm_next_resize);
/* Engine::seed(s) for one arena (Engine.hpp:242-245): load_env_state ends with seed(json["seed"]) (:347) */
int agarcl_seed_arena(agarcl_env *env, int32_t arena, uint32_t seed);
/* last seed of every arena, u32[num_arenas] (BaseEnvironment::seed_, the "seed" field of a snapshot, :225) */
int agarcl_get_seeds(agarcl_env *env, uint32_t *out_host);
/* raw words of one arena: ar_out i32[AGARCL_ARENA_WORDS] (agar_types.h AR_*), pl_out i32[players][AGARCL_PLAYER_WORDS] (PL_*, slot-major);
 * either may be NULL.  agarcl_player_words() returns AGARCL_PLAYER_WORDS of the library that is loaded (callers without the header size
 * their buffer by it). */
#define AGARCL_ARENA_WORDS 48
#define AGARCL_PLAYER_WORDS 24
int agarcl_get_arena_words(agarcl_env *env, int32_t arena, int32_t *ar_out, int32_t *pl_out);
int agarcl_player_words(void);

/* introspection */
int agarcl_num_arenas(agarcl_env *env);
int agarcl_players_per_arena(agarcl_env *env);
/* HBM bytes the engine reads+writes per arena-tick under the streaming model of DESIGN.md */
int64_t agarcl_state_bytes(agarcl_env *env);
/* HBM bytes the env has allocated (all SoA arrays, result rings, tables, and the pellet-event spill area that dense / crowded arena
 * configurations get: up to 256 KB per arena there) -- what `num_arenas` costs on this device */
int64_t agarcl_device_bytes(agarcl_env *env);


/* ---- stream ordering (the env's launches against work of the caller's own streams) --------------------------------------- */
/* the hipStream_t the env currently launches on (its private non-blocking stream until agarcl_set_stream adopts another) */
void *agarcl_get_stream(agarcl_env *env);
/* the env's stream waits (on the device; the host does not block) for everything enqueued so far on `producer_stream` -- e.g. the policy's
 * kernels that wrote the action tensors of the next agarcl_step_actions.  NULL = the legacy default stream. */
int agarcl_stream_wait(agarcl_env *env, void *producer_stream);
/* `consumer_stream` waits (on the device) for everything the env has enqueued so far -- e.g. before a learner reads rewards / observations */
int agarcl_stream_signal(agarcl_env *env, void *consumer_stream);

/* ---- sub-batch pipelining ------------------------------------------------------------------------------------------------
 * replaces: the reference's vectorised runner, which lets every engine run ahead on its own pool thread and waits once at the end
 * (/root/reference/agario/bots/benchmark.cpp:149-167: pool.schedule per game, one pool.wait()).  An agarcl_env steps ALL its arenas in one
 * launch, so every step lasts as long as its slowest arena while the wave slots of the finished ones stand empty.  A pipe splits
 * `num_arenas` into `sub_batches` contiguous arena ranges, each a complete agarcl_env (agarcl_pipe_env: use it with every call of this
 * header) on a HIP stream of its own; the streams are checked at creation to really execute concurrently (distinct hardware queues:
 * agarcl_pipe_concurrent), so sub-batch B's launch fills the SIMDs during A's tail and A's observation kernel runs under B's step.
 * Arenas never interact and seeds go by the GLOBAL arena index (agarcl_pipe_seed), so arena `first_j + a` of the pipe computes exactly
 * what arena `first_j + a` of one agarcl_env over all arenas computes -- only the waiting differs.
 * The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of them the legacy default stream's):
 * more than 3 concurrent sub-batches need that variable raised in the process environment before the first HIP call. */
typedef struct agarcl_pipe agarcl_pipe;
int agarcl_pipe_create(const agarcl_config *cfg, int32_t num_arenas, int32_t sub_batches, int32_t device, agarcl_pipe **out);
int agarcl_pipe_destroy(agarcl_pipe *pipe);
int agarcl_pipe_sub_batches(agarcl_pipe *pipe);
/* the j-th sub-batch: an agarcl_env over the arenas [first, first + count) of the pipe (agarcl_pipe_range); owned by the pipe */
agarcl_env *agarcl_pipe_env(agarcl_pipe *pipe, int32_t j);
int agarcl_pipe_range(agarcl_pipe *pipe, int32_t j, int32_t *first_arena, int32_t *count);
/* agarcl_seed over the whole pipe: seeds_host[num_arenas] by global arena index; NULL -> arena i gets base_seed + i */
int agarcl_pipe_seed(agarcl_pipe *pipe, const uint32_t *seeds_host, uint32_t base_seed);
/* how many of the sub-batch streams were verified at creation to execute concurrently with each other (== sub_batches unless the
 * runtime ran out of hardware queues; a smaller number only costs overlap, never results) */
int agarcl_pipe_concurrent(agarcl_pipe *pipe);
/* waits for every sub-batch's stream (the reference's single pool.wait()) */
int agarcl_pipe_sync(agarcl_pipe *pipe);
/* agarcl_stream_wait for all sub-batches at once: everything enqueued so far on `producer_stream` happens before whatever the sub-batches enqueue from
 * now on (device-side).  Through a flag word in HBM -- a one-lane kernel on the producer's stream publishes an epoch, a one-lane kernel at the head of
 * every sub-batch's stream spins until it sees it -- when the producer's stream was verified to run beside every sub-batch stream; through one HIP event
 * otherwise (or with AGARCL_PIPE_EVENTS=1): ~40 us of device-side latency per hand-over on this stack, which is what the flag words avoid */
int agarcl_pipe_fork(agarcl_pipe *pipe, void *producer_stream);
/* agarcl_stream_signal for all sub-batches at once: `consumer_stream` waits (device-side) for everything every sub-batch has enqueued so far */
int agarcl_pipe_join(agarcl_pipe *pipe, void *consumer_stream);
/* diagnostics: 1 if a spin of the flag-word fork / join ever ran into its 200 ms bound (synchronises the pipe) */
int agarcl_pipe_spin_timeouts(agarcl_pipe *pipe);


/* ---- diagnostics (not part of the drop-in surface; used by tests/ and scripts/) --------------------------------------- */
/* 1 = the single-launch fused step is in use, 0 = the two-kernel step (the choice never changes results) */
int agarcl_debug_fused(agarcl_env *env);
/* i32[num_arenas][2]: the front kernel's hand-over words of the last step (ticks done or -1, mass before) */
int agarcl_debug_qinfo(agarcl_env *env, int32_t *out_host);
/* -DAGAR_PROFILE builds only: per-phase cycle sums, u64[16] summed over arenas / u64[num_arenas][16] raw */
int agarcl_debug_prof(agarcl_env *env, unsigned long long *out16, int reset);
int agarcl_debug_prof_raw(agarcl_env *env, unsigned long long *out_host);
/* bytes the step kernels requested from memory since the last call with reset != 0, counted by the kernels themselves:
 * out[0] = arena-steps finished by the lean front part, out[1] = arena-steps that needed the general engine,
 * out[2] = pellet passes (each reads the arena's whole pellet array), out[3] = general ticks executed */
int agarcl_debug_work(agarcl_env *env, int64_t *out4, int reset);
/* the 16 raw running statistics words: [0] arena-steps the front part left unfinished, [1] OR of raised flags; diagnostic builds
 * only: [4..11] why the front part stopped (-DAGAR_PROFILE_REASONS, agar_core.inl AG_WHY).  (-DAGAR_PROFILE -DAGAR_PROFILE_LEVELS counts
 * the relaxation's visited levels / levels with a touching pair / touch passes per arena in slots 13 / 14 / 15 of agarcl_debug_prof.) */
int agarcl_debug_qstat(agarcl_env *env, int32_t *out16);
/* self-test of the self-collision relaxation's short square root against the correctly rounded sqrtf on all 2^32 float bit patterns
 * (runs a kernel, waits for it): out2[0] = patterns whose results differ in any bit, out2[1] = the lowest of them */
int agarcl_debug_sqrt_check(agarcl_env *env, unsigned long long *out2);

#ifdef __cplusplus
}
#endif
#endif /* AGARCL_BATCH_H */
