// Shared host/device definitions of the batched engine's HBM-resident SoA state.
// Layout rationale: DESIGN.md "Data layout in HBM".
#pragma once
#include <stdint.h>

// game constants: /root/reference/agario/core/settings.hpp:5-50, core/Entities.hpp:9-18
#define AG_CELL_MIN_SIZE 25u
#define AG_CELL_SPLIT_MINIMUM 50u
#define AG_SPLIT_DECEL 80.0f
#define AG_FOOD_SPEED 100.0f
#define AG_FOOD_DECEL 80.0f
#define AG_CELL_POP_SIZE 25u
#define AG_PLAYER_CELL_LIMIT 14
#define AG_FOOD_HITS 7
#define AG_MAX_MASS 22500u
#define AG_NEW_MASS_NO_SPLIT 22000u
#define AG_ANTI_TEAM_TICKS 3600
#define AG_PELLET_MASS 1u
#define AG_FOOD_MASS 10u
#define AG_VIRUS_MASS 100u
#define AG_PELLET_GRID 510
#define AG_VIRUS_GRID 25

// per-player int32 words ([A][P][PL_WORDS]); floats are bit-cast
enum {
  PL_NCELLS = 0, PL_ACTION, PL_TX, PL_TY, PL_SPLIT_CD, PL_FEED_CD, PL_ELAPSED, PL_LAST_DECAY, PL_ANTI_TEAM,
  PL_FOOD_EATEN, PL_HIGHEST_MASS, PL_CELLS_EATEN, PL_VIRUSES_EATEN, PL_MIN_MASS, PL_NVTICKS, PL_PID, PL_KIND,
  PL_SAFE_X, PL_SAFE_Y,  // single-player arenas: where the pellet-free disc of radius AR_SAFE was measured (agar_core.inl quiet_ticks)
  PL_PASSES,             // diagnostics: how often this arena's pellet array has been read from memory (slot 0 only; never part of a blob)
  // single-player arenas: the ONE pellet the disc does not exclude -- the nearest one at the time of the pass (position as f32 bits, index;
  // index < 0: none).  With it the disc reaches to the SECOND nearest pellet, and eating the tracked one needs no pass (quiet_ticks)
  PL_CAND_X, PL_CAND_Y, PL_CAND_IDX, PL_SPARE,
  PL_WORDS = 24
};
// per-arena int32 words ([A][AR_WORDS])
enum {
  AR_TICKS = 0, AR_CLOCK, AR_IDC, AR_NEXT_PID, AR_NPEL, AR_NVIR, AR_NFOOD, AR_FLAGS, AR_MTIDX,
  AR_NEVP, AR_NEVV, AR_DONE, AR_RESPAWNED, AR_ORDER0 /* AG_MAX_PLAYERS slots: player slots in the engine's iteration order */,
  AR_HM_BUCKETS = AR_ORDER0 + 32, AR_HM_RESIZE,  // rehash-policy state of the players map (survives reset, GameState.hpp:61-67)
  AR_SAFE,  // f32 bits S: no pellet within S + radius of (PL_SAFE_X, PL_SAFE_Y) of player 0; 0 = unknown
  AR_WORDS = 48
};
#define AG_MAX_PLAYERS 32   // (bench/main.cpp's Tick/30: 30 ExampleBots in one arena)
#define AG_PACKED_SLOTS 64  // ring of packed (reward, done) result buffers: step k of an env writes slot k % 64
#define AG_CC 32        // cell capacity per player (reference: unbounded vector, nominal limit 14)
#define AG_EV_MIN 256   // pellet eat events per arena-tick (AgDims::EC) and candidate records of one cell's replay (AgDims::KC): at least this; dense small arenas get more (agarcl_create)
#define AG_EVV_CAP 32   // virus eat events per arena-tick (at most one per player and tick in practice; AG_MAX_PLAYERS of them fit)
#define AG_VT_CAP 256   // virus_eaten_ticks kept per player (the reference vector is unbounded; > 256 virus meals inside 3600 ticks raises a flag)
#define AG_LUT_SIZE (1 << 19)
#define AG_ANTI_LUT 256

// cell fields, player-major in LDS and HBM: [player][field][AG_CC]
enum { CF_X = 0, CF_Y, CF_VX, CF_VY, CF_SX, CF_SY, CF_M, CF_ID, CF_DL, CF_FIELDS,
       // derived, persisted so a launch needs no dependent table lookups: radius / max-speed cache of the cell,
       // valid iff CF_CMC == CF_M (agar_core.inl: Cells::cmc/crad/cms)
       CF_CMC = CF_FIELDS, CF_CRAD, CF_CMS, CF_ALL };
// The three per-arena word arrays (ar, pl, cells) are TILE-TRANSPOSED in HBM: arenas are grouped in tiles of TS = 2^ts_lg
// (AgDims::ts_lg, chosen per env: 0 or 6) and inside a tile the arena index is the fastest one -- word w of arena a of an
// array with R words per arena lives at
//     ((a / TS) * R + w) * TS + a % TS.
// TS = 1 is the plain arena-major layout: an arena's words are contiguous, which is what the general engine (one wavefront
// per arena) and the 16-lanes-per-arena front part of small batches want -- few cache lines per arena.
// TS = 64 serves the big batches, where the lean front kernel gives every arena one lane (or two, or four): lane l of a
// wavefront reads word w of arena 64 t + l, so a wave-instruction moves one contiguous 256-byte run and only the words that
// are used ever leave HBM (arena-major: every lane touches its own cache lines, 64 lines per load instruction, whole lines
// for a few words).  Measured on MI355X, C2, us per step, TS 1 / 64: 4096 arenas 9.2 / 9.9, 65536: 26.4 / 22.0, 262144:
// 65.6 / 45.2; the full-ruleset and multi-player workloads (general engine only) 449 / 457 and 219 / 240.
// The macros below expect `ag_ts_lg` (the env's AgDims::ts_lg) in scope.
#define AG_TILE_BASE(a, R) (((((size_t)(a) >> ag_ts_lg) * (size_t)(R)) << ag_ts_lg) + ((size_t)(a) & (((size_t)1 << ag_ts_lg) - 1)))  // offset of word 0 of arena a
#define AG_TW(w) ((size_t)(w) << ag_ts_lg)                                                        // offset of word w from there
#define AG_TILE_ARENAS(A) ((((((size_t)(A) ? (size_t)(A) : 1) - 1) >> ag_ts_lg) + 1) << ag_ts_lg)  // arenas allocated
// one arena's block of each array (gs: AgState pointer); word w of a block is [AG_TW(w)]
#define AG_AR_PTR(gs, a) ((gs)->ar + AG_TILE_BASE(a, AR_WORDS))
#define AG_PL_PTR(gs, a, p) ((gs)->pl + AG_TILE_BASE(a, (gs)->d.P * PL_WORDS) + AG_TW((p) * PL_WORDS))
#define AG_CELLS_PTR(gs, a, p) ((gs)->cells + AG_TILE_BASE(a, (gs)->d.P * (CF_ALL * AG_CC)) + AG_TW((p) * (CF_ALL * AG_CC)))
// word index of field f of cell slot i inside a player's cell block (cell-major), in tile-transposed HBM and in a
// contiguous host copy of one arena's block
#define AG_CELL_HW(f, i) ((i) * CF_ALL + (f))
#define AG_CELL_W(f, i) AG_TW(AG_CELL_HW(f, i))

struct AgDims {
  int A;         // arenas
  int P;         // player slots per arena (agents + bots)
  int n_agents;  // RL-controlled players per arena
  int PC;        // pellet capacity per arena (multiple of 64)
  int VC;        // virus capacity
  int FC;        // food capacity
  int EC;        // pellet eat events per arena-tick kept in LDS (behind the foods): AG_EV_MIN
  int KC;        // candidate records of ONE cell's ordered replay (LDS, behind the events): >= AG_EV_MIN, <= pellet capacity (a cell cannot reach more)
  int EX;        // further eat events of the tick spill to HBM (ev_p, behind the first EC of each arena): 0 except in dense arenas (agarcl_create)
  int ts_lg;     // log2 of the tile size of the transposed word arrays (0: arena-major, 6: tiles of 64 arenas)
};

struct AgParams {
  float W;                 // arena width == height
  int target_pellets, target_viruses;
  int mass_decay, squared, agent_mass, regen, mode;  // Engine.hpp:362-416
  float dt, dt10;          // (float)dt and (float)(dt*10) (Engine.hpp:613,674)
  int recomb_ticks;        // 10 s in ticks
  int reward_type, c_death;
  float pel_r;             // radius of a pellet (lut_r[1]): random_location's margin when pellets regenerate
  int screen_respawn;      // ScreenEnvironment's hook: respawn dead agents right after the ticks, in every mode (ScreenEnvironment.hpp:233-243)
  int example_bots;        // ExampleBots added after the agents and the mode's bots at every reset (bench/main.cpp:21-24: engine.add_player<ExampleBot>())
  int pgw, pgh, vgw, vgh;  // pellet / virus grid dims (Engine.hpp:964-965,1210-1211)
};

// HBM-resident descriptor of one env; kernels take a pointer to it (scalar loads on demand keep the
// SGPR budget for the hot path)
struct AgState {
  AgDims d; AgParams g;
  // pellets [A][PC] interleaved (x,y) pairs + ids
  float *pel_xy; int32_t *pel_id;
  // viruses [A][VC]
  float *vir_x, *vir_y, *vir_vx, *vir_vy; int32_t *vir_mass, *vir_hits, *vir_id;
  // foods [A][FC]
  float *food_x, *food_y, *food_vx, *food_vy; int32_t *food_id;
  // cells [A][P][AG_CC][CF_ALL] as 32-bit words, tile-transposed (AG_TILE_BASE / AG_CELL_W)
  uint32_t *cells;
  int32_t *pl;      // [A][P][PL_WORDS], tile-transposed
  int32_t *vticks;  // [A][P][AG_VT_CAP]
  int32_t *ar;      // [A][AR_WORDS], tile-transposed
  uint64_t *mt;     // [A][312]
  int32_t *rnd;     // [A][35] glibc rand() ring + position (Player colours, bots' fallbacks)
  int32_t *scratch; // [A][AGM_WORDS] multi-player working memory (null when P == 1)
  // inputs
  const float *act_dxdy;  // [A][n_agents][2]
  const int32_t *act;     // [A][n_agents]
  // outputs
  double *rewards;        // [A][n_agents]
  uint8_t *dones;         // [A][n_agents]
  int32_t *masses;        // [A][n_agents]
  float *packed;          // [AG_PACKED_SLOTS][A][n_agents][2] (reward, done) as f32, ring indexed by step number: what gets gathered across GPUs
  int32_t *counts;        // [A][4]
  int32_t *ev_p;          // [A][EC + EX]: the tick's pellet eat events -- the first EC exported from LDS by arena_store, the rest written directly (spill)
  int32_t *ev_v;          // [A][AG_EVV_CAP]
  // read-only tables (mass -> fp32), host generated (Engine.hpp:1296-1302, core/utils.hpp:8-11)
  const float *lut_r, *lut_ms, *lut_ss, *lut_anti;
  int32_t *qinfo;         // [A][2] hand-over from k_quiet to k_step: ticks done (-1: nothing), agent mass before the step
  int32_t *qstat;         // [4] running totals the host samples now and then: [0] arena-steps the front part left unfinished,
                          //     [1] OR of every capacity flag raised (the flag watch of agarcl_poll_flags)
  int32_t *qcount;        // [2] number of arenas k_quiet left unfinished, ping-pong by launch parity (k_step exits at once on 0)
  int32_t *qlist;         // [2][A] the arenas k_quiet left unfinished (same parity), in arrival order: k_step's work list
  int32_t *sched;         // [2] k_step's work counter (items beyond the grid are drawn from it), ping-pong by launch parity
  uint32_t *cost;         // [A] shader-clock cycles the last k_step visit of every arena took (what k_order sorts by)
  int32_t *order;         // [A] arenas by descending cost (k_order): the order in which k_step hands them out when the batch exceeds its grid
  unsigned long long *prof;  // [16] phase cycle sums (diagnostic builds only; may be null)
};
// The three pointers the lean front part needs before it can request an arena's state.  They travel as kernel arguments
// (preloaded into SGPRs at wave start, see build.py) so that the state loads do not wait for a round trip to the descriptor.
struct AgHot { int32_t *ar; int32_t *pl; uint32_t *cells; };
