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
  PL_WORDS = 20
};
// per-arena int32 words ([A][AR_WORDS])
enum {
  AR_TICKS = 0, AR_CLOCK, AR_IDC, AR_NEXT_PID, AR_NPEL, AR_NVIR, AR_NFOOD, AR_FLAGS, AR_MTIDX,
  AR_NEVP, AR_NEVV, AR_DONE, AR_RESPAWNED, AR_ORDER0 /* AG_MAX_PLAYERS slots: player slots in the engine's iteration order */,
  AR_HM_BUCKETS = AR_ORDER0 + 16, AR_HM_RESIZE,  // rehash-policy state of the players map (survives reset, GameState.hpp:61-67)
  AR_SAFE,  // f32 bits S: no pellet within S + radius of (PL_SAFE_X, PL_SAFE_Y) of player 0; 0 = unknown
  AR_WORDS = 32
};
#define AG_MAX_PLAYERS 16
#define AG_PACKED_SLOTS 16  // ring of packed (reward, done) result buffers: step k of an env writes slot k % 16
#define AG_CC 32        // cell capacity per player (reference: unbounded vector, nominal limit 14)
#define AG_EV_CAP 256   // pellet eat events per arena-tick
#define AG_EVV_CAP 16   // virus eat events per arena-tick (<= players)
#define AG_CAND_CAP 1024 // words of the ordered-replay candidate list: 256 records of (key, index, x, y)
#define AG_VT_CAP 256   // virus_eaten_ticks kept per player (the reference vector is unbounded; > 256 virus meals inside 3600 ticks raises a flag)
#define AG_LUT_SIZE (1 << 19)
#define AG_ANTI_LUT 256

// cell fields, player-major in LDS and HBM: [player][field][AG_CC]
enum { CF_X = 0, CF_Y, CF_VX, CF_VY, CF_SX, CF_SY, CF_M, CF_ID, CF_DL, CF_FIELDS,
       // derived, persisted so a launch needs no dependent table lookups: radius / max-speed cache of the cell,
       // valid iff CF_CMC == CF_M (agar_core.inl: Cells::cmc/crad/cms)
       CF_CMC = CF_FIELDS, CF_CRAD, CF_CMS, CF_ALL };
// word index of field f of cell slot i inside a player's HBM cell block: CELL-major, so that one cell's 12 words are one
// contiguous 48-byte run (a single-cell arena touches one cache line instead of twelve; a lane that stages slot i into LDS
// reads three 16-byte words).  The LDS copy stays field-major.
#define AG_CELL_W(f, i) ((i) * CF_ALL + (f))

struct AgDims {
  int A;         // arenas
  int P;         // player slots per arena (agents + bots)
  int n_agents;  // RL-controlled players per arena
  int PC;        // pellet capacity per arena (multiple of 64)
  int VC;        // virus capacity
  int FC;        // food capacity
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
  // cells [A][P][AG_CC][CF_ALL] as 32-bit words (AG_CELL_W)
  uint32_t *cells;
  int32_t *pl;      // [A][P][PL_WORDS]
  int32_t *vticks;  // [A][P][AG_VT_CAP]
  int32_t *ar;      // [A][AR_WORDS]
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
  int32_t *ev_p;          // [A][AG_EV_CAP]
  int32_t *ev_v;          // [A][AG_EVV_CAP]
  // read-only tables (mass -> fp32), host generated (Engine.hpp:1296-1302, core/utils.hpp:8-11)
  const float *lut_r, *lut_ms, *lut_ss, *lut_anti;
  int32_t *qinfo;         // [A][2] hand-over from k_quiet to k_step: ticks done (-1: nothing), agent mass before the step
  int32_t *qstat;         // [4] running totals the host samples now and then: [0] arena-steps the front part left unfinished,
                          //     [1] OR of every capacity flag raised (the flag watch of agarcl_poll_flags)
  int32_t *qcount;        // [2] number of arenas k_quiet left unfinished, ping-pong by launch parity (k_step exits at once on 0)
  int32_t *qlist;         // [2][A] the arenas k_quiet left unfinished (same parity), in arrival order: k_step's work list
  unsigned long long *prof;  // [16] phase cycle sums (diagnostic builds only; may be null)
};
// The three pointers the lean front part needs before it can request an arena's state.  They travel as kernel arguments
// (preloaded into SGPRs at wave start, see build.py) so that the state loads do not wait for a round trip to the descriptor.
struct AgHot { int32_t *ar; int32_t *pl; uint32_t *cells; };
