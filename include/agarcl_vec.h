/* agarcl_vec.h -- one host call per step of the batched RL surface (libagarcl_hip.so; Python side: agarcl_amd/vector_env.py AgarioVectorEnv).
 *
 * replaces, for all arenas of an agarcl_env at once, what /root/reference/gym_agario/AgarioEnv.py:85-132 does per env on the host:
 *   step   = take_actions + env step (:99-103) + observation (:106) + done = the engine's done flag or -- for an episodic env -- `number_steps`
 *            steps played, compared before this step is counted (:111-112) + the step counter (:123), and the reset a user performs when an
 *            episode has ended (:125-132), here on the device in the same call ("same step" auto-reset);
 *   plus the episode statistics a RecordEpisodeStatistics wrapper would keep.
 * agarcl_vec_step enqueues, on the env's stream and without waiting for anything:  the step kernel(s)  ->  ONE bookkeeping + masked-reset
 * launch (a workgroup per arena: its first lane settles the arena's episode, the wavefront resets the arena if the episode ended and leaves
 * at once otherwise)  ->  the observation kernel.  Plain device pointers, no torch types.
 *
 * Several agents per arena: an arena's episode ends when ANY agent is done or the cut-off strikes, and the whole arena is reset.  Row i of an
 * ended arena gets done = 1 if agent i itself was done (or the cut-off struck), otherwise truncated = 1: its episode was cut short by the
 * reset, so a learner must not bootstrap across that boundary with terminated = False (the reference never resets on its own; this is the
 * vector surface's rule).  Every row of an ended arena gets its final_return; ep_return restarts at 0 for all of them. */
#ifndef AGARCL_VEC_H
#define AGARCL_VEC_H
#include <stdint.h>
#include "agarcl_batch.h"
#ifdef __cplusplus
extern "C" {
#endif

#define AGARCL_OBS_NONE 0
#define AGARCL_OBS_GRID 1     /* agarcl_grid_obs into `obs` (persistent tensor: on_device = 2); obs_arg = grid_size, cells, others, viruses, pellets */
#define AGARCL_OBS_SCREEN 2   /* agarcl_screen_obs; obs_arg = width, height, agent_view */
#define AGARCL_OBS_RAM 3      /* agarcl_ram_obs; obs_arg = k_cells, k_pellets, k_viruses, k_others */

typedef struct agarcl_vec_spec {
  int32_t number_steps;   /* the episodic cut-off (AgarioEnv.py:111-112) */
  int32_t episodic;       /* != 0: env_type 0 (the cut-off applies) */
  int32_t reset_ids;      /* forwarded to the reset of ended arenas (agarcl_reset) */
  int32_t obs_kind;       /* AGARCL_OBS_* */
  int32_t obs_arg[6];
  int32_t ticks;          /* engine ticks per step; <= 0 -> the env's ticks_per_step */
  int32_t reset_flagged;  /* != 0: an arena that carries a capacity flag (agarcl_get_flags) is treated like an ended episode: reset in the same step,
                             its rows truncated = 1 (unless done), final_return / final_length written -- a diverged arena never feeds a learner */
  int32_t reserved[4];
} agarcl_vec_spec;

/* HBM buffers of the caller, A = agarcl_num_arenas(env), n = num_agents: all written by agarcl_vec_step / agarcl_vec_reset */
typedef struct agarcl_vec_buffers {
  int32_t *steps;         /* [A]    in/out: steps played in the running episode */
  float *reward;          /* [A][n] this step's reward (BaseEnvironment::step, f64 -> f32) */
  uint8_t *done;          /* [A][n] 0 / 1: terminated (engine flag or cut-off) */
  uint8_t *truncated;     /* [A][n] 0 / 1: the arena was reset under an agent that was not done (several agents only) */
  uint8_t *ended;         /* [A]    0 / 1: the arena's episode ended in this step and the arena was reset */
  float *ep_return;       /* [A][n] in/out: running return of the current episode */
  float *final_return;    /* [A][n] rewritten where ended: the finished episode's return */
  int32_t *final_length;  /* [A]    rewritten where ended: its length */
  void *obs;              /* the observation tensor of obs_kind (NULL with AGARCL_OBS_NONE) */
} agarcl_vec_buffers;

/* all arenas start a new episode (BaseEnvironment::reset for every arena), counters and returns restart, `obs` = the first observation */
int agarcl_vec_reset(agarcl_env *env, const agarcl_vec_spec *spec, const agarcl_vec_buffers *buf);
/* one vector step: dxdy_dev f32 [A][n][2], act_dev i32 [A][n] (HBM; read when the step kernel executes).  flags_seen (nullable): the
 * capacity-flag watch of agarcl_poll_flags after this step's sample (never blocks).  The auto-reset does NOT restart the flag watch: an
 * arena that diverged is reported even when its episode has ended since. */
int agarcl_vec_step(agarcl_env *env, const agarcl_vec_spec *spec, const agarcl_vec_buffers *buf, const float *dxdy_dev, const int32_t *act_dev,
                    uint32_t *flags_seen);
#ifdef __cplusplus
}
#endif
#endif /* AGARCL_VEC_H */
