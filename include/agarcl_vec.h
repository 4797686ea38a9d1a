/* Episode bookkeeping for the batched user surface, one launch per step (agarcl_amd/libagarcl_vec.so, source agarcl_amd/csrc_vec/).
 * Replaces, for N arenas at once, what /root/reference/gym_agario/AgarioEnv.py:105-123 does per env on the host after the engine's step:
 * done = the engine's done flag or -- for an episodic env -- `number_steps` steps played (compared before this step is counted), the step
 * counter, and the episode statistics a RecordEpisodeStatistics wrapper would keep.  Plain device pointers, no torch types.
 *   dones u8 [A][n], rewards f64 [A][n]: the engine's result arrays of the step just taken (agarcl_batch.h: agarcl_device_results);
 *   steps i32 [A] (in/out): steps played in the running episode; reward_out f32 [A][n]; done_out u8 [A][n] (0 / 1);
 *   ended_out u8 [A]: 1 where any agent's episode ended -- the mask for agarcl_reset_device;
 *   ep_return f32 [A][n] (in/out): running return; final_return f32 [A][n], final_length i32 [A]: rewritten where ended_out is 1.
 * `stream`: the hipStream_t the engine runs on.  Returns 0, 1 (bad argument) or 2 (launch failed). */
#ifndef AGARCL_VEC_H
#define AGARCL_VEC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int agarcl_vec_post(void *stream, const uint8_t *dones, const double *rewards, int32_t num_arenas, int32_t num_agents, int32_t number_steps, int32_t episodic,
                    int32_t *steps, float *reward_out, uint8_t *done_out, uint8_t *ended_out, float *ep_return, float *final_return, int32_t *final_length);
#ifdef __cplusplus
}
#endif
#endif /* AGARCL_VEC_H */
