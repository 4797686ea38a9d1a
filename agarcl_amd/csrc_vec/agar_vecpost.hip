// Episode bookkeeping of the batched user surface (agarcl_amd/vector_env.py: AgarioVectorEnv) as ONE launch: what AgarioEnv.step does per
// env on the host (/root/reference/gym_agario/AgarioEnv.py:105-123 -- done = engine done or, for an episodic env, number_steps played,
// compared before the step is counted; the step counter) plus the episode statistics, for all arenas on the device.  In torch this is a
// dozen small element-wise launches per step, ~80 us of host time -- nine times the engine's own step at 4096 arenas.
// A separate library (libagarcl_vec.so) on purpose: it needs nothing of the engine but the pointers of its result arrays and a stream.
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared (agarcl_amd/build.py build_vecpost).  C ABI: include/agarcl_vec.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/agarcl_vec.h"

__global__ void __launch_bounds__(256) k_vec_post(const uint8_t *__restrict__ dones, const double *__restrict__ rewards, int A, int n, int number_steps, int episodic,
                                                  int32_t *steps, float *reward_out, uint8_t *done_out, uint8_t *ended_out, float *ep_return, float *final_return, int32_t *final_length) {
  const int a = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (a >= A) return;
  const int played = steps[a];
  const bool timeout = episodic != 0 && played >= number_steps;   // (before this step is counted: AgarioEnv.py:111-112)
  bool any = false;
  for (int i = 0; i < n; i++) { const bool d = dones[(size_t)a * n + i] != 0 || timeout; done_out[(size_t)a * n + i] = d ? 1 : 0; any = any || d; }
  for (int i = 0; i < n; i++) {
    const size_t k = (size_t)a * n + i;
    const float r = (float)rewards[k];
    reward_out[k] = r;
    const float e = ep_return[k] + r;
    if (any) { final_return[k] = e; ep_return[k] = 0.0f; } else ep_return[k] = e;
  }
  if (any) final_length[a] = played + 1;
  steps[a] = any ? 0 : played + 1;
  ended_out[a] = any ? 1 : 0;
}

extern "C" int agarcl_vec_post(void *stream, const uint8_t *dones, const double *rewards, int32_t num_arenas, int32_t num_agents, int32_t number_steps, int32_t episodic,
                               int32_t *steps, float *reward_out, uint8_t *done_out, uint8_t *ended_out, float *ep_return, float *final_return, int32_t *final_length) {
  if (!dones || !rewards || !steps || !reward_out || !done_out || !ended_out || !ep_return || !final_return || !final_length || num_arenas < 1 || num_agents < 1) return 1;
  hipLaunchKernelGGL(k_vec_post, dim3((unsigned)((num_arenas + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dones, rewards, (int)num_arenas, (int)num_agents, (int)number_steps,
                     (int)episodic, steps, reward_out, done_out, ended_out, ep_return, final_return, final_length);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
