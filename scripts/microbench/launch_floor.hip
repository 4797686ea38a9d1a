// Launch-floor microbenchmark for sizing the front kernel (diagnostic; build + run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_floor scripts/microbench/launch_floor.hip && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct Desc { int A; float *a; float *b; int *c; };
__global__ void k_empty() {}
__global__ void k_chain(const Desc *d) {  // kernarg -> descriptor -> data -> store
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d->A) return;
  d->b[i] = d->a[i] + 1.0f;
}
__global__ void k_direct(int A, const float *a, float *b) {  // kernarg -> data -> store
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A) return;
  b[i] = a[i] + 1.0f;
}
__global__ void k_spin(const Desc *d, int iters, unsigned long long *out) {  // dependent fp32 chain: cycles per dependent VALU op
  float x = d->a[threadIdx.x];
  unsigned long long t0 = __builtin_readcyclecounter(); unsigned long long w0 = wall_clock64();
  for (int i = 0; i < iters; i++) x = x * 1.0001f + 0.5f;
  unsigned long long t1 = __builtin_readcyclecounter(); unsigned long long w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
  d->b[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
// early-exit kernel that nevertheless declares k_step's resources (dynamic LDS + 128 VGPRs): what does dispatch cost?
__global__ void __attribute__((amdgpu_num_vgpr(128))) k_heavy_exit(const Desc *d) {
  extern __shared__ unsigned char lds[];
  if (d->c[0] == 0) return;
  lds[threadIdx.x] = 1; d->b[blockIdx.x * blockDim.x + threadIdx.x] = lds[(threadIdx.x + 1) % blockDim.x];
}
template <class F> static float timeit(hipStream_t s, int n, F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; i++) f();
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int i = 0; i < n; i++) f();
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.0f / n;
}
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int A = 4096 * 64;
  float *a, *b; int *c; Desc h, *d; unsigned long long *out;
  CK(hipMalloc(&a, A * 4)); CK(hipMalloc(&b, A * 4)); CK(hipMalloc(&c, A * 4)); CK(hipMalloc(&d, sizeof(Desc))); CK(hipMalloc(&out, 16));
  CK(hipMemset(a, 0, A * 4)); h.A = A; h.a = a; h.b = b; h.c = c; CK(hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice)); CK(hipDeviceSynchronize());
  for (int grid : {256, 1024, 4096}) {
    int blk = A / grid > 1024 ? 1024 : A / grid; int g2 = A / blk;
    printf("grid %5d x %4d: empty %.2f us, chain %.2f us, direct %.2f us per launch (back-to-back, same stream)\n", g2, blk,
           timeit(s, 1000, [&] { hipLaunchKernelGGL(k_empty, dim3(g2), dim3(blk), 0, s); }),
           timeit(s, 1000, [&] { hipLaunchKernelGGL(k_chain, dim3(g2), dim3(blk), 0, s, d); }),
           timeit(s, 1000, [&] { hipLaunchKernelGGL(k_direct, dim3(g2), dim3(blk), 0, s, A, a, b); }));
  }
  printf("grid  4096 x   64: empty %.2f us, chain %.2f us, direct %.2f us\n",
         timeit(s, 1000, [&] { hipLaunchKernelGGL(k_empty, dim3(4096), dim3(64), 0, s); }),
         timeit(s, 1000, [&] { hipLaunchKernelGGL(k_chain, dim3(4096), dim3(64), 0, s, d); }),
         timeit(s, 1000, [&] { hipLaunchKernelGGL(k_direct, dim3(4096), dim3(64), 0, s, 4096 * 64, a, b); }));
  CK(hipMemset(c, 0, 64)); CK(hipDeviceSynchronize());
  printf("early exit with k_step-like resources: 4096 x 64 thr, 8848 B LDS: %.2f us;  1024 x 256 thr, 35392 B LDS: %.2f us;  4096 x 64 thr, no LDS: %.2f us;  256 x 64, 8848 B LDS: %.2f us\n",
         timeit(s, 1000, [&] { hipLaunchKernelGGL(k_heavy_exit, dim3(4096), dim3(64), 8848, s, d); }),
         timeit(s, 1000, [&] { hipLaunchKernelGGL(k_heavy_exit, dim3(1024), dim3(256), 35392, s, d); }),
         timeit(s, 1000, [&] { hipLaunchKernelGGL(k_heavy_exit, dim3(4096), dim3(64), 0, s, d); }),
         timeit(s, 1000, [&] { hipLaunchKernelGGL(k_heavy_exit, dim3(256), dim3(64), 8848, s, d); }));
  printf("two kernels per step (empty + empty): %.2f us\n", timeit(s, 1000, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); hipLaunchKernelGGL(k_empty, dim3(4096), dim3(64), 0, s); }));
  for (int iters : {1000, 100000}) {
    hipLaunchKernelGGL(k_spin, dim3(1024), dim3(64), 0, s, d, iters, out); CK(hipStreamSynchronize(s));
    unsigned long long o[2]; CK(hipMemcpy(o, out, 16, hipMemcpyDeviceToHost));
    printf("spin %d dependent mul+add pairs: %llu shader cycles, %llu ticks of the 100 MHz clock -> %.2f cycles per dependent op, shader clock ~%.0f MHz\n", iters, o[0], o[1], o[0] / (2.0 * iters), o[1] ? o[0] * 100.0 / o[1] : 0.0);
  }
  return 0;
}
