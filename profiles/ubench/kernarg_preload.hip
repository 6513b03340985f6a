// kernarg_preload.hip -- does preloading the kernel arguments into SGPRs (-mllvm -amdgpu-kernarg-preload-count=N)
// shorten a dependent chain of small kernels on gfx950?  Each kernel of the chain is the optimiser tail in miniature:
// arguments -> one round of global reads -> a little arithmetic -> one store, 48 workgroups of 1024 threads; kernel i
// reads what kernel i-1 wrote.  Built twice (with / without the flag), the per-kernel time of a 2000-kernel chain is
// the figure.
//   hipcc --offload-arch=gfx950 -O3 -o kernarg_preload_off kernarg_preload.hip
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o kernarg_preload_on kernarg_preload.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void k_link(const double* __restrict__ p, double* __restrict__ q,
                                               const double* __restrict__ a, const double* __restrict__ b, int n,
                                               double s) {
  const int i = blockIdx.x * 1024 + threadIdx.x;
  if (i < n) q[i] = p[i] * s + a[i] * b[i];
}

__global__ void k_empty(int) {}

int main() {
  const int n = 48 * 1024;
  double *p, *q, *a, *b;
  CHECK(hipMalloc(&p, n * 8)); CHECK(hipMalloc(&q, n * 8)); CHECK(hipMalloc(&a, n * 8)); CHECK(hipMalloc(&b, n * 8));
  CHECK(hipMemset(p, 0, n * 8)); CHECK(hipMemset(q, 0, n * 8)); CHECK(hipMemset(a, 0, n * 8)); CHECK(hipMemset(b, 0, n * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 5; ++rep) {
    const int K = 2000;
    CHECK(hipEventRecord(e0));
    for (int k = 0; k < K; ++k) {
      hipLaunchKernelGGL(k_link, dim3(48), dim3(1024), 0, 0, p, q, a, b, n, 0.5);
      double* t = p; p = q; q = t;
    }
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("chain of %d dependent small kernels: %.3f us per kernel\n", K, ms * 1e3 / K);
    CHECK(hipEventRecord(e0));
    for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_empty, dim3(48), dim3(1024), 0, 0, k);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("chain of %d empty kernels:           %.3f us per kernel\n", K, ms * 1e3 / K);
  }
  return 0;
}
