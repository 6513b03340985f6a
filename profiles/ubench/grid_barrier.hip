// grid_barrier.hip -- cost and coherence of a device-wide barrier between co-resident workgroups on gfx950
// (8 XCDs, one L2 each).  Each round every workgroup publishes a value with plain stores, crosses the
// barrier, and reads another workgroup's value with plain loads; mismatches are counted.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// monotone counter: round r completes when counter >= (r + 1) * n_wg.  Bounded spin: a lost workgroup turns
// into a reported failure, never a hang.
__device__ inline bool grid_barrier(unsigned* counter, unsigned target, int* fail) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { *fail = 1; ok = false; break; }
    }
  }
  __syncthreads();
  return ok;
}

template <int PAYLOAD>   // floats published per workgroup per round (plain stores)
__global__ void k_rounds(unsigned* counter, float* buf, int rounds, int* fail, int* mismatches, long long* cyc) {
  const int wg = blockIdx.x, n = gridDim.x;
  const long long t0 = clock64();
  int bad = 0;
  for (int r = 0; r < rounds; ++r) {
    float* mine = buf + ((size_t)(r & 1) * n + wg) * PAYLOAD;
    for (int i = threadIdx.x; i < PAYLOAD; i += blockDim.x) mine[i] = (float)(r * 1000 + wg) + i * 0.001f;
    if (!grid_barrier(counter, (unsigned)(r + 1) * n, fail)) return;
    const int other = (wg + 37) % n;
    const float* theirs = buf + ((size_t)(r & 1) * n + other) * PAYLOAD;
    for (int i = threadIdx.x; i < PAYLOAD; i += blockDim.x)
      if (theirs[i] != (float)(r * 1000 + other) + i * 0.001f) ++bad;
  }
  if (bad) atomicAdd(mismatches, bad);
  if (threadIdx.x == 0) cyc[wg] = clock64() - t0;
}

int main() {
  unsigned* counter; float* buf; int *fail, *mism; long long* cyc;
  const int max_wg = 512, rounds = 200;
  CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&buf, sizeof(float) * 2 * max_wg * 4096));
  CHECK(hipMalloc(&fail, 4)); CHECK(hipMalloc(&mism, 4)); CHECK(hipMalloc(&cyc, 8 * max_wg));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int n_wg : {48, 157, 256}) {
    for (int payload : {1, 3072}) {
      float best = 1e30f; int f = 0, m = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(fail, 0, 4)); CHECK(hipMemset(mism, 0, 4));
        CHECK(hipEventRecord(e0));
        if (payload == 1) hipLaunchKernelGGL(k_rounds<1>, dim3(n_wg), dim3(256), 0, 0, counter, buf, rounds, fail, mism, cyc);
        else hipLaunchKernelGGL(k_rounds<3072>, dim3(n_wg), dim3(256), 0, 0, counter, buf, rounds, fail, mism, cyc);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        int ff, mm; CHECK(hipMemcpy(&ff, fail, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&mm, mism, 4, hipMemcpyDeviceToHost));
        f |= ff; m += mm;
      }
      printf("grid barrier: %3d workgroups x 256 threads, %4d floats published per workgroup per round: "
             "%.2f us per round (write + barrier + read)  timeouts=%d mismatches=%d\n",
             n_wg, payload, best * 1000.0f / rounds, f, m);
    }
  }
  return 0;
}
