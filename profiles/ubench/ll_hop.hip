// ll_hop.hip -- latency of one producer->consumer hop between workgroups of ONE kernel through 16-byte
// "LL" lines {lo, seq, hi, seq} (data-flow synchronisation, no barrier, no atomics), on gfx950:
// 48 producer workgroups publish 64 float64 each, 51 consumer workgroups of 1024 threads poll all 3072 lines.
// Chains of H hops (consumer i republishes for consumer i+1) separate the hop cost from the launch cost.
// Memory: uncached (hipDeviceMallocUncached) vs ordinary hipMalloc with sc0 sc1 accesses.
//   hipcc --offload-arch=gfx950 -O3 -o ll_hop ll_hop.hip && ./ll_hop
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int line_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ll_store(line_t* dst, line_t v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ line_t ll_load(const line_t* src) {
  line_t v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
  return v;
}
constexpr int N = 3072, NP = 48, NC = 51;

// stage s (0..hops): workgroups [s*NP, (s+1)*NP) wait for stage s-1's lines (all N of them, spread over the
// 1024 threads), then publish their own 64 lines of stage s.  Stage 0 publishes immediately.
__global__ __launch_bounds__(1024) void k_chain(line_t* buf, unsigned seq, int hops, int* bad) {
  const int stage = blockIdx.x / NP, b = blockIdx.x % NP;
  if (stage > 0) {
    const line_t* src = buf + (size_t)(stage - 1) * N;
    for (int i = threadIdx.x; i < N; i += 1024) {
      line_t l = ll_load(src + i);
      unsigned spins = 0;
      while (l.y != seq || l.w != seq) { if (++spins > (1u << 22)) { atomicExch(bad, 1); break; } __builtin_amdgcn_s_sleep(1); l = ll_load(src + i); }
      if (l.x != (unsigned)(i + stage - 1)) atomicExch(bad, 2);
    }
    __syncthreads();
  }
  if (threadIdx.x < 64) {
    const int i = b * 64 + threadIdx.x;
    ll_store(buf + (size_t)stage * N + i, line_t{(unsigned)(i + stage), seq, 7u, seq});
  }
}

int main() {
  int* bad; CHECK(hipMalloc(&bad, 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int kind = 0; kind < 2; ++kind) {
    line_t* buf;
    const size_t bytes = sizeof(line_t) * N * 8;
    if (kind == 0) CHECK(hipExtMallocWithFlags((void**)&buf, bytes, hipDeviceMallocUncached));
    else CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMemset(buf, 0, bytes)); CHECK(hipMemset(bad, 0, 4));
    unsigned seq = 0;
    for (int hops : {0, 1, 2, 4}) {
      float best = 1e9f;
      for (int rep = 0; rep < 20; ++rep) {
        ++seq;
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain, dim3(NP * (hops + 1)), dim3(1024), 0, 0, buf, seq, hops, bad);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      int hb; CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      printf("%-28s %d hop(s): %.2f us per launch (event bracket), errors=%d\n",
             kind == 0 ? "uncached memory" : "hipMalloc + sc0 sc1 accesses", hops, best * 1000.f, hb);
    }
    CHECK(hipFree(buf));
  }
  return 0;
}
