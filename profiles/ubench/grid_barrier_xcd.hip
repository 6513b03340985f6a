// grid_barrier_xcd.hip -- the one variant of a device-wide barrier DESIGN.md left open: XCD-hierarchical.
// Workgroups are dealt round-robin to the 8 XCDs (checked here against HW_REG_XCC_ID).  Level 1: one counter per XCD, bumped
// by that XCD's workgroups only; the LAST arriver of an XCD (known from the value its atomic returns) bumps the global counter
// (level 2: 8 arrivals per round instead of 157), waits for it, then raises its XCD's release word; everybody else polls the
// release word of its own XCD.  Same payload protocol as grid_barrier.hip: every workgroup publishes PAYLOAD floats with plain
// stores, crosses the barrier, reads another workgroup's (on another XCD) with plain loads; mismatches are counted.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier_xcd grid_barrier_xcd.hip && ./grid_barrier_xcd
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Sync {                      // one cache line per word
  unsigned xcd_count[8][32];
  unsigned xcd_release[8][32];
  unsigned global_count[32];
};

__device__ inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u; }   // HW_REG_XCC_ID

template <bool HIER>
__device__ inline bool grid_barrier(Sync* s, int xcd, unsigned n_in_xcd, unsigned n_xcd, unsigned n_wg, unsigned round, int* fail) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    if (!HIER) {
      __hip_atomic_fetch_add(&s->global_count[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&s->global_count[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (round + 1) * n_wg) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) { *fail = 1; ok = false; break; }
      }
    } else {
      const unsigned prev = __hip_atomic_fetch_add(&s->xcd_count[xcd][0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (prev + 1 == (round + 1) * n_in_xcd) {                       // last of this XCD: cross-XCD hop
        __hip_atomic_fetch_add(&s->global_count[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(&s->global_count[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (round + 1) * n_xcd) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 22)) { *fail = 1; ok = false; break; }
        }
        __hip_atomic_store(&s->xcd_release[xcd][0], round + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(&s->xcd_release[xcd][0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < round + 1) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 22)) { *fail = 1; ok = false; break; }
        }
      }
    }
  }
  __syncthreads();
  return ok;
}

template <int PAYLOAD, bool HIER>
__global__ void k_rounds(Sync* s, float* buf, int rounds, int* fail, int* mismatches, int* xcd_wrong) {
  const int wg = blockIdx.x, n = gridDim.x;
  const int xcd = wg & 7;
  if (threadIdx.x == 0 && xcc_id() != (unsigned)xcd) atomicAdd(xcd_wrong, 1);
  const unsigned n_in_xcd = (n - xcd + 7) / 8, n_xcd = n < 8 ? n : 8;
  int bad = 0;
  for (int r = 0; r < rounds; ++r) {
    float* mine = buf + ((size_t)(r & 1) * n + wg) * PAYLOAD;
    for (int i = threadIdx.x; i < PAYLOAD; i += blockDim.x) mine[i] = (float)(r * 1000 + wg) + i * 0.001f;
    if (!grid_barrier<HIER>(s, xcd, n_in_xcd, n_xcd, n, r, fail)) return;
    const int other = (wg + 37) % n;                                   // 37 is odd: another XCD
    const float* theirs = buf + ((size_t)(r & 1) * n + other) * PAYLOAD;
    for (int i = threadIdx.x; i < PAYLOAD; i += blockDim.x)
      if (theirs[i] != (float)(r * 1000 + other) + i * 0.001f) ++bad;
  }
  if (bad) atomicAdd(mismatches, bad);
}

template <int PAYLOAD, bool HIER>
int run(Sync* s, float* buf, int* fail, int* mism, int* wrong, int n_wg, int rounds, hipEvent_t e0, hipEvent_t e1) {
  float best = 1e30f; int f = 0, m = 0, w = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemset(s, 0, sizeof(Sync))); CHECK(hipMemset(fail, 0, 4)); CHECK(hipMemset(mism, 0, 4)); CHECK(hipMemset(wrong, 0, 4));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rounds<PAYLOAD, HIER>), dim3(n_wg), dim3(256), 0, 0, s, buf, rounds, fail, mism, wrong);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    int ff, mm, ww; CHECK(hipMemcpy(&ff, fail, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&mm, mism, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&ww, wrong, 4, hipMemcpyDeviceToHost));
    f |= ff; m += mm; w += ww;
  }
  printf("%-13s %3d workgroups, %4d floats published per workgroup and round: %6.2f us per round  timeouts=%d mismatches=%d "
         "workgroups not on XCD (id %% 8): %d\n", HIER ? "hierarchical" : "one counter", n_wg, PAYLOAD, best * 1000.0f / rounds, f, m, w);
  return 0;
}

int main() {
  Sync* s; float* buf; int *fail, *mism, *wrong;
  const int max_wg = 512, rounds = 200;
  CHECK(hipMalloc(&s, sizeof(Sync))); CHECK(hipMalloc(&buf, sizeof(float) * 2 * max_wg * 4096));
  CHECK(hipMalloc(&fail, 4)); CHECK(hipMalloc(&mism, 4)); CHECK(hipMalloc(&wrong, 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int n_wg : {48, 158, 256}) {
    if (run<1, false>(s, buf, fail, mism, wrong, n_wg, rounds, e0, e1)) return 1;
    if (run<1, true>(s, buf, fail, mism, wrong, n_wg, rounds, e0, e1)) return 1;
    if (run<3072, false>(s, buf, fail, mism, wrong, n_wg, rounds, e0, e1)) return 1;
    if (run<3072, true>(s, buf, fail, mism, wrong, n_wg, rounds, e0, e1)) return 1;
  }
  return 0;
}
