// graph_gap.hip -- what does the boundary between two dependent kernels cost, launched one by one on a stream
// versus replayed from a captured hipGraph?  The pair mimics an Adam step of the headline workload: a 158-workgroup
// kernel that is busy for ~25 us, then a 54-workgroup kernel busy for ~1 us, each depending on the one before.
//   hipcc --offload-arch=gfx950 -O3 -o graph_gap graph_gap.hip && ./graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void k_busy(long long ticks, int* sink) {           // ticks of the 100 MHz wall clock
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
  if (ticks < 0) sink[0] = 1;
}

// the same with the memory behaviour of the real pair: A holds 115 KB of LDS and ends by writing a 12 KB row per
// workgroup; B reads all 158 rows (16 row slices x 54 column blocks, as k_reduce_adam does) and writes 12 KB
// STORE 0: no row written, 1: plain stores, 2: system-scope (write-through) stores, 3: non-temporal stores
template <int STORE>
__global__ void k_busy_rows(long long ticks, float* rows, int R) {
  extern __shared__ float lds[];
  const long long t0 = wall_clock64();
  lds[threadIdx.x] = (float)t0;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
  for (int i = threadIdx.x; i < R; i += 256) {
    float* p = rows + (size_t)blockIdx.x * R + i;
    const float v = lds[threadIdx.x] + i;
    if (STORE == 1) *p = v;
    if (STORE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (STORE == 3) __builtin_nontemporal_store(v, p);
  }
}
template <bool READ>
__global__ void k_busy_sum(long long ticks, const float* rows, int R, int n_rows, float* out) {
  const long long t0 = wall_clock64();
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
  float acc = 0.0f;
  if (READ && col < R) for (int r = slice; r < n_rows; r += 4) acc += rows[(size_t)r * R + col];
  __shared__ float part[256];
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 64 && col < R) out[col] = part[threadIdx.x] + part[threadIdx.x + 64] + part[threadIdx.x + 128] + part[threadIdx.x + 192];
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}

// B with RG row groups: grid = 54 x RG blocks, block (cb, rg) sums rows rg*4+slice, +4*RG, ... of 64 columns with
// 8 independent accumulators per thread; writes one partial row per row group (the final combine is not modelled)
template <int RG>
__global__ void k_sum_wide(long long ticks, const float* rows, int R, int n_rows, float* out) {
  const long long t0 = wall_clock64();
  const int cb = blockIdx.x % 54, rg = blockIdx.x / 54;
  const int col = cb * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int r0 = rg * 4 + slice, step = 4 * RG;
  int r = r0;
  for (; r + 7 * step < n_rows; r += 8 * step)
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += rows[(size_t)(r + u * step) * R + col];
  for (; r < n_rows; r += step) a[0] += rows[(size_t)r * R + col];
  __shared__ float part[256];
  part[threadIdx.x] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (threadIdx.x < 64) out[(size_t)rg * R + col] = part[threadIdx.x] + part[threadIdx.x + 64] + part[threadIdx.x + 128] + part[threadIdx.x + 192];
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  int* sink; hipMalloc(&sink, 16);
  hipStream_t st; hipStreamCreate(&st);
  const int PAIRS = 300, CHUNK = 50;
  for (int variant = 0; variant < 3; ++variant) {
    const long long a_ticks = variant == 2 ? 0 : 2500, b_ticks = variant == 2 ? 0 : 100;   // 25 us + 1 us; or empty kernels
    const int reps = variant == 1 ? 1 : 1;
    (void)reps;
    // one by one
    for (int warm = 0; warm < 2; ++warm) {
      hipStreamSynchronize(st);
      const double t0 = now_us();
      for (int i = 0; i < PAIRS; ++i) {
        hipLaunchKernelGGL(k_busy, dim3(158), dim3(256), 0, st, a_ticks, sink);
        hipLaunchKernelGGL(k_busy, dim3(54), dim3(256), 0, st, b_ticks, sink);
      }
      hipStreamSynchronize(st);
      const double dt = (now_us() - t0) / PAIRS;
      if (warm) printf("busy %5.1f + %4.1f us, launched one by one : %6.2f us per pair -> %5.2f us of boundaries per pair\n",
                       a_ticks / 100.0, b_ticks / 100.0, dt, dt - (a_ticks + b_ticks) / 100.0);
    }
    // captured graph of CHUNK pairs
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < CHUNK; ++i) {
      hipLaunchKernelGGL(k_busy, dim3(158), dim3(256), 0, st, a_ticks, sink);
      hipLaunchKernelGGL(k_busy, dim3(54), dim3(256), 0, st, b_ticks, sink);
    }
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int warm = 0; warm < 2; ++warm) {
      hipStreamSynchronize(st);
      const double t0 = now_us();
      for (int i = 0; i < PAIRS / CHUNK; ++i) hipGraphLaunch(ge, st);
      hipStreamSynchronize(st);
      const double dt = (now_us() - t0) / PAIRS;
      if (warm) printf("busy %5.1f + %4.1f us, hipGraph of %d pairs     : %6.2f us per pair -> %5.2f us of boundaries per pair\n",
                       a_ticks / 100.0, b_ticks / 100.0, CHUNK, dt, dt - (a_ticks + b_ticks) / 100.0);
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    if (variant == 0) {   // same, but the second kernel does not wait for the first (no barrier bit) -- a lower bound
      hipStream_t s2; hipStreamCreate(&s2);
      hipStreamSynchronize(st);
      const double t0 = now_us();
      for (int i = 0; i < PAIRS; ++i) hipLaunchKernelGGL(k_busy, dim3(158), dim3(256), 0, st, a_ticks, sink);
      hipStreamSynchronize(st);
      const double dt = (now_us() - t0) / PAIRS;
      printf("busy %5.1f us alone, one by one             : %6.2f us per launch -> %5.2f us of boundary\n", a_ticks / 100.0, dt, dt - a_ticks / 100.0);
      hipStreamDestroy(s2);
    }
  }
  {  // the pair with memory behaviour
    const int R = 3072 + 384, NR = 158;
    float *rows, *out; hipMalloc(&rows, (size_t)NR * R * 4); hipMalloc(&out, 8 * R * 4);
    auto run = [&](const char* what, auto ka, auto kb) {
      for (int warm = 0; warm < 2; ++warm) {
        hipStreamSynchronize(st);
        const double t0 = now_us();
        for (int i = 0; i < PAIRS; ++i) {
          hipLaunchKernelGGL(ka, dim3(NR), dim3(256), 1024, st, 2500LL, rows, R);
          hipLaunchKernelGGL(kb, dim3((R + 63) / 64), dim3(256), 0, st, 100LL, rows, R, NR, out);
        }
        hipStreamSynchronize(st);
        const double dt = (now_us() - t0) / PAIRS;
        if (warm) printf("busy 25.0 + 1.0 us, %-66s: %6.2f us per pair -> %5.2f us beyond the busy time\n", what, dt, dt - 26.0);
      }
    };
    run("A writes nothing,                    B reads nothing", k_busy_rows<0>, k_busy_sum<false>);
    run("A writes 158 rows of 13.8 KB (plain), B reads nothing", k_busy_rows<1>, k_busy_sum<false>);
    run("A writes nothing,                    B sums the 158 rows", k_busy_rows<0>, k_busy_sum<true>);
    run("A writes rows (plain stores),        B sums the rows", k_busy_rows<1>, k_busy_sum<true>);
    auto runw = [&](const char* what, auto ka, auto kb, int rg) {
      for (int warm = 0; warm < 2; ++warm) {
        hipStreamSynchronize(st);
        const double t0 = now_us();
        for (int i = 0; i < PAIRS; ++i) {
          hipLaunchKernelGGL(ka, dim3(NR), dim3(256), 1024, st, 2500LL, rows, R);
          hipLaunchKernelGGL(kb, dim3(54 * rg), dim3(256), 0, st, 100LL, rows, R, NR, out);
        }
        hipStreamSynchronize(st);
        const double dt = (now_us() - t0) / PAIRS;
        if (warm) printf("busy 25.0 + 1.0 us, %-66s: %6.2f us per pair -> %5.2f us beyond the busy time\n", what, dt, dt - 26.0);
      }
    };
    runw("A writes rows (plain), B = 54 x 1 blocks, 8 loads in flight", k_busy_rows<1>, k_sum_wide<1>, 1);
    runw("A writes rows (plain), B = 54 x 2 blocks, 8 loads in flight", k_busy_rows<1>, k_sum_wide<2>, 2);
    runw("A writes rows (plain), B = 54 x 4 blocks, 8 loads in flight", k_busy_rows<1>, k_sum_wide<4>, 4);
    runw("A writes rows (plain), B = 54 x 8 blocks, 5 loads in flight", k_busy_rows<1>, k_sum_wide<8>, 8);
    runw("A writes nothing,      B = 54 x 4 blocks, 8 loads in flight", k_busy_rows<0>, k_sum_wide<4>, 4);
    run("A writes rows (system-scope stores), B sums the rows", k_busy_rows<2>, k_busy_sum<true>);
    run("A writes rows (non-temporal stores), B sums the rows", k_busy_rows<3>, k_busy_sum<true>);
  }
  return 0;
}
