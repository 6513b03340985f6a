// lds_rates.hip -- LDS -> VGPR delivery rate on gfx950 for the access shapes the fused PINN kernels use.
//   hipcc --offload-arch=gfx950 -O3 -o lds_rates lds_rates.hip && ./lds_rates
// One workgroup; every wave issues 32 ds_read per iteration (no VALU in between) and waits once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITER 512
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// MODE 0: b128 lane-contiguous   1: b128 broadcast (all lanes one address)   2: b32 lane-contiguous
// MODE 3: b64 lane-contiguous    4: b32 broadcast
template <int MODE>
__global__ void k_lds(float* out, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) sh[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned base;
  if (MODE == 0) base = lane * 16;
  else if (MODE == 1) base = (threadIdx.x >> 6) * 64;
  else if (MODE == 2) base = lane * 4;
  else if (MODE == 3) base = lane * 8;
  else base = (threadIdx.x >> 6) * 64;
  v4f acc = {0, 0, 0, 0};
  const long long t0 = clock64();
  for (int it = 0; it < N_ITER; ++it) {
    v4f r[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (MODE == 0 || MODE == 1)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(base), "n"(i * 1040));
      else if (MODE == 3) {
        v2f t;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t) : "v"(base), "n"(i * 1040));
        r[i] = v4f{t.x, t.y, 0, 0};
      } else {
        float t;
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t) : "v"(base), "n"(i * 1040));
        r[i] = v4f{t, 0, 0, 0};
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" ::"v"(r[i]));
    acc += r[0];
  }
  const long long t1 = clock64();
  out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
  if (lane == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int bytes_per_lane, float* out, long long* cyc) {
  for (int threads : {64, 256, 512, 1024}) {
    hipLaunchKernelGGL((k_lds<MODE>), dim3(1), dim3(threads), 65536, 0, out, cyc);
    hipLaunchKernelGGL((k_lds<MODE>), dim3(1), dim3(threads), 65536, 0, out, cyc);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(threads / 64);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (auto v : h) mx = v > mx ? v : mx;
    const double per = (double)mx / (N_ITER * 32.0);
    printf("%-28s waves=%2d  %6.2f cycles per read per wave -> %6.1f B/clk per CU\n", name, threads / 64,
           per, (threads / 64) * 64.0 * bytes_per_lane / per);
  }
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 4096);
  run<0>("ds_read_b128 contiguous", 16, out, cyc);
  run<1>("ds_read_b128 broadcast", 16, out, cyc);
  run<3>("ds_read_b64 contiguous", 8, out, cyc);
  run<2>("ds_read_b32 contiguous", 4, out, cyc);
  run<4>("ds_read_b32 broadcast", 4, out, cyc);
  return 0;
}
