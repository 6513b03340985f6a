// mfma4x4.hip -- operand layout and issue rate of v_mfma_f32_4x4x1_16B_f32 on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o mfma4x4 mfma4x4.hip && ./mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k_layout(float* out) {
  const int l = threadIdx.x;
  // A = 100 + lane, B = 1000 * (lane + 1): D[r] = A_src * B_src identifies which lanes meet
  const float a = 1.0f + l, b = 1000.0f * (l + 1);
  v4f c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

#define N_ITER 100
template <int KIND>
__global__ void k_rate(float* out, long long* cyc, float seed) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) sh[i] = seed + i;
  __syncthreads();
  v4f acc[5];
  for (int i = 0; i < 5; ++i) acc[i] = v4f{seed, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  float a = seed + lane, b = seed * 2 + lane;
  const v4f* shv = (const v4f*)sh;
  const long long t0 = clock64();
  for (int it = 0; it < N_ITER; ++it) {
    if (KIND == 0) {        // 100 MFMAs 4x4x1, 5 accumulators round-robin
#pragma unroll
      for (int m = 0; m < 100; ++m) acc[m % 5] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m % 5], 0, 0, 0);
    } else if (KIND == 1) { // the same + 30 ds_read_b128 feeding the B operand (the layer pattern)
      v4f in[20];
#pragma unroll
      for (int k = 0; k < 20; ++k) in[k] = shv[k * 65 + lane];
#pragma unroll
      for (int k = 0; k < 20; ++k) {
        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, in[k].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, in[k].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, in[k].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, in[k].w, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_4x4x1f32(b, in[k].x, acc[4], 0, 0, 0);
      }
      __builtin_amdgcn_s_barrier();
    } else {                // 25 MFMAs 16x16x4 (same MAC count as 100 4x4x1)
#pragma unroll
      for (int m = 0; m < 25; ++m) acc[m % 5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % 5], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 5; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[threadIdx.x] = s;
  if (lane == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char* name, float* out, long long* cyc) {
  for (int threads : {256, 512}) {
    hipLaunchKernelGGL((k_rate<KIND>), dim3(1), dim3(threads), 40000, 0, out, cyc, 1.0f);
    hipLaunchKernelGGL((k_rate<KIND>), dim3(1), dim3(threads), 40000, 0, out, cyc, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(threads / 64);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (auto v : h) mx = v > mx ? v : mx;
    printf("%-48s waves/SIMD=%d  %8.1f cycles per iteration\n", name, threads / 256, (double)mx / N_ITER);
  }
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 4096);
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out);
  std::vector<float> h(256);
  (void)hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost);
  printf("layout probe: D[lane][r] = A(lane_a) * B(lane_b): (lane_a, lane_b) per (lane, r)\n");
  for (int l : {0, 1, 2, 3, 4, 5, 17, 63}) {
    printf("  lane %2d:", l);
    for (int r = 0; r < 4; ++r) {
      const double v = h[l * 4 + r];
      int la = -1, lb = -1;
      for (int x = 0; x < 64 && la < 0; ++x)
        for (int y = 0; y < 64; ++y)
          if ((double)(1.0f + x) * (double)(1000.0f * (y + 1)) == v) { la = x; lb = y; break; }
      printf("  r%d=(a:%2d,b:%2d)", r, la, lb);
    }
    printf("\n");
  }
  run<0>("100 x mfma_4x4x1 (5 accumulators)", out, cyc);
  run<1>("20 ds_read_b128 + 100 x mfma_4x4x1 + barrier", out, cyc);
  run<2>("25 x mfma_16x16x4 (5 accumulators)", out, cyc);
  return 0;
}
