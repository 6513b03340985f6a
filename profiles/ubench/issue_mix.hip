// issue_mix.hip -- cost of the instruction mixes used by k_fused20r's GEMV inner loop, one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o issue_mix issue_mix.hip && ./issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 200
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// KIND 0: 10 x [2 readlane + 4 pk_fma (SGPR-pair operand)]      (the GEMV body)
// KIND 1: 10 x [4 pk_fma (VGPR operand)]
// KIND 2: KIND 0 + 1 MFMA 16x16x4 f32 per group, two alternating accumulators
// KIND 3: 10 x [1 MFMA] only (two alternating accumulators)
// KIND 4: 10 x [4 v_accvgpr_read + 4 v_mul]
// KIND 5: 10 x [2 readlane + 4 v_fma_f32 (SGPR operand)]
// KIND 6: KIND 1 + 1 MFMA per group
template <int KIND>
__global__ void k_mix(float* out, long long* cyc, float seed) {
  v2f acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = v2f{seed + i, seed - i};
  v2f in = {seed * 0.5f + threadIdx.x, seed * 0.25f};
  float wv = seed + threadIdx.x;
  v4f m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0};
  float a0 = 1, a1 = 2, a2 = 3, a3 = 4;
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a0) : "v"(wv));
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a1) : "v"(wv));
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a2) : "v"(wv));
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a3) : "v"(wv));
  float f0 = seed, f1 = seed, f2 = seed, f3 = seed;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int g = 0; g < 10; ++g) {
      if (KIND == 0 || KIND == 2 || KIND == 5) {
        v2f w;
        w.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv), 2 * g));
        w.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv), 2 * g + 1));
        if (KIND == 5) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f0) : "v"(in.x), "s"(w.x));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f1) : "v"(in.y), "s"(w.x));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f2) : "v"(in.x), "s"(w.y));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f3) : "v"(in.y), "s"(w.y));
        } else {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[(2 * g) & 7]) : "v"(in), "s"(w));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[(2 * g + 1) & 7]) : "v"(in), "s"(w));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[(2 * g + 2) & 7]) : "v"(in), "s"(w));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[(2 * g + 3) & 7]) : "v"(in), "s"(w));
        }
      }
      if (KIND == 1 || KIND == 6) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[(4 * g + q) & 7]) : "v"(in), "v"(in));
      }
      if (KIND == 2 || KIND == 3 || KIND == 6) {
        if (g & 1) m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(in.x, in.y, m1, 0, 0, 0);
        else m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(in.x, in.y, m0, 0, 0, 0);
      }
      if (KIND == 4) {
        float r0, r1, r2, r3;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r0) : "a"(a0));
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r1) : "a"(a1));
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r2) : "a"(a2));
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r3) : "a"(a3));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f0) : "v"(r0));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f1) : "v"(r1));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f2) : "v"(r2));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f3) : "v"(r3));
      }
    }
  }
  const long long t1 = clock64();
  float s = f0 + f1 + f2 + f3 + m0.x + m0.y + m1.z + m1.w;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char* name, float* out, long long* cyc) {
  for (int threads : {256, 512}) {
    hipLaunchKernelGGL((k_mix<KIND>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipLaunchKernelGGL((k_mix<KIND>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(threads / 64);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (auto v : h) mx = v > mx ? v : mx;
    printf("%-52s waves/SIMD=%d  %7.1f cycles per group\n", name, threads / 256, (double)mx / (N_ITER * 10.0));
  }
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 4096);
  run<0>("2 readlane + 4 pk_fma(sgpr)", out, cyc);
  run<1>("4 pk_fma(vgpr)", out, cyc);
  run<5>("2 readlane + 4 fma(sgpr)", out, cyc);
  run<3>("1 mfma16x16x4f32", out, cyc);
  run<2>("2 readlane + 4 pk_fma(sgpr) + 1 mfma", out, cyc);
  run<6>("4 pk_fma(vgpr) + 1 mfma", out, cyc);
  run<4>("4 accvgpr_read + 4 v_mul", out, cyc);
  return 0;
}
