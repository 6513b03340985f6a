// valu_rates.hip -- gfx950 issue-rate microbenchmarks that size the fused PINN kernel's VALU/LDS budget.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
// Reports shader cycles (s_memtime) per wave-instruction for independent chains, 1 wave per SIMD
// (256-thread block, one block) and 2 waves per SIMD (512-thread block).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITER 256

template <int KIND>
__global__ void k_rate(float* out, long long* cyc, float seed) {
  float a[16];
  typedef float float2v __attribute__((ext_vector_type(2)));
  float2v p[8];
  double d[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) { p[i] = float2v{a[2 * i], a[2 * i + 1]}; d[i] = a[i]; }
  const float m = seed * 0.5f, c = seed * 0.25f;
  const float2v m2{m, m}, c2{c, c};
  const double md = m, cd = c;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < N_ITER; ++it) {
    if (KIND == 0) {          // 16 independent v_fma_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    } else if (KIND == 1) {   // 8 independent v_pk_fma_f32 (= 16 FMAs per lane)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
    } else if (KIND == 2) {   // 8 independent v_fma_f64
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
    } else if (KIND == 3) {   // 16 v_exp_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
    } else if (KIND == 4) {   // 16 v_rcp_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
    } else if (KIND == 5) {   // 16 v_fma_f32 with an SGPR operand
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(m), "v"(c));
    } else if (KIND == 6) {   // 16 v_mul_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
    } else if (KIND == 7) {   // 1 dependent chain of v_fma_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
    }
  }
  const long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y + (float)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// LDS read throughput: each wave issues 16 ds_read_b128 (conflict-free, lane-contiguous) per iteration
__global__ void k_lds(float* out, long long* cyc) {
  __shared__ float4 sh[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = float4{(float)i, 0, 0, 0};
  __syncthreads();
  float acc = 0;
  const int lane = threadIdx.x & 63;
  const long long t0 = clock64();
  for (int it = 0; it < N_ITER; ++it) {
    float4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = sh[(i * 65 + lane + it) & 4095];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const long long t1 = clock64();
  out[threadIdx.x] = acc;
  if (lane == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

// the fused kernel's forward-layer LDS pattern: per k one lane-contiguous b128 (inputs) and one
// wave-uniform b128 + b32 (weights), 20 k per "layer", consumed by 20 FMAs each
__global__ void k_layer(float* out, long long* cyc) {
  __shared__ float4 sh[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = float4{(float)i, 1, 2, 3};
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[20];
#pragma unroll
  for (int i = 0; i < 20; ++i) acc[i] = 0;
  const float* shf = (const float*)sh;
  const long long t0 = clock64();
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int k = 0; k < 20; ++k) {
      const float4 in = sh[k * 65 + lane];
      const float4 w = sh[2048 + (k * 4 + (wave & 3)) * 2];
      const float w4 = shf[4 * (2048 + (k * 4 + (wave & 3)) * 2) + 4];
      acc[0] += in.x * w.x; acc[1] += in.y * w.x; acc[2] += in.z * w.x; acc[3] += in.w * w.x;
      acc[4] += in.x * w.y; acc[5] += in.y * w.y; acc[6] += in.z * w.y; acc[7] += in.w * w.y;
      acc[8] += in.x * w.z; acc[9] += in.y * w.z; acc[10] += in.z * w.z; acc[11] += in.w * w.z;
      acc[12] += in.x * w.w; acc[13] += in.y * w.w; acc[14] += in.z * w.w; acc[15] += in.w * w.w;
      acc[16] += in.x * w4; acc[17] += in.y * w4; acc[18] += in.z * w4; acc[19] += in.w * w4;
    }
    __builtin_amdgcn_s_barrier();
  }
  const long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 20; ++i) s += acc[i];
  out[threadIdx.x] = s;
  if (lane == 0) cyc[wave] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int threads, int per_iter, float* out, long long* cyc) {
  hipLaunchKernelGGL((k_rate<KIND>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.0f);
  hipLaunchKernelGGL((k_rate<KIND>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.0f);
  hipDeviceSynchronize();
  std::vector<long long> h(threads / 64);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  long long mx = 0;
  for (auto v : h) mx = v > mx ? v : mx;
  printf("%-34s threads=%4d  %7.2f cycles per wave-instruction (slowest wave)\n", name, threads,
         (double)mx / ((double)N_ITER * per_iter));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
  for (int threads : {256, 512, 1024}) {
    run<0>("v_fma_f32 x16 independent", threads, 16, out, cyc);
    run<1>("v_pk_fma_f32 x8 independent", threads, 8, out, cyc);
    run<2>("v_fma_f64 x8 independent", threads, 8, out, cyc);
    run<3>("v_exp_f32 x16", threads, 16, out, cyc);
    run<4>("v_rcp_f32 x16", threads, 16, out, cyc);
    run<5>("v_fma_f32 x16 (SGPR operand)", threads, 16, out, cyc);
    run<6>("v_mul_f32 x16", threads, 16, out, cyc);
    run<7>("v_fma_f32 dependent chain", threads, 16, out, cyc);
  }
  for (int threads : {64, 256, 512}) {
    hipLaunchKernelGGL(k_lds, dim3(1), dim3(threads), 0, 0, out, cyc);
    hipLaunchKernelGGL(k_lds, dim3(1), dim3(threads), 0, 0, out, cyc);
    hipDeviceSynchronize();
    std::vector<long long> h(threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (auto v : h) mx = v > mx ? v : mx;
    printf("ds_read_b128 x16 + 64 v_add       threads=%4d  %7.2f cycles per ds_read_b128 per wave (slowest wave)\n",
           threads, (double)mx / (N_ITER * 16.0));
  }
  for (int threads : {64, 256, 512}) {
    hipLaunchKernelGGL(k_layer, dim3(1), dim3(threads), 0, 0, out, cyc);
    hipLaunchKernelGGL(k_layer, dim3(1), dim3(threads), 0, 0, out, cyc);
    hipDeviceSynchronize();
    std::vector<long long> h(threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (auto v : h) mx = v > mx ? v : mx;
    printf("layer pattern (60 ds_read + 400 fma + barrier) threads=%4d  %8.1f cycles per layer (slowest wave)\n",
           threads, (double)mx / N_ITER);
  }
  return 0;
}
