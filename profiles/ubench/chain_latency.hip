// chain_latency.hip -- cycles per step of the dependent chain inside the L-BFGS two-loop recursion
// (csrc/kernels_optim.h k_lbc_coef_apply): lane i's float64 value becomes final, is broadcast, every other lane does
// one fma with it, the next lane's value becomes final ...   One wave, 64 steps per pass, s_memtime around 16 passes.
// Variants of the broadcast:
//   0  v_readlane_b32 x2 -> SGPR pair -> v_fma_f64 (constant lane index)           (what the kernel does)
//   1  same + s_cselect x2 between readlane and fma
//   2  ds_bpermute_b32 x2 -> v_fma_f64
//   3  LDS: owner lane writes 8 bytes, everybody reads them back (broadcast read)
//   4  no broadcast at all: v_fma_f64 on the lane's own value (the floor: the fma's own dependent latency)
//   5  v_readlane x2 -> v_fma_f64 -> v_fma_f64 (two dependent fma per step: the backward loop's shape is 1 + 1 independent)
//   6  float32: v_readlane x1 -> v_fma_f32
//   7  DPP row_shr:1 x2 (systolic neighbour hand-over inside a row of 16) -> v_fma_f64
//   hipcc --offload-arch=gfx950 -O3 -o chain_latency chain_latency.hip && ./chain_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ double rl(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

template <int KIND>
__global__ __launch_bounds__(64) void k_chain(const double* __restrict__ u, double* out, long long* cyc, int m1) {
  __shared__ double sh[64];
  const int lane = threadIdx.x;
  double uu[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) uu[k] = u[k * 64 + lane];
  double acc = 1.0 + lane * 1e-3, acc2 = 0.5;
  float accf = 1.0f + lane * 1e-3f;
  sh[lane] = 0.0;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int pass = 0; pass < 16; ++pass) {
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      if (KIND == 0) { const double b = rl(acc, k); acc = __builtin_fma(-b, uu[k], acc); }
      if (KIND == 1) { const double r = rl(acc, k); const double b = k < m1 ? r : 0.0; acc = __builtin_fma(-b, uu[k], acc); }
      if (KIND == 2) {
        const int lo = __builtin_amdgcn_ds_bpermute(k * 4, __double2loint(acc)), hi = __builtin_amdgcn_ds_bpermute(k * 4, __double2hiint(acc));
        acc = __builtin_fma(-__hiloint2double(hi, lo), uu[k], acc);
      }
      if (KIND == 3) {
        if (lane == k) sh[0] = acc;
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
        const double b = *(volatile double*)&sh[0];
        acc = __builtin_fma(-b, uu[k], acc);
      }
      if (KIND == 4) { acc = __builtin_fma(-acc, uu[k], acc); }
      if (KIND == 5) { const double b = rl(acc, k); acc2 = __builtin_fma(-b, uu[k], acc2); acc = __builtin_fma(-acc2, uu[k], acc); }
      if (KIND == 6) {
        const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(accf), k));
        accf = __builtin_fmaf(-b, (float)uu[k], accf);
      }
      if (KIND == 7) {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(acc), 0x111, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(acc), 0x111, 0xf, 0xf, false);
        acc = __builtin_fma(-__hiloint2double(hi, lo), uu[k], acc);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[lane] = acc + acc2 + accf;
  if (lane == 0) cyc[0] = t1 - t0;
}

template <int KIND>
int run(const char* name, const double* u, double* out, long long* cyc) {
  long long best = 1ll << 60;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(k_chain<KIND>, dim3(1), dim3(64), 0, 0, u, out, cyc, 51);
    long long h; CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    if (h < best) best = h;
  }
  printf("%-86s %7.1f cycles per step\n", name, best / (16.0 * 64.0));
  return 0;
}

int main() {
  double *u, *out; long long* cyc;
  CHECK(hipMalloc(&u, 64 * 64 * 8)); CHECK(hipMalloc(&out, 64 * 8)); CHECK(hipMalloc(&cyc, 8));
  CHECK(hipMemset(u, 0, 64 * 64 * 8));
  run<4>("4  fma on the lane's own value (floor: dependent v_fma_f64 latency)", u, out, cyc);
  run<0>("0  v_readlane x2 -> v_fma_f64", u, out, cyc);
  run<1>("1  v_readlane x2 -> s_cselect x2 -> v_fma_f64", u, out, cyc);
  run<5>("5  v_readlane x2 -> v_fma_f64 -> v_fma_f64", u, out, cyc);
  run<2>("2  ds_bpermute x2 -> v_fma_f64", u, out, cyc);
  run<3>("3  LDS write by the owner, broadcast read -> v_fma_f64", u, out, cyc);
  run<7>("7  DPP row_shr:1 x2 -> v_fma_f64 (neighbour hand-over)", u, out, cyc);
  run<6>("6  float32: v_readlane -> v_fma_f32", u, out, cyc);
  return 0;
}
