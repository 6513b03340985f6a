// mfma_f64_4x4x4.hip -- operand/result lane maps, issue rate and VALU co-issue of v_mfma_f64_4x4x4_4b_f64
// (and v_mfma_f64_16x16x4_f64 for comparison) on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_4x4x4 mfma_f64_4x4x4.hip && ./mfma_f64_4x4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

// one wave per block; block (la, lb): A is one-hot at lane la, B one-hot at lane lb -> which D lanes light up
__global__ void k_layout(double* out) {
  const int l = threadIdx.x, la = blockIdx.x >> 6, lb = blockIdx.x & 63;
  const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
  double c = 0.0;
  c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
  out[(size_t)blockIdx.x * 64 + l] = c;
}

#define N_ITER 50
template <int KIND>
__global__ void k_rate(double* out, long long* cyc, double seed) {
  const int lane = threadIdx.x & 63;
  double a = seed + lane, b = seed * 2 + lane;
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = seed * i;
  v4d big[4];
  for (int i = 0; i < 4; ++i) big[i] = v4d{seed, 0, 0, 0};
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i;
  const long long t0 = clock64();
  for (int it = 0; it < N_ITER; ++it) {
    if (KIND == 0) {          // 100 independent-ish 4x4x4 (8 accumulators round-robin)
#pragma unroll
      for (int m = 0; m < 100; ++m) acc[m % 8] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[m % 8], 0, 0, 0);
    } else if (KIND == 1) {   // 100 dependent 4x4x4 (one accumulator)
#pragma unroll
      for (int m = 0; m < 100; ++m) acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[0], 0, 0, 0);
    } else if (KIND == 2) {   // 25 x 16x16x4 f64 (4 accumulators): the same MAC count as 100 x 4x4x4
#pragma unroll
      for (int m = 0; m < 25; ++m) big[m % 4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, big[m % 4], 0, 0, 0);
    } else if (KIND == 3) {   // 400 v_fma_f64 (8 chains): the same MAC count on the vector ALU
#pragma unroll
      for (int m = 0; m < 400; ++m) v[m % 8] = __builtin_fma(v[m % 8], a, b);
    } else if (KIND == 4) {   // 100 x 4x4x4 interleaved with 100 v_fma_f64: do the pipes overlap?
#pragma unroll
      for (int m = 0; m < 100; ++m) {
        acc[m % 8] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[m % 8], 0, 0, 0);
        v[m % 8] = __builtin_fma(v[m % 8], a, b);
      }
    } else if (KIND == 5) {   // 100 x 4x4x4 interleaved with 200 f32 fma
      float w[8];
      for (int i = 0; i < 8; ++i) w[i] = (float)v[i];
#pragma unroll
      for (int m = 0; m < 100; ++m) {
        acc[m % 8] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[m % 8], 0, 0, 0);
        w[m % 8] = __builtin_fmaf(w[m % 8], (float)a, (float)b);
        w[(m + 4) % 8] = __builtin_fmaf(w[(m + 4) % 8], (float)a, (float)b);
      }
      for (int i = 0; i < 8; ++i) v[i] = w[i];
    } else if (KIND == 6) {   // 100 x 4x4x4 + 40 ds_bpermute (the transposes of the weight-gradient operands)
#pragma unroll
      for (int m = 0; m < 100; ++m) {
        acc[m % 8] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[m % 8], 0, 0, 0);
        if (m % 5 == 0) {
          const int src = ((lane & 3) * 4 + ((lane >> 2) & 3) + (lane & 48)) << 2;
          int lo = __double2loint(v[m % 8]), hi = __double2hiint(v[m % 8]);
          lo = __builtin_amdgcn_ds_bpermute(src, lo);
          hi = __builtin_amdgcn_ds_bpermute(src, hi);
          v[m % 8] = __hiloint2double(hi, lo);
        }
      }
    } else if (KIND == 7) {   // 20 x (exp + divide) in f64: the tanh of 5 features x 4... per layer cost reference
#pragma unroll
      for (int m = 0; m < 20; ++m) {
        const double e = exp(v[m % 8] * 1e-3);
        v[m % 8] = (1.0 - e) / (1.0 + e);
      }
    }
  }
  const long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i] + v[i];
  for (int i = 0; i < 4; ++i) s += big[i].x + big[i].y + big[i].z + big[i].w;
  out[threadIdx.x] = s;
  if (lane == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char* name, double* out, long long* cyc) {
  for (int threads : {256, 512}) {
    hipLaunchKernelGGL((k_rate<KIND>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.0);
    hipLaunchKernelGGL((k_rate<KIND>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.0);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(threads / 64);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (auto v : h) mx = v > mx ? v : mx;
    printf("%-64s waves/SIMD=%d  %8.1f cycles per iteration\n", name, threads / 256, (double)mx / N_ITER);
  }
}

int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 4096 * 64 * 8); (void)hipMalloc(&cyc, 4096);
  hipLaunchKernelGGL(k_layout, dim3(4096), dim3(64), 0, 0, out);
  std::vector<double> h(4096 * 64);
  (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  // D[ld] != 0 for the pair (la, lb): same block, same k; ld = f(block, i(la), j(lb))
  printf("v_mfma_f64_4x4x4_4b: for A one-hot at lane la and B one-hot at lane lb, the D lane that is 1 (or -)\n");
  int n_hits = 0;
  std::vector<int> hit(4096, -1);
  for (int p = 0; p < 4096; ++p)
    for (int l = 0; l < 64; ++l)
      if (h[(size_t)p * 64 + l] != 0.0) { hit[p] = l; ++n_hits; }
  printf("pairs with a product: %d (expected 4 blocks x 4 k x 4 i x 4 j = 256)\n", n_hits);
  printf("la: lb->ld ...\n");
  for (int la = 0; la < 64; ++la) {
    printf("  A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (hit[la * 64 + lb] >= 0) printf("  B%2d->D%2d", lb, hit[la * 64 + lb]);
    printf("\n");
  }
  // hypothesis check: A[i][k] at lane 16b + 4k + i ; B[k][j] at lane 16b + 4k + j ; D[i][j] at lane 16b + 4i + j
  int ok_h1 = 1, ok_h2 = 1;
  for (int b = 0; b < 4; ++b) for (int k = 0; k < 4; ++k) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
    if (hit[(16 * b + 4 * k + i) * 64 + 16 * b + 4 * k + j] != 16 * b + 4 * i + j) ok_h1 = 0;
    if (hit[(16 * b + 4 * i + k) * 64 + 16 * b + 4 * j + k] != 16 * b + 4 * j + i) ok_h2 = 0;
  }
  printf("H1 {A: 16b+4k+i, B: 16b+4k+j, D: 16b+4i+j}: %s\n", ok_h1 ? "HOLDS" : "no");
  printf("H2 {A: 16b+4i+k, B: 16b+4j+k, D: 16b+4j+i}: %s\n", ok_h2 ? "HOLDS" : "no");
  run<0>("100 x mfma_f64_4x4x4 (8 accumulators)", out, cyc);
  run<1>("100 x mfma_f64_4x4x4 (dependent chain)", out, cyc);
  run<2>("25 x mfma_f64_16x16x4 (4 accumulators)", out, cyc);
  run<3>("400 x v_fma_f64 (8 chains)", out, cyc);
  run<4>("100 x mfma_f64_4x4x4 + 100 x v_fma_f64 interleaved", out, cyc);
  run<5>("100 x mfma_f64_4x4x4 + 200 x v_fma_f32 interleaved", out, cyc);
  run<6>("100 x mfma_f64_4x4x4 + 40 ds_bpermute_b32", out, cyc);
  run<7>("20 x f64 (exp, divide)", out, cyc);
  return 0;
}
