// lds_patterns.hip -- what a 32-byte (a, z_x, z_t, z_xx) tile entry costs to read / write from LDS on gfx950 in the
// access shapes of k_t16_fused (csrc/kernels_tile16f.h), for three tile layouts:
//   AoS   [row][17] x 32 B, two ds_*_b128 per entry                      (the layout of rounds 1-4)
//   SoA2  2 planes [row][17] x 16 B, two ds_*_b128 per entry
//   SoA4  4 planes [row][17] x  8 B, four ds_*_b64 per entry
// shapes (lane = 16 g + m):  B  entry (row g, point m)        -- layer-GEMM B operand
//                            T  entry (row m, point g)        -- both operands of the weight-gradient tiles
//                            W  entry (row 4 g, point m)      -- epilogue writes of a matrix-instruction result tile
//   hipcc --offload-arch=gfx950 -O3 -o lds_patterns lds_patterns.hip && ./lds_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITER 256
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int PD = 17, ROWS = 112;

template <int LAYOUT, int SHAPE, bool WRITE>
__global__ void k(float* out, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  for (int i = threadIdx.x; i < 65536 / 4 * 2; i += blockDim.x) sh[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
  const int e = SHAPE == 0 ? g * PD + m : SHAPE == 1 ? m * PD + g : 4 * g * PD + m;
  const unsigned esz = LAYOUT == 0 ? 32 : LAYOUT == 1 ? 16 : 8;
  const unsigned plane = ROWS * PD * esz;
  const unsigned base = e * esz;
  v4f acc = {0, 0, 0, 0};
  const long long t0 = clock64();
  for (int it = 0; it < N_ITER; ++it) {
    v4f r[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {                // 8 entries per iteration, 4 rows apart (one k-step) each
      const unsigned a = base + i * 4 * PD * esz;
      if (LAYOUT == 0) {
        if (WRITE) {
          asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(acc));
          asm volatile("ds_write_b128 %0, %1 offset:16" ::"v"(a), "v"(acc));
        } else {
          asm volatile("ds_read_b128 %0, %1" : "=v"(r[2 * i]) : "v"(a));
          asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(r[2 * i + 1]) : "v"(a));
        }
      } else if (LAYOUT == 1) {
        if (WRITE) {
          asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(acc));
          asm volatile("ds_write_b128 %0, %1" ::"v"(a + plane), "v"(acc));
        } else {
          asm volatile("ds_read_b128 %0, %1" : "=v"(r[2 * i]) : "v"(a));
          asm volatile("ds_read_b128 %0, %1" : "=v"(r[2 * i + 1]) : "v"(a + plane));
        }
      } else {
        v2f t0_, t1_, t2_, t3_;
        if (WRITE) {
          const v2f w = {acc.x, acc.y};
          asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(w));
          asm volatile("ds_write_b64 %0, %1" ::"v"(a + plane), "v"(w));
          asm volatile("ds_write_b64 %0, %1" ::"v"(a + 2 * plane), "v"(w));
          asm volatile("ds_write_b64 %0, %1" ::"v"(a + 3 * plane), "v"(w));
        } else {
          asm volatile("ds_read_b64 %0, %1" : "=v"(t0_) : "v"(a));
          asm volatile("ds_read_b64 %0, %1" : "=v"(t1_) : "v"(a + plane));
          asm volatile("ds_read_b64 %0, %1" : "=v"(t2_) : "v"(a + 2 * plane));
          asm volatile("ds_read_b64 %0, %1" : "=v"(t3_) : "v"(a + 3 * plane));
          r[2 * i] = v4f{t0_.x, t0_.y, t1_.x, t1_.y};
          r[2 * i + 1] = v4f{t2_.x, t2_.y, t3_.x, t3_.y};
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!WRITE) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(r[i]));
      acc += r[0];
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
  if (lane == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

template <int LAYOUT, int SHAPE, bool WRITE>
static void run(const char* name, float* out, long long* cyc) {
  for (int threads : {64, 512}) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<LAYOUT, SHAPE, WRITE>), dim3(1), dim3(threads), 160 * 1024, 0, out, cyc);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(threads / 64);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (auto v : h) mx = v > mx ? v : mx;
    const double per = (double)mx / (N_ITER * 8.0);
    printf("%-44s waves=%d  %7.2f cycles per 32-byte entry per wave -> %6.1f B/clk per CU\n", name, threads / 64, per,
           (threads / 64) * 64.0 * 32 / per);
  }
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 4096);
  (void)hipFuncSetAttribute((const void*)k<0, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define ALL(L, S, W, n) (void)hipFuncSetAttribute((const void*)k<L, S, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); run<L, S, W>(n, out, cyc)
  ALL(0, 0, false, "AoS  read  B (row g, point m)");
  ALL(1, 0, false, "SoA2 read  B");
  ALL(2, 0, false, "SoA4 read  B");
  ALL(0, 1, false, "AoS  read  T (row m, point g)");
  ALL(1, 1, false, "SoA2 read  T");
  ALL(2, 1, false, "SoA4 read  T");
  ALL(0, 2, true, "AoS  write W (row 4 g, point m)");
  ALL(1, 2, true, "SoA2 write W");
  ALL(2, 2, true, "SoA4 write W");
  ALL(0, 0, true, "AoS  write B");
  ALL(2, 0, true, "SoA4 write B");
  return 0;
}
