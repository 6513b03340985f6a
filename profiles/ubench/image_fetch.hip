// image_fetch.hip -- how long does a workgroup wait for its 23 KB weight image at kernel start?
// 158 workgroups x 4 waves (the headline launch) all fetch the same image, which a 1-workgroup kernel
// rewrote just before (as the Adam update does).  Per variant: shader cycles from kernel entry until
// s_waitcnt vmcnt(0), median / min / max over all waves.
//   hipcc --offload-arch=gfx950 -O3 -o image_fetch image_fetch.hip && ./image_fetch
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int PIECES = 24;                  // 1-KiB pieces (256 floats)
constexpr int NF = PIECES * 256;

__global__ void k_rewrite(float* img, float v) {
  for (int i = threadIdx.x; i < NF; i += blockDim.x) img[i] = v + i;
}

// KIND 0: 6 x global_load_lds_dwordx4 per wave (the kernel's form)
// KIND 1: 1 x global_load_lds_dwordx4 per wave (4 KiB per workgroup)
// KIND 2: 6 x global_load_dwordx4 per wave into registers, then ds_write_b128
// KIND 3: wave 0 alone issues all 24 DMA pieces
// KIND 4: 24 x global_load_lds_dword (256 B pieces), 6 KiB per workgroup... x4 = whole image at dword width
// KIND 5: KIND 0 twice in a row (second time: whatever the first left in the caches)
// KIND 6: one plain 4-byte load per lane (the coordinate load), for reference
// KIND 7: KIND 0 but only the first 12 pieces (forward half), 3 per wave
template <int KIND>
__global__ __launch_bounds__(256) void k_fetch(const float* __restrict__ img, long long* __restrict__ cyc,
                                                float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const long long t0 = __builtin_readcyclecounter();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long t1 = 0, t2 = 0;
  float acc = 0.0f;
  if (KIND == 0 || KIND == 5 || KIND == 7) {
    const int np = KIND == 7 ? 12 : PIECES;
    for (int c = wave; c < np; c += 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + c * 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(wl + c * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t1 = __builtin_readcyclecounter();
    if (KIND == 5) {
      for (int c = wave; c < PIECES; c += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + c * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(wl + NF + c * 256), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t2 = __builtin_readcyclecounter();
    }
  } else if (KIND == 1) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + wave * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(wl + wave * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t1 = __builtin_readcyclecounter();
  } else if (KIND == 2) {
    v4f r[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) r[m] = *reinterpret_cast<const v4f*>(img + (wave + 4 * m) * 256 + lane * 4);
#pragma unroll
    for (int m = 0; m < 6; ++m) *reinterpret_cast<v4f*>(wl + (wave + 4 * m) * 256 + lane * 4) = r[m];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    t1 = __builtin_readcyclecounter();
  } else if (KIND == 3) {
    if (wave == 0)
      for (int c = 0; c < PIECES; ++c)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + c * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(wl + c * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t1 = __builtin_readcyclecounter();
  } else if (KIND == 4) {
    for (int c = wave; c < 4 * PIECES; c += 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + c * 64 + lane),
                                       (__attribute__((address_space(3))) void*)(wl + c * 64), 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t1 = __builtin_readcyclecounter();
  } else if (KIND == 6) {
    acc = img[(blockIdx.x & 15) * 256 + threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t1 = __builtin_readcyclecounter();
  }
  __syncthreads();
  acc += wl[(threadIdx.x * 7) % NF];
  if (acc == 12345.678f) sink[0] = acc;
  if (lane == 0) {
    cyc[(blockIdx.x * 4 + wave) * 2 + 0] = t1 - t0;
    cyc[(blockIdx.x * 4 + wave) * 2 + 1] = t2 ? t2 - t1 : 0;
  }
}

template <int KIND>
static void run(const char* name, float* img, long long* cyc, float* sink, bool rewrite) {
  const int WG = 158, NW = WG * 4;
  std::vector<long long> h(NW * 2), a, b;
  for (int it = 0; it < 12; ++it) {
    if (rewrite) hipLaunchKernelGGL(k_rewrite, dim3(1), dim3(256), 0, 0, img, (float)it);
    hipLaunchKernelGGL(k_fetch<KIND>, dim3(WG), dim3(256), 2 * NF * 4, 0, img, cyc, sink);
    hipDeviceSynchronize();
    if (it < 2) continue;
    hipMemcpy(h.data(), cyc, NW * 16, hipMemcpyDeviceToHost);
    for (int w = 0; w < NW; ++w) { a.push_back(h[2 * w]); if (h[2 * w + 1]) b.push_back(h[2 * w + 1]); }
  }
  std::sort(a.begin(), a.end());
  printf("%-64s %s  median %6lld  min %6lld  p90 %6lld  max %6lld", name, rewrite ? "rewritten" : "untouched",
         a[a.size() / 2], a.front(), a[a.size() * 9 / 10], a.back());
  if (!b.empty()) { std::sort(b.begin(), b.end()); printf("   | second pass median %lld max %lld", b[b.size() / 2], b.back()); }
  printf("\n");
}

int main() {
  float *img, *sink; long long* cyc;
  hipMalloc(&img, NF * 4 + 4096); hipMalloc(&sink, 16); hipMalloc(&cyc, 158 * 4 * 16);
  hipFuncSetAttribute((const void*)k_fetch<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NF * 4);
  printf("# cycles (s_memtime-class counter: __builtin_readcyclecounter) from kernel entry to data landed; 158 WG x 4 waves\n");
  for (int rw = 1; rw >= 0; --rw) {
    run<6>("one 4-byte load per lane", img, cyc, sink, rw);
    run<1>("1 x global_load_lds_dwordx4 per wave (4 KiB / WG)", img, cyc, sink, rw);
    run<7>("3 x global_load_lds_dwordx4 per wave (12 KiB / WG)", img, cyc, sink, rw);
    run<0>("6 x global_load_lds_dwordx4 per wave (24 KiB / WG)", img, cyc, sink, rw);
    run<3>("24 x global_load_lds_dwordx4 from wave 0 alone", img, cyc, sink, rw);
    run<4>("24 x global_load_lds_dword per wave (24 KiB / WG)", img, cyc, sink, rw);
    run<2>("6 x global_load_dwordx4 + ds_write_b128 per wave", img, cyc, sink, rw);
    run<5>("6 x global_load_lds_dwordx4 per wave, twice", img, cyc, sink, rw);
  }
  return 0;
}
