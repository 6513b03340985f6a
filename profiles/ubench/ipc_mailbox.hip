// ipc_mailbox.hip -- feasibility + latency of a flag/mailbox exchange between two PROCESSES through
// hipIpc-shared uncached device memory (the building block of a one-shot all-reduce for the 24 KB gradient).
//   ./ipc_mailbox A handle_file &   ./ipc_mailbox B handle_file
// A owns the mailbox; B writes payload + sequence flag into it round after round, A's kernel polls the flag,
// checks the payload and answers through a second flag that B polls (ping-pong): round-trip time / 2 = one-way.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int PAYLOAD = 3072;   // doubles
struct Box { unsigned long long flag_ab, flag_ba; double data[PAYLOAD]; int bad, timeout; };

__device__ bool wait_flag(const unsigned long long* f, unsigned long long want) {
  for (unsigned spins = 0; spins < (1u << 24); ++spins) {
    if (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= want) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// B: writes round r's payload, fences, raises flag_ba = r; then waits for flag_ab = r
__global__ void k_B(Box* box, int rounds) {
  for (int r = 1; r <= rounds; ++r) {
    for (int i = threadIdx.x; i < PAYLOAD; i += blockDim.x) box->data[i] = r * 1000.0 + i;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_store(&box->flag_ba, (unsigned long long)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (!wait_flag(&box->flag_ab, r)) box->timeout = 1;
    }
    __syncthreads();
  }
}
// A: waits for flag_ba = r, verifies, raises flag_ab = r
__global__ void k_A(Box* box, int rounds) {
  __shared__ int ok;
  for (int r = 1; r <= rounds; ++r) {
    if (threadIdx.x == 0) ok = wait_flag(&box->flag_ba, r);
    __syncthreads();
    if (!ok) { if (threadIdx.x == 0) box->timeout = 2; return; }
    int bad = 0;
    for (int i = threadIdx.x; i < PAYLOAD; i += blockDim.x) {
      const double v = __builtin_nontemporal_load(&box->data[i]);
      if (v != r * 1000.0 + i) ++bad;
    }
    if (bad) atomicAdd(&box->bad, bad);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&box->flag_ab, (unsigned long long)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const bool isA = argv[1][0] == 'A';
  const int rounds = 2000;
  CHECK(hipSetDevice(0));
  Box* box = nullptr;
  if (isA) {
    CHECK(hipExtMallocWithFlags((void**)&box, sizeof(Box), hipDeviceMallocUncached));
    CHECK(hipMemset(box, 0, sizeof(Box)));
    CHECK(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    CHECK(hipIpcGetMemHandle(&h, box));
    FILE* f = fopen(argv[2], "wb"); fwrite(&h, sizeof(h), 1, f); fclose(f);
    char done[512]; snprintf(done, sizeof done, "%s.ready", argv[2]);
    f = fopen(done, "wb"); fclose(f);
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_A, dim3(1), dim3(256), 0, 0, box, rounds);
    CHECK(hipDeviceSynchronize());
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    Box hb; CHECK(hipMemcpy(&hb, box, sizeof(Box), hipMemcpyDeviceToHost));
    printf("A: %d rounds, %.2f us per round trip (includes waiting for B to start), bad=%d timeout=%d\n", rounds,
           us / rounds, hb.bad, hb.timeout);
  } else {
    char done[512]; snprintf(done, sizeof done, "%s.ready", argv[2]);
    for (int i = 0; i < 600 && access(done, F_OK) != 0; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(50));
    hipIpcMemHandle_t h;
    FILE* f = fopen(argv[2], "rb"); if (!f || fread(&h, sizeof(h), 1, f) != 1) { printf("B: no handle\n"); return 1; } fclose(f);
    CHECK(hipIpcOpenMemHandle((void**)&box, h, hipIpcMemLazyEnablePeerAccess));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_B, dim3(1), dim3(256), 0, 0, box, rounds);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("B: %d rounds in %.3f ms -> %.2f us per round trip (24 KB payload one way + flag back)\n", rounds, ms,
           ms * 1000 / rounds);
    CHECK(hipIpcCloseMemHandle(box));
  }
  return 0;
}
