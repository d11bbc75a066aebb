// Feasibility probe for the peer-to-peer reduction transport: two PROCESSES on one device exchange
// hipIpc handles of uncached buffers, then run kernels that push LL words (value half + sequence flag in
// one 8-byte store) into each other's buffer and spin (bounded) on their own.  Prints the round-trip
// time per exchange.  Build: hipcc --offload-arch=gfx950 -O2 -o ipc_probe ipc_probe.hip
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <sys/wait.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s -> %s\n", getpid(), #x, hipGetErrorString(e_)); _exit(3); } } while (0)

__device__ __forceinline__ void ll_store(uint64_t* p, uint32_t data, uint32_t seq) {
  __hip_atomic_store(p, ((uint64_t)seq << 32) | data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool ll_load(const uint64_t* p, uint32_t seq, uint32_t& data) {
  const uint64_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  data = (uint32_t)v;
  return (uint32_t)(v >> 32) == seq;
}

// every thread t < count pushes value[t] to the peer and waits for the peer's value[t]
__global__ void k_xchg(uint64_t* mine, uint64_t* peer, const double* val, double* out, int count, uint32_t seq, int* err) {
  const int t = threadIdx.x;
  if (t >= count) return;
  const uint64_t bits = __double_as_longlong(val[t]);
  uint64_t* dst = peer + (size_t)(seq & 1) * 2 * 64 + 2 * t;
  ll_store(dst, (uint32_t)bits, seq);
  ll_store(dst + 1, (uint32_t)(bits >> 32), seq);
  const uint64_t* src = mine + (size_t)(seq & 1) * 2 * 64 + 2 * t;
  uint32_t lo = 0, hi = 0;
  long spins = 0;
  while (!(ll_load(src, seq, lo) && ll_load(src + 1, seq, hi))) {
    if (++spins > 200000000L) { *err = 1; return; }
    __builtin_amdgcn_s_sleep(1);
  }
  out[t] = val[t] + __longlong_as_double(((uint64_t)hi << 32) | lo);
}

int main() {
  int ab[2], ba[2];
  if (pipe(ab) || pipe(ba)) return 1;
  const pid_t child = fork();
  const int me = child ? 0 : 1;
  const int rd = me == 0 ? ba[0] : ab[0], wr = me == 0 ? ab[1] : ba[1];
  CK(hipSetDevice(0));
  uint64_t* mine = nullptr;
  CK(hipExtMallocWithFlags((void**)&mine, 4096, hipDeviceMallocUncached));
  CK(hipMemset(mine, 0, 4096));
  CK(hipDeviceSynchronize());
  hipIpcMemHandle_t hm, hp;
  CK(hipIpcGetMemHandle(&hm, mine));
  if (write(wr, &hm, sizeof hm) != sizeof hm) return 2;
  if (read(rd, &hp, sizeof hp) != sizeof hp) return 2;
  uint64_t* peer = nullptr;
  CK(hipIpcOpenMemHandle((void**)&peer, hp, hipIpcMemLazyEnablePeerAccess));
  double *val, *out; int* err;
  CK(hipMalloc(&val, 64 * 8)); CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&err, 4));
  CK(hipMemset(err, 0, 4));
  double hv[64]; for (int i = 0; i < 64; ++i) hv[i] = (me + 1) * 1000.0 + i;
  CK(hipMemcpy(val, hv, sizeof hv, hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreate(&s));
  const int rounds = 2000;
  // handshake so both processes start launching at about the same time
  char c = 'x'; if (write(wr, &c, 1) != 1 || read(rd, &c, 1) != 1) return 2;
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 1; r <= rounds; ++r) hipLaunchKernelGGL(k_xchg, dim3(1), dim3(64), 0, s, mine, peer, val, out, 42, (uint32_t)r, err);
  CK(hipStreamSynchronize(s));
  auto t1 = std::chrono::steady_clock::now();
  double ho[64]; int he = 0;
  CK(hipMemcpy(ho, out, sizeof ho, hipMemcpyDeviceToHost)); CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
  const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / rounds;
  printf("[rank %d] err=%d out[0]=%.1f out[41]=%.1f (expect 3000.0 3082.0)  %.2f us per exchange kernel\n", me, he, ho[0], ho[41], us);
  fflush(stdout);
  if (write(wr, &c, 1) != 1 || read(rd, &c, 1) != 1) return 2;   // keep mappings alive until both are done
  CK(hipIpcCloseMemHandle(peer));
  if (me == 0) { int st; waitpid(child, &st, 0); }
  return he;
}
