// handover.hip -- what it costs to hand an 8-byte word from one workgroup to another on gfx950 (8 XCDs, one L2 each), by the scope bits of
// the store and of the polling load, for two workgroups on the SAME XCD and on DIFFERENT XCDs.  The raster sweep (cspm_chain.h) hands every
// pixel's plane to its successors this way, 1 616 times in a row: on its critical path a hand-over costs ~5 us with agent-scope (sc1)
// accesses (profiles/r05_sweep_critical_path.txt).  Ping-pong between two workgroups, `n` round trips; one-way latency = time / (2 n).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int LD, int ST>  // 0: no scope bits, 1: sc0, 2: sc1, 3: sc0 sc1
__device__ __forceinline__ void st_u64(unsigned long long *p, unsigned long long v) {
  if (ST == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (ST == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  if (ST == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (ST == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int LD, int ST>
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long *p) {
  unsigned long long v;
  if (LD == 0) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// LD == 4: the poll is a SCALAR load with glc (misses the scalar cache, served by the XCD's L2): it does not queue behind the CU's vector
// memory instructions -- but it is only coherent with stores that went through the same L2, i.e. from the same XCD
template <>
__device__ __forceinline__ unsigned long long ld_u64<4, 2>(const unsigned long long *p) {
  unsigned long long v;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)p)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)p >> 32));
  const unsigned long long q = (unsigned long long)lo | ((unsigned long long)hi << 32);
  asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(q) : "memory");
  return v;
}
template <>
__device__ __forceinline__ void st_u64<4, 2>(unsigned long long *p, unsigned long long v) { st_u64<2, 2>(p, v); }

// block `a` and block `b` play; everybody else leaves.  flags[0] is written by a, flags[16] by b (separate cache lines).
// `load` > 0: waves 1.. of the two playing workgroups keep the CU's vector memory path busy with 63-lane gathers of 12-byte elements
// (the sweep's own traffic) until the game is over.
template <int LD, int ST>
__global__ void k_pingpong(unsigned long long *flags, int a, int b, int n, long long *out, unsigned *xcc_out, const char *img, unsigned *sink) {
  if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
  const bool first = (int)blockIdx.x == a;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) xcc_out[first ? 0 : 1] = xcc & 0xF;
  if (threadIdx.x >= 64) {  // the hammering waves
    typedef unsigned u3 __attribute__((ext_vector_type(3)));
    typedef u3 u3a __attribute__((aligned(4)));
    const int lane = threadIdx.x & 63, lr = lane / 7, j = lane - lr * 7;
    unsigned acc = 0;
    int ox = (threadIdx.x >> 6) * 97;
    while (ld_u64<2, 2>(flags + 32) == 0ull) {
#pragma unroll
      for (int st = 0; st < 5; ++st) {
        const u3 v = *reinterpret_cast<const u3a *>(img + ((size_t)(lr * 1400 + ox + j + 7 * st)) * 12);
        acc += v.x ^ v.y ^ v.z;
      }
      ox = (ox + 1) % 1200;
    }
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
    return;
  }
  if (threadIdx.x != 0) return;
  unsigned long long *mine = flags + (first ? 0 : 16), *theirs = flags + (first ? 16 : 0);
  const long long t0 = wall_clock64();
  long long spins = 0;
  for (int i = 1; i <= n; ++i) {
    if (first) st_u64<LD, ST>(mine, (unsigned long long)i);
    while (ld_u64<LD, ST>(theirs) < (unsigned long long)i) {
      if (++spins > 50000000LL) { out[2] = -1; return; }  // the store never became visible to this kind of load
    }
    if (!first) st_u64<LD, ST>(mine, (unsigned long long)i);
  }
  if (first) { out[0] = wall_clock64() - t0; out[1] = spins; }
  st_u64<2, 2>(flags + 32, 1ull);  // game over: the hammering waves leave
}

template <int LD, int ST>
static void run(const char *name, int a, int b, int load_waves = 0) {
  static char *img = nullptr;
  static unsigned *sink = nullptr;
  if (!img) { hipMalloc(&img, 1400 * 64 * 12 + 4096); hipMemset(img, 1, 1400 * 64 * 12 + 4096); hipMalloc(&sink, 4096); }
  unsigned long long *flags;
  long long *out;
  unsigned *xcc;
  hipMalloc(&flags, 4096);
  hipMalloc(&out, 64);
  hipMalloc(&xcc, 64);
  hipMemset(flags, 0, 4096);
  hipMemset(out, 0, 64);
  const int n = 2000;
  hipLaunchKernelGGL((k_pingpong<LD, ST>), dim3(64), dim3(64 * (1 + load_waves)), 0, 0, flags, a, b, n, out, xcc, img, sink);
  hipDeviceSynchronize();
  long long h[3];
  unsigned hx[2];
  hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  hipMemcpy(hx, xcc, sizeof hx, hipMemcpyDeviceToHost);
  if (load_waves) printf("[%d gather waves beside] ", load_waves);
  if (h[2] < 0) printf("%-34s blocks %2d (XCC %u) <-> %2d (XCC %u): NEVER VISIBLE (timed out)\n", name, a, hx[0], b, hx[1]);
  else printf("%-34s blocks %2d (XCC %u) <-> %2d (XCC %u): one way %7.0f ns  (%.1f polls per hand-over)\n", name, a, hx[0], b, hx[1], h[0] * 10.0 / (2.0 * n),
              (double)h[1] / n);
  hipFree(flags); hipFree(out); hipFree(xcc);
}

int main() {
  for (int pass = 0; pass < 2; ++pass) {
    const int a = 0, b = pass == 0 ? 8 : 1;  // workgroups are dealt round-robin to the XCDs: 0 and 8 share one, 0 and 1 do not
    printf("%s\n", pass == 0 ? "-- same XCD" : "-- different XCDs");
    run<2, 2>("store sc1, load sc1 (agent scope)", a, b);
    run<3, 3>("store sc0 sc1, load sc0 sc1 (system)", a, b);
    run<1, 2>("store sc1, load sc0", a, b);
    run<1, 1>("store sc0, load sc0", a, b);
    run<1, 3>("store sc0 sc1, load sc0", a, b);
    run<0, 2>("store sc1, load plain", a, b);
    run<4, 2>("store sc1, SCALAR load glc", a, b);
    for (int lw : {4, 9}) {
      run<2, 2>("store sc1, load sc1 (agent scope)", a, b, lw);
      run<4, 2>("store sc1, SCALAR load glc", a, b, lw);
    }
  }
  return 0;
}
