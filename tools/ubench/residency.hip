// residency.hip -- how many workgroups of W waves are resident per CU on gfx950, by VGPRs per lane and LDS per workgroup.
// Every workgroup increments a device counter, records the maximum it ever sees, spins ~200 us, decrements.  The raster sweep
// (cspm_chain.h: 5-wave workgroups) stayed at ~2 resident workgroups per CU whatever was launched: which resource caps it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NREG>
__global__ void k_res(unsigned *ctr, unsigned *mx, long long ticks, float *sink) {
  extern __shared__ float smem[];
  float r[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) r[i] = (float)(threadIdx.x + i);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned now = atomicAdd(ctr, 1u) + 1u;
    atomicMax(mx, now);
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = r[i] * 1.0001f + r[(i + 1) % NREG];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) s += r[i];
  if (s == 12345.f) sink[threadIdx.x] = s + smem[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) atomicSub(ctr, 1u);
}

template <int NREG>
static void run(int waves, int lds_bytes, int wg_per_cu) {
  unsigned *d;
  float *sink;
  hipMalloc(&d, 8);
  hipMalloc(&sink, 4096);
  hipMemset(d, 0, 8);
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_res<NREG>));
  if (lds_bytes > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(k_res<NREG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(k_res<NREG>, dim3(256 * wg_per_cu), dim3(waves * 64), lds_bytes, 0, d, d + 1, 20000LL, sink);  // 200 us at 100 MHz
  hipDeviceSynchronize();
  unsigned h[2];
  hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("waves/WG %d  VGPRs %3d  LDS %6d B  launched %4d (%d per CU): max resident %4u = %.2f per CU\n", waves, fa.numRegs, lds_bytes, 256 * wg_per_cu, wg_per_cu, h[1],
         h[1] / 256.0);
  hipFree(d);
  hipFree(sink);
}

int main() {
  for (int waves : {5, 4, 8}) {
    for (int lds : {1024, 20000, 33000}) {
      run<40>(waves, lds, 4);
      run<80>(waves, lds, 4);
      run<100>(waves, lds, 4);
      run<120>(waves, lds, 4);
    }
  }
  return 0;
}
