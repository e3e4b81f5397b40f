// gather_cost.hip -- what a wave64 gather costs the CU's L1 return path on gfx950, by width and by access pattern.
// The raster sweep (cspm_chain.h) is bound by that path: 382 gathers of 12-byte elements per pixel, 63 lanes each, 9 window rows x 7
// columns per instruction.  Every CU runs `waves` waves that issue `n` dependent-free gathers of one kind from an L2-resident array
// with the sweep's address pattern (rows `pitch` elements apart, 7 consecutive elements per row, 5 steps of 7 columns, 4 passes of 9
// rows); reported: nanoseconds and shader cycles per gather instruction and CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int BYTES>
__global__ __launch_bounds__(320) void k_gather(const char *base, int pitch_elems, int rows, int n, unsigned *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane / 7, j = lane - lr * 7;
  // each workgroup walks its own window: a 45-row x 45-column region whose origin moves one column per item, like sweep pixels
  unsigned acc = 0;
  int ox = (blockIdx.x * 37) % (pitch_elems - 64), oy = (blockIdx.x * 11 + wave * 3) % (rows - 48);
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int st = 0; st < 5; ++st) {
        const int row = oy + p * 9 + (lr < 9 ? lr : 0), col = ox + j + 7 * st;
        const char *a = base + ((size_t)row * pitch_elems + col) * BYTES;
        if constexpr (BYTES == 4) acc += *reinterpret_cast<const unsigned *>(a);
        else if constexpr (BYTES == 8) { const uint2 v = *reinterpret_cast<const uint2 *>(a); acc += v.x ^ v.y; }
        else if constexpr (BYTES == 12) {
          typedef unsigned u3 __attribute__((ext_vector_type(3)));
          typedef u3 u3a __attribute__((aligned(4)));
          const u3 v = *reinterpret_cast<const u3a *>(a);
          acc += v.x ^ v.y ^ v.z;
        } else if constexpr (BYTES == 16) { const uint4 v = *reinterpret_cast<const uint4 *>(a); acc += v.x ^ v.y ^ v.z ^ v.w; }
        else {  // BYTES == 816: 8-byte elements, 16-byte loads of two neighbours -- half of the addresses are only 8-byte aligned
          typedef unsigned u4 __attribute__((ext_vector_type(4)));
          typedef u4 u4a __attribute__((aligned(8)));
          const u4 v = *reinterpret_cast<const u4a *>(base + ((size_t)row * pitch_elems + col) * 8);
          acc += v.x ^ v.y ^ v.z ^ v.w;
        }
      }
    }
    ox = (ox + 1) % (pitch_elems - 64);
  }
  if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// The same bytes by LDS-DMA: 9 window rows x 30 contiguous 16-byte pieces (480 B per row) per pass = 270 pieces = 5 instructions of
// 64 lanes (the last one 14 lanes), 4 passes per item -- what staging a pass's rows of ONE view through LDS would issue.
__global__ __launch_bounds__(320) void k_dma(const char *base, int pitch_elems, int rows, int n, unsigned *out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem + (unsigned)wave * 5120u;
  int ox = (blockIdx.x * 37) % (pitch_elems - 64), oy = (blockIdx.x * 11 + wave * 3) % (rows - 48);
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int e = i * 64 + lane, r = e / 30, pc = e - r * 30;
        const unsigned voff = (unsigned)(((size_t)(oy + p * 9 + (r < 9 ? r : 0)) * pitch_elems + ox) * 12 + pc * 16);
        if (e < 270) {
          unsigned keep;
          const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)i * 1024u);
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    ox = (ox + 1) % (pitch_elems - 64);
  }
  if (n < 0) out[threadIdx.x] = smem[threadIdx.x];
}
static void run_dma(const char *d, int pitch, int rows, unsigned *dout, int wg_per_cu) {
  const int ncu = 256, n = 400;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_dma, dim3(ncu * wg_per_cu), dim3(320), 5 * 5120, 0, d, pitch, rows, 20, dout);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k_dma, dim3(ncu * wg_per_cu), dim3(320), 5 * 5120, 0, d, pitch, rows, n, dout);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double per_cu = (double)wg_per_cu * 5 * n * 20;
  const double ns = ms * 1e6 / per_cu;
  printf("%-28s %d workgroups/CU: %7.2f ms, %6.1f ns = %6.1f cycles (2.4 GHz) per DMA instruction and CU, %5.1f B/cycle/CU\n", "LDS-DMA dwordx4, row runs", wg_per_cu, ms, ns,
         ns * 2.4, 270.0 * 16 / 5 / (ns * 2.4));
}

template <int BYTES>
static void run(const char *d, int pitch, int rows, unsigned *dout, int wg_per_cu, const char *name) {
  const int ncu = 256, n = 400;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_gather<BYTES>, dim3(ncu * wg_per_cu), dim3(320), 0, 0, d, pitch, rows, 20, dout);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k_gather<BYTES>, dim3(ncu * wg_per_cu), dim3(320), 0, 0, d, pitch, rows, n, dout);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double gathers_per_cu = (double)wg_per_cu * 5 * n * 20;  // waves per workgroup x items x gathers per item
  const double ns = ms * 1e6 / gathers_per_cu;
  printf("%-28s %d workgroups/CU: %7.2f ms, %6.1f ns = %6.1f cycles (2.4 GHz) per gather and CU, %5.1f B/cycle/CU\n", name, wg_per_cu, ms, ns, ns * 2.4,
         63.0 * (BYTES == 816 ? 16 : BYTES) / (ns * 2.4));
}

int main() {
  const int pitch = 1400, rows = 400;  // a KITTI-size level-0 image: 2.2 - 9 MB, L2 / Infinity-Cache resident
  char *d;
  unsigned *dout;
  hipMalloc(&d, (size_t)pitch * rows * 16 + 4096);
  hipMemset(d, 1, (size_t)pitch * rows * 16 + 4096);
  hipMalloc(&dout, 4096);
  for (int wg : {1, 2, 3}) {
    run<4>(d, pitch, rows, dout, wg, "dword   (4 B elements)");
    run<8>(d, pitch, rows, dout, wg, "dwordx2 (8 B elements)");
    run<12>(d, pitch, rows, dout, wg, "dwordx3 (12 B elements)");
    run<16>(d, pitch, rows, dout, wg, "dwordx4 (16 B elements)");
    run<816>(d, pitch, rows, dout, wg, "dwordx4 of two 8 B elements");
    run_dma(d, pitch, rows, dout, wg);
  }
  return 0;
}
