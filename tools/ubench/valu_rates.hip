// valu_rates.hip -- issue rate of the VALU instructions the tap engine is made of (gfx950), measured as
// wave-instructions per SIMD per nanosecond with every SIMD holding 8 waves of independent dependency chains.
// Used to price the tap loop (DESIGN.md "roofline"): the engine is VALU-issue bound, and f64 / conversion
// instructions cost more issue cycles than 32-bit ones.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096
#define CHAINS 8

#define KERNEL(name, decl, body, sink)                                          \
  __global__ __launch_bounds__(256) void name(double *out, int n) {             \
    decl;                                                                       \
    for (int i = 0; i < n; ++i) {                                               \
      _Pragma("unroll") for (int k = 0; k < CHAINS; ++k) { body; }              \
    }                                                                           \
    sink;                                                                       \
  }

#define D8 double a[CHAINS]; double b = (double)threadIdx.x * 1.0000001 + 1.5; for (int k = 0; k < CHAINS; ++k) a[k] = b + k
#define S8 double s = 0; for (int k = 0; k < CHAINS; ++k) s += a[k]; if (s == 123.456) out[threadIdx.x] = s
#define I8 int a[CHAINS]; int b = threadIdx.x * 2654435 + 17; for (int k = 0; k < CHAINS; ++k) a[k] = b + k
#define SI8 int s = 0; for (int k = 0; k < CHAINS; ++k) s += a[k]; if (s == 123456) out[threadIdx.x] = s

KERNEL(k_add_f64, D8, asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b)), S8)
KERNEL(k_mul_f64, D8, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b)), S8)
KERNEL(k_fma_f64, D8, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b)), S8)
KERNEL(k_min_f64, D8, asm volatile("v_min_f64 %0, |%0|, %1" : "+v"(a[k]) : "v"(b)), S8)
KERNEL(k_fract_f64, D8, asm volatile("v_fract_f64 %0, %0" : "+v"(a[k])), S8)
KERNEL(k_trunc_f64, D8, asm volatile("v_trunc_f64 %0, %0" : "+v"(a[k])), S8)
KERNEL(k_cmp_f64, D8, asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a[k]) : "v"(b), "v"(((int *)&a[k])[0]), "v"(1) : "vcc"), S8)
KERNEL(k_cvt_i32_f64, D8, int t; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t) : "v"(a[k])); asm volatile("" : "+v"(a[k]) : "v"(t)), S8)
KERNEL(k_cvt_f64_i32, I8, double t; asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(t) : "v"(a[k])); asm volatile("" : "+v"(a[k]) : "v"(t)), SI8)
KERNEL(k_add_u32, I8, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b)), SI8)
KERNEL(k_sad_u8, I8, asm volatile("v_sad_u8 %0, %0, %1, 0" : "+v"(a[k]) : "v"(b)), SI8)
KERNEL(k_mad_i24, I8, asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b)), SI8)
KERNEL(k_med3_i32, I8, asm volatile("v_med3_i32 %0, %0, %1, 7" : "+v"(a[k]) : "v"(b)), SI8)
KERNEL(k_lshl_add, I8, asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[k]) : "v"(b)), SI8)
KERNEL(k_cndmask, I8, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : ), SI8)
KERNEL(k_cmp_u32, I8, asm volatile("v_cmp_lt_u32 vcc, %0, %1\n" : : "v"(a[k]), "v"(b) : "vcc"), SI8)
KERNEL(k_mov_dpp, I8, asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k])), SI8)
KERNEL(k_cvt_f32_f64, D8, float t; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(t) : "v"(a[k])); asm volatile("" : "+v"(a[k]) : "v"(t)), S8)
KERNEL(k_fma_f32, I8, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b)), SI8)

// LDS: random 8-byte table reads (the weight / colour LUT gathers) and conflict-free 8-byte reads
__global__ __launch_bounds__(256) void k_lds_b64_random(double *out, int n) {
  __shared__ double tab[768];
  for (int i = threadIdx.x; i < 768; i += 256) tab[i] = i;
  __syncthreads();
  unsigned idx = threadIdx.x * 2654435761u;
  double s = 0;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
      idx = idx * 1664525u + 1013904223u;
      s += tab[(idx >> 10) % 768];
    }
  }
  if (s == 123.456) out[threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_lds_b64_linear(double *out, int n) {
  __shared__ double tab[768];
  for (int i = threadIdx.x; i < 768; i += 256) tab[i] = i;
  __syncthreads();
  double s = 0;
  int base = threadIdx.x & 63;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
      double v;
      asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((base + ((i + k) & 63)) * 8));
      s += v;
    }
  }
  if (s == 123.456) out[threadIdx.x] = s;
}

struct K { const char *name; void (*fn)(double *, int); int per_iter; };

int main() {
  double *out;
  hipMalloc(&out, 4096);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 8;  // 8 blocks x 4 waves = 32 waves per CU = 8 per SIMD
  std::vector<K> ks = {
      {"v_add_f64", k_add_f64, CHAINS}, {"v_mul_f64", k_mul_f64, CHAINS}, {"v_fma_f64", k_fma_f64, CHAINS}, {"v_min_f64 |x|", k_min_f64, CHAINS},
      {"v_fract_f64", k_fract_f64, CHAINS}, {"v_trunc_f64", k_trunc_f64, CHAINS}, {"v_cmp_lt_f64 + v_cndmask", k_cmp_f64, 2 * CHAINS},
      {"v_cvt_i32_f64", k_cvt_i32_f64, CHAINS}, {"v_cvt_f64_i32", k_cvt_f64_i32, CHAINS}, {"v_cvt_f32_f64", k_cvt_f32_f64, CHAINS},
      {"v_add_u32", k_add_u32, CHAINS}, {"v_sad_u8", k_sad_u8, CHAINS}, {"v_mad_i32_i24", k_mad_i24, CHAINS}, {"v_med3_i32", k_med3_i32, CHAINS},
      {"v_lshl_add_u32", k_lshl_add, CHAINS}, {"v_cndmask_b32", k_cndmask, CHAINS}, {"v_cmp_lt_u32", k_cmp_u32, CHAINS}, {"v_mov_b32_dpp", k_mov_dpp, CHAINS},
      {"v_fma_f32", k_fma_f32, CHAINS}, {"lds b64 random 768", k_lds_b64_random, CHAINS}, {"lds b64 linear", k_lds_b64_linear, CHAINS},
  };
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("%d CUs, clock %d MHz (max)\n", cus, prop.clockRate / 1000);
  printf("%-28s %12s %14s %16s\n", "instruction", "ms", "winstr/ns/SIMD", "cycles@2.4GHz");
  for (auto &k : ks) {
    hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, ITER);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * ITER * k.per_iter;  // wave-instructions
    const double per_simd_ns = winstr / (cus * 4.0) / (ms * 1e6);
    printf("%-28s %12.3f %14.4f %16.2f\n", k.name, ms, per_simd_ns, 2.4 / per_simd_ns);
  }
  return 0;
}
