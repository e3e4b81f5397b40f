// valu_issue.hip -- how many shader cycles one wave64 VALU instruction occupies a gfx950 SIMD, by instruction class and by
// waves per SIMD.  Straight-line blocks of 128 independent instructions (16 registers, dependency distance 16) so that loop
// overhead is < 3 %; the shader clock is read with s_memtime next to the 100 MHz s_memrealtime, so the result is in real
// shader cycles whatever DVFS does.  This is the peak the tap engine's VALU-issue roofline is priced against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define BLOCK128(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X)

struct Stamp { unsigned long long clk, real; };

#define KERNEL_D(name, INS)                                                                    \
  __global__ __launch_bounds__(1024) void name(Stamp *st, double *out, int n) {                 \
    double a[16];                                                                              \
    double b = (double)threadIdx.x * 1.0000001 + 1.5;                                          \
    for (int k = 0; k < 16; ++k) a[k] = b + k;                                                 \
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();                 \
    for (int i = 0; i < n; ++i) { BLOCK128(INS) }                                              \
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();                 \
    double s = 0;                                                                              \
    for (int k = 0; k < 16; ++k) s += a[k];                                                    \
    if (s == 123.456) out[threadIdx.x] = s;                                                    \
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c1 - c0, r1 - r0};                            \
  }
#define KERNEL_I(name, INS)                                                                    \
  __global__ __launch_bounds__(1024) void name(Stamp *st, double *out, int n) {                 \
    int a[16];                                                                                 \
    int b = threadIdx.x * 2654435 + 17;                                                        \
    for (int k = 0; k < 16; ++k) a[k] = b + k;                                                 \
    unsigned long long msk = __builtin_amdgcn_ballot_w64((threadIdx.x & 3) != 0);              \
    asm volatile("s_mov_b64 vcc, %0" : : "s"(msk) : "vcc");                                    \
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();                 \
    for (int i = 0; i < n; ++i) { BLOCK128(INS) }                                              \
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();                 \
    int s = 0;                                                                                 \
    for (int k = 0; k < 16; ++k) s += a[k];                                                    \
    if (s == 123456) out[threadIdx.x] = s;                                                     \
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c1 - c0, r1 - r0};                            \
  }

#define I_ADD_F64(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_MUL_F64(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_FMA_F64(k) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b));
#define I_MIN_F64(k) asm volatile("v_min_f64 %0, |%0|, %1" : "+v"(a[k]) : "v"(b));
#define I_FRACT_F64(k) asm volatile("v_fract_f64 %0, %0" : "+v"(a[k]));
#define I_CVT_I32_F64(k) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(((int *)&a[k])[0]) : "v"(a[k]));
#define I_CVT_F64_I32(k) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[k]) : "v"(((int *)&a[k])[0]));
#define I_ADD_U32(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_SAD_U8(k) asm volatile("v_sad_u8 %0, %0, %1, 0" : "+v"(a[k]) : "v"(b));
#define I_MAD_I24(k) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b));
#define I_MED3(k) asm volatile("v_med3_i32 %0, %0, %1, 7" : "+v"(a[k]) : "v"(b));
#define I_LSHL(k) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[k]));
#define I_CNDMASK(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b));
// variants of the select, to explain the 22.7 cycles of the row above: mask in an SGPR pair written before the loop (VOP3 form);
// VCC written once before the loop; destination different from the sources
#define I_CNDMASK_SGPR(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "s"(msk));
#define I_CNDMASK_VCCSET(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b), "s"(msk));
#define I_CNDMASK_DST(k) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[k]) : "v"(a[((k) + 5) & 15]), "v"(b));
#define I_CMP_U32(k) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
#define I_CMP_F64(k) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[(k) & 7]), "v"(b) : "vcc");
#define I_FMA_F32(k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b));
#define I_PK_FMA_F32(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b));
#define I_MOV_DPP(k) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
// the shape of the tap loop: 2 of 3 instructions 32-bit, 1 of 3 f64
#define I_MIX(k) asm volatile("v_add_f64 %0, %0, %2\n v_sad_u8 %1, %1, %1, 0\n v_mad_i32_i24 %1, %1, %1, %1" : "+v"(a[k]), "+v"(((int *)&a[(k + 8) & 15])[1]) : "v"(b));

KERNEL_D(k_add_f64, I_ADD_F64)
KERNEL_D(k_mul_f64, I_MUL_F64)
KERNEL_D(k_fma_f64, I_FMA_F64)
KERNEL_D(k_min_f64, I_MIN_F64)
KERNEL_D(k_fract_f64, I_FRACT_F64)
KERNEL_D(k_cvt_i32_f64, I_CVT_I32_F64)
KERNEL_D(k_cvt_f64_i32, I_CVT_F64_I32)
KERNEL_D(k_cmp_f64, I_CMP_F64)
KERNEL_D(k_pk_fma_f32, I_PK_FMA_F32)
KERNEL_I(k_add_u32, I_ADD_U32)
KERNEL_I(k_sad_u8, I_SAD_U8)
KERNEL_I(k_mad_i24, I_MAD_I24)
KERNEL_I(k_med3, I_MED3)
KERNEL_I(k_lshl, I_LSHL)
KERNEL_I(k_cndmask, I_CNDMASK)
KERNEL_I(k_cndmask_sgpr, I_CNDMASK_SGPR)
KERNEL_I(k_cndmask_vccset, I_CNDMASK_VCCSET)
KERNEL_I(k_cndmask_dst, I_CNDMASK_DST)
KERNEL_I(k_cmp_u32, I_CMP_U32)
KERNEL_I(k_fma_f32, I_FMA_F32)
KERNEL_I(k_mov_dpp, I_MOV_DPP)

struct K { const char *name; void (*fn)(Stamp *, double *, int); int per; };

int main(int argc, char **argv) {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  double *out;
  Stamp *st;
  (void)hipMalloc(&out, 4096);
  (void)hipMalloc(&st, sizeof(Stamp) * cus * 8);
  std::vector<K> ks = {{"v_add_f64", k_add_f64, 128}, {"v_mul_f64", k_mul_f64, 128}, {"v_fma_f64", k_fma_f64, 128}, {"v_min_f64", k_min_f64, 128},
                       {"v_fract_f64", k_fract_f64, 128}, {"v_cvt_i32_f64", k_cvt_i32_f64, 128}, {"v_cvt_f64_i32", k_cvt_f64_i32, 128},
                       {"v_cmp_lt_f64", k_cmp_f64, 128}, {"v_pk_fma_f32", k_pk_fma_f32, 128}, {"v_add_u32", k_add_u32, 128}, {"v_sad_u8", k_sad_u8, 128},
                       {"v_mad_i32_i24", k_mad_i24, 128}, {"v_med3_i32", k_med3, 128}, {"v_lshlrev_b32", k_lshl, 128}, {"v_cndmask_b32", k_cndmask, 128},
                       {"cndmask sgpr", k_cndmask_sgpr, 128}, {"cndmask vcc set", k_cndmask_vccset, 128}, {"cndmask dst!=src", k_cndmask_dst, 128},
                       {"v_cmp_lt_u32", k_cmp_u32, 128}, {"v_fma_f32", k_fma_f32, 128}, {"v_mov_b32_dpp", k_mov_dpp, 128}};
  const int n = 2048;
  printf("%d CUs.  One workgroup of 256*w threads per CU = w co-resident waves per SIMD (w = 8: two workgroups of 1024).\n"
         "cycles = shader cycles per wave-instruction per SIMD = kernel wall time x shader clock / (instructions per SIMD); the clock is\n"
         "s_memtime / s_memrealtime measured inside the kernel.\n", cus);
  printf("%-16s", "waves/SIMD:");
  for (int w : {1, 2, 4, 8}) printf(" %8d", w);
  printf("   clock(GHz)\n");
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (auto &k : ks) {
    printf("%-16s", k.name);
    double ghz = 0;
    for (int w : {1, 2, 4, 8}) {
      const int blocks = w == 8 ? 2 * cus : cus, threads = w == 8 ? 1024 : 256 * w;
      hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(threads), 0, 0, st, out, 8);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(threads), 0, 0, st, out, n);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<Stamp> h(blocks);
      (void)hipMemcpy(h.data(), st, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost);
      double clk = 0, real = 0;
      for (auto &s : h) { clk += (double)s.clk; real += (double)s.real; }
      clk /= blocks; real /= blocks;
      ghz = clk / (real * 10.0);  // s_memrealtime ticks at 100 MHz
      const double instr_per_simd = (double)w * n * k.per;
      printf(" %8.2f", ms * 1e6 * ghz / instr_per_simd);
    }
    printf("   %.2f\n", ghz);
  }
  return 0;
}
