// Issue rate of the MFMA shapes considered for the fused gradient kernel (cycles per instruction per SIMD, back to back, 8 independent
// accumulators, 1 and 2 waves per SIMD).  hipcc --offload-arch=gfx950 -O2 tools/microbench_mfma.hip -o build/tools/microbench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef double v4d __attribute__((ext_vector_type(4)));

template <int KIND> __global__ void k(double* out, int iters) {
  const int l = threadIdx.x;
  double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  float af = (float)a, bf = (float)b;
  long long t0 = clock64();
  if constexpr (KIND == 0) {            // f64 4x4x4 4 blocks
    double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[u], 0, 0, 0);
    double s = 0; for (int u = 0; u < 8; ++u) s += c[u];
    out[blockIdx.x * blockDim.x + l] = s;
  } else if constexpr (KIND == 1) {     // f64 16x16x4
    v4d c[8];
    for (int u = 0; u < 8; ++u) c[u] = (v4d){0, 0, 0, 0};
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[u], 0, 0, 0);
    double s = 0; for (int u = 0; u < 8; ++u) s += c[u][0] + c[u][3];
    out[blockIdx.x * blockDim.x + l] = s;
  } else if constexpr (KIND == 2) {     // f32 4x4x1 16 blocks
    v4f c[8];
    for (int u = 0; u < 8; ++u) c[u] = (v4f){0, 0, 0, 0};
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(af, bf, c[u], 0, 0, 0);
    float s = 0; for (int u = 0; u < 8; ++u) s += c[u][0] + c[u][3];
    out[blockIdx.x * blockDim.x + l] = s;
  } else if constexpr (KIND == 3) {     // f32 16x16x1 4 blocks
    v16f c[4];
    for (int u = 0; u < 4; ++u) for (int q = 0; q < 16; ++q) c[u][q] = 0;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 4; ++u) { c[u] = __builtin_amdgcn_mfma_f32_16x16x1f32(af, bf, c[u], 0, 0, 0); c[u] = __builtin_amdgcn_mfma_f32_16x16x1f32(bf, af, c[u], 0, 0, 0); }
    float s = 0; for (int u = 0; u < 4; ++u) s += c[u][0] + c[u][15];
    out[blockIdx.x * blockDim.x + l] = s;
  } else {                              // v_fma_f64 VALU reference
    double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = __builtin_fma(a, b, c[u]);
    double s = 0; for (int u = 0; u < 8; ++u) s += c[u];
    out[blockIdx.x * blockDim.x + l] = s;
  }
  long long t1 = clock64();
  if (l == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0);
}

template <int KIND> void run(const char* name, int fmas_per_instr) {
  double* d; (void)hipMalloc(&d, ((1 << 20) + 8) * sizeof(double));
  const int iters = 20000;
  for (int wps = 1; wps <= 2; ++wps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<KIND><<<256 * 4, 64 * 4 * wps / 4>>>(d, 10);          // warm
    (void)hipEventRecord(e0);
    k<KIND><<<256, 256 * wps>>>(d, iters);                    // 256 blocks = 1 per CU, 4*wps waves each = wps waves per SIMD
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 8 * wps;
    const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
    printf("%-22s waves/SIMD %d: %.3f ms  -> %.1f cycles/instr at 2.4 GHz (%.1f FMA/clk/SIMD), %.1f TFLOP/s chip\n", name, wps, ms, cyc, fmas_per_instr / cyc,
           256.0 * 4 * instr_per_simd * fmas_per_instr * 2 / (ms * 1e-3) / 1e12);
  }
  (void)hipFree(d);
}
int main() {
  run<0>("mfma_f64_4x4x4_4b", 256);
  run<1>("mfma_f64_16x16x4", 1024);
  run<2>("mfma_f32_4x4x1_16b", 256);
  run<3>("mfma_f32_16x16x1_4b", 1024);
  run<4>("v_fma_f64 (64 lanes)", 64);
  return 0;
}
