// Do MFMA and VALU instructions of two wavefronts on one SIMD run side by side on gfx950?  Block = 8 waves (2 per SIMD): waves 0-3 run
// an MFMA loop, waves 4-7 a VALU loop (same SIMDs); compared with each loop alone.  fp32: mfma_f32_4x4x1 + v_pk_fma_f32; fp64:
// mfma_f64_4x4x4 + v_fma_f64.   hipcc --offload-arch=gfx950 -O2 tools/microbench_mfma_overlap.hip -o build/tools/microbench_mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool F64> __global__ void k(double* out, int iters, int mode) {   // mode 1: mfma only, 2: valu only, 3: both
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool do_mfma = (w < 4) && (mode & 1), do_valu = (w >= 4) && (mode & 2);
  double res = 0;
  if (do_mfma) {
    if constexpr (F64) {
      double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3, c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[u], 0, 0, 0);
      for (int u = 0; u < 8; ++u) res += c[u];
    } else {
      float a = 1.0f + l * 1e-3f, b = 1.0f - l * 1e-3f;
      v4f c[8];
      for (int u = 0; u < 8; ++u) c[u] = (v4f){0, 0, 0, 0};
      for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[u], 0, 0, 0);
      for (int u = 0; u < 8; ++u) res += c[u][0] + c[u][3];
    }
  }
  if (do_valu) {
    if constexpr (F64) {
      double a = 1.0 + l * 1e-9, b = 1e-9, c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < 4 * iters; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = __builtin_fma(a, c[u], b);
      for (int u = 0; u < 8; ++u) res += c[u];
    } else {
      f32x2 a = {1.0f + l * 1e-7f, 1.0f}, b = {1e-9f, 1e-9f}, c[8];
      for (int u = 0; u < 8; ++u) c[u] = (f32x2){0, 0};
      for (int i = 0; i < 2 * iters; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = __builtin_elementwise_fma(a, c[u], b);
      for (int u = 0; u < 8; ++u) res += c[u].x + c[u].y;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}
template <bool F64> void run(const char* name) {
  double* d; (void)hipMalloc(&d, (256 * 512) * sizeof(double));
  const int iters = 20000;
  float t[4] = {0, 0, 0, 0};
  for (int mode = 1; mode <= 3; ++mode) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<F64><<<256, 512>>>(d, 10, mode);
    (void)hipEventRecord(e0);
    k<F64><<<256, 512>>>(d, iters, mode);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&t[mode], e0, e1);
  }
  printf("%s: mfma alone %.3f ms, valu alone %.3f ms, both on the same SIMDs %.3f ms  (sum %.3f, max %.3f) -> %s\n", name, t[1], t[2], t[3], t[1] + t[2],
         t[1] > t[2] ? t[1] : t[2], t[3] < 0.75 * (t[1] + t[2]) ? "they overlap" : "they serialise (shared datapath)");
  (void)hipFree(d);
}
int main() { run<false>("fp32 mfma_4x4x1 + v_pk_fma_f32"); run<true>("fp64 mfma_4x4x4 + v_fma_f64"); return 0; }
