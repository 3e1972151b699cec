// Probe of the lane layouts of v_mfma_f64_4x4x4_4b_f64 and v_mfma_f32_4x4x1_16b_f32 on gfx950 (used by the fused gradient kernels).
// Result on MI355X (profiles/r02_mfma_layout.txt):
//   f64 4x4x4, 4 blocks:  A[i][k] of block b <- lane 16 k + 4 b + i;  B[k][j] <- lane 16 k + 4 b + j;  D[i][j] -> lane 16 i + 4 b + j
//   f32 4x4x1, 16 blocks: A[i] of block b <- lane 4 b + i;  B[j] <- lane 4 b + j;  D[i][j] -> register i of lane 4 b + j
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_layout_test.hip -o build/tools/mfma_layout_test && build/tools/mfma_layout_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float v4f __attribute__((ext_vector_type(4)));

// one-hot probes: for every source lane q, a(q) = 1 and everything else 0 (b = 1 everywhere): D(l) = 1 iff lane q feeds lane l through A
__global__ void k_f64(double* out) {
  const int l = threadIdx.x;
  for (int q = 0; q < 64; ++q) {
    out[q * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(l == q ? 1.0 : 0.0, 1.0, 0.0, 0, 0, 0);
    out[4096 + q * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, l == q ? 1.0 : 0.0, 0.0, 0, 0, 0);
  }
}
__global__ void k_f32(float* out) {
  const int l = threadIdx.x;
  v4f c = {0.f, 0.f, 0.f, 0.f};
  v4f d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), 1.0f, c, 0, 0, 0);      // D[i][j] = A[i]: which lane's a?
  for (int i = 0; i < 4; ++i) out[4 * l + i] = d[i];
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(l + 1), c, 0, 0, 0);          // D[i][j] = B[j]
  for (int i = 0; i < 4; ++i) out[256 + 4 * l + i] = d[i];
}
int main() {
  double* d64; float* d32;
  (void)hipMalloc(&d64, 8192 * 8); (void)hipMalloc(&d32, 512 * 4);
  k_f64<<<1, 64>>>(d64); k_f32<<<1, 64>>>(d32);
  static double h64[8192]; float h32[512];
  (void)hipMemcpy(h64, d64, sizeof(h64), hipMemcpyDeviceToHost); (void)hipMemcpy(h32, d32, sizeof(h32), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int i = l / 16, b = (l % 16) / 4, j = l % 4;
    for (int q = 0; q < 64; ++q) {
      const int k = q / 16;
      const bool expA = (q == 16 * k + 4 * b + i), expB = (q == 16 * k + 4 * b + j);
      if ((h64[q * 64 + l] != 0.0) != expA || (h64[4096 + q * 64 + l] != 0.0) != expB) ++bad;
    }
  }
  printf("f64 4x4x4: D(lane 16 i + 4 b + j) = sum_k A(lane 16 k + 4 b + i) * B(lane 16 k + 4 b + j): %s (%d mismatches)\n", bad ? "NO" : "confirmed", bad);
  bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i)
      if (h32[4 * l + i] - 1 != 4 * (l / 4) + i || h32[256 + 4 * l + i] - 1 != l) ++bad;
  printf("f32 4x4x1: D[i](lane l) = A(lane 4 (l/4) + i) * B(lane l): %s (%d mismatches)\n", bad ? "NO" : "confirmed", bad);
  return 0;
}
