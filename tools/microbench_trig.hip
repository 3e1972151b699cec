// Accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) on gfx950 against fp64 libm, and the issue rate of
// v_pk_fma_f32 with an SGPR-pair operand.  Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_trig.hip -o /tmp/mbt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void k_trig_err(double* maxerr, int n) {
  // y in [-0.5, 0.5] revolutions
  double es = 0, ec = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float y = (float)(((double)i + 0.5) / n - 0.5);
    const float s = __builtin_amdgcn_sinf(y), c = __builtin_amdgcn_cosf(y);
    const double rs = sinpi(2.0 * (double)y), rc = cospi(2.0 * (double)y);
    es = fmax(es, fabs((double)s - rs));
    ec = fmax(ec, fabs((double)c - rc));
  }
  for (int o = 32; o; o >>= 1) { es = fmax(es, __shfl_xor(es, o)); ec = fmax(ec, __shfl_xor(ec, o)); }
  if ((threadIdx.x & 63) == 0) {
    // atomic max on non-negative doubles via their integer pattern
    atomicMax(reinterpret_cast<unsigned long long*>(&maxerr[0]), (unsigned long long)__double_as_longlong(es));
    atomicMax(reinterpret_cast<unsigned long long*>(&maxerr[1]), (unsigned long long)__double_as_longlong(ec));
  }
}

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pkfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// lifting inner loop with the pbflux pair taken from scalar loads (uniform address -> s_load, SGPR-pair operand)
template <bool SGPR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_lift(float* out, const float* __restrict__ pin, float seed) {
  constexpr int HC = 32, NSRC = 512;
  __shared__ __attribute__((aligned(16))) float lp[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) lp[i] = pin[i];
  __syncthreads();
  f2 acc_re[HC], acc_im[HC];
#pragma unroll
  for (int j = 0; j < HC; ++j) { acc_re[j] = (f2)(0.f); acc_im[j] = (f2)(0.f); }
  const float th = seed * (threadIdx.x + 1);
  for (int s = 0; s < NSRC; ++s) {
    const float a = th * (s + 1);
    f2 zre = {1.0f - a, 1.0f + a}, zim = {a, -a}, NT = {-0.5f * a, 0.5f * a}, SS = {a, -a};
    const float* row = pin + (size_t)(s & 63) * 64;      // uniform
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      f2 p0;
      if (SGPR) { p0 = (f2){row[2 * j], row[2 * j + 1]}; }
      else { const float2 pv = *reinterpret_cast<const float2*>(&lp[(s & 63) * 64 + 2 * j]); p0 = (f2){pv.x, pv.y}; }
      acc_re[j] = pkfma(p0, zre, acc_re[j]);
      acc_im[j] = pkfma(p0, zim, acc_im[j]);
      const f2 x1 = pkfma(NT, zim, zre);
      const f2 y1 = pkfma(SS, x1, zim);
      zre = pkfma(NT, y1, x1);
      zim = y1;
    }
  }
  f2 r = (f2)(0.f);
#pragma unroll
  for (int j = 0; j < HC; ++j) r += acc_re[j] + acc_im[j];
  out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
}

template <typename K>
static void run(const char* name, K kern, int blocks, float* dout, const float* dpin) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dout, dpin, 1e-7f);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dout, dpin, 1e-7f);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double terms = (double)blocks * 256 * 512 * 64;
  printf("%-28s blocks=%5d  %8.3f ms  %7.3f T terms/s\n", name, blocks, best, terms / best * 1e-9);
}

int main() {
  double* derr; CK(hipMalloc(&derr, 16)); CK(hipMemset(derr, 0, 16));
  hipLaunchKernelGGL(k_trig_err, dim3(1024), dim3(256), 0, 0, derr, 1 << 28);
  double herr[2]; CK(hipMemcpy(herr, derr, 16, hipMemcpyDeviceToHost));
  printf("v_sin_f32 max abs err on [-0.5,0.5] rev: %.3e   v_cos_f32: %.3e\n", herr[0], herr[1]);
  float *dout, *dpin; CK(hipMalloc(&dout, 4 * 256 * 8192)); CK(hipMalloc(&dpin, 4 * 64 * 64));
  CK(hipMemset(dpin, 0, 4 * 64 * 64));
  run("lift LDS  2 w/SIMD", k_lift<false>, 4096, dout, dpin);
  run("lift SGPR 2 w/SIMD", k_lift<true>, 4096, dout, dpin);
  return 0;
}
