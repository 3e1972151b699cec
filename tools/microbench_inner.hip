// Inner-loop microbenchmark for the packed-fp32 sky-sum kernel (gfx950): how close to the
// v_pk_fma_f32 issue peak can the rotate+accumulate recurrence get as a function of
//   NCH   independent recurrence chains interleaved in one wave's instruction stream
//   LDS   whether the pbflux operand pair comes from an LDS broadcast read
//   waves per SIMD (grid size / launch bounds)
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_inner.hip -o /tmp/mbi
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pkfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

constexpr int HC = 32;      // pairs per source-tile (CT = 64)
constexpr int NSRC = 512;   // sources per thread

template <int NCH, bool LDS, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void k_inner(float* out, const float* pin, float seed) {
  __shared__ __attribute__((aligned(16))) float lp[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) lp[i] = pin[i];
  __syncthreads();
  f2 acc_re[HC], acc_im[HC];
#pragma unroll
  for (int j = 0; j < HC; ++j) { acc_re[j] = (f2)(0.f); acc_im[j] = (f2)(0.f); }
  const float th = seed * (threadIdx.x + 1);
  for (int s = 0; s < NSRC; s += NCH) {
    f2 zre[NCH], zim[NCH], RR[NCH], RI[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float a = th * (s + c + 1);
      zre[c] = (f2){1.0f - a, 1.0f + a}; zim[c] = (f2){a, -a};
      RR[c] = (f2){1.0f - a * a, 1.0f - a * a}; RI[c] = (f2){-a, a};
    }
#pragma unroll
    for (int j = 0; j < HC; j += 2) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        f2 p0, p1;
        if (LDS) {
          const float4 pv = *reinterpret_cast<const float4*>(&lp[((s + c) & 63) * 64 + 2 * j]);
          p0 = (f2){pv.x, pv.y}; p1 = (f2){pv.z, pv.w};
        } else {
          p0 = (f2){seed, seed * 2}; p1 = (f2){seed * 3, seed * 4};
        }
        acc_re[j] = pkfma(p0, zre[c], acc_re[j]);
        acc_im[j] = pkfma(p0, zim[c], acc_im[j]);
        f2 t0 = zim[c] * RI[c], t1 = zre[c] * RI[c];
        f2 nre = pkfma(zre[c], RR[c], t0), nim = pkfma(zim[c], RR[c], -t1);
        acc_re[j + 1] = pkfma(p1, nre, acc_re[j + 1]);
        acc_im[j + 1] = pkfma(p1, nim, acc_im[j + 1]);
        t0 = nim * RI[c]; t1 = nre * RI[c];
        zre[c] = pkfma(nre, RR[c], t0); zim[c] = pkfma(nim, RR[c], -t1);
      }
    }
  }
  f2 r = (f2)(0.f);
#pragma unroll
  for (int j = 0; j < HC; ++j) r += acc_re[j] + acc_im[j];
  out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
}

// Lifting (3-shear) rotation: x1 = x - t y; y1 = y + s x1; x2 = x1 - t y1  -- 3 dependent FMAs instead of 2 mul + 2 fma
template <int NCH, bool LDS, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void k_inner_lift(float* out, const float* pin, float seed) {
  __shared__ __attribute__((aligned(16))) float lp[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) lp[i] = pin[i];
  __syncthreads();
  f2 acc_re[HC], acc_im[HC];
#pragma unroll
  for (int j = 0; j < HC; ++j) { acc_re[j] = (f2)(0.f); acc_im[j] = (f2)(0.f); }
  const float th = seed * (threadIdx.x + 1);
  for (int s = 0; s < NSRC; s += NCH) {
    f2 zre[NCH], zim[NCH], NT[NCH], SS[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float a = th * (s + c + 1);
      zre[c] = (f2){1.0f - a, 1.0f + a}; zim[c] = (f2){a, -a};
      NT[c] = (f2){-0.5f * a, 0.5f * a}; SS[c] = (f2){a, -a};
    }
#pragma unroll
    for (int j = 0; j < HC; ++j) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        f2 p0;
        if (LDS) {
          const float2 pv = *reinterpret_cast<const float2*>(&lp[((s + c) & 63) * 64 + 2 * j]);
          p0 = (f2){pv.x, pv.y};
        } else {
          p0 = (f2){seed, seed * 2};
        }
        acc_re[j] = pkfma(p0, zre[c], acc_re[j]);
        acc_im[j] = pkfma(p0, zim[c], acc_im[j]);
        const f2 x1 = pkfma(NT[c], zim[c], zre[c]);
        const f2 y1 = pkfma(SS[c], x1, zim[c]);
        zre[c] = pkfma(NT[c], y1, x1);
        zim[c] = y1;
      }
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
  const double terms = (double)blocks * 256 * NSRC * 64;           // 64 channels per source per thread
  printf("%-28s blocks=%5d  %8.3f ms  %7.3f T terms/s  (%.2f T pk-inst/s)\n", name, blocks, best, terms / best * 1e-9,
         terms / best * 1e-9 * 3);
}

int main() {
  float *dout, *dpin; CK(hipMalloc(&dout, 4 * 256 * 8192)); CK(hipMalloc(&dpin, 4 * 64 * 64));
  CK(hipMemset(dpin, 0, 4 * 64 * 64));
  const int cus = 256;
  printf("ideal at 2.4 GHz: 256 CU x 4 SIMD x 64 lanes / (12 cycles per term) = 13.1 T terms/s\n");
  run("1 chain  reg  1 wave/SIMD", k_inner<1, false, 1>, cus * 1 * 8, dout, dpin);
  run("1 chain  reg  2 waves/SIMD", k_inner<1, false, 2>, cus * 2 * 8, dout, dpin);
  run("2 chains reg  1 wave/SIMD", k_inner<2, false, 1>, cus * 1 * 8, dout, dpin);
  run("2 chains reg  2 waves/SIMD", k_inner<2, false, 2>, cus * 2 * 8, dout, dpin);
  run("4 chains reg  2 waves/SIMD", k_inner<4, false, 2>, cus * 2 * 8, dout, dpin);
  run("1 chain  LDS  2 waves/SIMD", k_inner<1, true, 2>, cus * 2 * 8, dout, dpin);
  run("2 chains LDS  1 wave/SIMD", k_inner<2, true, 1>, cus * 1 * 8, dout, dpin);
  run("2 chains LDS  2 waves/SIMD", k_inner<2, true, 2>, cus * 2 * 8, dout, dpin);
  run("4 chains LDS  2 waves/SIMD", k_inner<4, true, 2>, cus * 2 * 8, dout, dpin);
  printf("lifting rotation (5 packed inst per pair of terms; ideal 15.7 T terms/s at 2.4 GHz):\n");
  run("lift 1 chain  reg  2 w/SIMD", k_inner_lift<1, false, 2>, cus * 2 * 8, dout, dpin);
  run("lift 1 chain  LDS  2 w/SIMD", k_inner_lift<1, true, 2>, cus * 2 * 8, dout, dpin);
  run("lift 2 chains LDS  2 w/SIMD", k_inner_lift<2, true, 2>, cus * 2 * 8, dout, dpin);
  run("lift 2 chains LDS  1 w/SIMD", k_inner_lift<2, true, 1>, cus * 1 * 8, dout, dpin);
  return 0;
}
