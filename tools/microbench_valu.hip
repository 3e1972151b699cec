// VALU issue-rate microbenchmark for gfx950 (MI355X).
// Purpose: re-derive on the GPU box the peak VALU slot rates that the sky-sum
// roofline is priced against, and decide whether
// packed fp32 (v_pk_fma_f32) buys anything over scalar v_fma_f32 on CDNA4.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_valu.hip -o gpurun_out/microbench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int ITERS = 4096;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

__global__ void k_fma_f32(float* out, float b, float c) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    REP8(OP) REP8(OP)
#undef OP
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_fma_f32_sgpr(float* out, float b, float c) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "s"(b), "v"(c));
    REP8(OP) REP8(OP)
#undef OP
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_pk_fma_f32(float* out, float b, float c) {
  f32x2 a[8]; f32x2 bb = {b, b}, cc = {c, c};
  for (int i = 0; i < 8; ++i) { a[i].x = threadIdx.x * 1e-3f + i; a[i].y = a[i].x + 0.5f; }
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(bb), "v"(cc));
    REP8(OP) REP8(OP)
#undef OP
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_pk_mul_f32(float* out, float b, float c) {
  f32x2 a[8]; f32x2 bb = {b, b};
  for (int i = 0; i < 8; ++i) { a[i].x = threadIdx.x * 1e-3f + i; a[i].y = a[i].x + 0.5f; }
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(bb));
    REP8(OP) REP8(OP)
#undef OP
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_fma_f64(float* out, float bf, float cf) {
  double a[8]; double b = bf, c = cf;
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    REP8(OP) REP8(OP)
#undef OP
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}

__global__ void k_mul_f64(float* out, float bf, float cf) {
  double a[8]; double b = bf;
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    REP8(OP) REP8(OP)
#undef OP
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}

__global__ void k_add_f64(float* out, float bf, float cf) {
  double a[8]; double b = bf;
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    REP8(OP) REP8(OP)
#undef OP
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}

__global__ void k_sin_f32(float* out, float b, float c) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
    REP8(OP) REP8(OP)
#undef OP
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mul_f32(float* out, float b, float c) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITERS; ++it) {
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    REP8(OP) REP8(OP)
#undef OP
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// fma_f32 interleaved with broadcast LDS reads (all lanes same address), 1 ds_read_b128 per 16 fma
__global__ void k_fma_f32_lds(float* out, float b, float c) {
  __shared__ float4 tab[256];
  tab[threadIdx.x] = make_float4(b, c, b, c);
  __syncthreads();
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  float4 acc4 = make_float4(0, 0, 0, 0);
  for (int it = 0; it < ITERS; ++it) {
    float4 v = tab[it & 255];
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    REP8(OP) REP8(OP)
#undef OP
    acc4.x += v.x; acc4.y += v.y;
  }
  float s = acc4.x + acc4.y; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int blocks, int threads, double ops_per_thread_iter, double flops_per_op, float* dout) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dout, 1.0000001f, 1e-9f);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dout, 1.0000001f, 1e-9f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double insts = (double)blocks * threads * ITERS * ops_per_thread_iter;   // lane-instructions
  double rate = insts / (best * 1e-3);                                     // lane-inst/s
  printf("%-18s blocks=%5d thr=%4d  %8.3f ms  %8.2f T lane-inst/s  %8.2f TFLOP/s\n",
         name, blocks, threads, best, rate * 1e-12, rate * flops_per_op * 1e-12);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d  clock=%d kHz  arch=%s\n", p.name, p.multiProcessorCount, p.clockRate, p.gcnArchName);
  int cus = p.multiProcessorCount;
  float* dout; CK(hipMalloc(&dout, sizeof(float) * cus * 8 * 1024));
  for (int wps : {1, 2, 4, 8}) {   // waves per SIMD: blocks of 256 thr = 4 waves = 1 wave/SIMD per block/CU
    int blocks = cus * wps;
    printf("--- %d wave(s) per SIMD ---\n", wps);
    run("v_fma_f32", k_fma_f32, blocks, 256, 16, 2, dout);
    run("v_fma_f32(sgpr)", k_fma_f32_sgpr, blocks, 256, 16, 2, dout);
    run("v_mul_f32", k_mul_f32, blocks, 256, 16, 1, dout);
    run("v_pk_fma_f32", k_pk_fma_f32, blocks, 256, 16, 4, dout);
    run("v_pk_mul_f32", k_pk_mul_f32, blocks, 256, 16, 2, dout);
    run("v_fma_f64", k_fma_f64, blocks, 256, 16, 2, dout);
    run("v_mul_f64", k_mul_f64, blocks, 256, 16, 1, dout);
    run("v_add_f64", k_add_f64, blocks, 256, 16, 1, dout);
    run("v_sin_f32", k_sin_f32, blocks, 256, 16, 1, dout);
    run("v_fma_f32+lds", k_fma_f32_lds, blocks, 256, 16, 2, dout);
  }
  CK(hipFree(dout));
  return 0;
}
