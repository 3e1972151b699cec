// Cost and safety of "the last split to arrive sums the partial cubes" on MI355X (8 XCDs, one L2 each, not coherent with each other).
// Investigation for DESIGN 7 item 2 -- NOT product code.  Round 4 built the arrival reduction into the sky-sum kernels with device-scope
// release / acquire fences (bit-identical, but every release is an L2 write-back: 2.6x slower on config 2) and reverted it.  This
// microbenchmark isolates the protocol: G groups of S wavefronts, every wavefront writes a partial of 64 x K complex128, counts itself
// in, and the last one of a group sums the S partials in order.  Variants:
//   0  plain stores, __builtin_amdgcn_fence(release / acquire, "agent")                    (the form that was built)
//   1  partials written and read with agent-scope relaxed atomics (sc1: past the L2), a wavefront-level wait for the stores, the
//      counter an agent-scope relaxed atomic                                                 (no cache maintenance at all)
//   2  no arrival: a second launch sums the partials                                         (what ships: k_reduce_partials)
// Every variant's result is compared with the exact expected sums (small integers in doubles); prints time per launch and mismatches.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_arrive.hip -o gpurun_out/microbench_arrive
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// group g, split s, element e (0 .. 64 K - 1): value re = (s + 1) * (e % 97 + 1) + g % 5 + round, im = -(re)
__device__ __forceinline__ double val(int g, int s, int e, int round) { return (double)((s + 1) * (e % 97 + 1) + g % 5 + round); }

template <int VARIANT>
__global__ __launch_bounds__(256) void k_arrive(double2* __restrict__ part, double2* __restrict__ out, uint32_t* __restrict__ ctr, int G, int S, int K,
                                                int round, int spin) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= G * S) return;
  const int g = item % G, s = item / G;           // consecutive wavefronts belong to different groups: a group's splits sit on different CUs / XCDs
  const size_t n = (size_t)64 * K;                // complex elements per partial
  double2* mine = part + ((size_t)s * G + g) * n;
  // some arithmetic first so that the wavefronts do not all arrive at once
  double acc = 0.0;
  for (int i = 0; i < spin * (1 + (item % 3)); ++i) acc = __builtin_fma(acc, 0.999, 1.0 / (double)(i + 1));
  const double eps = acc > 1e300 ? 1.0 : 0.0;     // always 0: keeps the loop alive
  for (int e = lane; e < (int)n; e += 64) {
    const double v = val(g, s, e, round) + eps;
    if (VARIANT == 1) {
      __hip_atomic_store(&mine[e].x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&mine[e].y, -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      mine[e] = make_double2(v, -v);
    }
  }
  if (VARIANT == 2) return;
  if (VARIANT == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"), __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): the sc1 stores have been acknowledged
  uint32_t old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(&ctr[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
  if (old != (uint32_t)(S - 1)) return;
  if (VARIANT == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (lane == 0) __hip_atomic_store(&ctr[g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int e = lane; e < (int)n; e += 64) {
    double ar = 0.0, ai = 0.0;
    for (int sp = 0; sp < S; ++sp) {
      const double2* p = part + ((size_t)sp * G + g) * n + e;
      if (VARIANT == 1) {
        ar += __hip_atomic_load(&p->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ai += __hip_atomic_load(&p->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        const double2 v = *p; ar += v.x; ai += v.y;
      }
    }
    out[(size_t)g * n + e] = make_double2(ar, ai);
  }
}

__global__ void k_reduce(const double2* __restrict__ part, double2* __restrict__ out, size_t total, int S) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    double ar = 0.0, ai = 0.0;
    for (int sp = 0; sp < S; ++sp) { const double2 v = part[(size_t)sp * total + i]; ar += v.x; ai += v.y; }
    out[i] = make_double2(ar, ai);
  }
}

template <int VARIANT>
static void run(const char* name, int G, int S, int K, int spin, int reps) {
  const size_t n = (size_t)64 * K, total = (size_t)G * n;
  double2 *part, *out; uint32_t* ctr;
  CK(hipMalloc(&part, total * S * sizeof(double2))); CK(hipMalloc(&out, total * sizeof(double2))); CK(hipMalloc(&ctr, G * sizeof(uint32_t)));
  CK(hipMemset(ctr, 0, G * sizeof(uint32_t)));
  std::vector<double2> h(total);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = (G * S + 3) / 4;
  long bad_launches = 0, bad_elems = 0;
  float ms_sum = 0.f;
  for (int r = 0; r < reps; ++r) {
    CK(hipMemsetAsync(out, 0xff, total * sizeof(double2), 0));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_arrive<VARIANT>, dim3(blocks), dim3(256), 0, 0, part, out, ctr, G, S, K, r, spin);
    if (VARIANT == 2) hipLaunchKernelGGL(k_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, part, out, total, S);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) ms_sum += ms;
    CK(hipMemcpy(h.data(), out, total * sizeof(double2), hipMemcpyDeviceToHost));
    long bad = 0;
    for (int g = 0; g < G; ++g)
      for (size_t e = 0; e < n; ++e) {
        double want = 0.0;
        for (int s = 0; s < S; ++s) want += (double)((s + 1) * ((int)(e % 97) + 1) + g % 5 + r);
        const double2 v = h[(size_t)g * n + e];
        if (!(v.x == want && v.y == -want)) ++bad;
      }
    if (bad) { ++bad_launches; bad_elems += bad; }
  }
  printf("{\"variant\": \"%s\", \"groups\": %d, \"splits\": %d, \"wavefronts\": %d, \"KiB_per_wavefront\": %.1f, \"spin\": %d, \"launches\": %d, "
         "\"us_per_launch\": %.2f, \"launches_with_wrong_sums\": %ld, \"wrong_elements\": %ld}\n",
         name, G, S, G * S, n * 16 / 1024.0, spin, reps, ms_sum / (reps - 1) * 1e3, bad_launches, bad_elems);
  fflush(stdout);
  CK(hipFree(part)); CK(hipFree(out)); CK(hipFree(ctr));
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  // config 2: 16 tiles x 3 baseline waves = 48 groups of 42 splits, 64 x 16 complex128 per wavefront
  // one rank's share of the headline at N = 8: 16 tiles x 120 baseline waves = 1920 groups of 8 splits, 64 x 64 per wavefront
  const int cases[2][3] = {{48, 42, 16}, {1920, 8, 64}};
  for (int c = 0; c < 2; ++c)
    for (int spin : {0, 2000}) {
      const int r = c == 0 ? reps : (reps / 8 > 12 ? reps / 8 : 12);      // (the large case is checked on the host element by element)
      run<2>("separate reduce launch", cases[c][0], cases[c][1], cases[c][2], spin, r);
      run<0>("agent-scope fences", cases[c][0], cases[c][1], cases[c][2], spin, r);
      run<1>("sc1 atomics, no cache maintenance", cases[c][0], cases[c][1], cases[c][2], spin, r);
    }
  return 0;
}
