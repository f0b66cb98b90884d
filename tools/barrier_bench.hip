// barrier_bench.hip -- what does a device-wide barrier of co-resident workgroups cost on this part? (round 6, design input for k_relax_epochs)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/barrier_bench tools/barrier_bench.hip && /tmp/barrier_bench
// Variants: flat (one counter, everybody polls one word), tree (G group counters on separate 256-byte lines, one top counter, one release word
// per group), with thread 0 fencing (release before arrive, acquire after) or without fences (the floor of the atomics alone); a workload line
// per variant where every thread also writes and reads a word between barriers (so that the L2 write-back / invalidate has something to do).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define AG __HIP_MEMORY_SCOPE_AGENT

struct Bar { uint32_t w[64 * 66]; };   // line k at w[64 * k]: 0 = top counter, 1 = flat gen, 2.. = group counters, 34.. = group release words

template <int MODE>   // 0 flat, 1 tree
__device__ __forceinline__ void gsync(Bar* b, uint32_t nblocks, uint32_t groups, uint32_t& gen, bool fence) {
  __syncthreads();
  gen++;
  if (threadIdx.x == 0) {
    if (fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (MODE == 0) {
      if (__hip_atomic_fetch_add(&b->w[0], 1u, __ATOMIC_RELAXED, AG) == nblocks - 1u) {
        __hip_atomic_store(&b->w[0], 0u, __ATOMIC_RELAXED, AG);
        __hip_atomic_store(&b->w[64], gen, __ATOMIC_RELAXED, AG);
      } else while (__hip_atomic_load(&b->w[64], __ATOMIC_RELAXED, AG) != gen) __builtin_amdgcn_s_sleep(1);
    } else {
      const uint32_t g = blockIdx.x % groups, members = (nblocks - g + groups - 1u) / groups;
      uint32_t* cnt = &b->w[64 * (2 + g)];
      uint32_t* rel = &b->w[64 * (34 + g)];
      bool released = false;
      if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, AG) == members - 1u) {
        __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, AG);
        if (__hip_atomic_fetch_add(&b->w[0], 1u, __ATOMIC_RELAXED, AG) == groups - 1u) {
          __hip_atomic_store(&b->w[0], 0u, __ATOMIC_RELAXED, AG);
          for (uint32_t k = 0; k < groups; k++) __hip_atomic_store(&b->w[64 * (34 + k)], gen, __ATOMIC_RELAXED, AG);
          released = true;
        }
      }
      if (!released) while (__hip_atomic_load(rel, __ATOMIC_RELAXED, AG) != gen) __builtin_amdgcn_s_sleep(1);
    }
    if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int MODE>
__global__ void __launch_bounds__(256) k_bar(Bar* b, uint32_t iters, uint32_t groups, uint32_t fence, uint32_t* data, uint32_t work, uint32_t* bad) {
  uint32_t gen = __hip_atomic_load(MODE == 0 ? &b->w[64] : &b->w[64 * (34 + blockIdx.x % groups)], __ATOMIC_RELAXED, AG);
  const uint32_t n = gridDim.x * 256u, me = blockIdx.x * 256u + threadIdx.x;
  uint32_t errs = 0;
  for (uint32_t i = 0; i < iters; i++) {
    if (work) data[me] = i * 7u + me;                         // everybody writes its word ...
    gsync<MODE>(b, gridDim.x, groups, gen, fence != 0);
    if (work) {                                               // ... and reads somebody else's (another workgroup, another XCD) after the barrier
      const uint32_t other = (me + 256u * 37u + 11u) % n;
      if (data[other] != i * 7u + other) errs++;
      gsync<MODE>(b, gridDim.x, groups, gen, fence != 0);
    }
  }
  if (errs) atomicAdd(bad, errs);
}

int main(int argc, char** argv) {
  int dev = 0; CHK(hipSetDevice(dev));
  int ncu = 0; CHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  Bar* b; uint32_t* data; uint32_t* bad;
  CHK(hipMalloc(&b, sizeof(Bar))); CHK(hipMalloc(&data, 1024 * 256 * 4)); CHK(hipMalloc(&bad, 4));
  hipStream_t st; CHK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const uint32_t iters = 2000;
  printf("CUs %d; %u iterations per launch; microseconds per barrier\n", ncu, iters);
  printf("%-6s %-7s %-6s %-6s %-5s %10s %8s\n", "mode", "blocks", "groups", "fence", "work", "us/barrier", "stale");
  for (int work = 0; work <= 1; work++)
    for (int fence = 0; fence <= 1; fence++)
      for (int blocks : {64, 128, 256})
        for (int groups : {0, 8, 16, 32}) {
          if (blocks > ncu) continue;
          if (work && !fence && groups) continue;             // (unfenced + work only once: it shows the stale reads the fences prevent)
          CHK(hipMemsetAsync(b, 0, sizeof(Bar), st)); CHK(hipMemsetAsync(bad, 0, 4, st));
          uint32_t it = iters, g = groups ? groups : 1, f = fence, w = work;
          void* args[] = {&b, &it, &g, &f, &data, &w, &bad};
          const void* fn = groups ? (const void*)k_bar<1> : (const void*)k_bar<0>;
          CHK(hipLaunchCooperativeKernel(fn, dim3(blocks), dim3(256), args, 0, st));   // warm-up
          CHK(hipEventRecord(e0, st));
          CHK(hipLaunchCooperativeKernel(fn, dim3(blocks), dim3(256), args, 0, st));
          CHK(hipEventRecord(e1, st));
          CHK(hipEventSynchronize(e1));
          float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
          uint32_t hb = 0; CHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
          printf("%-6s %-7d %-6d %-6d %-5d %10.2f %8u\n", groups ? "tree" : "flat", blocks, groups, fence, work, 1000.0 * ms / (iters * (work ? 2 : 1)), hb);
          fflush(stdout);
        }
  return 0;
}
