// Micro-benchmark (GPU box): where does the dispatcher put the workgroups of an under-filled grid?
// Launches `nwg` workgroups of 256 threads with `lds` bytes of dynamic LDS; every workgroup spins for ~`us` microseconds and records
// (XCC id, SE/CU id from HW_ID, start and end of s_memrealtime).  Prints the histogram "workgroups per CU" and the launch span.
// build: hipcc -O3 --offload-arch=gfx950 scripts/micro/wg_census.hip -o scripts/micro/wg_census ; run: wg_census nwg lds_bytes us [threads]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

struct Rec { unsigned xcc, hwid; unsigned long long t0, t1; };

__global__ void census(Rec* out, unsigned long long ticks) {
  extern __shared__ float lds[];
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  lds[threadIdx.x] = (float)t0;
  unsigned long long t1;
  do { t1 = __builtin_amdgcn_s_memrealtime(); } while (t1 - t0 < ticks);
  if (threadIdx.x == 0) { out[blockIdx.x].xcc = xcc & 0xf; out[blockIdx.x].hwid = hwid; out[blockIdx.x].t0 = t0; out[blockIdx.x].t1 = t1; }
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 480;
  const int lds = argc > 2 ? atoi(argv[2]) : 50 * 1024;
  const int us = argc > 3 ? atoi(argv[3]) : 100;
  const int threads = argc > 4 ? atoi(argv[4]) : 256;
  Rec* d; hipMalloc(&d, sizeof(Rec) * nwg);
  hipFuncSetAttribute(reinterpret_cast<const void*>(census), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(census, dim3(nwg), dim3(threads), lds, 0, d, (unsigned long long)us * 100ull);   // s_memrealtime ticks at 100 MHz
    hipDeviceSynchronize();
  }
  std::vector<Rec> h(nwg);
  hipMemcpy(h.data(), d, sizeof(Rec) * nwg, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_cu;
  std::map<unsigned, std::vector<int>> ids;
  unsigned long long tmin = ~0ull, tmax = 0, smax = 0;
  for (int b = 0; b < nwg; ++b) {
    // HW_ID (gfx9): [3:0] wave, [5:4] simd, [6] pipe, [11:8] cu, [12] sh, [15:13] se
    const unsigned cu = (h[b].hwid >> 8) & 0xf, sh = (h[b].hwid >> 12) & 1, se = (h[b].hwid >> 13) & 7;
    const unsigned key = (h[b].xcc << 12) | (se << 8) | (sh << 4) | cu;
    per_cu[key]++;
    ids[key].push_back(b);
    if (h[b].t0 < tmin) tmin = h[b].t0;
    if (h[b].t1 > tmax) tmax = h[b].t1;
    if (h[b].t0 > smax) smax = h[b].t0;
  }
  std::map<int, int> hist;
  for (auto& kv : per_cu) hist[kv.second]++;
  printf("nwg %d threads %d lds %d B spin %d us: %zu distinct CUs used; launch span %.1f us; last start %.1f us after first\n", nwg, threads, lds, us,
         per_cu.size(), (tmax - tmin) / 100.0, (smax - tmin) / 100.0);
  for (auto& kv : hist) printf("  CUs holding %d workgroup(s): %d\n", kv.first, kv.second);
  int shown = 0;
  for (auto& kv : ids) {
    if ((int)kv.second.size() >= 3 && shown < 4) {
      printf("  e.g. xcc %u se %u cu %u holds blocks:", kv.first >> 12, (kv.first >> 8) & 0xf, kv.first & 0xf);
      for (int b : kv.second) printf(" %d(start +%.1f us)", b, (h[b].t0 - tmin) / 100.0);
      printf("\n");
      ++shown;
    }
  }
  return 0;
}
