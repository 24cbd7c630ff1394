// Micro-benchmark: what does it cost a one-wave-per-SIMD MFMA stream to issue its own operand loads?
// Each of the 4 waves of a workgroup (one per SIMD) runs groups of 8 x v_mfma_f32_32x32x2_f32 (= 512 cycles of matrix work) and, per
// group, NL loads of 1 KB per wave-instruction whose results are only awaited many groups later (L2-hot 64 KB buffer per CU):
//   mode 0: global_load_dwordx4 (saddr form)        mode 1: ds_read_b128 from LDS
//   mode 2: the loads are issued by a SECOND wave on the same SIMD (8 waves per workgroup: 4 MFMA waves + 4 loader waves that
//           copy global -> LDS with global_load_dwordx4 + ds_write_b128), the MFMA wave issues none
// Build: hipcc --offload-arch=gfx950 -O3 vmem_issue.hip -o vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NL>
__global__ __launch_bounds__(512) void k(const float* __restrict__ buf, float* out, int iters) {
  __shared__ f32x4 lds[4096];  // 64 KB
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  const float* base = buf + (size_t)blockIdx.x * 16384;  // 64 KB per workgroup
  if (wave >= 4) {  // loader waves (mode 2 only)
    if (MODE != 2) return;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (((it * NL + l) & 63) * 64 + lane) * 4);
        lds[((it * NL + l) & 63) * 64 + lane] = v;
        acc += v;
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc.x;
    return;
  }
  f32x16 a0 = {}, a1 = {};
  f32x4 ring[4][NL > 0 ? NL : 1];
  float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  f32x4 sum = {0, 0, 0, 0};
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE != 2) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          const int slot = ((it + u) * NL + l) & 63;
          if (MODE == 0) ring[u][l] = *reinterpret_cast<const f32x4*>(base + (slot * 64 + lane) * 4);
          else ring[u][l] = lds[slot * 64 + lane];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 2) {
#pragma unroll
        for (int l = 0; l < NL; ++l) sum += ring[(u + 1) & 3][l];  // consume what was loaded three groups ago
      }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + sum.x + sum.y;
}

template <int MODE, int NL>
float run(const float* buf, float* d, int threads) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NL>), dim3(256), dim3(threads), 0, 0, buf, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NL>), dim3(256), dim3(threads), 0, 0, buf, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 2.4e6f / iters;  // cycles per group of 8 MFMAs @ 2.4 GHz
}

int main() {
  float *buf, *d;
  hipMalloc(&buf, 256 * 65536); hipMemset(buf, 0, 256 * 65536);
  hipMalloc(&d, 256 * 512 * 4);
  printf("cycles per group of 8 MFMAs (ideal 512 @ 64/MFMA):\n");
  printf("no loads                        : %.0f\n", run<0, 0>(buf, d, 256));
  printf("own global_load_dwordx4 x1/x2/x4: %.0f %.0f %.0f\n", run<0, 1>(buf, d, 256), run<0, 2>(buf, d, 256), run<0, 4>(buf, d, 256));
  printf("own ds_read_b128        x1/x2/x4: %.0f %.0f %.0f\n", run<1, 1>(buf, d, 256), run<1, 2>(buf, d, 256), run<1, 4>(buf, d, 256));
  printf("loader wave global->LDS x1/x2/x4: %.0f %.0f %.0f\n", run<2, 1>(buf, d, 512), run<2, 2>(buf, d, 512), run<2, 4>(buf, d, 512));
  return 0;
}
