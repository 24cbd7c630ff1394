// Micro-benchmark: wave-uniform tables streamed through the SCALAR cache (s_load_dwordx16 -> SGPR operands of v_pk_fma_f32)
// against the same tables as broadcast ds_read_b128 from LDS -- the question behind moving the dense mixture tables of the
// trajectory kernel's V wave out of LDS.
//   * 8 waves per workgroup (2 per SIMD), one workgroup per CU, like the trajectory kernel at B = 65 536
//   * every wave streams a table of F bytes cyclically, 32 dwords per batch, double-buffered, and spends 2 packed vector
//     instructions per coordinate pair on it (the shared-scale logit loop: t = y - m, acc = fma(t, t, acc))
//   * mode 0: s_load_dwordx16 x 2 per batch (constant address space)   mode 1: ds_read_b128 x 8 per batch (broadcast)   mode 2: no loads
// Reports cycles per batch per wave (s_memtime) as a function of F.
// Build: hipcc --offload-arch=gfx950 -O3 smem_stream.hip -o smem_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned long long f16v __attribute__((ext_vector_type(8)));  // eight SGPR pairs = one s_load_dwordx16
typedef const f16v __attribute__((address_space(4))) * c16p;

// hipcc unpacks a packed op with a scalar-register operand into two v_sub_f32: written in assembly
__device__ __forceinline__ f2 pk_sub_s(f2 a, unsigned long long s) {
  f2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "s"(s));
  return r;
}
__device__ __forceinline__ void use16(const f16v& q, const f2 (&y)[8], f2& a0, f2& a1) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const f2 t = pk_sub_s(y[j], q[j]);
    if (j & 1) a1 = __builtin_elementwise_fma(t, t, a1);
    else a0 = __builtin_elementwise_fma(t, t, a0);
  }
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ tab, int batches, int iters, float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (MODE == 1) {
    for (int i = threadIdx.x; i < batches * 32; i += blockDim.x) lds[i] = tab[i];
    __syncthreads();
  }
  f2 y[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) y[j] = f2{threadIdx.x * 1e-3f + j, threadIdx.x * 2e-3f - j};
  f2 a0 = {0, 0}, a1 = {0, 0};
  c16p p = (c16p)(unsigned long long)tab;
  const float4* l4 = reinterpret_cast<const float4*>(lds);
  const unsigned long long t0 = __builtin_readcyclecounter();
  f16v qa, qb;
  if (MODE == 0) { qa = p[0]; qb = p[1]; }
  for (int it = 0; it < iters; ++it) {
    for (int b = 0; b < batches; ++b) {
      const int nb = b + 1 < batches ? b + 1 : 0;
      if (MODE == 0) {
        const f16v na = p[2 * nb], nbv = p[2 * nb + 1];
        __builtin_amdgcn_sched_barrier(0);
        use16(qa, y, a0, a1);
        use16(qb, y, a0, a1);
        __builtin_amdgcn_sched_barrier(0);
        qa = na; qb = nbv;
      } else if (MODE == 1) {
        float4 q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = l4[b * 8 + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f2 t0v = y[(2 * j) & 7] - f2{q[j].x, q[j].y}, t1v = y[(2 * j + 1) & 7] - f2{q[j].z, q[j].w};
          a0 = __builtin_elementwise_fma(t0v, t0v, a0);
          a1 = __builtin_elementwise_fma(t1v, t1v, a1);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const f2 t = y[j & 7] - f2{(float)b, (float)j};
          if (j & 1) a1 = __builtin_elementwise_fma(t, t, a1);
          else a0 = __builtin_elementwise_fma(t, t, a0);
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a0.y + a1.x + a1.y;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE>
void run(const char* name, int nwg, int waves, const float* d_tab, float* d_out, unsigned long long* d_cyc) {
  printf("%s, %d workgroups x %d waves\n", name, nwg, waves);
  for (int kb : {2, 4, 8, 12, 14, 16, 18, 20, 24, 32, 48, 64}) {
    const int batches = kb * 1024 / 128, iters = 64;
    const size_t sh = MODE == 1 ? (size_t)kb * 1024 : 0;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(64 * waves), sh, 0, d_tab, batches, iters, d_out, d_cyc);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(64 * waves), sh, 0, d_tab, batches, iters, d_out, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[8]; hipMemcpy(c, d_cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("  table %2d KB: %7.1f cycles per 32-dword batch per wave (wave 0; memtime ticks x 21 if 100 MHz), kernel %.3f ms = %.1f ns per batch\n",
           kb, (double)c[0] / ((double)batches * iters), ms, ms * 1e6 / ((double)batches * iters));
  }
}

int main() {
  float *d_tab, *d_out; unsigned long long* d_cyc;
  std::vector<float> h(65536 / 4 * 4, 0.5f);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 97) * 0.01f;
  hipMalloc(&d_tab, h.size() * 4); hipMalloc(&d_out, 512 * 512 * 4); hipMalloc(&d_cyc, 64);
  hipMemcpy(d_tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<2>("no loads (pure vector work: 32 packed instructions per batch)", 256, 8, d_tab, d_out, d_cyc);
  run<0>("scalar loads", 256, 8, d_tab, d_out, d_cyc);
  run<0>("scalar loads, one wave per SIMD", 256, 4, d_tab, d_out, d_cyc);
  run<1>("LDS broadcast reads", 256, 8, d_tab, d_out, d_cyc);
  run<1>("LDS broadcast reads, one wave per SIMD", 256, 4, d_tab, d_out, d_cyc);
  return 0;
}
