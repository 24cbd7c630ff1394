// Test hooks: expose the exact random stream sdeh_simulate_fwd consumes (include/sdeh.h: sdeh_debug_*).
#include "sdeh_traj.hpp"

namespace sdeh {

__global__ void debug_philox_kernel(unsigned long long seed, unsigned long long offset, long long row_offset, int step,
                                    int block, long long n, uint32_t* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const U4 r = philox_block(seed, offset, (unsigned long long)(row_offset + i), step, block);
  out[4 * i + 0] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

__global__ void debug_normals_kernel(unsigned long long seed, unsigned long long offset, long long row_offset, int step,
                                     int dim, long long n, float* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int jb = 0; 4 * jb < dim; ++jb) {
    float v[4];
    box_muller4(philox_block(seed, offset, (unsigned long long)(row_offset + i), step, jb), v);
    for (int q = 0; q < 4; ++q)
      if (4 * jb + q < dim) out[i * dim + 4 * jb + q] = v[q];
  }
}

__global__ void debug_gelu_kernel(const float* in, long long n, float* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = act_gelu(in[i]);
}

}  // namespace sdeh

extern "C" {

int32_t sdeh_debug_philox(uint64_t seed, uint64_t offset, int64_t row_offset, int32_t step, int32_t block, int64_t n,
                          uint32_t* out, void* stream) {
  if (out == nullptr || n < 1) return SDEH_ERR_INVALID;
  hipLaunchKernelGGL(sdeh::debug_philox_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (unsigned long long)seed, (unsigned long long)offset, (long long)row_offset, step, block,
                     (long long)n, out);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

int32_t sdeh_debug_normals(uint64_t seed, uint64_t offset, int64_t row_offset, int32_t step, int32_t dim, int64_t n,
                           float* out, void* stream) {
  if (out == nullptr || n < 1 || dim < 1) return SDEH_ERR_INVALID;
  hipLaunchKernelGGL(sdeh::debug_normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (unsigned long long)seed, (unsigned long long)offset, (long long)row_offset, step, dim,
                     (long long)n, out);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

int32_t sdeh_debug_gelu(const float* in, int64_t n, float* out, void* stream) {
  if (in == nullptr || out == nullptr || n < 1) return SDEH_ERR_INVALID;
  hipLaunchKernelGGL(sdeh::debug_gelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in,
                     (long long)n, out);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

}  // extern "C"
