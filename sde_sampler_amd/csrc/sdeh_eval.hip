// Evaluation-side kernels that are not templated on the state dimension: the bookkeeping of the Sinkhorn iteration
// (eval/sinkhorn.py:112-178) and the one-pass sample statistics behind get_metrics (eval/metrics.py:70-184).
#include "sdeh_common.hpp"

namespace sdeh {

// ---------------------------------------------------------------------------------------------------------
// Sinkhorn bookkeeping.  flags (ints / float bits): [0] done  [1] iterations run  [2] max|du| bits  [3] max|dv| bits
//                                                  [4] last max|du| (float)  [5] last max|dv| (float)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sink_init_kernel(float* u, float* v, float* log_a, float* log_b, const float* w_x,
                                                        const float* w_y, long long n, long long m, float eps, int* flags) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i == 0)
    for (int k = 0; k < 8; ++k) flags[k] = 0;
  // eval/sinkhorn.py:122-125: w_x = ones(n)/n, w_y = ones(m)/m * (n/m)
  if (i < n) {
    log_a[i] = logf(w_x != nullptr ? w_x[i] : 1.0f / (float)n);
    u[i] = 0.0f;
  }
  if (i < m) {
    const float lb = logf(w_y != nullptr ? w_y[i] : (1.0f / (float)m) * (float)((double)n / (double)m));
    log_b[i] = lb;
    v[i] = eps * lb;  // eval/sinkhorn.py:141
  }
}

// pot[i] <- eps (log_w[i] - logsumexp over the split partials); records max |pot_new - pot_old| in *err_bits
__global__ __launch_bounds__(256) void sink_finalize_kernel(const float* __restrict__ part_m, const float* __restrict__ part_s,
                                                            int splits, long long np, const float* __restrict__ log_w,
                                                            float eps, float* __restrict__ pot, int* err_bits, const int* done) {
  if (*done != 0) return;
  __shared__ float sh[4];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  float err = 0.0f;
  if (i < np) {
    float mm = -INFINITY;
    for (int s = 0; s < splits; ++s) mm = fmaxf(mm, part_m[(long long)s * np + i]);
    float ss = 0.0f;
    for (int s = 0; s < splits; ++s) {
      const float pm = part_m[(long long)s * np + i];
      if (pm > -INFINITY) ss += part_s[(long long)s * np + i] * expf(pm - mm);
    }
    const float nv = eps * (log_w[i] - (mm + logf(ss)));
    err = fabsf(nv - pot[i]);
    pot[i] = nv;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) err = fmaxf(err, __shfl_xor(err, o));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = err;
  __syncthreads();
  if (threadIdx.x == 0) {
    err = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    atomicMax(err_bits, __float_as_int(err));  // non-negative floats order like their bit patterns
  }
}

__global__ void sink_check_kernel(int* flags, float thresh) {
  if (flags[0] != 0) return;
  const float eu = __int_as_float(flags[2]), ev = __int_as_float(flags[3]);
  flags[1] += 1;
  flags[4] = flags[2];
  flags[5] = flags[3];
  flags[2] = flags[3] = 0;
  if (eu < thresh && ev < thresh) flags[0] = 1;  // eval/sinkhorn.py:166-167
}

__global__ __launch_bounds__(256) void sink_dist_final_kernel(const float* __restrict__ part, int nb, const int* flags,
                                                              float* out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) acc += (double)part[i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
    out[1] = (float)flags[1];
    out[2] = __int_as_float(flags[4]);
    out[3] = __int_as_float(flags[5]);
  }
}

int launch_sink_init(float* u, float* v, float* log_a, float* log_b, const float* w_x, const float* w_y, long long n,
                     long long m, float eps, int* flags, hipStream_t st) {
  const long long mx = n > m ? n : m;
  hipLaunchKernelGGL(sink_init_kernel, dim3((unsigned)((mx + 255) / 256)), dim3(256), 0, st, u, v, log_a, log_b, w_x, w_y, n, m,
                     eps, flags);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}
int launch_sink_finalize(const float* pm, const float* ps, int splits, long long np, const float* log_w, float eps, float* pot,
                         int* err_bits, const int* done, hipStream_t st) {
  hipLaunchKernelGGL(sink_finalize_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, pm, ps, splits, np, log_w, eps,
                     pot, err_bits, done);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}
int launch_sink_check(int* flags, float thresh, hipStream_t st) {
  hipLaunchKernelGGL(sink_check_kernel, dim3(1), dim3(1), 0, st, flags, thresh);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}
int launch_sink_dist_final(const float* part, int nb, const int* flags, float* out, hipStream_t st) {
  hipLaunchKernelGGL(sink_dist_final_kernel, dim3(1), dim3(256), 0, st, part, nb, flags, out);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------
// Sample statistics (eval/metrics.py:70-184 + distr/base.py:12-17 EXPECTATION_FNS), one pass over samples[B,d]:
//   out[0] = B            out[1] = sum w          out[2] = sum w^2       out[3] = rows inside `domain` (-1 without one)
//   out[4..8)  = sum_i f_k(x_i)        k = square, abs, sum, square_minus_sum   (f_k sums over the coordinates)
//   out[8..12) = sum_i w_i f_k(x_i)
//   out[12 .. 12+d) = per-coordinate mean        out[12+d .. 12+2d) = per-coordinate M2 = sum (x - mean)^2
// Two kernels: per-block partials (Welford per column, plain sums for the rest), then a Chan merge over the blocks.
// A thread owns column tid % d of the rows (tid / d) + k (256 / d): consecutive threads read consecutive floats.
// ---------------------------------------------------------------------------------------------------------
constexpr int kStatHead = 12;

__global__ __launch_bounds__(256) void stats_partial_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ domain, long long B, int d,
                                                            float* __restrict__ part) {
  __shared__ float sh_n[256], sh_mean[256], sh_m2[256], sh_acc[4][kStatHead];
  const int tid = threadIdx.x;
  const int rpb = 256 / d;                       // rows in flight per block pass (d <= 256)
  const int c = tid % d, rsub = tid / d;
  const bool active = rsub < rpb;
  float n = 0.0f, mean = 0.0f, m2 = 0.0f;
  for (long long r = (long long)blockIdx.x * rpb + rsub; active && r < B; r += (long long)gridDim.x * rpb) {
    const float v = x[r * d + c];
    n += 1.0f;
    const float delta = v - mean;
    mean += delta / n;
    m2 = fmaf(delta, v - mean, m2);
  }
  sh_n[tid] = active ? n : 0.0f; sh_mean[tid] = mean; sh_m2[tid] = m2;
  // row-wise quantities: thread per row
  float acc[kStatHead];
#pragma unroll
  for (int k = 0; k < kStatHead; ++k) acc[k] = 0.0f;
  for (long long r = (long long)blockIdx.x * 256 + tid; r < B; r += (long long)gridDim.x * 256) {
    float sq = 0.0f, ab = 0.0f, sm = 0.0f;
    bool inside = true;
    for (int k = 0; k < d; ++k) {
      const float v = x[r * d + k];
      sq = fmaf(v, v, sq); ab += fabsf(v); sm += v;
      if (domain != nullptr) inside = inside && (domain[2 * k] <= v) && (v <= domain[2 * k + 1]);
    }
    const float wi = w != nullptr ? w[r] : 1.0f;
    acc[0] += 1.0f; acc[1] += wi; acc[2] = fmaf(wi, wi, acc[2]); acc[3] += inside ? 1.0f : 0.0f;
    const float f[4] = {sq, ab, sm, sq - sm};
#pragma unroll
    for (int k = 0; k < 4; ++k) { acc[4 + k] += f[k]; acc[8 + k] = fmaf(wi, f[k], acc[8 + k]); }
  }
#pragma unroll
  for (int k = 0; k < kStatHead; ++k) {
    float v = acc[k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((tid & 63) == 0) sh_acc[tid >> 6][k] = v;
  }
  __syncthreads();
  float* out = part + (size_t)blockIdx.x * (kStatHead + 3 * d);
  if (tid < kStatHead) out[tid] = sh_acc[0][tid] + sh_acc[1][tid] + sh_acc[2][tid] + sh_acc[3][tid];
  if (tid < d) {  // Chan merge of the rpb threads that share this column
    float N = 0.0f, MEAN = 0.0f, M2 = 0.0f;
    for (int q = 0; q < rpb; ++q) {
      const int t = q * d + tid;
      const float nb = sh_n[t];
      if (nb == 0.0f) continue;
      const float tot = N + nb, delta = sh_mean[t] - MEAN;
      MEAN += delta * (nb / tot);
      M2 += sh_m2[t] + delta * delta * (N * nb / tot);
      N = tot;
    }
    out[kStatHead + tid] = N; out[kStatHead + d + tid] = MEAN; out[kStatHead + 2 * d + tid] = M2;
  }
}

__global__ __launch_bounds__(256) void stats_final_kernel(const float* __restrict__ part, int nb, int d, int has_domain,
                                                          float* __restrict__ out) {
  const int tid = threadIdx.x;
  const int stride = kStatHead + 3 * d;
  if (tid < kStatHead) {
    double a = 0.0;
    for (int b = 0; b < nb; ++b) a += (double)part[(size_t)b * stride + tid];
    out[tid] = (tid == 3 && !has_domain) ? -1.0f : (float)a;
  }
  for (int c = tid; c < d; c += 256) {
    double N = 0.0, MEAN = 0.0, M2 = 0.0;
    for (int b = 0; b < nb; ++b) {
      const float* p = part + (size_t)b * stride + kStatHead;
      const double n = p[c];
      if (n == 0.0) continue;
      const double tot = N + n, delta = (double)p[d + c] - MEAN;
      MEAN += delta * (n / tot);
      M2 += (double)p[2 * d + c] + delta * delta * (N * n / tot);
      N = tot;
    }
    out[kStatHead + c] = (float)MEAN;
    out[kStatHead + d + c] = (float)M2;
  }
}

int launch_sample_stats(const float* x, const float* w, const float* domain, long long B, int d, float* scratch, int nb,
                        float* out, hipStream_t st) {
  hipLaunchKernelGGL(stats_partial_kernel, dim3(nb), dim3(256), 0, st, x, w, domain, B, d, scratch);
  hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(256), 0, st, scratch, nb, d, domain != nullptr ? 1 : 0, out);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

}  // namespace sdeh
