// Weight-gradient contraction of the training backward:
//     dW_k = sum_n Dt[k+1][:, n] act(Zt[k][:, n])^T      db_k = sum_n Dt[k+1][:, n]      (n over the N = T*B rows)
// i.e. the per-parameter sums the reference's autograd accumulates through its T per-step Linear backward calls
// (models/mlp.py:114-122 under losses/oc.py:232-256).  The planes the backward kernels write are coordinate-major
// ([C][N], N contiguous), so both MFMA operands are K-contiguous: lane (j, h) streams row j as float4s along n and the
// contraction index is simply enumerated in the order the lanes hold it -- no LDS staging, no transposition.  The
// activation is applied to Z on the fly (the host used to run one GELU kernel, one split-K bmm, one sum and one bias
// reduction per layer over planes that reach 1.7 GB each at B = 65 536: 10 passes over HBM instead of one).
// One wave = one chunk of n = one [64, 64] partial (+ [64] bias partial); the chunks are summed by the caller.  Layers of the wide
// networks (m, c up to 256) are tiled into [64, 64] blocks over blockIdx.y / .z, each block streaming its own 64 + 64 rows.
#include "sdeh_traj.hpp"

namespace sdeh {

constexpr int kActIdentity = 3;  // SDEH_ACT_IDENTITY: Z already holds activations (or plain inputs)

template <int ACT>
__device__ __forceinline__ float wg_act(float v) {
  if constexpr (ACT == kActIdentity) return v;
  else return act_ct<ACT>(v);
}

constexpr int kWgV = 4;               // float4 per lane, tile and iteration
constexpr int kWgStep = 8 * kWgV;     // rows of n per iteration: lane half h covers [base + 16 h, base + 16 h + 16) -> whole 128-byte lines

struct WgTile {
  float4 v[kWgV];
};

// one tile row of this lane for the iteration starting at `base`; rows >= nrows and n >= n_end read as zero
__device__ __forceinline__ void wg_load(WgTile& t, const float* __restrict__ row, bool on, long long n, long long n_end, bool fast) {
  if (!on) {
#pragma unroll
    for (int k = 0; k < kWgV; ++k) t.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else if (fast) {  // whole iteration inside the chunk, rows 16-byte aligned
#pragma unroll
    for (int k = 0; k < kWgV; ++k) t.v[k] = *reinterpret_cast<const float4*>(row + n + 4 * k);
  } else {
#pragma unroll
    for (int k = 0; k < kWgV; ++k) {
      const long long q = n + 4 * k;
      t.v[k].x = q < n_end ? row[q] : 0.0f;
      t.v[k].y = q + 1 < n_end ? row[q + 1] : 0.0f;
      t.v[k].z = q + 2 < n_end ? row[q + 2] : 0.0f;
      t.v[k].w = q + 3 < n_end ? row[q + 3] : 0.0f;
    }
  }
}

template <int ACT>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ D, int m, const float* __restrict__ Z, int c,
                                                    long long N, long long chunk, long long n_chunks,
                                                    float* __restrict__ part_w, float* __restrict__ part_b, int mp_blocks, int cp_blocks) {
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int wv = threadIdx.x >> 6;
  // 64-channel networks (one [64, 64] block): the four waves of a workgroup take four chunks.  Wide layers (m, c up to 256): the
  // [m, c] product is tiled into [64, 64] blocks and the four waves take a 2 x 2 arrangement of blocks of the SAME chunk -- two waves
  // share each row of D and of Z, which then comes from the CU's L1 instead of L2 for the second one.  The partial of a chunk is the
  // padded matrix [mp][ldw] (multiples of 64; 64-channel networks: [64][64], unchanged).
  const bool blocked = mp_blocks > 1 || cp_blocks > 1;
  const long long ck = blocked ? (long long)blockIdx.x : (long long)blockIdx.x * 4 + wv;
  if (ck >= n_chunks) return;
  const int mb = blocked ? 2 * (int)blockIdx.y + (wv >> 1) : 0, cb = blocked ? 2 * (int)blockIdx.z + (wv & 1) : 0;
  if (mb >= mp_blocks || cb >= cp_blocks) return;
  const long long n_begin = ck * chunk;
  const long long n_end = n_begin + chunk < N ? n_begin + chunk : N;
  const int ldw = 64 * cp_blocks, mp = 64 * mp_blocks;
  D += (long long)mb * 64 * N;
  Z += (long long)cb * 64 * N;
  m = m - 64 * mb < 64 ? m - 64 * mb : 64;
  c = c - 64 * cb < 64 ? c - 64 * cb : 64;
  const bool two_d = m > 32, two_z = c > 32;
  // float4 loads when every row start is 16-byte aligned (n_begin is a multiple of 8)
  const bool vec = (N & 3) == 0 && ((reinterpret_cast<unsigned long long>(D) | reinterpret_cast<unsigned long long>(Z)) & 15) == 0;
  const bool d0 = j < m, d1 = 32 + j < m, z0 = j < c, z1 = 32 + j < c;
  const float* __restrict__ Dr0 = D + (long long)(d0 ? j : 0) * N;
  const float* __restrict__ Dr1 = D + (long long)(d1 ? 32 + j : 0) * N;
  const float* __restrict__ Zr0 = Z + (long long)(z0 ? j : 0) * N;
  const float* __restrict__ Zr1 = Z + (long long)(z1 ? 32 + j : 0) * N;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;
  float bs0 = 0.0f, bs1 = 0.0f;

  // lane (j, h) holds n = base + 16 h + {0..15}: k-step s contracts the pair (base + s, base + 16 + s).
  // The next iteration's 16 float4 are in flight while this one's 64 MFMAs and 32 activations issue.
  WgTile cd0, cd1, cz0, cz1, nd0, nd1, nz0, nz1;
  {
    const bool fast = vec && n_begin + kWgStep <= n_end;
    const long long n = n_begin + 16 * h;
    wg_load(cd0, Dr0, d0, n, n_end, fast);
    wg_load(cd1, Dr1, d1, n, n_end, fast);
    wg_load(cz0, Zr0, z0, n, n_end, fast);
    wg_load(cz1, Zr1, z1, n, n_end, fast);
  }
  for (long long base = n_begin; base < n_end; base += kWgStep) {
    const long long nb = base + kWgStep;
    if (nb < n_end) {
      const bool fast = vec && nb + kWgStep <= n_end;
      const long long n = nb + 16 * h;
      wg_load(nd0, Dr0, d0, n, n_end, fast);
      wg_load(nd1, Dr1, d1, n, n_end, fast);
      wg_load(nz0, Zr0, z0, n, n_end, fast);
      wg_load(nz1, Zr1, z1, n, n_end, fast);
    }
#pragma unroll
    for (int k = 0; k < kWgV; ++k) {
      const float dd0[4] = {cd0.v[k].x, cd0.v[k].y, cd0.v[k].z, cd0.v[k].w};
      const float dd1[4] = {cd1.v[k].x, cd1.v[k].y, cd1.v[k].z, cd1.v[k].w};
      // rows >= c: the loads already returned zeros, but act(0) need not be 0 -> mask after the activation.  Elements past n_end
      // carry D = 0, so their activation value is immaterial.
      float a0[4] = {wg_act<ACT>(cz0.v[k].x), wg_act<ACT>(cz0.v[k].y), wg_act<ACT>(cz0.v[k].z), wg_act<ACT>(cz0.v[k].w)};
      float a1[4] = {wg_act<ACT>(cz1.v[k].x), wg_act<ACT>(cz1.v[k].y), wg_act<ACT>(cz1.v[k].z), wg_act<ACT>(cz1.v[k].w)};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (!z0) a0[s] = 0.0f;
        if (!z1) a1[s] = 0.0f;
      }
      bs0 += (dd0[0] + dd0[1]) + (dd0[2] + dd0[3]);
      bs1 += (dd1[0] + dd1[1]) + (dd1[2] + dd1[3]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0][0] = SDEH_MFMA(dd0[s], a0[s], acc[0][0]);
        if (two_z) acc[0][1] = SDEH_MFMA(dd0[s], a1[s], acc[0][1]);
        if (two_d) {
          acc[1][0] = SDEH_MFMA(dd1[s], a0[s], acc[1][0]);
          if (two_z) acc[1][1] = SDEH_MFMA(dd1[s], a1[s], acc[1][1]);
        }
      }
    }
    cd0 = nd0; cd1 = nd1; cz0 = nz0; cz1 = nz1;
  }

  // partial [64][64]: accumulator q of lane (j, h) in tile (a, b) is element (32 a + rho(q, h), 32 b + j)
  float* __restrict__ pw = part_w + ck * ((long long)mp * ldw) + (long long)mb * 64 * ldw + cb * 64;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) pw[(32 * a + rho(q, h)) * ldw + 32 * b + j] = acc[a][b][q];
  bs0 += __shfl_xor(bs0, 32);
  bs1 += __shfl_xor(bs1, 32);
  if (h == 0 && cb == 0) {
    part_b[ck * mp + mb * 64 + j] = bs0;
    part_b[ck * mp + mb * 64 + 32 + j] = bs1;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Sum of the per-chunk partials: out[i][e] = sum_k part[i][k][e].  Two passes (groups of kSumGroup chunks, then the groups), one
// thread per element, consecutive threads on consecutive elements; no atomics and no host-zeroed semaphores -- the library
// reduction this replaces (torch's multi-block sum) is not safe to replay inside a hipGraph on this ROCm stack: a second
// reduction captured right after it comes back corrupted from the second replay on (its output block holds what look like the
// first reduction's semaphore counters / staging values; plain memset -> kernel ordering is fine, tests/perf/rocm_graph_memset_order.py)
// (tests/perf/rocm_graph_two_reductions.py reproduces it with PyTorch alone).
// ---------------------------------------------------------------------------------------------------------
constexpr int kSumGroup = 32;

__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ in, long long n_in, long long width,
                                                          long long group, float* __restrict__ out, long long n_out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long g = blockIdx.y, item = blockIdx.z;
  if (e >= width) return;
  const long long k0 = g * group, k1 = k0 + group < n_in ? k0 + group : n_in;
  const float* __restrict__ src = in + (item * n_in + k0) * width + e;
  float acc0 = 0.0f, acc1 = 0.0f;  // two chains: fixed order, hence deterministic
  long long k = k0;
  for (; k + 1 < k1; k += 2) {
    acc0 += src[0];
    acc1 += src[width];
    src += 2 * width;
  }
  if (k < k1) acc0 += src[0];
  out[(item * n_out + g) * width + e] = acc0 + acc1;
}

int launch_partial_sums(const float* part, long long n_items, long long n_chunks, long long width, float* scratch, float* out,
                        hipStream_t stream) {
  const unsigned bx = width >= 256 ? 256 : 64;
  const long long groups = (n_chunks + kSumGroup - 1) / kSumGroup;
  const dim3 grid_a((unsigned)((width + bx - 1) / bx), (unsigned)groups, (unsigned)n_items);
  hipLaunchKernelGGL(partial_sum_kernel, grid_a, dim3(bx), 0, stream, part, n_chunks, width, (long long)kSumGroup, scratch, groups);
  const dim3 grid_b((unsigned)((width + bx - 1) / bx), 1, (unsigned)n_items);
  hipLaunchKernelGGL(partial_sum_kernel, grid_b, dim3(bx), 0, stream, scratch, groups, width, groups, out, 1LL);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

// Several such sums (the fused backward finishes with three to five: weight, time-embedding and scale partials) in ONE pair of launches
// -- one launch when no sum has more than kSumGroup chunks (batches below ~1e3 trajectories, where a replayed optimisation step is a chain
// of 5 us dispatches).  Same two-level order of additions as launch_partial_sums, sum by sum.
__global__ __launch_bounds__(256) void multi_sum_kernel(SumJob job, int pass) {
  int s = 0;
  while (s + 1 < job.n && (int)blockIdx.x >= job.first_block[s + 1]) ++s;
  const SumSeg seg = job.s[s];
  const long long groups = (seg.n_chunks + kSumGroup - 1) / kSumGroup;
  const long long e = (long long)((int)blockIdx.x - job.first_block[s]) * 256 + threadIdx.x;
  const long long g = blockIdx.y;
  if (e >= seg.width || g >= groups || (pass == 1 && groups == 1)) return;
  const float* __restrict__ in = pass == 0 ? seg.in : seg.mid;
  const long long n_in = pass == 0 ? seg.n_chunks : groups, group = pass == 0 ? kSumGroup : groups;
  float* __restrict__ out = (pass == 1 || groups == 1) ? seg.out : seg.mid;
  const long long k0 = g * group, k1 = k0 + group < n_in ? k0 + group : n_in;
  const float* __restrict__ src = in + k0 * seg.width + e;
  float acc0 = 0.0f, acc1 = 0.0f;
  long long k = k0;
  for (; k + 1 < k1; k += 2) {
    acc0 += src[0];
    acc1 += src[seg.width];
    src += 2 * seg.width;
  }
  if (k < k1) acc0 += src[0];
  out[(pass == 0 && groups > 1 ? g : 0) * seg.width + e] = acc0 + acc1;
}

int launch_partial_sums_multi(SumJob job, hipStream_t stream) {
  long long max_groups = 1;
  int blocks = 0;
  for (int s = 0; s < job.n; ++s) {
    job.first_block[s] = blocks;
    blocks += (int)((job.s[s].width + 255) / 256);
    const long long groups = (job.s[s].n_chunks + kSumGroup - 1) / kSumGroup;
    max_groups = groups > max_groups ? groups : max_groups;
  }
  job.first_block[job.n] = blocks;
  hipLaunchKernelGGL(multi_sum_kernel, dim3((unsigned)blocks, (unsigned)max_groups), dim3(256), 0, stream, job, 0);
  if (max_groups > 1) hipLaunchKernelGGL(multi_sum_kernel, dim3((unsigned)blocks, 1), dim3(256), 0, stream, job, 1);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

int launch_weight_grad(const float* D, int m, const float* Z, int c, long long N, int act, long long chunk, float* part_w,
                       float* part_b, hipStream_t stream) {
  const long long n_chunks = (N + chunk - 1) / chunk;
  const int mpb = (m + 63) / 64, cpb = (c + 63) / 64;
  const bool blocked = mpb > 1 || cpb > 1;
  const dim3 grid(blocked ? (unsigned)n_chunks : (unsigned)((n_chunks + 3) / 4), blocked ? (unsigned)((mpb + 1) / 2) : 1u,
                  blocked ? (unsigned)((cpb + 1) / 2) : 1u);
  switch (act) {
    case SDEH_ACT_GELU_ERF:
      hipLaunchKernelGGL(wgrad_kernel<SDEH_ACT_GELU_ERF>, grid, dim3(256), 0, stream, D, m, Z, c, N, chunk, n_chunks, part_w, part_b, mpb, cpb);
      break;
    case SDEH_ACT_SILU:
      hipLaunchKernelGGL(wgrad_kernel<SDEH_ACT_SILU>, grid, dim3(256), 0, stream, D, m, Z, c, N, chunk, n_chunks, part_w, part_b, mpb, cpb);
      break;
    case SDEH_ACT_RELU:
      hipLaunchKernelGGL(wgrad_kernel<SDEH_ACT_RELU>, grid, dim3(256), 0, stream, D, m, Z, c, N, chunk, n_chunks, part_w, part_b, mpb, cpb);
      break;
    case kActIdentity:
      hipLaunchKernelGGL(wgrad_kernel<kActIdentity>, grid, dim3(256), 0, stream, D, m, Z, c, N, chunk, n_chunks, part_w, part_b, mpb, cpb);
      break;
    default:
      return SDEH_ERR_INVALID;
  }
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

}  // namespace sdeh
