// The NICE flow target of BASELINE configs[4] (reference distr/nice.py): unnorm_log_prob(x) and score(x) = d unnorm_log_prob / d x for a
// batch of rows, as the losses ask of a target (losses/oc.py:225 terminal cost; models/reparam.py:185-197 LerpTargetCtrl's score term).
//
//   f(x):  for every coupling (nice.py:64-97, reverse = False): x viewed as [B, d/2, 2]; the "off" half feeds an MLP
//          (in_block Linear + ReLU, hidden - 1 mid blocks Linear + ReLU, out_block Linear), the "on" half is shifted by its output;
//          then z = x * exp(scale) (nice.py:100-120);  log p = sum_j -(softplus(z_j) + softplus(-z_j)) + sum_j scale_j (nice.py:17-27, 176-189)
//   score: the reverse pass -- g_z = -tanh(z / 2), g_x = g_z * exp(scale), then through the couplings backwards:
//          g_off += J_MLP(off)^T g_on (additive couplings: the "on" gradient passes unchanged).  The reference gets it from autograd
//          (distr/base.py:130-137); the weights are constants (nice.py:271-273), so only d / d x is propagated.
//
// Everything is GEMM-shaped: per coupling hidden + 1 products forward and as many backward, [B, K] x [K, N] with K, N in {d/2, mid_dim}
// (98 / 500 in the reference's checkpoint format).  One kernel serves them all: a workgroup of eight waves owns a 64 (rows of the batch) x
// 128 (outputs) tile, both operands are staged through LDS k-major ([k][row], so that the 32x32x2 fp32 MFMA's operand fetch -- lane (j, h)
// reads element j of k-row 2 s + h -- is one conflict-free ds_read_b32), the next k-tile's global loads are in flight while the current
// one is multiplied, and is stored into the second LDS buffer behind the multiplication (one barrier per k-tile of 32).  The weight operand is read in the layout nn.Linear stores it ([out, in]): k-contiguous for a forward layer
// (transposed on its way into LDS), n-contiguous for the reverse pass (copied as is) -- no packed or transposed copies of the parameters
// exist, they are re-read on every call like every other parameter of this library.  Epilogue: + bias, + residual (the coupling's shift
// joins the "on" half in place), ReLU, or the ReLU mask of the reverse pass (the stored activation of the layer below > 0).
// fp32 throughout (the matrix instruction is bitwise an fmaf chain); the summation order of a dot product differs from the reference's
// BLAS, which is what the parity tolerance of tests/test_hip_nice.py covers.
#include <type_traits>

#include "sdeh_traj.hpp"

namespace sdeh {

constexpr int kNgBN = 128, kNgBK = 32;
// LDS row strides (floats).  The transposing stores put lane (row r of 8, k quad q of 8) at [4 q + e][r]: with a stride = 2 (mod 8) the 64 lanes
// of a wave land on every bank exactly twice (the minimum for 64 lanes on 32 banks); the n-contiguous copy is a 16-byte store: stride = 0 (mod 4)
template <bool TRANS_B> constexpr int kNgSW = TRANS_B ? kNgBN + 2 : kNgBN + 4;

struct NiceGemm {
  const float* X; long long ldx;    // [M, K], k contiguous
  const float* W; long long ldw;    // TRANS_B: [N, K] (k contiguous: nn.Linear.weight of a forward layer); else [K, N] (n contiguous)
  const float* bias;                // [N] or null
  const float* addend; long long lda;  // [M, N] or null (may be Y itself)
  const float* mask; long long ldm;    // [M, N] or null: the result is kept where mask > 0, zero elsewhere
  float* Y; long long ldy;
  long long M;
  int N, K, relu;
};

__device__ __forceinline__ float4 ng_load4(const float* __restrict__ row, int k, int K, bool ok, bool aligned) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!ok) return v;
  if (aligned && k + 3 < K) return *reinterpret_cast<const float4*>(row + k);
  if (k < K) v.x = row[k];
  if (k + 1 < K) v.y = row[k + 1];
  if (k + 2 < K) v.z = row[k + 2];
  if (k + 3 < K) v.w = row[k + 3];
  return v;
}

// Eight waves (two per SIMD: one wave's LDS / global latency is the other's matrix time) own a 64 x 128 tile as a 2 x 4 arrangement of 32 x 32
// accumulator tiles; k-tiles of 32, two LDS buffers (the next tile is stored while the current one is multiplied: one barrier per k-tile).
// VEC: every operand row is 16-byte aligned and whole float4s are in or out of range (all but the two products per coupling that read
// in_block.weight, whose rows are 98 floats): the loads of a k-tile are then UNCONDITIONAL 16-byte loads from clamped addresses with the
// out-of-range values selected to zero afterwards -- a fixed instruction sequence, so that the compiler can count the loads in flight
// (`s_waitcnt vmcnt(3)`: wait for the older tile, leave the newer one in flight).  With loads under branches it waited for ALL of them
// in every iteration (vmcnt(0)): 1.77 us per k-tile = the memory latency, 0.55 of the matrix rate inside the loop.
// BM = 64: eight waves (2 x 4 accumulator tiles); BM = 32: four waves (1 x 4) -- twice the workgroups with half the work each, for batches
// whose 64-row tiles would not fill the chip (nice_gemm: batch <= kNgSmallBatch).  A row's arithmetic does not depend on the tile height.
template <bool TRANS_B, bool VEC, int BM>
__global__ __launch_bounds__(8 * BM) void nice_gemm_kernel(const NiceGemm G) {
  constexpr int kNgBM = BM, kNgSX = BM + 2, NR = kNgBN / BM;  // NR: passes of the workgroup over the 128 weight rows (TRANS_B) / 32 k-rows of a tile
  __shared__ __attribute__((aligned(16))) float Xs[2][kNgBK * kNgSX];
  constexpr int SW = kNgSW<TRANS_B>;
  __shared__ __attribute__((aligned(16))) float Ws[2][kNgBK * SW];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = BM == 64 ? (w & 1) : 0, wn = BM == 64 ? (w >> 1) : w;
  const long long m0 = (long long)blockIdx.x * kNgBM;
  const int n0 = (int)blockIdx.y * kNgBN;
  const int K = G.K, N = G.N;
  const bool ax = (G.ldx & 3) == 0 && (reinterpret_cast<unsigned long long>(G.X) & 15) == 0;
  const bool aw = (G.ldw & 3) == 0 && (reinterpret_cast<unsigned long long>(G.W) & 15) == 0;

  // this thread's share of a k-tile: one float4 of X (row xm, k quad xq), two float4 of W
  const int xm = tid >> 3, xq = tid & 7;
  const bool xok = m0 + xm < G.M;
  const float* __restrict__ xrow = G.X + (xok ? m0 + xm : G.M - 1) * G.ldx;
  // two register sets: the global loads of k-tile kt + 2 are issued while tile kt is multiplied and tile kt + 1 (loaded an iteration
  // earlier) moves into the other LDS buffer -- two iterations of latency tolerance for a chain that has one workgroup per CU at B = 4096
  float4 rx[2], rw[2][NR];
  const int K4 = (K + 3) & ~3;  // (X rows are padded with zeros to a multiple of 4; VEC weights have K % 4 == 0 resp. N % 4 == 0)
  auto load_tile = [&](int kt, auto SET) {
    constexpr int S = decltype(SET)::value;
    const int k0 = kt * kNgBK;
    if constexpr (VEC) {  // raw loads from clamped addresses; out-of-range values are zeroed where the tile is stored (store_tile)
      const int kq = k0 + 4 * xq;
      const int kc = kq < K4 ? kq : K4 - 4;
      rx[S] = *reinterpret_cast<const float4*>(xrow + kc);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if constexpr (TRANS_B) {  // W [N, K]: row n = n0 + tid / 8 + 64 r, k quad tid % 8
          const int n = n0 + xm + BM * r;
          rw[S][r] = *reinterpret_cast<const float4*>(G.W + (long long)(n < N ? n : N - 1) * G.ldw + kc);
        } else {                  // W [K, N]: k row tid / 32 + 16 r, n quad tid % 32
          const int k = k0 + (tid >> 5) + (BM / 4) * r, nq = n0 + 4 * (tid & 31);
          rw[S][r] = *reinterpret_cast<const float4*>(G.W + (long long)(k < K ? k : K - 1) * G.ldw + (nq < N ? nq : N - 4));
        }
      }
    } else {
      rx[S] = ng_load4(xrow, k0 + 4 * xq, K, xok, ax);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if constexpr (TRANS_B) {
          const int n = n0 + xm + BM * r;
          rw[S][r] = ng_load4(G.W + (long long)(n < N ? n : 0) * G.ldw, k0 + 4 * xq, K, n < N, aw);
        } else {
          const int k = k0 + (tid >> 5) + (BM / 4) * r;
          rw[S][r] = ng_load4(G.W + (long long)(k < K ? k : 0) * G.ldw, n0 + 4 * (tid & 31), N, k < K, aw);
        }
      }
    }
  };
  auto store_tile = [&](auto SET, int kt) {  // register set S (holding k-tile kt) -> LDS buffer S
    constexpr int S = decltype(SET)::value;
    float4 vx = rx[S], vw[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) vw[r] = rw[S][r];
    if constexpr (VEC) {  // (component-wise selects: a ternary on the float4 struct is lowered to a select of ADDRESSES, i.e. scratch)
      const int k0 = kt * kNgBK, kq = k0 + 4 * xq;
      const bool okx = xok && kq < K4;
      vx.x = okx ? vx.x : 0.0f; vx.y = okx ? vx.y : 0.0f; vx.z = okx ? vx.z : 0.0f; vx.w = okx ? vx.w : 0.0f;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        bool ok;
        if constexpr (TRANS_B) ok = n0 + xm + BM * r < N && kq < K;
        else ok = k0 + (tid >> 5) + (BM / 4) * r < K && n0 + 4 * (tid & 31) < N;
        vw[r].x = ok ? vw[r].x : 0.0f; vw[r].y = ok ? vw[r].y : 0.0f; vw[r].z = ok ? vw[r].z : 0.0f; vw[r].w = ok ? vw[r].w : 0.0f;
      }
    }
    float* __restrict__ xs = Xs[S];
    float* __restrict__ wsb = Ws[S];
    xs[(4 * xq + 0) * kNgSX + xm] = vx.x;
    xs[(4 * xq + 1) * kNgSX + xm] = vx.y;
    xs[(4 * xq + 2) * kNgSX + xm] = vx.z;
    xs[(4 * xq + 3) * kNgSX + xm] = vx.w;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if constexpr (TRANS_B) {
        const int n = xm + BM * r;
        wsb[(4 * xq + 0) * SW + n] = vw[r].x;
        wsb[(4 * xq + 1) * SW + n] = vw[r].y;
        wsb[(4 * xq + 2) * SW + n] = vw[r].z;
        wsb[(4 * xq + 3) * SW + n] = vw[r].w;
      } else {
        *reinterpret_cast<float4*>(wsb + ((tid >> 5) + (BM / 4) * r) * SW + 4 * (tid & 31)) = vw[r];
      }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  const int nk = (K + kNgBK - 1) / kNgBK;
  load_tile(0, I0{});
  load_tile(nk > 1 ? 1 : 0, I1{});
  store_tile(I0{}, 0);
  __syncthreads();
  const int xoff = h * kNgSX + 32 * wm + j, woff = h * SW + 32 * wn + j;
  // D rows = outputs n (A operand: the weights), D columns = batch rows m (B operand): lane (j, h) ends up with 16 outputs of row m0 + 32 wm + j
  auto step = [&](int kt, auto CUR, auto NXT) {
    constexpr int B = decltype(CUR)::value;
    load_tile(kt + 2 < nk ? kt + 2 : nk - 1, CUR);  // (set B held tile kt: in LDS since the previous iteration; past the end: the last tile again)
    SDEH_FENCE();  // the loads stay at the top of the iteration (the scheduler otherwise sinks them behind the matrix instructions)
    const float* __restrict__ xa = Xs[B] + xoff;
    const float* __restrict__ wa = Ws[B] + woff;
    float av[kNgBK / 2], bv[kNgBK / 2];
#pragma unroll
    for (int s = 0; s < kNgBK / 2; ++s) {
      av[s] = wa[2 * s * SW];
      bv[s] = xa[2 * s * kNgSX];
    }
#pragma unroll
    for (int s = 0; s < kNgBK / 2; ++s) acc = SDEH_MFMA(av[s], bv[s], acc);
    SDEH_FENCE();  // ... and the only wait on global memory at its bottom, for the tile loaded an iteration ago (vmcnt(1 + NR): the newer tile stays in flight)
    store_tile(NXT, kt + 1 < nk ? kt + 1 : nk - 1);  // (LDS buffer 1 - B was last read in iteration kt - 1, behind that iteration's barrier; past the end: nobody reads it)
    __syncthreads();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    step(kt, I0{}, I1{});
    if (kt + 1 < nk) step(kt + 1, I1{}, I0{});
  }

  // ---- epilogue: lane (j, h) holds outputs n = n0 + 32 wn + 8 g + 4 h + e of batch row m ----------------------------------------------
  const long long m = m0 + 32 * wm + j;
  if (m >= G.M) return;
  const bool vec = (N & 3) == 0 && (G.ldy & 3) == 0 && (G.addend == nullptr || (G.lda & 3) == 0) && (G.mask == nullptr || (G.ldm & 3) == 0) &&
                   (G.bias == nullptr || (reinterpret_cast<unsigned long long>(G.bias) & 15) == 0);  // (a bias that is a view at an odd offset: element-wise)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = n0 + 32 * wn + 8 * g + 4 * h;
    if (n >= N) continue;
    float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    if (vec) {  // (n + 3 < N: both multiples of 4)
      if (G.bias != nullptr) {
        const float4 bb = *reinterpret_cast<const float4*>(G.bias + n);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
      }
      if (G.addend != nullptr) {
        const float4 aa = *reinterpret_cast<const float4*>(G.addend + m * G.lda + n);
        v[0] += aa.x; v[1] += aa.y; v[2] += aa.z; v[3] += aa.w;
      }
      if (G.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
      }
      if (G.mask != nullptr) {
        const float4 mm = *reinterpret_cast<const float4*>(G.mask + m * G.ldm + n);
        v[0] = mm.x > 0.0f ? v[0] : 0.0f; v[1] = mm.y > 0.0f ? v[1] : 0.0f;
        v[2] = mm.z > 0.0f ? v[2] : 0.0f; v[3] = mm.w > 0.0f ? v[3] : 0.0f;
      }
      *reinterpret_cast<float4*>(G.Y + m * G.ldy + n) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= N) continue;
        float y = v[e];
        if (G.bias != nullptr) y += G.bias[n + e];
        if (G.addend != nullptr) y += G.addend[m * G.lda + n + e];
        if (G.relu) y = fmaxf(y, 0.0f);
        if (G.mask != nullptr) y = G.mask[m * G.ldm + n + e] > 0.0f ? y : 0.0f;
        G.Y[m * G.ldy + n + e] = y;
      }
    }
  }
}

template <int BM>
static int nice_gemm_bm(const NiceGemm& g, bool trans_b, bool vec, hipStream_t st) {
  const dim3 grid((unsigned)((g.M + BM - 1) / BM), (unsigned)((g.N + kNgBN - 1) / kNgBN));
  if (trans_b) {
    if (vec) hipLaunchKernelGGL((nice_gemm_kernel<true, true, BM>), grid, dim3(8 * BM), 0, st, g);
    else hipLaunchKernelGGL((nice_gemm_kernel<true, false, BM>), grid, dim3(8 * BM), 0, st, g);
  } else {
    if (vec) hipLaunchKernelGGL((nice_gemm_kernel<false, true, BM>), grid, dim3(8 * BM), 0, st, g);
    else hipLaunchKernelGGL((nice_gemm_kernel<false, false, BM>), grid, dim3(8 * BM), 0, st, g);
  }
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

constexpr long long kNgSmallBatch = 4096;  // up to here 32-row tiles (measured, profiles/r06_nice_score_timing.txt; SDEH_NICE_BM forces a height: 0.92 vs 1.23 ms at B <= 2048, 1.30 vs 1.33 at 4096, 2.14 vs 2.02 at 8192)

static int nice_gemm(const NiceGemm& g, bool trans_b, hipStream_t st) {
  const bool al = (g.ldx & 3) == 0 && (g.ldw & 3) == 0 && ((reinterpret_cast<unsigned long long>(g.X) | reinterpret_cast<unsigned long long>(g.W)) & 15) == 0;
  // whole float4s in or out of range: the weights' contiguous extent a multiple of 4, at least one quad; X rows are zero-padded to a multiple of 4 by their owner
  const bool vec = al && (trans_b ? (g.K & 3) == 0 : (g.N & 3) == 0 && g.N >= 4) && g.K >= 4 && g.ldx >= ((g.K + 3) & ~3);
  static const char* const force = getenv("SDEH_NICE_BM");  // measurement aid, read once: "32" | "64"
  const bool small = force != nullptr ? force[0] == '3' : g.M <= kNgSmallBatch;
  return small ? nice_gemm_bm<32>(g, trans_b, vec, st) : nice_gemm_bm<64>(g, trans_b, vec, st);
}

// x [B, d] -> E = x[:, :, 0], O = x[:, :, 1] as [B, hp] (columns >= d / 2 zero)
__global__ void nice_split_kernel(const float* __restrict__ x, long long batch, int d, int hp, float* __restrict__ E, float* __restrict__ O) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch * hp) return;
  const long long b = i / hp;
  const int c = (int)(i % hp);
  const bool in = 2 * c < d;
  E[i] = in ? x[b * d + 2 * c] : 0.0f;
  O[i] = in ? x[b * d + 2 * c + 1] : 0.0f;
}

// the scaling layer and the logistic prior: one wave per row.  logp[b] = sum_j -(softplus(z_j) + softplus(-z_j)) + sum_j scale_j + lnc with
// z = x exp(scale); G = d logp / d (the coupling stack's output) = -tanh(z / 2) exp(scale)
__global__ __launch_bounds__(256) void nice_latent_kernel(const float* __restrict__ E, const float* __restrict__ O, long long batch, int d, int hp,
                                                          const float* __restrict__ scale, float lnc, float* __restrict__ logp,
                                                          float* __restrict__ GE, float* __restrict__ GO) {
  const int lane = threadIdx.x & 63;
  const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= batch) return;
  const int half = d / 2;
  float acc = 0.0f;
  for (int c = lane; c < hp; c += 64) {
    float ge = 0.0f, go = 0.0f;
    if (c < half) {
      const float s0 = scale[2 * c], s1 = scale[2 * c + 1];
      const float e0 = expf(s0), e1 = expf(s1);
      const float z0 = E[b * hp + c] * e0, z1 = O[b * hp + c] * e1;
      const float a0 = fabsf(z0), a1 = fabsf(z1);
      // softplus(z) + softplus(-z) = |z| + 2 log(1 + exp(-|z|))
      acc -= a0 + 2.0f * log1pf(expf(-a0));
      acc -= a1 + 2.0f * log1pf(expf(-a1));
      acc += s0 + s1;
      ge = -tanhf(0.5f * z0) * e0;
      go = -tanhf(0.5f * z1) * e1;
    }
    if (GE != nullptr) { GE[b * hp + c] = ge; GO[b * hp + c] = go; }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (logp != nullptr && lane == 0) logp[b] = acc + lnc;
}

__global__ void nice_merge_kernel(const float* __restrict__ GE, const float* __restrict__ GO, long long batch, int d, int hp, float* __restrict__ score) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch * d) return;
  const long long b = i / d;
  const int c = (int)(i % d);
  score[i] = (c & 1) ? GO[b * hp + (c >> 1)] : GE[b * hp + (c >> 1)];
}

long long nice_work_floats(const SdehNice& nn, long long batch, bool want_score) {
  const long long hp = (nn.dim / 2 + 3) & ~3, L = nn.n_mid + 1;
  const long long halves = (want_score ? 4 : 2) * batch * hp;
  const long long acts = (want_score ? (long long)nn.n_coupling * L + 2 : 2) * batch * nn.mid_dim;
  return halves + acts;
}

int launch_nice_eval(const SdehNice& nn, const float* x, long long batch, float* score, float* logp, float* work, hipStream_t st) {
  const int d = nn.dim, half = d / 2, hp = (half + 3) & ~3, mid = nn.mid_dim, L = nn.n_mid + 1;
  const bool ws = score != nullptr;
  float* E = work;
  float* O = E + batch * hp;
  float* GE = ws ? O + batch * hp : nullptr;
  float* GO = ws ? GE + batch * hp : nullptr;
  float* acts = (ws ? GO : O) + batch * hp;
  const long long plane = batch * mid;
  // score: H[c][l] kept for the reverse pass, then two adjoint planes; log-density only: two planes, ping-pong
  auto H = [&](int c, int l) { return ws ? acts + ((long long)c * L + l) * plane : acts + (long long)(l & 1) * plane; };
  float* dA[2] = {acts + (ws ? (long long)nn.n_coupling * L : 0) * plane, acts + (ws ? (long long)nn.n_coupling * L + 1 : 1) * plane};

  const long long ns = batch * hp;
  hipLaunchKernelGGL(nice_split_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, x, batch, d, hp, E, O);
  int rc = SDEH_OK;
  for (int c = 0; c < nn.n_coupling && rc == SDEH_OK; ++c) {
    float* on = nn.mask_config[c] ? E : O;
    float* off = nn.mask_config[c] ? O : E;
    NiceGemm g{};
    g.M = batch;
    g.X = off; g.ldx = hp; g.W = nn.in_w[c]; g.ldw = half; g.bias = nn.in_b[c]; g.relu = 1;
    g.Y = H(c, 0); g.ldy = mid; g.N = mid; g.K = half;
    rc = nice_gemm(g, true, st);
    for (int l = 1; l < L && rc == SDEH_OK; ++l) {
      g.X = H(c, l - 1); g.ldx = mid; g.W = nn.mid_w[c][l - 1]; g.ldw = mid; g.bias = nn.mid_b[c][l - 1];
      g.Y = H(c, l); g.K = mid;
      rc = nice_gemm(g, true, st);
    }
    if (rc != SDEH_OK) break;
    NiceGemm o{};
    o.M = batch;
    o.X = H(c, L - 1); o.ldx = mid; o.W = nn.out_w[c]; o.ldw = mid; o.bias = nn.out_b[c]; o.relu = 0;
    o.addend = on; o.lda = hp; o.Y = on; o.ldy = hp; o.N = half; o.K = mid;
    rc = nice_gemm(o, true, st);
  }
  if (rc != SDEH_OK) return rc;
  hipLaunchKernelGGL(nice_latent_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, st, E, O, batch, d, hp, nn.scale, nn.log_norm_const, logp, GE, GO);
  if (!ws) return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
  for (int c = nn.n_coupling - 1; c >= 0 && rc == SDEH_OK; --c) {
    float* g_on = nn.mask_config[c] ? GE : GO;
    float* g_off = nn.mask_config[c] ? GO : GE;
    int cur = 0;
    NiceGemm g{};
    g.M = batch;
    g.X = g_on; g.ldx = hp; g.W = nn.out_w[c]; g.ldw = mid; g.mask = H(c, L - 1); g.ldm = mid;
    g.Y = dA[cur]; g.ldy = mid; g.N = mid; g.K = half;
    rc = nice_gemm(g, false, st);
    for (int l = L - 1; l >= 1 && rc == SDEH_OK; --l) {
      g.X = dA[cur]; g.ldx = mid; g.W = nn.mid_w[c][l - 1]; g.ldw = mid; g.mask = H(c, l - 1);
      g.Y = dA[cur ^ 1]; g.K = mid;
      rc = nice_gemm(g, false, st);
      cur ^= 1;
    }
    if (rc != SDEH_OK) break;
    NiceGemm o{};
    o.M = batch;
    o.X = dA[cur]; o.ldx = mid; o.W = nn.in_w[c]; o.ldw = half;
    o.addend = g_off; o.lda = hp; o.Y = g_off; o.ldy = hp; o.N = half; o.K = mid;
    rc = nice_gemm(o, false, st);
  }
  if (rc != SDEH_OK) return rc;
  const long long nd = batch * d;
  hipLaunchKernelGGL(nice_merge_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, GE, GO, batch, d, hp, score);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

}  // namespace sdeh
