// Backward pass of a TimeEmbed sub-network (models/mlp.py:43-82) over its [T, .] table: the time embedding inside the
// FourierMLP and the gamma(t) network of the score controls depend on t only, so their parameter gradients are those of a small
// MLP evaluated at the T step times, with upstream gradient d loss / d table [T, dim_out] (the per-step sums of the backward
// kernels' planes).  The reference obtains them through autograd inside the one big graph; as ~50 tiny framework kernels per
// sub-network they were a quarter of a replayed training step (1.58 -> 1.17 ms with these parameters frozen).  Here: one
// workgroup per sub-network, rows in chunks of 32 through LDS, every gradient element owned by one thread (no atomics).
#include "sdeh_common.hpp"

namespace sdeh {

constexpr int kTeRows = 8;        // rows of t per chunk = per workgroup (1024 threads: the phases are short loops over 512 ... 8192 elements)
constexpr int kTeMaxHidden = 4;   // hidden layers this kernel keeps in LDS (the reference uses 1 and 3)

struct TeBwdArgs {
  SdehTimeEmbed te;   // parameters
  float* part;        // [n_chunks][P] per-chunk gradients, flat layout: phase | (w_k, b_k) for each hidden layer | out_w | out_b
  long long P;
  int act, n_steps;
  const float* ts;    // [n_steps]
  const float* gout;  // [n_steps, dim_out]
  float clip;         // |output| > clip: no gradient (torch.clamp's backward); +INF = no clamp
};

__device__ __forceinline__ float te_act(float v, int act) {
  if (act == SDEH_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  if (act == SDEH_ACT_SILU) return v / (1.0f + expf(-v));
  return fmaxf(v, 0.0f);
}
__device__ __forceinline__ float te_act_grad(float v, int act) {
  if (act == SDEH_ACT_GELU_ERF)
    return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * 0.3989422804014327f * expf(-0.5f * v * v);
  if (act == SDEH_ACT_SILU) {
    const float s = 1.0f / (1.0f + expf(-v));
    return s * fmaf(v, 1.0f - s, 1.0f);
  }
  return v > 0.0f ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(1024) void time_embed_bwd_kernel(const TeBwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int C = A.te.channels, H = A.te.n_hidden, DO = A.te.dim_out, T = A.n_steps, act = A.act;
  const int tid = threadIdx.x, nt = blockDim.x;
  float* feat = sh;                              // [kTeRows][2C]
  float* zb = feat + kTeRows * 2 * C;            // [H][kTeRows][C]   pre-activations, later d loss / d pre-activation
  float* ab = zb + H * kTeRows * C;              // [H][kTeRows][C]   activations
  float* gb = ab + H * kTeRows * C;              // [kTeRows][DO]     upstream gradient of the chunk
  float* wbuf = gb + kTeRows * DO;               // [C][2C + 1]       the current layer's weight, staged coalesced, padded rows
  // lanes index the output channel in the forward (rows of W, stride nin + 1: conflict-free) and the input channel in the
  // backward (columns): reading W straight from global memory costs a cache line per lane and k-step
  auto stage = [&](const float* __restrict__ w, int rows, int nin) {
    __syncthreads();
    for (int i = tid; i < rows * nin; i += nt) wbuf[(i / nin) * (nin + 1) + i % nin] = w[i];
    __syncthreads();
  };
  auto W = [&](int k) { return A.te.hidden_w[k]; };
  // flat layout of this chunk's gradients
  float* base = A.part + (long long)blockIdx.x * A.P;
  float* g_phase = base;
  auto gw = [&](int k) { return base + C + (k == 0 ? 0 : (long long)(2 * C * C + C) + (long long)(k - 1) * (C * C + C)); };
  auto gbias = [&](int k) { return gw(k) + (long long)C * (k == 0 ? 2 * C : C); };
  float* g_ow = gw(H);                       // one past the last hidden layer
  float* g_ob = g_ow + (long long)DO * C;

  {
    const int r0 = blockIdx.x * kTeRows;
    const int nr = T - r0 < kTeRows ? T - r0 : kTeRows;
    // ---- features [sin(c t + phi), cos(c t + phi)]
    for (int i = tid; i < nr * C; i += nt) {
      const int r = i / C, c = i % C;
      const float arg = A.te.coeff[c] * A.ts[r0 + r] + A.te.phase[c];
      feat[r * 2 * C + c] = sinf(arg);
      feat[r * 2 * C + C + c] = cosf(arg);
    }
    __syncthreads();
    // ---- forward through the hidden layers
    for (int k = 0; k < H; ++k) {
      const int nin = k == 0 ? 2 * C : C;
      const float* in = k == 0 ? feat : ab + (k - 1) * kTeRows * C;
      stage(W(k), C, nin);
      const float* __restrict__ b = A.te.hidden_b[k];
      for (int i = tid; i < nr * C; i += nt) {
        const int r = i / C, c = i % C;
        float acc = b[c];
        const float* wr = wbuf + c * (nin + 1);
        const float* ir = in + r * nin;
        for (int j = 0; j < nin; ++j) acc = fmaf(wr[j], ir[j], acc);
        zb[(k * kTeRows + r) * C + c] = acc;
        ab[(k * kTeRows + r) * C + c] = te_act(acc, act);
      }
      __syncthreads();
    }
    const float* alast = ab + (H - 1) * kTeRows * C;
    // ---- output (for the clamp mask) and upstream gradient of the chunk
    stage(A.te.out_w, DO, C);
    for (int i = tid; i < nr * DO; i += nt) {
      const int r = i / DO, o = i % DO;
      float acc = A.te.out_b[o];
      const float* wr = wbuf + o * (C + 1);
      for (int c = 0; c < C; ++c) acc = fmaf(wr[c], alast[r * C + c], acc);
      const float g = A.gout[(size_t)(r0 + r) * DO + o];
      gb[r * DO + o] = (acc >= -A.clip && acc <= A.clip) ? g : 0.0f;
    }
    __syncthreads();
    // ---- out layer: weight / bias gradients; d loss / d a_last -> d loss / d z_last (in place of z_last)
    for (int i = tid; i < DO * C; i += nt) {
      const int o = i / C, c = i % C;
      float s = 0.0f;
      for (int r = 0; r < nr; ++r) s = fmaf(gb[r * DO + o], alast[r * C + c], s);
      g_ow[i] = s;
    }
    for (int o = tid; o < DO; o += nt) {
      float s = 0.0f;
      for (int r = 0; r < nr; ++r) s += gb[r * DO + o];
      g_ob[o] = s;
    }
    for (int i = tid; i < nr * C; i += nt) {
      const int r = i / C, c = i % C;
      float da = 0.0f;
      for (int o = 0; o < DO; ++o) da = fmaf(gb[r * DO + o], wbuf[o * (C + 1) + c], da);
      float* z = zb + ((H - 1) * kTeRows + r) * C + c;
      *z = da * te_act_grad(*z, act);
    }
    __syncthreads();
    // ---- back through the hidden layers
    for (int k = H - 1; k >= 0; --k) {
      const int nin = k == 0 ? 2 * C : C;
      const float* in = k == 0 ? feat : ab + (k - 1) * kTeRows * C;
      const float* dz = zb + k * kTeRows * C;
      for (int i = tid; i < C * nin; i += nt) {
        const int c = i / nin, j = i % nin;
        float s = 0.0f;
        for (int r = 0; r < nr; ++r) s = fmaf(dz[r * C + c], in[r * nin + j], s);
        gw(k)[i] = s;
      }
      for (int c = tid; c < C; c += nt) {
        float s = 0.0f;
        for (int r = 0; r < nr; ++r) s += dz[r * C + c];
        gbias(k)[c] = s;
      }
      stage(W(k), C, nin);  // (its barriers also separate the reads of a[k-1] / feat above from the in-place updates below)
      if (k > 0) {
        for (int i = tid; i < nr * C; i += nt) {
          const int r = i / C, j = i % C;
          float da = 0.0f;
          for (int c = 0; c < C; ++c) da = fmaf(dz[r * C + c], wbuf[c * (C + 1) + j], da);
          float* z = zb + ((k - 1) * kTeRows + r) * C + j;
          *z = da * te_act_grad(*z, act);
        }
      } else {
        // d / d phase_c = sum_r [ dfeat_sin(r,c) cos(arg) - dfeat_cos(r,c) sin(arg) ]; per-row terms parked in a[0] (free now)
        for (int i = tid; i < nr * C; i += nt) {
          const int r = i / C, c = i % C;
          float ds = 0.0f, dc = 0.0f;
          for (int cc = 0; cc < C; ++cc) {
            const float d = dz[r * C + cc];
            ds = fmaf(d, wbuf[cc * (2 * C + 1) + c], ds);
            dc = fmaf(d, wbuf[cc * (2 * C + 1) + C + c], dc);
          }
          ab[r * C + c] = ds * feat[r * 2 * C + C + c] - dc * feat[r * 2 * C + c];
        }
        __syncthreads();
        for (int c = tid; c < C; c += nt) {
          float s = 0.0f;
          for (int r = 0; r < nr; ++r) s += ab[r * C + c];
          g_phase[c] = s;
        }
      }
      __syncthreads();
    }
  }
}

long long time_embed_param_floats(const SdehTimeEmbed& te) {
  const long long C = te.channels, H = te.n_hidden, DO = te.dim_out;
  return C + (2 * C * C + C) + (H - 1) * (C * C + C) + DO * C + DO;
}

int launch_partial_sums(const float* part, long long n_items, long long n_chunks, long long width, float* scratch, float* out,
                        hipStream_t stream);

// workspace: n_chunks * P partials, then the scratch of the sum over the chunks
int launch_time_embed_bwd(const SdehTimeEmbed& te, int act, const float* ts, int n_steps, const float* gout, float clip,
                          float* workspace, float* grad_flat, hipStream_t stream) {
  if (te.n_hidden < 1 || te.n_hidden > kTeMaxHidden) return SDEH_ERR_UNSUPPORTED;
  const int wrows = te.channels > te.dim_out ? te.channels : te.dim_out;
  const size_t lds = ((size_t)kTeRows * (2 * te.channels + 2 * te.n_hidden * te.channels + te.dim_out) +
                      (size_t)wrows * (2 * te.channels + 1)) * sizeof(float);
  if (lds > 64 * 1024) return SDEH_ERR_UNSUPPORTED;
  const long long P = time_embed_param_floats(te);
  const long long n_chunks = (n_steps + kTeRows - 1) / kTeRows;
  TeBwdArgs A;
  A.te = te; A.part = workspace; A.P = P; A.act = act; A.n_steps = n_steps; A.ts = ts; A.gout = gout; A.clip = clip;
  hipLaunchKernelGGL(time_embed_bwd_kernel, dim3((unsigned)n_chunks), dim3(1024), lds, stream, A);
  if (hipGetLastError() != hipSuccess) return SDEH_ERR_HIP;
  return launch_partial_sums(workspace, 1, n_chunks, P, workspace + n_chunks * P, grad_flat, stream);
}

long long time_embed_workspace_floats(const SdehTimeEmbed& te, int n_steps) {
  const long long P = time_embed_param_floats(te);
  const long long n_chunks = (n_steps + kTeRows - 1) / kTeRows;
  return n_chunks * P + ((n_chunks + 31) / 32) * P;
}

}  // namespace sdeh
