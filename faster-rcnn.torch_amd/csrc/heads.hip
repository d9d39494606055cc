// heads.hip -- the anchor nets' TRAINING path on the sampled positions only, every anchor net in one launch per step of the chain.
//
// objective.lua:91-140 reads the anchor nets' outputs (models/model_utilities.lua:31-34: k x k valid convolution, PReLU, 1 x 1
// convolution to 18 planes) at the sampled anchors only -- a few hundred positions of maps with thousands -- and
// delta_outputs[1..4] are non-zero only there, so neither the forward nor the backward pass of an anchor net needs the rest of
// its map while training (Detector.lua:33 does: the evaluate-mode path stays the dense convolution).  Per anchor net, with P
// sampled positions, n filters, ckk = Cin k k:
//   forward : COL[P][ckk] = patches of the input map | HX[n][P] = W COL^T + b (split over K: partial sums folded by the next
//             kernel) | HY = prelu(HX) | OUT[18][P] = W1 HY | out map[.][pos] = OUT + b1
//   loss    : frcnn_rpn_loss on the maps (rpn.hip), as before
//   backward: D[18][P] = delta map[.][pos], gb1 += rowsum | GH[n][P] = W1^T D | GH *= prelu'(HX), gb += rowsum, gslope += ...
//             | DX[P][ckk] = GH^T W | input-map gradient += DX (atomics) | gW1 += D HY^T | gW += GH COL
// Rounds 1-5 ran the forward part as dense convolutions on a stream per anchor net and the backward part as ten launches per
// anchor net on those streams: 58 launches on five streams, which the runtime multiplexes onto four hardware queues together
// with the caller's stream -- the chains serialised in pairs and the caller's stream waited for them (tools/r6_hwq.sh).  Here
// every step of the chain is ONE launch for all anchor nets (blockIdx.y = the anchor net; the products: gemm_f32_group), the whole
// chain runs on one stream: 13 launches.
#include "kernels.h"

namespace frcnn {

__device__ __forceinline__ float hd_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double hd_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// COL[p][(c, ky, kx)] = in[c][y_p + ky][x_p + kx]   (valid convolution: always inside the map); pos = y * Wo + x
__global__ __launch_bounds__(256) void hd_im2col_kernel(HeadJobs g) {
  const HeadJob& j = g.j[blockIdx.y];
  const int kk = j.k * j.k, ckk = j.Cin * kk;
  const long total = (long)j.P * ckk;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int pi = (int)(t / ckk), q = (int)(t - (long)pi * ckk);
    const int kx = q % j.k, ky = (q / j.k) % j.k, c = q / kk;
    const int y = j.pos[pi] / j.Wo, x = j.pos[pi] - y * j.Wo;
    j.COL[t] = j.in[((size_t)c * j.H + y + ky) * j.W + x + kx];
  }
}

// HX[c][p] = bias3[c] + sum_s slab[s][c][p] (the K splits of the product, in order), HY = prelu(HX)
__global__ __launch_bounds__(256) void hd_bias_act_kernel(HeadJobs g) {
  const HeadJob& j = g.j[blockIdx.y];
  const long total = (long)j.n * j.P;
  const float a = *j.slope;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int c = (int)(t / j.P);
    float v = j.bias3[c];
    int s = 0;
    for (; s + 7 < j.hx_splits; s += 8) {   // (eight independent loads in flight; summed in split order)
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = j.hx_slab[(size_t)(s + u) * total + t];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += x[u];
    }
    for (; s < j.hx_splits; ++s) v += j.hx_slab[(size_t)s * total + t];
    j.HX[t] = v;
    j.HY[t] = v > 0.f ? v : a * v;
  }
}

// out[c][pos[p]] = OUT[c][p] + bias1[c]
__global__ __launch_bounds__(256) void hd_scatter_kernel(HeadJobs g) {
  const HeadJob& j = g.j[blockIdx.y];
  const long hw1 = (long)j.Ho * j.Wo, total = (long)FRCNN_HEAD_OUT * j.P;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int c = (int)(t / j.P), pi = (int)(t - (long)c * j.P);
    j.out[(size_t)c * hw1 + j.pos[pi]] = j.OUT[t] + j.bias1[c];
  }
}

// D[c][p] = delta[c][pos[p]], gbias1[c] += sum_p D[c][p]: one block per output plane
__global__ __launch_bounds__(256) void hd_gather_delta_kernel(HeadJobs g) {
  __shared__ float sh[4];
  const HeadJob& j = g.j[blockIdx.y];
  const int c = blockIdx.x;
  const long hw1 = (long)j.Ho * j.Wo;
  float sb = 0.f;
  for (int pi = threadIdx.x; pi < j.P; pi += 256) {
    const float v = j.delta[(size_t)c * hw1 + j.pos[pi]];
    j.D[(size_t)c * j.P + pi] = v;
    sb += v;
  }
  sb = hd_wave_sum(sb);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sb;
  __syncthreads();
  if (threadIdx.x == 0 && j.P > 0) unsafeAtomicAdd(j.gbias1 + c, (sh[0] + sh[1]) + (sh[2] + sh[3]));
}

// PReLU backward on [n][P]: GH[c][p] *= (HX > 0 ? 1 : slope); gbias3[c] += sum_p of the result; gslope += sum over HX <= 0 of
// HX * (incoming gradient) (carried in double: one number summed over a layer with cancellation).  One block per filter.
__global__ __launch_bounds__(256) void hd_act_backward_kernel(HeadJobs g) {
  __shared__ float sh[4];
  __shared__ double shd[4];
  const HeadJob& j = g.j[blockIdx.y];
  const int c = blockIdx.x;
  if (c >= j.n) return;
  const float a = *j.slope;
  float sb = 0.f;
  double sa = 0.0;
  for (int pi = threadIdx.x; pi < j.P; pi += 256) {
    const size_t t = (size_t)c * j.P + pi;
    const float gy = j.GH[t], x = j.HX[t];
    float r = gy;
    if (!(x > 0.f)) { r = a * gy; sa += (double)x * (double)gy; }
    j.GH[t] = r;
    sb += r;
  }
  sb = hd_wave_sum(sb);
  sa = hd_wave_sum_d(sa);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = sb; shd[threadIdx.x >> 6] = sa; }
  __syncthreads();
  if (threadIdx.x == 0 && j.P > 0) {
    unsafeAtomicAdd(j.gbias3 + c, (sh[0] + sh[1]) + (sh[2] + sh[3]));
    unsafeAtomicAdd(j.gslope, (float)((shd[0] + shd[1]) + (shd[2] + shd[3])));
  }
}

// gin[c][y_p + ky][x_p + kx] += DX[p][(c, ky, kx)]   (the patches of different anchors overlap: atomics)
__global__ __launch_bounds__(256) void hd_col2im_kernel(HeadJobs g) {
  const HeadJob& j = g.j[blockIdx.y];
  const int kk = j.k * j.k, ckk = j.Cin * kk;
  const long total = (long)j.P * ckk;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int pi = (int)(t / ckk), q = (int)(t - (long)pi * ckk);
    const int kx = q % j.k, ky = (q / j.k) % j.k, c = q / kk;
    const int y = j.pos[pi] / j.Wo, x = j.pos[pi] - y * j.Wo;
    unsafeAtomicAdd(j.gin + ((size_t)c * j.H + y + ky) * j.W + x + kx, j.DX[t]);
  }
}

static int hd_grid(const HeadJobs& g, int mode) {   // blocks along x: enough for the largest job, grid-stride inside
  long most = 1;
  for (int i = 0; i < g.n; ++i) {
    const HeadJob& j = g.j[i];
    const long t = mode == 0 ? (long)j.P * j.Cin * j.k * j.k : mode == 1 ? (long)j.n * j.P : (long)FRCNN_HEAD_OUT * j.P;
    most = std::max(most, t);
  }
  return (int)std::min<long>(cdivl(most, 256), 4096);
}

int heads_im2col(const HeadJobs& g, hipStream_t s) {
  FR_LAUNCH(KC_ELEMWISE, 0, 0, s, hd_im2col_kernel, dim3(hd_grid(g, 0), g.n), dim3(256), 0, g);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}
int heads_bias_act(const HeadJobs& g, hipStream_t s) {
  FR_LAUNCH(KC_ELEMWISE, 0, 0, s, hd_bias_act_kernel, dim3(hd_grid(g, 1), g.n), dim3(256), 0, g);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}
int heads_scatter(const HeadJobs& g, hipStream_t s) {
  FR_LAUNCH(KC_ELEMWISE, 0, 0, s, hd_scatter_kernel, dim3(hd_grid(g, 2), g.n), dim3(256), 0, g);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}
int heads_gather_delta(const HeadJobs& g, hipStream_t s) {
  FR_LAUNCH(KC_ELEMWISE, 0, 0, s, hd_gather_delta_kernel, dim3(FRCNN_HEAD_OUT, g.n), dim3(256), 0, g);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}
int heads_act_backward(const HeadJobs& g, hipStream_t s) {
  int nmax = 1;
  for (int i = 0; i < g.n; ++i) nmax = std::max(nmax, g.j[i].n);
  FR_LAUNCH(KC_ELEMWISE, 0, 0, s, hd_act_backward_kernel, dim3(nmax, g.n), dim3(256), 0, g);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}
int heads_col2im(const HeadJobs& g, hipStream_t s) {
  FR_LAUNCH(KC_ELEMWISE, 0, 0, s, hd_col2im_kernel, dim3(hd_grid(g, 0), g.n), dim3(256), 0, g);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

}  // namespace frcnn
