// gemm.hip -- nn.Linear forward / backward of the classification network
// (models/model_utilities.lua:82,99,103; objective.lua:164,179; Detector.lua:101) as one strided
// fp32 GEMM on v_mfma_f32_32x32x2_f32.  C[M][N] (=|+=) A[M][K] * B[K][N] (+ bias[n]).
//   forward : Y[R][O]  = X[R][I]  * W[O][I]^T      A k-contiguous, B k-contiguous
//   dgrad   : gX[R][I] = gY[R][O] * W[O][I]        A k-contiguous, B n-contiguous
//   wgrad   : gW[O][I] += gY[R][O]^T * X[R][I]     A m-contiguous, B n-contiguous
// 64x64 block tile (2x2 waves of one 32x32 MFMA tile), BK = 32, split-K over gridDim.z with
// fp32 atomics when the tile grid alone cannot fill 256 CUs (R is a few hundred rows at most).
#include "kernels.h"

namespace frcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GB 64
#define GBK 32
#define GP 65  // LDS pitch (odd)

struct GemmArgs {
  const float* A; long sAm, sAk;
  const float* B; long sBk, sBn;
  float* C; long ldc;
  const float* bias;
  int M, N, K, kPerSplit, out_mode;  // 0 store, 1 add, 2 atomic
};

// Operand tile loader: 64 (rows: m or n) x 32 (k) floats -> LDS image T[k][row] (pitch GP).
// KC: the operand is contiguous along k (else along the row index).  VEC: 16-byte loads are legal
// (strides multiple of 4 floats, base 16-byte aligned).  All loads of the tile are issued first with
// clamped indices, zero fill by select, then the LDS writes: one latency exposure per k-block.
template <bool KC, bool VEC>
struct TileLoader {
  float4 v4[2];
  float v1[8];
  __device__ __forceinline__ void load(const float* __restrict__ P, long sRow, long sK, int row0, int nRows, int k0,
                                       int kEnd, int tid) {
    if (VEC) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        int r, k;
        if (KC) { k = (tid & 7) * 4; r = (tid >> 3) + 32 * it; }
        else    { r = (tid & 15) * 4; k = (tid >> 4) + 16 * it; }
        const int rr = min(row0 + r, nRows - (KC ? 1 : 4)), kk = min(k0 + k, kEnd - (KC ? 4 : 1));
        v4[it] = *reinterpret_cast<const float4*>(P + (long)rr * sRow + (long)kk * sK);
      }
    } else {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        int r, k;
        if (KC) { k = tid & 31; r = (tid >> 5) + 8 * it; }
        else    { r = tid & 63; k = (tid >> 6) + 4 * it; }
        const int rr = min(row0 + r, nRows - 1), kk = min(k0 + k, kEnd - 1);
        v1[it] = P[(long)rr * sRow + (long)kk * sK];
      }
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ T, int row0, int nRows, int k0, int kEnd, int tid) {
    if (VEC) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        int r, k;
        if (KC) { k = (tid & 7) * 4; r = (tid >> 3) + 32 * it; }
        else    { r = (tid & 15) * 4; k = (tid >> 4) + 16 * it; }
        const float e[4] = {v4[it].x, v4[it].y, v4[it].z, v4[it].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rj = KC ? r : r + j, kj = KC ? k + j : k;
          // VEC tiles are only used when every 4-group is entirely inside or outside the matrix
          T[kj * GP + rj] = (row0 + rj < nRows && k0 + kj < kEnd) ? e[j] : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        int r, k;
        if (KC) { k = tid & 31; r = (tid >> 5) + 8 * it; }
        else    { r = tid & 63; k = (tid >> 6) + 4 * it; }
        T[k * GP + r] = (row0 + r < nRows && k0 + k < kEnd) ? v1[it] : 0.f;
      }
    }
  }
};

template <bool AK, bool BK_, bool AV, bool BV>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  __shared__ float As[GBK * GP];
  __shared__ float Bs[GBK * GP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, li = lane & 31;
  const int m0 = blockIdx.y * GB, n0 = blockIdx.x * GB;
  const int kbeg = blockIdx.z * p.kPerSplit;
  const int kend = min(kbeg + p.kPerSplit, p.K);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  TileLoader<AK, AV> la;
  TileLoader<BK_, BV> lb;

  for (int k0 = kbeg; k0 < kend; k0 += GBK) {
    la.load(p.A, p.sAm, p.sAk, m0, p.M, k0, kend, tid);
    lb.load(p.B, p.sBn, p.sBk, n0, p.N, k0, kend, tid);
    la.store(As, m0, p.M, k0, kend, tid);
    lb.store(Bs, n0, p.N, k0, kend, tid);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GBK / 2; ++kk) {
      float a = As[(kk * 2 + h) * GP + wm * 32 + li];
      float b = Bs[(kk * 2 + h) * GP + wn * 32 + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = n0 + wn * 32 + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (m < p.M && n < p.N) {
      float v = acc[r];
      if (p.bias && blockIdx.z == 0) v += p.bias[n];
      float* dst = p.C + (long)m * p.ldc + n;
      if (p.out_mode == 0) *dst = v;
      else if (p.out_mode == 1) *dst += v;
      else unsafeAtomicAdd(dst, v);
    }
  }
}

__global__ void gemm_init_kernel(float* C, long ldc, int M, int N, const float* bias) {
  long total = (long)M * N;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    int m = (int)(t / N), n = (int)(t % N);
    C[(long)m * ldc + n] = bias ? bias[n] : 0.f;
  }
}

int gemm_f32(const float* A, long sAm, long sAk, const float* B, long sBk, long sBn, float* C,
             long ldc, int M, int N, int K, int out_mode, const float* bias_n, hipStream_t s) {
  if (M <= 0 || N <= 0) return FRCNN_OK;
  GemmArgs p;
  p.A = A; p.sAm = sAm; p.sAk = sAk; p.B = B; p.sBk = sBk; p.sBn = sBn; p.C = C; p.ldc = ldc;
  p.bias = bias_n; p.M = M; p.N = N; p.K = K;
  int tm = cdiv(M, GB), tn = cdiv(N, GB);
  long tiles = (long)tm * tn;
  int splitK = 1;
  if (tiles < 512) splitK = (int)std::max<long>(1, std::min<long>(cdiv(K, 2 * GBK), cdivl(768, tiles)));
  p.kPerSplit = cdiv(cdiv(K, splitK), GBK) * GBK;
  splitK = cdiv(K, p.kPerSplit);
  p.out_mode = out_mode;
  if (splitK > 1) {
    if (out_mode == OUT_STORE) {
      long total = (long)M * N;
      int grid = (int)std::min<long>(cdivl(total, 256), 1024);
      FR_LAUNCH(KC_ELEMWISE, 0, total * 4.0, s, gemm_init_kernel, dim3(grid), dim3(256), 0, C, ldc, M, N,
                bias_n);
      p.bias = nullptr;
    }
    p.out_mode = 2;
  }
  dim3 grid(tn, tm, splitK);
  double flops = 2.0 * M * N * (double)K;
  double bytes = 4.0 * ((double)M * K + (double)K * N + (double)M * N);
  const bool ak = sAk == 1, bk = sBk == 1;
  // 16-byte loads need: contiguous dimension a multiple of 4 inside the matrix, strides multiple of 4, aligned base
  auto vec_ok = [](const float* P, bool kc, long sRow, long sK, int nRows, int Kdim, int kPer) {
    if (((uintptr_t)P & 15) != 0) return false;
    if (kc) return sK == 1 && (sRow % 4) == 0 && (Kdim % 4) == 0 && (kPer % 4) == 0;
    return sRow == 1 && (sK % 4) == 0 && (nRows % 4) == 0;
  };
  const bool av = vec_ok(A, ak, sAm, sAk, M, K, p.kPerSplit), bv = vec_ok(B, bk, sBn, sBk, N, K, p.kPerSplit);
  const int sel = (ak ? 8 : 0) | (bk ? 4 : 0) | (av ? 2 : 0) | (bv ? 1 : 0);
#define GEMM_CASE(AKv, BKv, AVv, BVv)                                                                        \
  case ((AKv ? 8 : 0) | (BKv ? 4 : 0) | (AVv ? 2 : 0) | (BVv ? 1 : 0)):                                      \
    FR_LAUNCH(KC_GEMM, flops, bytes, s, (gemm_kernel<AKv, BKv, AVv, BVv>), grid, dim3(256), 0, p);           \
    break;
  switch (sel) {
    GEMM_CASE(true, true, true, true) GEMM_CASE(true, true, true, false) GEMM_CASE(true, true, false, true)
    GEMM_CASE(true, true, false, false) GEMM_CASE(true, false, true, true) GEMM_CASE(true, false, true, false)
    GEMM_CASE(true, false, false, true) GEMM_CASE(true, false, false, false) GEMM_CASE(false, true, true, true)
    GEMM_CASE(false, true, true, false) GEMM_CASE(false, true, false, true) GEMM_CASE(false, true, false, false)
    GEMM_CASE(false, false, true, true) GEMM_CASE(false, false, true, false) GEMM_CASE(false, false, false, true)
    GEMM_CASE(false, false, false, false)
  }
#undef GEMM_CASE
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

}  // namespace frcnn
