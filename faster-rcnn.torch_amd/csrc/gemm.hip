// gemm.hip -- nn.Linear forward / backward of the classification network
// (models/model_utilities.lua:82,99,103; objective.lua:164,179; Detector.lua:101) as one strided
// fp32 GEMM on v_mfma_f32_32x32x2_f32.  C[M][N] (=|+=) A[M][K] * B[K][N] (+ bias[n]).
//   forward : Y[R][O]  = X[R][I]  * W[O][I]^T      A k-contiguous, B k-contiguous
//   dgrad   : gX[R][I] = gY[R][O] * W[O][I]        A k-contiguous, B n-contiguous
//   wgrad   : gW[O][I] += gY[R][O]^T * X[R][I]     A m-contiguous, B n-contiguous
// 64-row block tiles (2x2 waves of 32x32 MFMA tiles), BK = 32, split-K over gridDim.z into slabs + one
// reduce pass when the tile grid alone cannot fill 256 CUs (R is a few hundred rows at most).
#include <cstdlib>
#include "kernels.h"

namespace frcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GB 64
#ifndef GBK
#define GBK 32
#endif

struct GemmGroup { GemmJob job[GEMM_GROUP_MAX]; int start[GEMM_GROUP_MAX + 1]; int tm[GEMM_GROUP_MAX], tn[GEMM_GROUP_MAX]; int n; };

struct GemmArgs {
  const float* A; long sAm, sAk;
  const float* B; long sBk, sBn;
  float* C; long ldc;
  const float* bias;
  int M, N, K, kPerSplit, out_mode;  // 0 store, 1 add, 3 split-K slab
};

// One 32 x 32 accumulator tile to memory: rows m = mb + (r & 3) + 8 (r >> 2) (mb includes the lane half's 4 h), column n.
// The bias and -- in the accumulate mode -- the 16 old values are read in one batch with clamped addresses; a test, a bias
// load and an old-value load per element made the compiler emit load - wait - add - store sixteen times per tile.
__device__ __forceinline__ void gemm_store_tile(const f32x16& a, const GemmArgs& p, int mb, int n, int split) {
  const bool nok = n < p.N;
  const int nc = min(n, p.N - 1);
  const float bv = (p.bias && split == 0) ? p.bias[nc] : 0.f;
  float* const col = (p.out_mode == 3 ? p.C + (long)split * p.M * p.N : p.C) + nc;   // 3: split-K slab [split][M][N]
  const long rs = p.out_mode == 3 ? (long)p.N : p.ldc;
  if (p.out_mode == 1) {
    float old[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) old[r] = col[(long)min(mb + (r & 3) + 8 * (r >> 2), p.M - 1) * rs];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mb + (r & 3) + 8 * (r >> 2);
      if (nok && m < p.M) col[(long)m * rs] = old[r] + (a[r] + bv);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mb + (r & 3) + 8 * (r >> 2);
      if (nok && m < p.M) col[(long)m * rs] = a[r] + bv;
    }
  }
}


// Operand tile loader: ROWS (m or n) x GBK (k) floats -> LDS image T[k][row] (pitch ROWS+1).
// KC: the operand is contiguous along k (else along the row index).  VEC: 16-byte loads are legal
// (strides multiple of 4 floats, base 16-byte aligned).  All loads of the tile are issued first with
// clamped indices, zero fill by select, then the LDS writes: one latency exposure per k-block.
template <int ROWS, bool KC, bool VEC>
struct TileLoader {
  static constexpr int NV = ROWS * GBK / 1024;   // float4 per thread
  static constexpr int NS = ROWS * GBK / 256;    // scalars per thread
  static constexpr int PITCH = ROWS + 1;
  float4 v4[NV];
  float v1[VEC ? 1 : NS];
  __device__ __forceinline__ static void coord_v(int tid, int it, int& r, int& k) {
    if (KC) { k = (tid % (GBK / 4)) * 4; r = tid / (GBK / 4) + (1024 / GBK) * it; }
    else    { r = (tid % (ROWS / 4)) * 4; k = tid / (ROWS / 4) + (1024 / ROWS) * it; }
  }
  __device__ __forceinline__ static void coord_s(int tid, int it, int& r, int& k) {
    if (KC) { k = tid % GBK; r = tid / GBK + (256 / GBK) * it; }
    else    { r = tid % ROWS; k = tid / ROWS + (256 / ROWS) * it; }
  }
  __device__ __forceinline__ void load(const float* __restrict__ P, long sRow, long sK, int row0, int nRows, int k0,
                                       int kEnd, int tid) {
    if (VEC) {
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        int r, k;
        coord_v(tid, it, r, k);
        const int rr = min(row0 + r, nRows - (KC ? 1 : 4)), kk = min(k0 + k, kEnd - (KC ? 4 : 1));
        v4[it] = *reinterpret_cast<const float4*>(P + (long)rr * sRow + (long)kk * sK);
      }
    } else {
#pragma unroll
      for (int it = 0; it < NS; ++it) {
        int r, k;
        coord_s(tid, it, r, k);
        const int rr = min(row0 + r, nRows - 1), kk = min(k0 + k, kEnd - 1);
        v1[it] = P[(long)rr * sRow + (long)kk * sK];
      }
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ T, int row0, int nRows, int k0, int kEnd, int tid) {
    if (VEC) {
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        int r, k;
        coord_v(tid, it, r, k);
        const float e[4] = {v4[it].x, v4[it].y, v4[it].z, v4[it].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rj = KC ? r : r + j, kj = KC ? k + j : k;
          // VEC tiles are only used when every 4-group is entirely inside or outside the matrix
          T[kj * PITCH + rj] = (row0 + rj < nRows && k0 + kj < kEnd) ? e[j] : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < NS; ++it) {
        int r, k;
        coord_s(tid, it, r, k);
        T[k * PITCH + r] = (row0 + r < nRows && k0 + k < kEnd) ? v1[it] : 0.f;
      }
    }
  }
};

// Block tile TM x TN (64 or 128 each), 2x2 waves, wave tile (TM/2) x (TN/2) of 32x32 MFMA tiles.  (bx, by, bz) = the block's
// column tile, row tile and K split: blockIdx for a launch of ONE product, decoded from a job table for a grouped launch.
template <int TM, int TN, bool AK, bool BK_, bool AV, bool BV>
__device__ __forceinline__ void gemm_tile(const GemmArgs& p, int bx, int by, int bz, float* As, float* Bs) {
  constexpr int PA = TM + 1, PB = TN + 1, MT = TM / 64, NT = TN / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, li = lane & 31;
  const int m0 = by * TM, n0 = bx * TN;
  const int kbeg = bz * p.kPerSplit;
  const int kend = min(kbeg + p.kPerSplit, p.K);
  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  TileLoader<TM, AK, AV> la;
  TileLoader<TN, BK_, BV> lb;

  for (int k0 = kbeg; k0 < kend; k0 += GBK) {
    la.load(p.A, p.sAm, p.sAk, m0, p.M, k0, kend, tid);
    lb.load(p.B, p.sBn, p.sBk, n0, p.N, k0, kend, tid);
    la.store(As, m0, p.M, k0, kend, tid);
    lb.store(Bs, n0, p.N, k0, kend, tid);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GBK / 2; ++kk) {
      float a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = As[(kk * 2 + h) * PA + wm * (TM / 2) + i * 32 + li];
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = Bs[(kk * 2 + h) * PB + wn * (TN / 2) + j * 32 + li];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * (TN / 2) + j * 32 + li;
#pragma unroll
    for (int i = 0; i < MT; ++i)
      gemm_store_tile(acc[i][j], p, m0 + wm * (TM / 2) + i * 32 + 4 * h, n, bz);
  }
}

template <int TM, int TN, bool AK, bool BK_, bool AV, bool BV>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  __shared__ float As[GBK * (TM + 1)];
  __shared__ float Bs[GBK * (TN + 1)];
  gemm_tile<TM, TN, AK, BK_, AV, BV>(p, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// Several independent products in ONE launch (the anchor nets' sparse training path, net.cpp): job j owns the blocks
// start[j] .. start[j + 1], numbered column tile fastest, then row tile, then K split.  64 x 64 tiles, scalar loads (the
// operands sit at any offset of the flat parameter vector); every job of a launch has the same operand orientations.
template <bool AK, bool BK_>
__global__ __launch_bounds__(256) void gemm_group_kernel(GemmGroup g) {
  __shared__ float As[GBK * (GB + 1)];
  __shared__ float Bs[GBK * (GB + 1)];
  int j = 0;
  while (j + 1 < g.n && (int)blockIdx.x >= g.start[j + 1]) ++j;
  const int b = blockIdx.x - g.start[j];
  const int tiles = g.tn[j] * g.tm[j];
  const int bz = b / tiles, r = b - bz * tiles;
  GemmArgs p;
  const GemmJob& q = g.job[j];
  p.A = q.A; p.sAm = q.sAm; p.sAk = q.sAk; p.B = q.B; p.sBk = q.sBk; p.sBn = q.sBn; p.C = q.C; p.ldc = q.ldc; p.bias = nullptr;
  p.M = q.M; p.N = q.N; p.K = q.K; p.kPerSplit = q.kPerSplit; p.out_mode = q.out_mode;
  gemm_tile<GB, GB, AK, BK_, false, false>(p, r % g.tn[j], r / g.tn[j], bz, As, Bs);
}

// ------------------------------------------------------------------------------------------------------
// DMA variant (the large cnet GEMMs): both operand tiles go global -> LDS by `global_load_lds_dwordx4`, no
// VGPR payload, no transposing LDS stores, two LDS stages (the next K block streams in under the MFMAs), one
// barrier per K block.  The register-staged kernel above measured 33-50 % MFMA-busy (the staging waves starve
// for issue slots beside the MFMA waves, see conv.hip); this one is used whenever every operand is 16-byte
// addressable.  Block tile 64 x TN, 2x2 waves, K block 32.
//   operand contiguous along its row index (A m-contiguous / B n-contiguous): LDS image T[k][row], fragment
//     reads T[k][row0 + lane] are consecutive words;
//   operand contiguous along k: LDS image T[row][chunk ^ (row & 7)] of 16-byte chunks (4 consecutive k): a
//     lane fetches one chunk with ds_read_b128 and feeds 4 MFMAs; the XOR swizzle makes 8 consecutive rows hit
//     8 different bank quads.  The DMA writes LDS linearly, so the swizzle is applied to the GLOBAL address each
//     lane reads.  Within a group of 8 k the lower wave half takes k = 0..3 and the upper half k = 4..7 for both
//     operands (MFMA is indifferent to the order of k as long as A and B agree).
//   K tail: chunks / k rows at or beyond kend read a 16-byte zero page instead (DMA cannot zero-fill).
// ------------------------------------------------------------------------------------------------------
#ifndef GD_BK
#define GD_BK 32
#endif
#ifndef GD_STAGES
#define GD_STAGES 1
#endif
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

template <int ROWS, bool KCONT>
__device__ __forceinline__ void gd_issue(const float* __restrict__ P, long sRow, long sK, int row0, int nRows, int k0,
                                         int kend, float* lds, int wave_u, int lane) {
  constexpr int NINST = ROWS * GD_BK / 256;          // wave instructions per tile (1 KiB each)
#pragma unroll
  for (int j = 0; j < NINST / 4; ++j) {
    const int inst = wave_u + 4 * j;                 // instruction index, dealt round-robin to the 4 waves
    const int slot = inst * 64 + lane;               // 16-byte slot in the linear LDS image
    const float* src;
    if (KCONT) {
      const int r = slot / (GD_BK / 4), cphys = slot % (GD_BK / 4);
      const int k = k0 + ((cphys ^ (r & 7)) << 2);
      const int rr = min(row0 + r, nRows - 1);
      src = k < kend ? P + (long)rr * sRow + k : g_zero16;
    } else {
      const int kr = slot / (ROWS / 4), r4 = (slot % (ROWS / 4)) * 4;
      const int k = k0 + kr;
      const int rr = min(row0 + r4, nRows - 4);
      src = k < kend ? P + (long)k * sK + rr : g_zero16;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + inst * 256), 16, 0, 0);
  }
}

template <int TM, int TN, bool AK, bool BK_>
__global__ __launch_bounds__(256) void gemm_dma_kernel(GemmArgs p) {
  constexpr int MT = TM / 64, NT = TN / 64;
  constexpr int AF = TM * GD_BK, BF = TN * GD_BK;    // floats per stage
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* As = gsm;                      // [GD_STAGES][AF]
  float* Bs = gsm + GD_STAGES * AF;     // [GD_STAGES][BF]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, li = lane & 31;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int kbeg = blockIdx.z * p.kPerSplit;
  const int kend = min(kbeg + p.kPerSplit, p.K);
  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  auto stage = [&](int k0, int buf) {
    gd_issue<TM, AK>(p.A, p.sAm, p.sAk, m0, p.M, k0, kend, As + buf * AF, wave, lane);
    gd_issue<TN, BK_>(p.B, p.sBn, p.sBk, n0, p.N, k0, kend, Bs + buf * BF, wave, lane);
  };
  // The compiler does not count an LDS-DMA load as a pending LDS write: without an explicit vmcnt(0) the barrier
  // would let the MFMAs read a tile that is still in flight.
#define GD_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
  if (GD_STAGES == 2) stage(kbeg, 0);
  GD_DMA_WAIT();
  __syncthreads();
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += GD_BK) {
    if (GD_STAGES == 2) {
      if (k0 + GD_BK < kend) stage(k0 + GD_BK, cur ^ 1);
    } else {
      stage(k0, 0);
      GD_DMA_WAIT();
      __syncthreads();
    }
    const float* A_ = As + cur * AF;
    const float* B_ = Bs + cur * BF;
#pragma unroll
    for (int g = 0; g < GD_BK / 8; ++g) {
      float a[MT][4], b[NT][4];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int arow = wm * (TM / 2) + mt * 32 + li;
        if (AK) {
          const float4 v = *reinterpret_cast<const float4*>(A_ + (arow * (GD_BK / 4) + ((2 * g + h) ^ (arow & 7))) * 4);
          a[mt][0] = v.x; a[mt][1] = v.y; a[mt][2] = v.z; a[mt][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[mt][j] = A_[(8 * g + 4 * h + j) * TM + arow];
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int brow = wn * (TN / 2) + nt * 32 + li;
        if (BK_) {
          const float4 v = *reinterpret_cast<const float4*>(B_ + (brow * (GD_BK / 4) + ((2 * g + h) ^ (brow & 7))) * 4);
          b[nt][0] = v.x; b[nt][1] = v.y; b[nt][2] = v.z; b[nt][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[nt][j] = B_[(8 * g + 4 * h + j) * TN + brow];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][j], b[nt][j], acc[mt][nt], 0, 0, 0);
    }
    if (GD_STAGES == 2) GD_DMA_WAIT();
    __syncthreads();   // next stage landed and every wave is done with `cur`
    if (GD_STAGES == 2) cur ^= 1;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * (TN / 2) + j * 32 + li;
#pragma unroll
    for (int i = 0; i < MT; ++i)
      gemm_store_tile(acc[i][j], p, m0 + wm * (TM / 2) + i * 32 + 4 * h, n, (int)blockIdx.z);
  }
}

// C[m][n] (= | +=) bias[n] + sum_s slab[s][m][n]: fixed summation order -> the result does not depend on which
// block finished first (fp32 atomics did, in the last ulp)
__global__ void gemm_reduce_kernel(const float* __restrict__ slab, int nSplit, int M, int N, const float* __restrict__ bias,
                                   float* __restrict__ C, long ldc, int accumulate) {
  const long total = (long)M * N;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int m = (int)(t / N), n = (int)(t - (long)m * N);
    float v = bias ? bias[n] : 0.f;
    for (int sI = 0; sI < nSplit; ++sI) v += slab[(size_t)sI * total + t];
    float* dst = C + (long)m * ldc + n;
    if (accumulate) *dst += v; else *dst = v;
  }
}

// library-owned split-K workspaces (one per stream that may run GEMMs concurrently; grown outside the steady state)
static void* g_gemm_ws[8] = {};
static size_t g_gemm_ws_bytes[8] = {};
static int gemm_workspace(size_t need, float** out, int slot) {
  if (need > g_gemm_ws_bytes[slot]) {
    if (g_gemm_ws[slot]) FR_HIP(hipFree(g_gemm_ws[slot]));
    g_gemm_ws[slot] = nullptr; g_gemm_ws_bytes[slot] = 0;
    FR_HIP(hipMalloc(&g_gemm_ws[slot], need));
    g_gemm_ws_bytes[slot] = need;
  }
  *out = (float*)g_gemm_ws[slot];
  return FRCNN_OK;
}

// (shared with gemmx.hip: the split-K workspaces and the slab fold)
int gemm_workspace_get(size_t need, float** out, int slot) { return gemm_workspace(need, out, slot); }
int gemm_reduce_slabs(const float* slab, int nSplit, int M, int N, const float* bias, float* C, long ldc, bool accumulate,
                      hipStream_t s) {
  const long total = (long)M * N;
  const int rgrid = (int)std::min<long>(cdivl(total, 256), 2048);
  FR_LAUNCH(KC_ELEMWISE, 0, total * 4.0 * (nSplit + 1), s, gemm_reduce_kernel, dim3(rgrid), dim3(256), 0, slab, nSplit, M, N, bias,
            C, ldc, accumulate ? 1 : 0);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

int gemm_f32(const float* A, long sAm, long sAk, const float* B, long sBk, long sBn, float* C,
             long ldc, int M, int N, int K, int out_mode, const float* bias_n, hipStream_t s, int ws_slot, GemmFold* defer) {
  if (defer) *defer = GemmFold{};
  if (M <= 0 || N <= 0) return FRCNN_OK;
  GemmArgs p;
  p.A = A; p.sAm = sAm; p.sAk = sAk; p.B = B; p.sBk = sBk; p.sBn = sBn; p.C = C; p.ldc = ldc;
  p.bias = bias_n; p.M = M; p.N = N; p.K = K;
  // 64-row tiles (more, shorter-lived blocks per CU: same finding as for the convolution tiles); 128 columns
  // when N is long.  Split K until one wave of blocks fills the resident slots, >= 256 K values per split, <= 16 splits.
  int TMs = 64;
  int TNs = (N >= 2048 && K >= 512) ? 128 : 64;
  if (const char* e = getenv("FRCNN_GEMM_TM")) TMs = atoi(e);
  if (const char* e = getenv("FRCNN_GEMM_TN")) TNs = atoi(e);
  int tm = cdiv(M, TMs), tn = cdiv(N, TNs);
  long tiles = (long)tm * tn;
  const long slots = TNs == 128 ? 1280 : 2048;
  int splitK = (int)std::max<long>(1, std::min<long>(std::min<long>(K / 256, 16), slots / tiles));
  if (const char* e = getenv("FRCNN_GEMM_SPLITK")) splitK = std::max(1, atoi(e));
  p.kPerSplit = cdiv(cdiv(K, splitK), GBK) * GBK;
  splitK = cdiv(K, p.kPerSplit);
  p.out_mode = out_mode;
  float* user_C = C;
  if (splitK > 1) {   // partial products go to slabs with plain stores; one pass folds them (+ bias, + accumulate)
    float* ws = nullptr;
    FR_TRY(gemm_workspace((size_t)splitK * M * N * 4, &ws, ws_slot & 7));
    p.C = ws; p.bias = nullptr; p.out_mode = 3;
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
  // the DMA path moves 16-byte chunks but only needs DWORD-aligned global addresses (the cnet weights sit at odd
  // offsets of the flat parameter vector: 12 089 683 pnet elements precede them)
  auto chunk_ok = [](const float* P, bool kc, long sRow, long sK, int nRows, int Kdim, int kPer) {
    if (((uintptr_t)P & 3) != 0) return false;
    if (kc) return sK == 1 && (Kdim % 4) == 0 && (kPer % 4) == 0;
    return sRow == 1 && (nRows % 4) == 0;
  };
  static const bool dma_unaligned = !(getenv("FRCNN_GEMM_DMA_UNALIGNED") && atoi(getenv("FRCNN_GEMM_DMA_UNALIGNED")) == 0);
  const bool ad = dma_unaligned ? chunk_ok(A, ak, sAm, sAk, M, K, p.kPerSplit) : av;
  const bool bd = dma_unaligned ? chunk_ok(B, bk, sBn, sBk, N, K, p.kPerSplit) : bv;
  // DMA kernel: every operand 16-byte addressable (k-contiguous: K and the row stride multiples of 4;
  // row-contiguous: row count and k stride multiples of 4), split boundaries on whole K blocks
  static const bool dma_on = !(getenv("FRCNN_GEMM_DMA") && atoi(getenv("FRCNN_GEMM_DMA")) == 0);
  const bool dma_ok = dma_on && ad && bd && (p.kPerSplit % GD_BK == 0 || splitK == 1) && K >= 64 &&
                      (ak ? true : (M % 4 == 0 && M >= 4)) && (bk ? true : (N % 4 == 0 && N >= 4));
  if (dma_ok) {
    const size_t lds = (size_t)GD_STAGES * (TMs + TNs) * GD_BK * 4;
#define GEMM_DMA(TMv, TNv, AKv, BKv)                                                                              \
  do {                                                                                                            \
    static bool attr_ = false;                                                                                    \
    if (!attr_) {                                                                                                 \
      FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_kernel<TMv, TNv, AKv, BKv>),              \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                        \
      attr_ = true;                                                                                               \
    }                                                                                                             \
    FR_LAUNCH(KC_GEMM, flops, bytes, s, (gemm_dma_kernel<TMv, TNv, AKv, BKv>), grid, dim3(256), lds, p);          \
  } while (0)
#define GEMM_DMA_T(TMv, TNv)                                                       \
  do {                                                                             \
    if (ak && bk) GEMM_DMA(TMv, TNv, true, true);                                  \
    else if (ak) GEMM_DMA(TMv, TNv, true, false);                                  \
    else if (bk) GEMM_DMA(TMv, TNv, false, true);                                  \
    else GEMM_DMA(TMv, TNv, false, false);                                         \
  } while (0)
    if (TMs == 128 && TNs == 128) GEMM_DMA_T(128, 128);
    else if (TMs == 128) GEMM_DMA_T(128, 64);
    else if (TNs == 128) GEMM_DMA_T(64, 128);
    else GEMM_DMA_T(64, 64);
#undef GEMM_DMA_T
#undef GEMM_DMA
  } else {
  const int sel = (ak ? 8 : 0) | (bk ? 4 : 0) | (av ? 2 : 0) | (bv ? 1 : 0);
#define GEMM_LAUNCH(TMv, TNv, AKv, BKv, AVv, BVv) \
  FR_LAUNCH(KC_GEMM, flops, bytes, s, (gemm_kernel<TMv, TNv, AKv, BKv, AVv, BVv>), grid, dim3(256), 0, p)
#define GEMM_CASE(AKv, BKv, AVv, BVv)                                                                        \
  case ((AKv ? 8 : 0) | (BKv ? 4 : 0) | (AVv ? 2 : 0) | (BVv ? 1 : 0)):                                      \
    if (TMs == 128 && TNs == 128) GEMM_LAUNCH(128, 128, AKv, BKv, AVv, BVv);                                 \
    else if (TMs == 128) GEMM_LAUNCH(128, 64, AKv, BKv, AVv, BVv);                                           \
    else if (TNs == 128) GEMM_LAUNCH(64, 128, AKv, BKv, AVv, BVv);                                           \
    else GEMM_LAUNCH(64, 64, AKv, BKv, AVv, BVv);                                                            \
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
#undef GEMM_LAUNCH
  }
  if (splitK > 1 && splitK <= 8 && defer && out_mode == OUT_STORE && ldc == N) {   // the consumer folds (see GemmFold; many slabs: the fold launch reads them coalesced)
    defer->slab = p.C; defer->nSplit = splitK; defer->bias = bias_n;
  } else if (splitK > 1) {
    long total = (long)M * N;
    int rgrid = (int)std::min<long>(cdivl(total, 256), 2048);
    FR_LAUNCH(KC_ELEMWISE, 0, total * 4.0 * (splitK + 1), s, gemm_reduce_kernel, dim3(rgrid), dim3(256), 0, (const float*)p.C,
              splitK, M, N, bias_n, user_C, ldc, out_mode == OUT_ADD ? 1 : 0);
  }
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// Grouped launch (see gemm_group_kernel).  A job with splits > 1 writes [split][M][N] partial-sum slabs to its C (out_mode is
// ignored: the caller folds them); otherwise C (=|+=) the product.
int gemm_f32_group(GemmJob* jobs, int n, hipStream_t s) {
  FR_CHECK(n >= 1 && n <= GEMM_GROUP_MAX, "gemm_f32_group: %d jobs", n);
  GemmGroup g;
  g.n = n;
  int at = 0;
  double flops = 0, bytes = 0;
  const bool ak = jobs[0].sAk == 1, bk = jobs[0].sBk == 1;
  for (int j = 0; j < n; ++j) {
    GemmJob& q = jobs[j];
    // (the orientation flags only choose which index runs along a wave's lanes when a tile is loaded -- the addresses use the
    // strides -- so a job whose unit stride is an accident of a dimension of 1 is still computed correctly under job 0's flags)
    const int splits = std::max(1, q.splits);
    q.kPerSplit = cdiv(cdiv(q.K, splits), GBK) * GBK;
    const int ns = (q.M > 0 && q.N > 0) ? cdiv(q.K, q.kPerSplit) : 0;
    if (splits > 1) { FR_CHECK(ns == splits, "gemm_f32_group: %d splits of K = %d leave one empty", splits, q.K); q.out_mode = 3; }
    g.job[j] = q;
    g.tm[j] = cdiv(q.M, GB); g.tn[j] = cdiv(q.N, GB);
    g.start[j] = at;
    at += g.tm[j] * g.tn[j] * ns;
    flops += 2.0 * q.M * q.N * (double)q.K;
    bytes += 4.0 * ((double)q.M * q.K + (double)q.K * q.N + (double)q.M * q.N);
  }
  g.start[n] = at;
  if (at == 0) return FRCNN_OK;
  if (ak && bk) FR_LAUNCH(KC_GEMM, flops, bytes, s, (gemm_group_kernel<true, true>), dim3(at), dim3(256), 0, g);
  else if (ak) FR_LAUNCH(KC_GEMM, flops, bytes, s, (gemm_group_kernel<true, false>), dim3(at), dim3(256), 0, g);
  else if (bk) FR_LAUNCH(KC_GEMM, flops, bytes, s, (gemm_group_kernel<false, true>), dim3(at), dim3(256), 0, g);
  else FR_LAUNCH(KC_GEMM, flops, bytes, s, (gemm_group_kernel<false, false>), dim3(at), dim3(256), 0, g);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

}  // namespace frcnn
