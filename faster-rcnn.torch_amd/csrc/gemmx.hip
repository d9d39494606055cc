// gemmx.hip -- the large nn.Linear of the classification network, Linear(kh*kw*planes, 1024) (models/model_utilities.lua:82,
// reached through objective.lua:164,179 and Detector.lua:101), in its three roles -- forward, input gradient, weight
// gradient -- in the SPLIT-bf16 operand form of convx.hip: fp32 tensors in and out, fp32 accumulation, every fp32 product
// formed from six exact bf16 x bf16 partial products of three-way split operands on v_mfma_f32_32x32x16_bf16 (arithmetic and
// accuracy: convx.hip; tests/test_gpu_elem.py::test_linear_fc1_split_bf16_form measures it against fp64 and against the
// fp32 matrix-core kernel of gemm.hip).  Round 5: forward and input gradient (the fp32-weight modes) also in the two-plane
// fp16 form of convx.hip (template parameter NP = 2, with option x3_f16: three partial products; the activation operand's planes
// and the in-register weight split are scaled by powers of two from the operands' magnitude records, amax.h).
//
// Who splits what.  The ACTIVATION operands (the pooled ROI features X [R][I], the gradient gY [R][O]) are small (R is a
// few hundred rows) and each is used by two products in two orientations: one pass (`split_planes_kernel`) writes their
// three bf16 planes once, row-major and transposed, and the GEMMs bring them to LDS by DMA with no VALU work at all.  The
// WEIGHTS (14.2 M values that change every optimiser step) are never split in HBM: a wave splits the 32 x 16 weight
// fragment it needs in registers (44 VALU instructions) and re-uses it for every row tile of the block (24 MFMAs with 128
// rows), so the split rides under the matrix pipe.
//   forward : Y[R][O]   = X[R][I]  W[O][I]^T + b     A = planes(X),    B = W fp32, k contiguous   (split K over I)
//   dgrad   : gX[R][I]  = gY[R][O] W[O][I]           A = planes(gY),   B = W fp32, n contiguous
//   wgrad   : gW[O][I] += gY^T X                     A = planes(gY^T), B = planes(X^T)            (k = r)
// Kernel geometry: see gemm_planes_kernel.
#include <cstdlib>

#include "kernels.h"
#include "amax.h"

namespace frcnn {

typedef float xf32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xbf16x2 __attribute__((ext_vector_type(2)));
typedef float xf32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 xf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 xf16x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(16))) unsigned g_xzero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ unsigned x_cvt2(float a, float b) {
  xf32x2 v = {a, b};
  xbf16x2 r = __builtin_convertvector(v, xbf16x2);   // v_cvt_pk_bf16_f32: round to nearest even
  return __builtin_bit_cast(unsigned, r);
}
// x = h + m + l exactly (three bf16 numbers each)
__device__ __forceinline__ void x_split8(const float* v, uint4& H, uint4& Mi, uint4& L) {
  unsigned hh[4], mm[4], ll[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = v[2 * j], x1 = v[2 * j + 1];
    hh[j] = x_cvt2(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, hh[j] << 16), r1 = x1 - __builtin_bit_cast(float, hh[j] & 0xFFFF0000u);
    mm[j] = x_cvt2(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, mm[j] << 16), s1 = r1 - __builtin_bit_cast(float, mm[j] & 0xFFFF0000u);
    ll[j] = x_cvt2(s0, s1);
  }
  H = make_uint4(hh[0], hh[1], hh[2], hh[3]);
  Mi = make_uint4(mm[0], mm[1], mm[2], mm[3]);
  L = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}

// ---- the two-plane fp16 form (convx.hip): x 2^e = h + l, e from the tensor's largest magnitude (its record, amax.h)
__device__ __forceinline__ unsigned x_cvt2h(float a, float b) {
  xf32x2 v = {a, b};
  xf16x2 r = __builtin_convertvector(v, xf16x2);
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void x_split8h(const float* v, uint4& H, uint4& L) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = v[2 * j], x1 = v[2 * j + 1];
    hh[j] = x_cvt2h(x0, x1);
    const xf16x2 hv = __builtin_bit_cast(xf16x2, hh[j]);
    ll[j] = x_cvt2h(x0 - (float)hv[0], x1 - (float)hv[1]);
  }
  H = make_uint4(hh[0], hh[1], hh[2], hh[3]);
  L = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}
__device__ __forceinline__ int gx_exp(float amax, int top) {
  const int be = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xFFu);
  if (be == 0 || be == 255) return 0;
  const int e = top - (be - 127);
  return e < -100 ? -100 : e > 100 ? 100 : e;
}
__device__ __forceinline__ float gx_pow2(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }

// ------------------------------------------------------------------------------------------ activation planes
// src [R][C] fp32  ->  P  [3][C/8][R][8] bf16   (the matrix as the k-contiguous operand with rows = R, k = C)
//                  and PT [3][Rp/8][C][8] bf16  (its transpose: rows = C, k = R, rows R..Rp-1 of k zero).
// K-GROUP-MAJOR: the 8 values of one row and one k group are one 16-byte entry, and the entries of consecutive ROWS are
// contiguous -- so the GEMM's LDS-DMA, which wants one entry per lane for 64 consecutive rows of a k group, reads 1 KB of
// consecutive memory per wave instruction (whole 128-byte lines; row-major planes made every lane touch a line of its own
// and the kernel ran at the rate of the texture addresser, 4x below this).  Either destination may be null.  One block =
// a 64 x 64 tile through LDS: coalesced reads, 16-byte coalesced writes both ways.
template <int NP>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, int R, int C, int Rp,
                                                           unsigned short* __restrict__ P, unsigned short* __restrict__ PT,
                                                           const float* __restrict__ amax) {
  __shared__ float tile[64 * 65];
  const float mul = NP == 2 ? gx_pow2(gx_exp(amax_load_block(amax), 14)) : 1.f;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, tid = threadIdx.x;
  for (int e = tid; e < 4096; e += 256) {
    const int r = e >> 6, c = e & 63;
    tile[r * 65 + c] = (r0 + r < R && c0 + c < C) ? src[(size_t)(r0 + r) * C + c0 + c] : 0.f;
  }
  __syncthreads();
  const size_t plane = (size_t)R * C, planeT = (size_t)C * Rp;
  for (int it = tid; it < 512; it += 256) {
    if (P) {   // 8 consecutive columns (one k group) of one row; consecutive threads take consecutive rows
      const int r = it & 63, c8 = (it >> 6) * 8;
      if (r0 + r < R && c0 + c8 < C) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile[r * 65 + c8 + j] * mul;
        uint4 H, Mi, L;
        unsigned short* d = P + ((size_t)((c0 + c8) >> 3) * R + r0 + r) * 8;
        if (NP == 2) {
          x_split8h(v, H, L);
          *reinterpret_cast<uint4*>(d) = H;
          *reinterpret_cast<uint4*>(d + plane) = L;
        } else {
          x_split8(v, H, Mi, L);
          *reinterpret_cast<uint4*>(d) = H;
          *reinterpret_cast<uint4*>(d + plane) = Mi;
          *reinterpret_cast<uint4*>(d + 2 * plane) = L;
        }
      }
    }
    if (PT) {  // 8 consecutive rows (one k group of the transpose) of one column; consecutive threads, consecutive columns
      const int c = it & 63, r8 = (it >> 6) * 8;
      if (c0 + c < C && r0 + r8 < Rp) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile[(r8 + j) * 65 + c];
        uint4 H, Mi, L;
        unsigned short* d = PT + ((size_t)((r0 + r8) >> 3) * C + c0 + c) * 8;
        if (NP == 2) {   // (round 6: the weight-gradient product in the two-plane form too)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= mul;
          x_split8h(v, H, L);
          *reinterpret_cast<uint4*>(d) = H;
          *reinterpret_cast<uint4*>(d + planeT) = L;
        } else {
          x_split8(v, H, Mi, L);
          *reinterpret_cast<uint4*>(d) = H;
          *reinterpret_cast<uint4*>(d + planeT) = Mi;
          *reinterpret_cast<uint4*>(d + 2 * planeT) = L;
        }
      }
    }
  }
}

int linear_x_rows_padded(int R) { return (R + 15) & ~15; }

int split_planes(const float* src, int R, int C, void* P, void* PT, hipStream_t s, const float* amax) {
  FR_CHECK(C % 8 == 0, "split_planes: %d columns (a multiple of 8 is needed for 16-byte plane entries)", C);
  const int Rp = linear_x_rows_padded(R);
  dim3 grid(cdiv(C, 64), cdiv(Rp, 64));
  if (amax)
    FR_LAUNCH(KC_ELEMWISE, 0, (double)R * C * 8.0, s, split_planes_kernel<2>, grid, dim3(256), 0,
              src, R, C, Rp, (unsigned short*)P, (unsigned short*)PT, amax);
  else
  FR_LAUNCH(KC_ELEMWISE, 0, (double)R * C * (4.0 + (P ? 6.0 : 0.0) + (PT ? 6.0 : 0.0)), s, split_planes_kernel<3>, grid, dim3(256), 0,
            src, R, C, Rp, (unsigned short*)P, (unsigned short*)PT, amax);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------ the product
struct GxArgs {
  const unsigned short* Ap;   // planes of A, k-group-major: element (plane, row, k) at plane * aPlane + ((k / 8) * aLd + row) * 8 + k % 8
  long aPlane, aLd;           //   (aLd = rows of the whole operand)
  const void* B;              // BMODE 0: planes like A (rows = n, bLd = rows of the whole operand); 1: fp32 [n][k] (k contiguous,
  long bPlane, bLd;           //       row stride bLd); 2: fp32 [k][n] (n contiguous, k stride bLd)
  float* C; long ldc;
  const float* bias;
  int M, N, K, kPerSplit, out_mode;   // 0 store, 1 add, 3 split-K slab [split][M][N]
  int tm, tn;                         // row / column tiles (the grid is one-dimensional: tm * tn * splits blocks)
  const float* amax_a;                // NP = 2 (fp32-weight modes): magnitude records of the A tensor and of the weights (amax.h)
  const float* amax_b;
};

// Block = TM (64 | 128 | 192 | 256) rows x 256 columns, EIGHT waves as 2 (rows) x 4 (columns): a wave owns TM / 2 rows x 64
// columns = TM / 64 x 2 accumulators of 32 x 32, i.e. with 256 rows 48 MFMAs per K step behind 12 A and 6 B fragment reads
// (0.375 KB of LDS reads per MFMA; the 128 x 32 wave tile of the first version read 0.625 KB, and with the eight waves
// released by the same barrier its LDS phase and its MFMA phase did not overlap).  What the shape buys, per K step of 16 and
// CU: 3072 matrix-pipe cycles per SIMD behind ONE barrier, 48 KB (planes x planes) or 40 KB (planes x fp32 weights) of
// LDS-DMA = 13 - 16 B/clk -- the first version (128 x 256 with 1536 cycles per barrier, or 64 x 256 with 768) asked for
// 19 - 29 B/clk, and its stripped variants showed both halves too slow on their own (Linear(13824,1024), 560 rows, forward:
// 156 us whole, 111 us with the DMA removed, 104 us with the MFMAs removed, 65 us DMA alone).  A weight fragment (fp32 in
// LDS) is split in registers by the two waves that need it.  K step = 16; a ring of 3 - 4 LDS stages filled by LDS-DMA
// ring - 1 steps ahead, counted vmcnt, one barrier per step; one block per CU.
#define GX_TN 256
#define GX_LDS_MAX (160 * 1024)
constexpr int gx_stage_bytes(int TM, int BMODE, int NP = 3) { return 32 * NP * TM + (BMODE == 0 ? 32 * NP * GX_TN : 64 * GX_TN); }
// Tiles of up to 128 rows are laid out for TWO blocks per CU (<= 80 KB of LDS, <= 128 registers): two independent blocks
// fill each other's barrier and fragment-latency bubbles (Linear(13824,1024) input gradient, 64-row tiles: 99 us with a
// ring of three = 66 KB, 121 us with a ring of four = 88 KB and one block per CU); taller tiles own the CU.
constexpr int gx_blocks_per_cu(int TM) { return TM <= 128 ? 2 : 1; }
constexpr int gx_ring(int TM, int BMODE, int NP = 3) {
  const int room = (GX_LDS_MAX - (NP == 2 ? 1024 : 0)) / gx_blocks_per_cu(TM) / gx_stage_bytes(TM, BMODE, NP);   // (NP = 2 has static words: amax.h)
  return room < 2 ? 2 : room > 4 ? 4 : room;
}
template <int N> __device__ __forceinline__ void gx_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TM, int WM, int BMODE, int NP = 3>
__global__ __launch_bounds__(512, 2 * gx_blocks_per_cu(TM)) void gemm_planes_kernel(GxArgs p) {
  constexpr int RW = TM / WM, MTW = RW / 32, NTW = WM;  // rows of one wave, its 32-row tiles and its 32-column tiles
  constexpr int SA = 32 * NP * TM;                      // bytes of one A stage: [plane NP][half 2][TM rows][8 x 16 bit]
  constexpr int SB = BMODE == 0 ? 32 * NP * GX_TN : 64 * GX_TN;   // B stage: planes | [256 n][4 chunks of 4 k] | [16 k][256 n] fp32
  constexpr int NA = SA / 1024, NB = SB / 1024;         // wave instructions (1 KB each) per stage
  constexpr int IA = (NA + 7) / 8, IB = NB / 8;         // ... per wave (A: the last ones may repeat a slot -- same bytes twice)
  static_assert(NB % 8 == 0, "B stage must deal evenly to the eight waves");
  constexpr int SS = SA + SB;
  constexpr int RING = gx_ring(TM, BMODE, NP);
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  float wmul = 1.f, out_mul = 1.f, out_mul2 = 1.f;   // NP = 2: weight scale; the inverse of both scales in two halves
  if (NP == 2) {
    const int ea = gx_exp(amax_load_block(p.amax_a), 14), ew = gx_exp(amax_load_block(p.amax_b), 14);
    wmul = gx_pow2(ew);
    const int et = -(ea + ew), e1 = et / 2;
    out_mul = gx_pow2(e1); out_mul2 = gx_pow2(et - e1);
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WM == 2 ? wave >> 2 : 0, wn = WM == 2 ? wave & 3 : wave;
  const int h = lane >> 5, li = lane & 31;
  // XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2); the virtual index gives
  // every XCD a CONTIGUOUS range, inside which the row tiles of one (column tile, K split) follow each other: the blocks
  // that read the same B tile -- the big operand: weights or X^T planes -- run together on one XCD and fetch it from HBM
  // once; the A tiles (activation planes, a few MB in all) stay L2-resident anyway.
  int v;
  {
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    v = xcd * q + min(xcd, r) + idx;
  }
  const int mt_id = v % p.tm;
  v /= p.tm;
  const int nt_id = v % p.tn, split = v / p.tn;
  const int m0 = mt_id * TM, n0 = nt_id * GX_TN;
  const int kbeg = split * p.kPerSplit;
  const int kend = min(kbeg + p.kPerSplit, p.K);   // (kend - kbeg is a multiple of 16: every stage is whole)
  const int nsteps = (kend - kbeg) >> 4;

  xf32x16 acc[MTW][NTW];
#pragma unroll
  for (int a = 0; a < MTW; ++a)
#pragma unroll
    for (int b = 0; b < NTW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- this lane's DMA sources at k = kbeg (one per wave instruction it issues) and what a K step adds to them
  const char* srcA[IA];
  const char* srcB[IB];
  int dstA[IA];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int q = (wave + 8 * i) % NA;     // (NA not a multiple of 8: the surplus instructions repeat a slot)
    const int s = q * 64 + lane;
    const int pl = s / (2 * TM), hh = (s / TM) & 1, row = s % TM;
    const int rr = min(m0 + row, p.M - 1);
    srcA[i] = reinterpret_cast<const char*>(p.Ap + (size_t)pl * p.aPlane + ((size_t)((kbeg >> 3) + hh) * p.aLd + rr) * 8);
    dstA[i] = q * 1024;
  }
  long stepB;
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int q = wave + 8 * i;
    const int s = q * 64 + lane;
    if (BMODE == 0) {
      const unsigned short* Bp = reinterpret_cast<const unsigned short*>(p.B);
      const int pl = s / (2 * GX_TN), hh = (s / GX_TN) & 1, row = s % GX_TN;
      const int rr = min(n0 + row, p.N - 1);
      srcB[i] = reinterpret_cast<const char*>(Bp + (size_t)pl * p.bPlane + ((size_t)((kbeg >> 3) + hh) * p.bLd + rr) * 8);
    } else if (BMODE == 1) {
      const float* Bf = reinterpret_cast<const float*>(p.B);
      const int row = s >> 2, cphys = s & 3;
      const int rr = min(n0 + row, p.N - 1);
      srcB[i] = reinterpret_cast<const char*>(Bf + (size_t)rr * p.bLd + kbeg + ((cphys ^ ((row >> 2) & 3)) << 2));
    } else {
      const float* Bf = reinterpret_cast<const float*>(p.B);
      const int kr = s >> 6, n4 = (s & 63) * 4;
      const int nn = min(n0 + n4, p.N - 4);
      srcB[i] = reinterpret_cast<const char*>(Bf + (size_t)(kbeg + kr) * p.bLd + nn);
    }
  }
  stepB = BMODE == 0 ? 32 * (long)p.bLd : BMODE == 1 ? 64 : 64 * (long)p.bLd;   // bytes per K step of 16
  const long stepA = 32 * (long)p.aLd;                                             // (two k groups of all rows)
  auto issue = [&](int step) {
    char* base = xsm + (step % RING) * SS;
#pragma unroll
    for (int i = 0; i < IA; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + (long)step * stepA),
                                       (__attribute__((address_space(3))) void*)(base + dstA[i]), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < IB; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + (long)step * stepB),
                                       (__attribute__((address_space(3))) void*)(base + SA + (wave + 8 * i) * 1024), 16, 0, 0);
  };

  for (int i = 0; i < RING - 1 && i < nsteps; ++i) issue(i);
  constexpr int PER = IA + IB;                  // DMA instructions of one stage, per wave
  const int arow = wm * RW + li;                // this lane's row (of the wave's first row tile) and column (of its first
  const int bcol = wn * 32 * NTW + li;          // column tile) inside the block tile
  for (int step = 0; step < nsteps; ++step) {
    // stage `step` has landed in this wave's part (DMA retires in order: at most the RING - 2 younger stages in flight)
    switch (min(nsteps - 1 - step, RING - 2)) {
      case 0: gx_wait_vmcnt<0>(); break;
      case 1: gx_wait_vmcnt<PER>(); break;
      default: gx_wait_vmcnt<2 * PER>(); break;
    }
    // ... in every wave's part, and every wave is done with stage step - 1, whose slot is refilled now.  A bare s_barrier:
    // __syncthreads() is fence + barrier, and the fence makes the compiler drain EVERY LDS-DMA in flight (s_waitcnt
    // vmcnt(0) before each barrier) -- the ring would prefetch nothing.  No wave writes LDS except by DMA, whose arrival the
    // counted wait above has established, so the barrier alone orders everything this loop needs.
    asm volatile("s_barrier" ::: "memory");
    if (step + RING - 1 < nsteps) issue(step + RING - 1);
    const char* A_ = xsm + (step % RING) * SS;
    const char* B_ = A_ + SA;
    // a block that owns its CU issues every fragment read of the step before the first MFMA (left alone the scheduler sinks
    // each group of reads to just before its use and waits lgkmcnt(0) five times a step); the two-per-CU tiles leave the
    // order to the compiler -- all fragments live at once would not fit their 128 registers
    uint4 bp[NTW][3], ap[MTW][3];
    auto load_a = [&](int pl) {
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
        ap[mt][pl] = *reinterpret_cast<const uint4*>(A_ + ((pl * 2 + h) * TM + arow + mt * 32) * 16);
    };
    if (BMODE == 0) {
      auto load_b = [&](int pl) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          bp[nt][pl] = *reinterpret_cast<const uint4*>(B_ + ((pl * 2 + h) * GX_TN + bcol + nt * 32) * 16);
      };
      if constexpr (NP == 2) { load_a(1); load_b(0); load_a(0); load_b(1); }
      else { load_a(2); load_b(0); load_a(0); load_b(2); load_a(1); load_b(1); }
    } else {
      float vv[NTW][8];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int col = bcol + nt * 32;
        if (BMODE == 1) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const float4 q = *reinterpret_cast<const float4*>(B_ + (col * 4 + ((2 * h + c) ^ ((col >> 2) & 3))) * 16);
            vv[nt][4 * c] = q.x; vv[nt][4 * c + 1] = q.y; vv[nt][4 * c + 2] = q.z; vv[nt][4 * c + 3] = q.w;
          }
        } else {
          const float* Bf = reinterpret_cast<const float*>(B_);
#pragma unroll
          for (int j = 0; j < 8; ++j) vv[nt][j] = Bf[(8 * h + j) * GX_TN + col];
        }
      }
      if constexpr (NP == 2) { load_a(1); load_a(0); } else { load_a(2); load_a(0); load_a(1); }
      if constexpr (gx_blocks_per_cu(TM) == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {   // the weight fragments are split while the A fragments are on their way
        uint4 H, Mi, L;
        if constexpr (NP == 2) {
#pragma unroll
          for (int j = 0; j < 8; ++j) vv[nt][j] *= wmul;
          x_split8h(vv[nt], H, L);
          bp[nt][0] = H; bp[nt][1] = L;
        } else {
          x_split8(vv[nt], H, Mi, L);
          bp[nt][0] = H; bp[nt][1] = Mi; bp[nt][2] = L;
        }
      }
    }
    if constexpr (gx_blocks_per_cu(TM) == 1) __builtin_amdgcn_sched_barrier(0);
    // smallest partial products first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h); plane index 0 = h, 1 = m, 2 = l
    // (two fp16 planes: (l,h) (h,l) (h,h); plane index 0 = h, 1 = l)
    constexpr int NQ = NP == 2 ? 3 : 6;
    constexpr int PA[6] = {NP == 2 ? 1 : 2, 0, NP == 2 ? 0 : 1, 1, 0, 0}, PB[6] = {0, NP == 2 ? 1 : 2, NP == 2 ? 0 : 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < NQ; ++t)
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          if constexpr (NP == 2)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(xf16x8, ap[mt][PA[t]]), __builtin_bit_cast(xf16x8, bp[nt][PB[t]]), acc[mt][nt], 0, 0, 0);
          else
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8, ap[mt][PA[t]]), __builtin_bit_cast(xbf16x8, bp[nt][PB[t]]), acc[mt][nt], 0, 0, 0);
        }
  }
  if (NP == 2) {   // undo the two tensors' scales
#pragma unroll
    for (int a = 0; a < MTW; ++a)
#pragma unroll
      for (int b = 0; b < NTW; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = (acc[a][b][r] * out_mul) * out_mul2;
  }

  // ---- epilogue: D layout col = lane & 31 (n), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (m).  The accumulate mode reads the
  // 16 old values of an accumulator tile in one batch (row clamped instead of branched around: a branch per element made
  // the compiler emit load - wait - add - store 128 times in a row, ~80 us of serialized round trips per launch).
  float* const cbase = p.out_mode == 3 ? p.C + (long)split * p.M * p.N : p.C;
  const long rstride = p.out_mode == 3 ? (long)p.N : p.ldc;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int n = n0 + bcol + nt * 32;
    const bool nok = n < p.N;
    float* const col = cbase + min(n, p.N - 1);
    const float bv = (p.bias && split == 0 && nok) ? p.bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const int mb = m0 + wm * RW + mt * 32 + 4 * h;
      if (p.out_mode == 1) {
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = col[(long)min(mb + (r & 3) + 8 * (r >> 2), p.M - 1) * rstride];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if (nok && m < p.M) col[(long)m * rstride] = old[r] + acc[mt][nt][r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if (nok && m < p.M) col[(long)m * rstride] = acc[mt][nt][r] + bv;
        }
      }
    }
  }
}

// role: 1 forward, 2 input gradient, 4 weight gradient.  Measured on Linear(13824, 1024) against the fp32 matrix-core
// kernel of gemm.hip (profiles/r03_other_configs.txt; product + plane split + slab fold): 560 rows 137 / 113 / 132 us against
// 179 / 184 / 196; 320 rows 92 / 95 / 92 against 105 / 120 / 110; 138 rows 72 / 63 / 63 against 70 / 71 / 65 -- the input
// gradient wins at every row count, forward and weight gradient from ~200 rows on (below, their plane passes and the K
// split's slabs cost what the faster product saves).
// FRCNN_GEMM_X / option "gemm_x_roles" = bit mask of the roles that take the split form (0 none, 7 all; -1 this rule).
static int g_gemm_x_roles = -2;   // -2: not decided yet (environment FRCNN_GEMM_X, default -1 = the rule below)
void set_gemm_x_roles(int mask) { g_gemm_x_roles = mask; }
int get_gemm_x_roles() {
  if (g_gemm_x_roles == -2) g_gemm_x_roles = getenv("FRCNN_GEMM_X") ? atoi(getenv("FRCNN_GEMM_X")) : -1;
  return g_gemm_x_roles;
}
bool linear_x_eligible(int role, int R, int I, int O) {
  const int mask = get_gemm_x_roles();
  if (!get_split_bf16() || I % 16 != 0 || O % 16 != 0 || R < 32 || (double)R * I * O < 1.0e9) return false;
  if (mask >= 0) return (mask & role) != 0;
  return role == 2 || R >= 192;
}

template <int TM, int WM, int BMODE, int NP = 3>
static int launch_gx(GxArgs& a, dim3 grid, double flops, double bytes, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes_kernel<TM, WM, BMODE, NP>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, GX_LDS_MAX - (NP == 2 ? 1024 : 0)));
    attr_set = true;
  }
  const size_t lds = (size_t)gx_ring(TM, BMODE, NP) * gx_stage_bytes(TM, BMODE, NP);
  FR_LAUNCH(KC_GEMM, flops, bytes, s, (gemm_planes_kernel<TM, WM, BMODE, NP>), grid, dim3(512), lds, a);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// Tile height and K split by a load model, in matrix-pipe cycles.  The busiest CU holds b = ceil(blocks / 256) blocks; a K
// step costs a block that is alone on its CU 768 cycles per 64 rows (12 MFMAs of 32 cycles on each of two waves per SIMD)
// at ~80 % plus ~300 for the barrier, the fragment latency and the DMA issue; two co-resident blocks (tiles of <= 128 rows)
// share the pipe at ~92 % and hide each other's fixed part; a block pays ~8000 cycles of prologue (first stages from HBM)
// and epilogue; split-K slabs are written and folded at ~1500 B/clk.
static void gx_plan(int M, int N, int K, int bmode, int splitK_max, int* TM, int* splitK) {
  double best = 1e300;
  for (int tmv : {256, 192, 128, 64}) {
    if (tmv == 256 && bmode != 0) continue;
    if (const char* e = getenv("FRCNN_GX_TM")) if (atoi(e) != tmv) continue;
    const long tiles = (long)cdiv(M, tmv) * cdiv(N, GX_TN);
    const int bpc = gx_blocks_per_cu(tmv);
    for (int sk = 1; sk <= splitK_max; ++sk) {
      const int kper = cdiv(cdiv(K, sk), 16) * 16;
      if (sk > 1 && kper < 256) break;
      const int sk_eff = cdiv(K, kper);
      const long b = cdivl(tiles * sk_eff, 256);
      const double mf = (tmv / 64) * 768.0, steps = kper / 16, fixed = 8000.0 + 12.0 * tmv;
      const double alone = steps * (mf / 0.80 + 300.0) + fixed, paired = steps * (2.0 * mf / 0.92 + 150.0) + fixed;
      double cost = bpc == 2 ? (b / 2) * paired + (b % 2) * alone : b * alone;
      if (sk_eff > 1) cost += 8.0 * sk_eff * (double)M * N / 1500.0 / 8.0 + 6000.0;   // slabs out and back in
      if (cost < best) { best = cost; *TM = tmv; *splitK = sk_eff; }
    }
  }
  if (const char* e = getenv("FRCNN_GX_SPLITK")) *splitK = std::max(1, atoi(e));
}

// wave layout by operand form: planes x planes -> 2 x 4 waves of TM / 2 rows x 64 columns (fewest fragment reads per MFMA);
// fp32 weights -> 1 x 8 waves of TM rows x 32 columns (each weight fragment is split by exactly one wave: 44 VALU
// instructions behind TM / 32 x 6 MFMAs)
template <int BMODE>
static int launch_gx_tm(int TM, GxArgs& a, dim3 grid, double flops, double bytes, hipStream_t s) {
  if constexpr (BMODE == 0) {
    if (a.amax_a) {   // two fp16 planes of both operands
      switch (TM) {
        case 256: return launch_gx<256, 2, 0, 2>(a, grid, flops, bytes, s);
        case 192: return launch_gx<192, 2, 0, 2>(a, grid, flops, bytes, s);
        case 128: return launch_gx<128, 2, 0, 2>(a, grid, flops, bytes, s);
        default: return launch_gx<64, 2, 0, 2>(a, grid, flops, bytes, s);
      }
    }
    switch (TM) {
      case 256: return launch_gx<256, 2, 0>(a, grid, flops, bytes, s);
      case 192: return launch_gx<192, 2, 0>(a, grid, flops, bytes, s);
      case 128: return launch_gx<128, 2, 0>(a, grid, flops, bytes, s);
      default: return launch_gx<64, 2, 0>(a, grid, flops, bytes, s);
    }
  } else if (a.amax_a) {   // two fp16 planes x fp32 weights split in registers
    switch (TM) {
      case 192: return launch_gx<192, 1, BMODE, 2>(a, grid, flops, bytes, s);
      case 128: return launch_gx<128, 1, BMODE, 2>(a, grid, flops, bytes, s);
      default: return launch_gx<64, 1, BMODE, 2>(a, grid, flops, bytes, s);
    }
  } else {
    switch (TM) {
      case 192: return launch_gx<192, 1, BMODE>(a, grid, flops, bytes, s);
      case 128: return launch_gx<128, 1, BMODE>(a, grid, flops, bytes, s);
      default: return launch_gx<64, 1, BMODE>(a, grid, flops, bytes, s);
    }
  }
}

static int gx_run(GxArgs& a, int bmode, int splitK_max, float* user_C, const float* bias, int out_mode, hipStream_t s, int ws_slot,
                  GemmFold* defer = nullptr) {
  if (defer) *defer = GemmFold{};
  FR_CHECK(a.K % 16 == 0, "gemm_planes: K = %d must be a multiple of 16", a.K);
  int TM = 128, splitK = 1;
  gx_plan(a.M, a.N, a.K, bmode, splitK_max, &TM, &splitK);
  const int tm = cdiv(a.M, TM), tn = cdiv(a.N, GX_TN);
  a.kPerSplit = cdiv(cdiv(a.K, splitK), 16) * 16;
  splitK = cdiv(a.K, a.kPerSplit);
  a.C = user_C; a.bias = bias; a.out_mode = out_mode;
  if (splitK > 1) {
    float* ws = nullptr;
    FR_TRY(gemm_workspace_get((size_t)splitK * a.M * a.N * 4, &ws, ws_slot & 7));
    a.C = ws; a.bias = nullptr; a.out_mode = 3;
  }
  a.tm = tm; a.tn = tn;
  dim3 grid(tn * tm * splitK);
  const double flops = 2.0 * a.M * a.N * (double)a.K;
  const double bytes = 4.0 * ((double)a.M * a.K + (double)a.K * a.N + (double)a.M * a.N);
  const int rc = bmode == 0 ? launch_gx_tm<0>(TM, a, grid, flops, bytes, s) : bmode == 1 ? launch_gx_tm<1>(TM, a, grid, flops, bytes, s)
                                                                                         : launch_gx_tm<2>(TM, a, grid, flops, bytes, s);
  FR_TRY(rc);
  if (splitK > 1 && splitK <= 8 && defer && out_mode == OUT_STORE && a.ldc == a.N) {   // the consumer folds (kernels.h: GemmFold; many slabs: the fold launch reads them coalesced)
    defer->slab = a.C; defer->nSplit = splitK; defer->bias = bias;
  } else if (splitK > 1) {
    FR_TRY(gemm_reduce_slabs(a.C, splitK, a.M, a.N, bias, user_C, a.ldc, out_mode == OUT_ADD, s));
  }
  return FRCNN_OK;
}

// Y[R][O] = X W^T + b ; Xp = planes of X, [3][I/8][R][8]
int linear_x_forward(const void* Xp, int R, int I, const float* W, const float* bias, int O, float* y, hipStream_t s, int ws_slot,
                     GemmFold* defer, const float* amax_x, const float* amax_w) {
  FR_CHECK(((uintptr_t)W & 3) == 0 && I % 16 == 0, "linear_x_forward: operand alignment");
  FR_CHECK((amax_x != nullptr) == (amax_w != nullptr), "linear_x_forward: the fp16 form needs the records of both operands");
  GxArgs a;
  a.amax_a = amax_x; a.amax_b = amax_w;
  a.Ap = (const unsigned short*)Xp; a.aPlane = (long)R * I; a.aLd = R;
  a.B = W; a.bPlane = 0; a.bLd = I;
  a.ldc = O; a.M = R; a.N = O; a.K = I;
  return gx_run(a, 1, 32, y, bias, OUT_STORE, s, ws_slot, defer);
}

// gX[R][I] (= | +=) gY W ; Gp = planes of gY, [3][O/8][R][8]
int linear_x_dgrad(const void* Gp, int R, int O, const float* W, int I, float* gx, int out_mode, hipStream_t s, int ws_slot,
                   GemmFold* defer, const float* amax_g, const float* amax_w) {
  FR_CHECK(((uintptr_t)W & 3) == 0 && I % 4 == 0 && O % 16 == 0, "linear_x_dgrad: operand alignment");
  FR_CHECK((amax_g != nullptr) == (amax_w != nullptr), "linear_x_dgrad: the fp16 form needs the records of both operands");
  GxArgs a;
  a.amax_a = amax_g; a.amax_b = amax_w;
  a.Ap = (const unsigned short*)Gp; a.aPlane = (long)R * O; a.aLd = R;
  a.B = W; a.bPlane = 0; a.bLd = I;
  a.ldc = I; a.M = R; a.N = I; a.K = O;
  return gx_run(a, 2, 4, gx, nullptr, out_mode, s, ws_slot, defer);
}

// gW[O][I] += gY^T X ; GpT = planes of gY^T [3][Rp/8][O][8], XpT = planes of X^T [3][Rp/8][I][8] (k = r; r >= R zero)
int linear_x_wgrad(const void* GpT, const void* XpT, int R, int O, int I, float* gw, hipStream_t s, int ws_slot, const float* amax_g,
                   const float* amax_x) {
  FR_CHECK((amax_g != nullptr) == (amax_x != nullptr), "linear_x_wgrad: the fp16 form needs the records of both operands");
  const int Rp = linear_x_rows_padded(R);
  GxArgs a;
  a.amax_a = amax_g; a.amax_b = amax_x;
  a.Ap = (const unsigned short*)GpT; a.aPlane = (long)O * Rp; a.aLd = O;
  a.B = XpT; a.bPlane = (long)I * Rp; a.bLd = I;
  a.ldc = I; a.M = O; a.N = I; a.K = Rp;
  return gx_run(a, 0, 1, gw, nullptr, OUT_ADD, s, ws_slot);
}

}  // namespace frcnn
