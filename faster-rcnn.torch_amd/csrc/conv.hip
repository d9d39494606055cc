// conv.hip -- nn.SpatialConvolution forward / updateGradInput / accGradParameters
// (reference call sites: models/model_utilities.lua:8,31,33 driven by objective.lua:71,189 and
// Detector.lua:33) as hand-written implicit-GEMM kernels on the gfx950 fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
// Layout: activations CHW fp32 exactly like the reference's tensors.  The input patch (with halo)
// of one output tile is staged ONCE in LDS and re-used by all k*k taps -- no im2col matrix ever
// exists in HBM.  Weights are pre-packed per step into a [K'][M] matrix (M fastest) whose K' order
// interleaves channel pairs, so that the two 32-lane halves of a wave (the two k-slices of
// v_mfma_f32_32x32x2_f32) address LDS at  lane_base + compile-time immediate.
//
// updateGradInput is the same kernel: a "full" correlation of gradOutput with the flipped,
// transposed filter bank (pad' = k-1-pad), fed by the second packed matrix.
#include <cstdlib>

#include "kernels.h"
#include "amax.h"
#include <type_traits>

namespace frcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
int conv_cc(int k) { return k == 1 ? 32 : (k == 3 ? 8 : 2); }
int conv_mpad(int M) { return M <= 64 ? 64 : cdiv(M, 128) * 128; }
size_t conv_pack_floats(int Kchan, int M, int k) {
  int cc = conv_cc(k);
  return (size_t)cdiv(Kchan, cc) * cc * k * k * conv_mpad(M);
}

// dst[row][m], row = ((cp*k+ky)*k+kx)*2+h, kc = 2cp+h.
//  mode 0 (fwd):   dst = W[m][kc][ky][kx]            (W is [O][C][k][k], M=O, Kchan=C)
//  mode 1 (dgrad): dst = W[kc][m][k-1-ky][k-1-kx]    (M=C, Kchan=O)
__global__ void pack_weights_kernel(const float* __restrict__ w, int O, int C, int k, int mode,
                                    int Kchan_pad, int M, int Mpad, float* __restrict__ dst) {
  long total = (long)Kchan_pad * k * k * Mpad;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long)gridDim.x * blockDim.x) {
    int m = (int)(t % Mpad);
    long row = t / Mpad;
    int h = (int)(row & 1);
    long r2 = row >> 1;
    int kx = (int)(r2 % k);
    r2 /= k;
    int ky = (int)(r2 % k);
    int cp = (int)(r2 / k);
    int kc = cp * 2 + h;
    float v = 0.f;
    if (mode == 0) {
      if (m < O && kc < C) v = w[(((long)m * C + kc) * k + ky) * k + kx];
    } else {
      if (m < C && kc < O) v = w[(((long)kc * C + m) * k + (k - 1 - ky)) * k + (k - 1 - kx)];
    }
    (void)M;
    dst[t] = v;
  }
}

// All packs of a model in ONE launch.  Blocks are dealt to jobs in proportion to their size
// (blk_begin/nblk).  Both layouts are transposes, done through a 64 x 64 LDS tile so that the canonical-layout
// reads and the packed-layout writes are 256-byte runs (16-byte accesses where the alignment allows):
//   mode 0: source matrix S[m = o][q = (c,ky,kx)]            -> dst[row(c,ky,kx)][o]
//   mode 1: source, per o: S[c][t = (ky,kx)] (contiguous)     -> dst[row(o,k-1-ky,k-1-kx)][c]
// (padding rows/columns of dst are zeroed once when the packed buffers are allocated)
#define PK_T 64
#define PK_P 65
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const float* __restrict__ weights, const PackJob* __restrict__ jobs, int njobs) {
  __shared__ float tile[PK_T * PK_P];
  int jb = 0;
  while (jb + 1 < njobs && (int)blockIdx.x >= jobs[jb + 1].blk_begin) ++jb;
  const PackJob j = jobs[jb];
  const float* __restrict__ w = weights + j.w_off;
  const int k = j.k, kk = k * k;
  const int tid = threadIdx.x;
  if (j.mode == 0) {
    const int Q = j.C * kk;
    const int tm = (j.O + PK_T - 1) / PK_T, tq = (Q + PK_T - 1) / PK_T;
    const bool vec = (Q & 3) == 0 && ((j.w_off & 3) == 0);
    const int l16 = tid & 15, r16 = tid >> 4;
    for (int tix = blockIdx.x - j.blk_begin; tix < tm * tq; tix += j.nblk) {
      const int m0 = (tix / tq) * PK_T, q0 = (tix % tq) * PK_T;
      if (vec) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + r16 + 16 * i, q = q0 + l16 * 4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (m < j.O && q < Q) v = *reinterpret_cast<const float4*>(w + (size_t)m * Q + q);
          float* t = tile + (r16 + 16 * i) * PK_P + l16 * 4;
          t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
        }
      } else {
        const int lq = tid & 63, rm = tid >> 6;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = m0 + rm + 4 * i, q = q0 + lq;
          tile[(rm + 4 * i) * PK_P + lq] = (m < j.O && q < Q) ? w[(size_t)m * Q + q] : 0.f;
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ql = r16 + 16 * i, q = q0 + ql;
        if (q < Q) {
          const int c = q / kk, t = q - c * kk;
          const long row = ((long)(c >> 1) * kk + t) * 2 + (c & 1);
          const float* tp = tile + (l16 * 4) * PK_P + ql;
          // columns m0..m0+63 exist in dst (Mpad is a multiple of 64); the tile holds 0 for m >= O
          *reinterpret_cast<float4*>(j.dst + row * j.Mpad + m0 + l16 * 4) = make_float4(tp[0], tp[PK_P], tp[2 * PK_P], tp[3 * PK_P]);
        }
      }
      __syncthreads();
    }
  } else {
    // work item = (group of G output channels o, 64-channel c tile): G*n*kk contiguous-per-o source floats
    const int G = min(16, (PK_T * PK_P) / (PK_T * kk));
    const int tc = (j.C + PK_T - 1) / PK_T, og = (j.O + G - 1) / G;
    const long nitem = (long)og * tc;
    for (long tix = blockIdx.x - j.blk_begin; tix < nitem; tix += j.nblk) {
      const int o0 = (int)(tix / tc) * G, c0 = (int)(tix % tc) * PK_T;
      const int n = min(PK_T, j.C - c0), ng = min(G, j.O - o0);
      const int per = n * kk;
      for (int e = tid; e < ng * per; e += 256) {
        const int g = e / per, r = e - g * per;
        tile[g * (PK_T * kk) + r] = w[((size_t)(o0 + g) * j.C + c0) * kk + r];
      }
      __syncthreads();
      for (int e = tid; e < ng * kk * PK_T; e += 256) {
        const int c = e & (PK_T - 1), gt = e >> 6;
        const int g = gt / kk, t = gt - g * kk;
        if (c < n) {
          const int o = o0 + g, tf = kk - 1 - t;   // flipped tap: (k-1-ky, k-1-kx)
          const long row = ((long)(o >> 1) * kk + tf) * 2 + (o & 1);
          j.dst[row * j.Mpad + c0 + c] = tile[g * (PK_T * kk) + c * kk + t];
        }
      }
      __syncthreads();
    }
  }
}

PackJob conv_pack_job(long w_off, int O, int C, int k, int mode, float* dst) {
  PackJob j;
  const int cc = conv_cc(k);
  const int kchan = mode == 0 ? C : O, M = mode == 0 ? O : C;
  j.w_off = w_off; j.O = O; j.C = C; j.k = k; j.mode = mode; j.dst = dst;
  j.Mpad = conv_mpad(M);
  j.total = (long)cdiv(kchan, cc) * cc * k * k * j.Mpad;
  j.blk_begin = 0; j.nblk = 1;
  return j;
}

// deals `total_blocks` blocks to the jobs in proportion to their element counts; returns the grid size
int conv_pack_assign_blocks(PackJob* jobs, int njobs, int total_blocks) {
  double sum = 0;
  for (int i = 0; i < njobs; ++i) sum += (double)jobs[i].total;
  int b = 0;
  for (int i = 0; i < njobs; ++i) {
    jobs[i].blk_begin = b;
    jobs[i].nblk = std::max(1, (int)(total_blocks * ((double)jobs[i].total / sum) + 0.5));
    b += jobs[i].nblk;
  }
  return b;
}

int conv_pack_weights_multi(const float* weights, const PackJob* jobs_dev, int njobs, int grid, hipStream_t s) {
  if (njobs <= 0) return FRCNN_OK;
  FR_LAUNCH(KC_ELEMWISE, 0, 0, s, pack_weights_multi_kernel, dim3(grid), dim3(256), 0, weights, jobs_dev, njobs);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

int conv_pack_weights(const float* w, int O, int C, int k, float* wf, float* wd, hipStream_t s) {
  int cc = conv_cc(k);
  if (wf) {
    int kp = cdiv(C, cc) * cc, mp = conv_mpad(O);
    long total = (long)kp * k * k * mp;
    int grid = (int)std::min<long>(cdivl(total, 256), 4096);
    FR_LAUNCH(KC_ELEMWISE, 0, total * 8.0, s, pack_weights_kernel, dim3(grid), dim3(256), 0, w, O, C, k,
              0, kp, O, mp, wf);
  }
  if (wd) {
    int kp = cdiv(O, cc) * cc, mp = conv_mpad(C);
    long total = (long)kp * k * k * mp;
    int grid = (int)std::min<long>(cdivl(total, 256), 4096);
    FR_LAUNCH(KC_ELEMWISE, 0, total * 8.0, s, pack_weights_kernel, dim3(grid), dim3(256), 0, w, O, C, k,
              1, kp, C, mp, wd);
  }
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------
// implicit GEMM: forward and input-gradient
// ------------------------------------------------------------------------------------------
struct IgemmArgs {
  const float* in;
  const float* in_slope;  // device scalar or null
  const float* in_scale;  // device [Cin] or null
  const float* wp;        // packed [Kp][Mpad]
  const float* bias;      // [M] or null
  float* out;             // [M][Ho][Wo]
  int Cin, H, W, M, Mpad, Ho, Wo, pad;
  int TH, TW, tilesX, tilesY, mTiles;
  int nChunks, splitK, chunksPerSplit;
  int out_mode;           // 0 store, 1 add, 3 split-K slab
  int dma_patch;          // input patch by LDS-DMA (no activation to fuse, tensor < 2 GiB)
  // fused 2x2 stride-2 ceil-mode max pool of act(out) (tile fixed to TH=4 x TW=32, single K split): null = off
  float* pool_out;             // [M][Hp][Wp]
  unsigned char* pool_idx;     // arg-max code dy*2+dx, first maximum wins
  const float* pool_slope;     // PReLU slope of the pooled activation (device scalar) or null
  const float* pool_scale;     // dropout scale [M] or null
  float* pool_amax;            // magnitude record of the pooled map (amax.h) or null
  int Hp, Wp;
};

#ifndef IG_TRACE
#define IG_TRACE 0
#endif
#if IG_TRACE
__device__ unsigned long long g_ig_trace[16 * 4096];
extern "C" int frcnn_debug_ig_trace(void* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ig_trace), sizeof(unsigned long long) * 16 * 4096); }
#define TR_NOW() __builtin_readcyclecounter()
#else
#define TR_NOW() 0ull
#endif
#ifndef IG_NTW
#define IG_NTW 2
#endif
#ifndef IG_BPC
#define IG_BPC 5
#endif
#define IG_MAXIT 4  // patch plane <= 1024 positions

// MODE 0: one LDS buffer, stage -> barrier -> MFMA -> barrier (latency covered by the other blocks of the CU)
// MODE 1: As and Bs double-buffered in LDS, one barrier per chunk (1x1: big chunks, 2 blocks per CU)
template <int CC, int MODE>
constexpr int igemm_blocks_per_cu(int KS, int BM) { return KS == 1 ? 2 : (BM == 64 && CC != 4 ? IG_BPC : 3); }

template <int KS, int CC, int BM, int MODE, int NIT>
__global__ __launch_bounds__(256, (igemm_blocks_per_cu<CC, MODE>(KS, BM))) void conv_igemm_kernel(IgemmArgs p) {
  constexpr int KC = CC * KS * KS;   // K rows per chunk
  constexpr int WM = BM / 2;         // 2x2 waves
  constexpr int MT = WM / 32;        // 32x32 tiles per wave along M
  constexpr int NTW = IG_NTW;        // ... along N (wave covers 32*NTW pixels)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned long long trR0 = IG_TRACE ? __builtin_amdgcn_s_memrealtime() : 0;
  unsigned long long tr0 = TR_NOW(), trS = 0, trB1 = 0, trC = 0, trB2 = 0, trT, trI = 0, trW = 0;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, li = lane & 31;

  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so block b and
  // b+8 share an L2.  The virtual index v walks one XCD's blocks consecutively, and consecutive v are the
  // M tiles of ONE pixel tile: the input patch they all stage is read from HBM once and then hits in that L2.
  const int nT = p.tilesX * p.tilesY;
  int v;
  {
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    v = xcd * q + min(xcd, r) + idx;
  }
  const int mt_id = v % p.mTiles;
  v /= p.mTiles;
  const int nt_id = v % nT;
  const int split = v / nT;
  const int ty0 = (nt_id / p.tilesX) * p.TH, tx0 = (nt_id % p.tilesX) * p.TW;
  const int m0 = mt_id * BM;
  const int PW = p.TW + KS - 1, PH = p.TH + KS - 1, plane = PH * PW;
  constexpr int planeP = NIT * 256;          // LDS stride of one patch channel (stores are unconditional)
  const int NT = p.TH * p.TW;
  const int HW = p.H * p.W;
  constexpr int bufFloats = KC * BM + CC * planeP;  // one LDS buffer: As[KC][BM] then Bs[CC][planeP]
  constexpr int aFloats = KC * BM;

  // this thread's patch positions (same for every channel and chunk): byte offsets inside a channel
  // plane, so that every patch load is `global_load_dword v, voff, s[base]` with a scalar channel base
  unsigned gofs[NIT];
  bool gok[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    int e = tid + it * 256;
    int r = e / PW, col = e - r * PW;
    int gy = ty0 - p.pad + r, gx = tx0 - p.pad + col;
    gok[it] = e < plane && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    gofs[it] = gok[it] ? (unsigned)(gy * p.W + gx) * 4u : 0u;   // clamped: loads are unconditional, zero fill by select
  }
  // LDS-DMA variant of the patch staging (launches whose input needs no PReLU / dropout scale): `buffer_load_dword
  // ... lds` moves one patch position per lane straight into Bs[cc][it*256 + tid] -- one instruction per 64 values
  // instead of load + select + ds_write, no staging registers.  Positions outside the image (and channels >= Cin)
  // fall outside the buffer resource's range and arrive as zeros.
  const bool dma_patch = MODE == 0 && p.dma_patch != 0;
  unsigned dofs[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) dofs[it] = gok[it] ? gofs[it] : 0x7FFFFFFFu;
  const __amdgpu_buffer_rsrc_t in_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, dma_patch ? p.Cin * HW * 4 : 0, 0x00020000);
  const bool has_slope = p.in_slope != nullptr, has_scale = p.in_scale != nullptr;
  const float slope = has_slope ? *p.in_slope : 1.f;

  // lane offsets of the MFMA operand reads inside a buffer
  const int aoff = h * BM + wm * WM + li;
  int boff[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    int q = wn * (32 * NTW) + nt * 32 + li;
    q = q < NT ? q : NT - 1;
    int ty = q / p.TW, tx = q - ty * p.TW;
    boff[nt] = h * planeP + ty * PW + tx;
  }

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NTW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // staging registers: only the patch values; the weight rows go global -> LDS by DMA
  float vb[CC][NIT];
  float scv = 1.f;   // lane cc holds the dropout scale of channel c0+cc
  // LDS-DMA geometry: one wave instruction moves 64 lanes x 16 B = 1 KiB = RPW consecutive rows of As
  constexpr int RPW = 1024 / (BM * 4), NDMA = (KC + 4 * RPW - 1) / (4 * RPW);
  constexpr bool DMA_FULL = KC % (4 * RPW) == 0;   // every wave instruction of a chunk is in range
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int dma_row = wave_u * RPW + lane / (BM / 4);
  const unsigned dma_voff = ((unsigned)dma_row * p.Mpad + (lane % (BM / 4)) * 4) * 4u;   // bytes inside a chunk
  const size_t dma_step = (size_t)4 * RPW * p.Mpad * 4;                                  // bytes between two instructions
  const size_t hw_bytes = (size_t)HW * 4;

  // A wave that stages shares its SIMD with waves that keep the matrix pipe's issue port saturated and is
  // granted an issue slot only every few tens of cycles (measured: ~6 k cycles for ~150 instructions), so
  // the staging code is written for instruction COUNT: scalar bases + fixed lane offsets, no per-element
  // branches, flags resolved once per chunk.
  auto stage_load = [&](int chunk, float* buf) {   // buf = As of the target buffer
    // weights: `global_load_lds_dwordx4` (LDS destination = wave-uniform base + lane*16, i.e. the linear
    // As[KC][BM] image).  The compiler does NOT treat the DMA as a pending LDS write at a barrier; completion is
    // covered because the patch loads below are issued AFTER it and vmcnt retires in order: stage_store waits for
    // them before the barrier, hence for the DMA too.
    const char* srcA = reinterpret_cast<const char*>(p.wp + ((size_t)chunk * KC) * p.Mpad + m0);
    float* dstA = buf + wave_u * RPW * BM;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      if (DMA_FULL || dma_row + i * 4 * RPW < KC)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA + i * dma_step + dma_voff),
                                         (__attribute__((address_space(3))) void*)(dstA + i * 4 * RPW * BM), 16, 0, 0);
    }
    const int c0 = chunk * CC;
    if (dma_patch) {
      float* dstB = buf + aFloats + wave_u * 64;
#pragma unroll
      for (int cc = 0; cc < CC; ++cc)
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (__attribute__((address_space(3))) void*)(dstB + cc * planeP + it * 256),
                                                   4, dofs[it], (unsigned)(c0 + cc) * (unsigned)hw_bytes, 0, 0);
      return;
    }
    const char* srcB = reinterpret_cast<const char*>(p.in + (size_t)c0 * HW);
    if (c0 + CC <= p.Cin) {
#pragma unroll
      for (int cc = 0; cc < CC; ++cc)
#pragma unroll
        for (int it = 0; it < NIT; ++it) vb[cc][it] = *reinterpret_cast<const float*>(srcB + cc * hw_bytes + gofs[it]);
    } else {   // last, partial chunk: out-of-range channels read channel Cin-1 and are zeroed by stage_store
#pragma unroll
      for (int cc = 0; cc < CC; ++cc) {
        const int back = c0 + cc < p.Cin ? 0 : c0 + cc - (p.Cin - 1);
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          vb[cc][it] = *reinterpret_cast<const float*>(srcB + (cc - back) * (long)hw_bytes + gofs[it]);
      }
    }
    if (has_scale) scv = p.in_scale[min(c0 + (lane & (CC - 1)), p.Cin - 1)];
  };
  // registers -> LDS; the producing layer's PReLU / dropout scale is applied here
  auto store_as = [&](float* bufB, auto slope_c, auto scale_c, int nvalid) {
    constexpr bool SLOPE = decltype(slope_c)::value, SCALE = decltype(scale_c)::value;
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
      float sc = 1.f;
      if (SCALE) sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, scv), cc));
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        float v = vb[cc][it];
        if (SLOPE) v = v > 0.f ? v : slope * v;
        if (SCALE) v *= sc;
        bufB[cc * planeP + tid + it * 256] = (gok[it] && cc < nvalid) ? v : 0.f;
      }
    }
  };
  auto stage_store = [&](int chunk, float* bufB) {
    if (dma_patch) {   // (the compiler does not count LDS-DMA as pending LDS writes at the barrier)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
    const int nvalid = p.Cin - chunk * CC;   // >= CC except in a partial last chunk
    if (has_slope) {
      if (has_scale) store_as(bufB, std::true_type{}, std::true_type{}, nvalid);
      else store_as(bufB, std::true_type{}, std::false_type{}, nvalid);
    } else {
      if (has_scale) store_as(bufB, std::false_type{}, std::true_type{}, nvalid);
      else store_as(bufB, std::false_type{}, std::false_type{}, nvalid);
    }
  };
  // MFMA over one staged chunk: K' order = ((cp*KS+ky)*KS+kx)*2 + h.  The operand fragments of
  // k-pair i+1 are read from LDS BEFORE the MFMAs of k-pair i are issued (register double buffer +
  // sched_group_barrier), so the ~100+ cycle LDS latency hides behind 4 x 64 cycles of matrix pipe.
  constexpr int NKP = (CC / 2) * KS * KS;
  auto compute = [&](const float* buf, const float* bufB, const int kbeg, const int kend) {
    float a[2][MT], b[2][NTW];
    auto frag = [&](int kp, float* fa, float* fb) {
      const int cp = kp / (KS * KS), ky = (kp / KS) % KS, kx = kp % KS;
      const int rowoff = cp * 2 * planeP + ky * PW + kx;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[mt] = buf[aoff + kp * 2 * BM + mt * 32];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) fb[nt] = bufB[boff[nt] + rowoff];
    };
    frag(kbeg, a[kbeg & 1], b[kbeg & 1]);
#pragma unroll
    for (int kp = kbeg; kp < kend; ++kp) {
      const int cur = kp & 1;
      if (kp + 1 < kend) frag(kp + 1, a[cur ^ 1], b[cur ^ 1]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][mt], b[cur][nt], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, MT + NTW, 0);  // DS reads of the next k-pair first
      __builtin_amdgcn_sched_group_barrier(0x008, MT * NTW, 0);  // then this k-pair's MFMAs
    }
  };

  const int cbeg = split * p.chunksPerSplit;
  const int cend = min(cbeg + p.chunksPerSplit, p.nChunks);
  if (MODE == 1) {
    // double-buffered LDS, one barrier per chunk: chunk k+1 is fetched into registers before the
    // MFMAs of chunk k and written to the other buffer after them.
    stage_load(cbeg, smem);
    stage_store(cbeg, smem + aFloats);
    __syncthreads();
    int cur = 0;
    for (int chunk = cbeg; chunk < cend; ++chunk) {
      const bool more = chunk + 1 < cend;
      float* nxt = smem + (cur ^ 1) * bufFloats;
      trT = TR_NOW();
      if (more) stage_load(chunk + 1, nxt);
      // the patch of chunk k+1 is written half way through the MFMAs (its loads have landed by then and
      // the ds_writes issue under the matrix pipe's backlog)
      compute(smem + cur * bufFloats, smem + cur * bufFloats + aFloats, 0, NKP / 2);
      if (more) stage_store(chunk + 1, nxt + aFloats);
      compute(smem + cur * bufFloats, smem + cur * bufFloats + aFloats, NKP / 2, NKP);
      if (IG_TRACE) { unsigned long long t = TR_NOW(); trC += t - trT; trT = t; }
      __syncthreads();
      if (IG_TRACE) { unsigned long long t = TR_NOW(); trB1 += t - trT; trT = t; }
      cur ^= 1;
    }
  } else {
    for (int chunk = cbeg; chunk < cend; ++chunk) {
      trT = TR_NOW();
      stage_load(chunk, smem);
#if IG_TRACE
      { unsigned long long t = TR_NOW(); trI += t - trT; trT = t; }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      { unsigned long long t = TR_NOW(); trW += t - trT; trT = t; }
#endif
      stage_store(chunk, smem + aFloats);
      if (IG_TRACE) { unsigned long long t = TR_NOW(); trS += t - trT; trT = t; }
      __syncthreads();
      if (IG_TRACE) { unsigned long long t = TR_NOW(); trB1 += t - trT; trT = t; }
      compute(smem, smem + aFloats, 0, NKP);
      if (IG_TRACE) { unsigned long long t = TR_NOW(); trC += t - trT; trT = t; }
      __syncthreads();
      if (IG_TRACE) { unsigned long long t = TR_NOW(); trB2 += t - trT; trT = t; }
    }
  }

  const unsigned long long trE = TR_NOW();
  // ---- epilogue: D layout col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (channel).
  // One exec mask per pixel column, row pointers advanced by the channel stride, the output mode resolved once.
  const long HoWo = (long)p.Ho * p.Wo;
  const bool add_bias = p.bias != nullptr && split == 0;
  const int mrow0 = m0 + wm * WM + 4 * h;
  auto store_tile = [&](auto mode_c) {
    constexpr int OM = decltype(mode_c)::value;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int q = wn * (32 * NTW) + nt * 32 + li;
      const int ty = q / p.TW, tx = q - ty * p.TW;
      const int oy = ty0 + ty, ox = tx0 + tx;
      if (q < NT && oy < p.Ho && ox < p.Wo) {
        float* col = p.out + (OM == 3 ? (size_t)split * p.M * HoWo : (size_t)0) + (size_t)oy * p.Wo + ox;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + mt * 32 + (r & 3) + 8 * (r >> 2);
            if (m < p.M) {
              float v = acc[mt][nt][r];
              if (add_bias) v += p.bias[m];
              float* dst = col + (size_t)m * HoWo;
              if (OM == 1) *dst += v; else *dst = v;
            }
          }
        }
      }
    }
  };
  // Epilogue with the block's max pool fused (last convolution of a backbone block): with the 4 x 32 tile a wave holds
  // two vertically adjacent output rows in acc[0][0] / acc[0][1] of the same lane and the horizontal neighbour in
  // lane^1, so a 2x2 window is two registers + two lane exchanges.  x = conv + bias is stored as usual (the backward
  // pass needs it); the pooled map takes act(x) = scale[m] * prelu(x) in the window order (0,0),(0,1),(1,0),(1,1),
  // first maximum wins -- the arithmetic of maxpool_act_forward_kernel.
  auto store_tile_pool = [&]() {
    const int oy0 = ty0 + wn * 2, ox = tx0 + li;
    const bool colok = ox < p.Wo, col1ok = ox + 1 < p.Wo, row0ok = oy0 < p.Ho, row1ok = oy0 + 1 < p.Ho;
    const float aslope = p.pool_slope ? *p.pool_slope : 1.f;
    const bool has_ps = p.pool_slope != nullptr;
    const size_t pofs = (size_t)(oy0 >> 1) * p.Wp + (ox >> 1);
    const size_t HpWp = (size_t)p.Hp * p.Wp;
    float am = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mrow0 + (r & 3) + 8 * (r >> 2);
      const bool mok = m < p.M;
      const int mc = mok ? m : p.M - 1;
      const float b = p.bias ? p.bias[mc] : 0.f;
      const float sc = p.pool_scale ? p.pool_scale[mc] : 1.f;
      const float v0 = acc[0][0][r] + b, v1 = acc[0][1][r] + b;
      if (mok && colok) {
        float* dst = p.out + (size_t)m * HoWo + (size_t)oy0 * p.Wo + ox;
        if (row0ok) dst[0] = v0;
        if (row1ok) dst[p.Wo] = v1;
      }
      float a0 = v0, a1 = v1;
      if (has_ps) { a0 = a0 > 0.f ? a0 : aslope * a0; a1 = a1 > 0.f ? a1 : aslope * a1; }
      if (p.pool_scale) { a0 *= sc; a1 *= sc; }
      const float n0 = __shfl_xor(a0, 1, 64), n1 = __shfl_xor(a1, 1, 64);
      if (!(li & 1) && mok && colok && row0ok) {
        float best = a0;
        int bi = 0;
        if (col1ok && n0 > best) { best = n0; bi = 1; }
        if (row1ok) {
          if (a1 > best) { best = a1; bi = 2; }
          if (col1ok && n1 > best) { best = n1; bi = 3; }
        }
        p.pool_out[(size_t)m * HpWp + pofs] = best;
        p.pool_idx[(size_t)m * HpWp + pofs] = (unsigned char)bi;
        am = fmaxf(am, fabsf(best));
      }
    }
    if (p.pool_amax) amax_store_block(am, p.pool_amax);
  };
  if constexpr (MT == 1 && NTW == 2 && KS == 3) {
    if (p.pool_out) { store_tile_pool(); goto done; }
  }
  if (p.out_mode == 0) store_tile(std::integral_constant<int, 0>{});
  else if (p.out_mode == 1) store_tile(std::integral_constant<int, 1>{});
  else store_tile(std::integral_constant<int, 3>{});
done:;
#if IG_TRACE
  if (tid == 0 && blockIdx.x < 4096) {
    unsigned long long* t = g_ig_trace + 16 * blockIdx.x;
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    t[0] = tr0; t[1] = trE; t[2] = TR_NOW(); t[3] = trS; t[4] = trB1; t[5] = trC; t[6] = trB2; t[7] = hwid; t[8] = xcc;
    t[9] = __builtin_amdgcn_s_memrealtime(); t[10] = trR0; t[11] = trI; t[12] = trW;
  }
#endif
}

// out[m][p] (= | +=) bias[m] + sum_s slab[s][m][p]
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, int nSplit, int M, long hw, const float* __restrict__ bias,
                                     float* __restrict__ out, int accumulate) {
  const long total = (long)M * hw;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    float v = bias ? bias[t / hw] : 0.f;
    for (int s = 0; s < nSplit; ++s) v += slab[(size_t)s * total + t];
    if (accumulate) out[t] += v; else out[t] = v;
  }
}

// library-owned split-K workspace (grown outside the steady state)
// (one per stream that may run split-K convolutions concurrently: slot 0 = caller's stream, 1 = side stream)
static void* g_ig_ws[8] = {};
static size_t g_ig_ws_bytes[8] = {};
static int ig_workspace(size_t need, float** out, int slot) {
  if (need > g_ig_ws_bytes[slot]) {
    if (g_ig_ws[slot]) FR_HIP(hipFree(g_ig_ws[slot]));
    g_ig_ws[slot] = nullptr; g_ig_ws_bytes[slot] = 0;
    FR_HIP(hipMalloc(&g_ig_ws[slot], need));
    g_ig_ws_bytes[slot] = need;
  }
  *out = (float*)g_ig_ws[slot];
  return FRCNN_OK;
}

// choose the output tile (TH x TW <= 128 pixels, patch plane <= 1024) that wastes the least work
static void choose_tile(int Ho, int Wo, int k, int maxNT, int* TH, int* TW) {
  long best = -1;
  int bth = 1, btw = 1;
  for (int tw = 1; tw <= std::min(Wo, maxNT); ++tw) {
    if (tw < 8 && Wo >= 8) continue;
    int th = std::min(maxNT / tw, Ho);
    if (th < 1) continue;
    if ((long)(th + k - 1) * (tw + k - 1) > 256 * IG_MAXIT) continue;
    long tiles = (long)cdiv(Ho, th) * cdiv(Wo, tw);
    long cost = tiles * maxNT * 64 + tiles * (th + k - 1) * (tw + k - 1);  // MFMA slots + halo traffic
    if (best < 0 || cost < best || (cost == best && tw > btw)) {
      best = cost; bth = th; btw = tw;
    }
  }
  *TH = bth; *TW = btw;
  if (const char* e = getenv("FRCNN_IG_TW")) {
    int tw = atoi(e);
    if (tw >= 1 && tw <= Wo && tw <= maxNT) { *TW = tw; *TH = std::max(1, std::min(maxNT / tw, Ho)); }
  }
}

template <int KS, int CC, int BM, int MODE, int NIT>
static int launch_igemm_n(IgemmArgs& a, int klass, double flops, hipStream_t s) {
  const size_t aB = (size_t)CC * KS * KS * BM * 4, bB = (size_t)CC * NIT * 256 * 4;
  size_t lds = MODE == 1 ? 2 * (aB + bB) : aB + bB;
  static bool attr_set = false;
  if (!attr_set) {
    FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<KS, CC, BM, MODE, NIT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));   // (a few static words: amax.h)
    attr_set = true;
  }
  int grid = a.tilesX * a.tilesY * a.mTiles * a.splitK;
  float* const late_amax = (a.pool_amax && grid > AMAX_MAX_BLOCKS) ? a.pool_amax : nullptr;   // more blocks than a record has entries:
  if (late_amax) a.pool_amax = nullptr;                                                       // the magnitude in a pass of its own
  double bytes = 4.0 * ((double)a.Cin * a.H * a.W + (double)a.M * a.Ho * a.Wo);
  FR_LAUNCH(klass, flops, bytes, s, (conv_igemm_kernel<KS, CC, BM, MODE, NIT>), dim3(grid), dim3(256), lds, a);
  FR_LAUNCH_CHECK();
  if (late_amax) FR_TRY(tensor_absmax(a.pool_out, (long)a.M * a.Hp * a.Wp, late_amax, s));
  return FRCNN_OK;
}

template <int KS, int CC, int BM, int MODE>
static int launch_igemm(IgemmArgs& a, int klass, double flops, hipStream_t s) {
  const int plane = (a.TH + KS - 1) * (a.TW + KS - 1);
  if (plane <= 256) return launch_igemm_n<KS, CC, BM, MODE, 1>(a, klass, flops, s);
  if (plane <= 512) return launch_igemm_n<KS, CC, BM, MODE, 2>(a, klass, flops, s);
  return launch_igemm_n<KS, CC, BM, MODE, 4>(a, klass, flops, s);
}

int conv_igemm(const float* in, int Cin, int H, int W, const float* in_slope, const float* in_scale,
               const float* wp, const float* bias, int M, int k, int pad, float* out, int out_mode,
               double algo_flops, hipStream_t s, int ws_slot, const IgemmPool* pool, bool* pool_fused) {
  IgemmArgs a;
  a.pool_out = nullptr; a.pool_idx = nullptr; a.pool_slope = nullptr; a.pool_scale = nullptr; a.pool_amax = nullptr; a.Hp = a.Wp = 0;
  if (pool_fused) *pool_fused = false;
  a.in = in; a.in_slope = in_slope; a.in_scale = in_scale; a.wp = wp; a.bias = bias; a.out = out;
  a.Cin = Cin; a.H = H; a.W = W; a.M = M; a.Mpad = conv_mpad(M);
  a.Ho = H + 2 * pad - k + 1; a.Wo = W + 2 * pad - k + 1; a.pad = pad;
  FR_CHECK(a.Ho > 0 && a.Wo > 0, "conv_igemm: empty output (%dx%d, k=%d, pad=%d)", H, W, k, pad);
  FR_CHECK(k == 1 || k == 3 || k == 5 || k == 7, "conv_igemm: unsupported kernel size %d", k);
  // 64-row M tiles for every spatial kernel: 32 accumulator registers per lane -> 5 blocks per CU (LDS 27 KB
  // each).  A CU serves its oldest block first, so blocks retire one after the other and the last one runs
  // alone with its staging exposed; smaller blocks make that tail shorter (measured on every vgg_small layer:
  // 64-row tiles are 0-14% faster than 128-row tiles).  1x1 keeps 128 rows (large K chunks, LDS double buffer).
  static const int ig_bm128 = getenv("FRCNN_IG_BM128") ? atoi(getenv("FRCNN_IG_BM128")) : 0;
  const int BM = (a.Mpad == 64 || (k > 1 && !ig_bm128)) ? 64 : 128;
  const int cc = conv_cc(k);
  a.nChunks = cdiv(Cin, cc);
  a.mTiles = a.Mpad / BM;
  // split K until one wave of blocks fills the resident slots, keeping >= ~200 K rows per split
  const long slots = BM == 64 && k > 1 ? 256 * IG_BPC : 768;
  const int minChunks = k == 1 ? 2 : std::max(1, cdiv(200, cc * k * k));
  auto splits_for = [&](long blocks) {
    return (int)std::min<long>(std::min<long>(std::max<long>(1, slots / blocks), 24), std::max(1, a.nChunks / minChunks));
  };
  // fused max pool: needs the 4 x 32 tile (two rows per wave, neighbours in lane^1), 64-row M tiles, a plain store and
  // a single K split (the pooled map is a function of the complete sum)
  static const int ig_pool = getenv("FRCNN_IG_POOL") ? atoi(getenv("FRCNN_IG_POOL")) : 1;
  bool fuse = false;
  if (pool && ig_pool && k == 3 && BM == 64 && IG_NTW == 2 && out_mode == OUT_STORE && !getenv("FRCNN_IG_SPLITK")) {
    const long blocks = (long)cdiv(a.Wo, 32) * cdiv(a.Ho, 4) * a.mTiles;
    fuse = (Cin <= 4 ? 1 : splits_for(blocks)) == 1;
    // ... and the fixed tile must not cost a round of blocks: 200x113 in 4x32 tiles is 812 blocks = up to 4 per CU where the
    // free choice (760) needs 3 -- measured 305 us instead of 235 + 8.5 (pool) on that layer
    int fth, ftw;
    choose_tile(a.Ho, a.Wo, k, 64 * IG_NTW, &fth, &ftw);
    const long free_blocks = (long)cdiv(a.Wo, ftw) * cdiv(a.Ho, fth) * a.mTiles;
    if (cdivl(blocks, 256) > cdivl(free_blocks, 256)) fuse = false;
  }
  if (fuse) {
    a.TH = 4; a.TW = 32;
    a.pool_out = pool->out; a.pool_idx = pool->idx; a.pool_slope = pool->slope; a.pool_scale = pool->scale; a.pool_amax = pool->amax;
    a.Hp = (a.Ho - 2 + 1) / 2 + 1; a.Wp = (a.Wo - 2 + 1) / 2 + 1;
    if (pool_fused) *pool_fused = true;
  } else {
    choose_tile(a.Ho, a.Wo, k, 64 * IG_NTW, &a.TH, &a.TW);
  }
  a.tilesX = cdiv(a.Wo, a.TW); a.tilesY = cdiv(a.Ho, a.TH);
  long blocks = (long)a.tilesX * a.tilesY * a.mTiles;
  int splitK = splits_for(blocks);
  if (const char* e = getenv("FRCNN_IG_SPLITK")) splitK = std::max(1, std::min(a.nChunks, atoi(e)));
  a.chunksPerSplit = cdiv(a.nChunks, splitK);
  a.splitK = cdiv(a.nChunks, a.chunksPerSplit);
  a.out_mode = out_mode;
  bool slab = false;
  if (a.splitK > 1) {
    // partial tiles go to slabs [split][M][Ho*Wo] with plain stores and one reduce pass adds the bias
    // (fp32 atomics on the shared result serialise in the memory-side atomic units: 2 splits cost 5-10%
    // more than the slab pass, 5-8 splits 50-70 us per launch)
    float* ws = nullptr;
    FR_TRY(ig_workspace((size_t)a.splitK * M * a.Ho * a.Wo * 4, &ws, ws_slot & 7));
    a.out = ws; a.out_mode = 3; a.bias = nullptr; slab = true;
  }
  static const int ig_dma = getenv("FRCNN_IG_DMA") ? atoi(getenv("FRCNN_IG_DMA")) : 1;
  a.dma_patch = ig_dma && k > 1 && !in_slope && !in_scale && (double)Cin * H * W * 4.0 < 2147483647.0 ? 1 : 0;
  if (algo_flops <= 0) algo_flops = 2.0 * M * Cin * k * k * (double)a.Ho * a.Wo;
  int klass = k == 3 ? KC_CONV_IGEMM_K3 : KC_CONV_IGEMM_OTHER;
  int rc;
  if (k == 3 && Cin <= 4) {  // first layer: a 4-channel chunk (the packed rows are ordered by channel pair)
    a.nChunks = 1; a.chunksPerSplit = 1; a.splitK = 1; a.out_mode = out_mode; a.bias = bias; a.out = out;
    rc = BM == 64 ? launch_igemm<3, 4, 64, 0>(a, klass, algo_flops, s)
                  : launch_igemm<3, 4, 128, 0>(a, klass, algo_flops, s);
  } else if (k == 3) {
    rc = BM == 64 ? launch_igemm<3, 8, 64, 0>(a, klass, algo_flops, s)
                  : launch_igemm<3, 8, 128, 0>(a, klass, algo_flops, s);
  } else if (k == 1) {
    rc = BM == 64 ? launch_igemm<1, 32, 64, 1>(a, klass, algo_flops, s)
                  : launch_igemm<1, 32, 128, 1>(a, klass, algo_flops, s);
  } else if (k == 5) {
    rc = BM == 64 ? launch_igemm<5, 2, 64, 0>(a, klass, algo_flops, s)
                  : launch_igemm<5, 2, 128, 0>(a, klass, algo_flops, s);
  } else {
    rc = BM == 64 ? launch_igemm<7, 2, 64, 0>(a, klass, algo_flops, s)
                  : launch_igemm<7, 2, 128, 0>(a, klass, algo_flops, s);
  }
  FR_TRY(rc);
  if (slab) {
    long total = (long)M * a.Ho * a.Wo;
    int grid = (int)std::min<long>(cdivl(total, 256), 4096);
    FR_LAUNCH(KC_ELEMWISE, 0, total * 4.0 * (a.splitK + 1), s, splitk_reduce_kernel, dim3(grid), dim3(256), 0,
              (const float*)a.out, a.splitK, M, (long)a.Ho * a.Wo, bias, out, out_mode == OUT_ADD ? 1 : 0);
    FR_LAUNCH_CHECK();
  }
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------
// weight gradient: gw[o][c][ky][kx] += sum_pix g[o][pix] * act(in)[c][pix + (ky,kx) - pad]
// GEMM view per tap: M = o (32/wave), N = c (32/wave), K = pixels.  One block = 64 o x 64 c x
// (TYS x KS taps) = 2x2 waves, each wave keeping all its taps in the accumulator file (3x3: nine
// 32x32 tiles = 144 AGPRs).  A block walks its share of <=64-pixel tiles (gradient tile + input
// patch with halo staged in LDS, odd pitches -> both operand reads conflict-free); two blocks per
// CU overlap one block's staging with the other's MFMAs.
// Split-K over pixel tiles: every block writes its partial tile to a slab [split][tap][o][c]
// with coalesced plain stores (fp32 atomics to the shared result serialise in the memory-side
// atomic units: measured 450-650 us per launch), then wgrad_reduce_kernel folds the slabs into the
// accumulating gradient tensor.
// ------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* in;
  const float* in_slope;
  const float* in_scale;
  const float* g;
  float* slab;   // [nSplit][KS*KS][O][Cin]
  int Cin, H, W, O, Ho, Wo, pad;
  int TH, TW, tilesX, tilesY;
  int oTiles, cTiles, kyGroups, nSplit;
  int dbg;  // tuning knobs (tools/bench_conv.py): bit0 skip epilogue, bit1 skip MFMA loop, bit2 skip staging
};

#define WG_NT 64     // staged gradient pixels per tile (TH*TW <= 64)
#define WG_NTP 65    // odd pitch -> conflict-free column reads
#define WG_PM 3      // patch slots per lane (patch plane <= 192)
#define WG_PP 193    // LDS pitch of a patch channel: odd (conflict-free) and >= 64*WG_PM (unconditional stores)

template <int KS, int TYS, int PM>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs p) {
  constexpr int NTAP = TYS * KS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* gs = smem;                  // [64][WG_NTP]
  float* ps = smem + 64 * WG_NTP;    // [64][PP], PP = 129 | 193
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave >> 1, wc = wave & 1;
  const int h = lane >> 5, li = lane & 31;

  int bid = blockIdx.x;
  const int ot = bid % p.oTiles; bid /= p.oTiles;
  const int ct = bid % p.cTiles; bid /= p.cTiles;
  const int kyg = bid % p.kyGroups;
  const int split = bid / p.kyGroups;
  const int o0 = ot * 64, c0 = ct * 64, ky0 = kyg * TYS;
  const int PW = p.TW + KS - 1, PHs = p.TH + TYS - 1;
  const int pplane = PHs * PW;
  constexpr int PP = PM == 2 ? 129 : WG_PP;   // odd pitch >= 64*PM
  const int NT = p.TH * p.TW, halfrows = p.TH >> 1;
  const int HW = p.H * p.W, HoWo = p.Ho * p.Wo;
  const bool has_slope = p.in_slope != nullptr, has_scale = p.in_scale != nullptr;
  const float slope = has_slope ? *p.in_slope : 1.f;
  const bool wave_active = (c0 + wc * 32 < p.Cin) && (o0 + wo * 32 < p.O);

  f32x16 acc[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- per-lane staging geometry (fixed for the whole launch)
  const int g_ty = lane / p.TW, g_tx = lane - g_ty * p.TW;
  const bool g_in = lane < NT;
  int p_r[PM], p_c[PM];
  bool p_in[PM];
#pragma unroll
  for (int m = 0; m < PM; ++m) {
    const int e = lane + 64 * m;
    p_r[m] = e / PW; p_c[m] = e - p_r[m] * PW;
    p_in[m] = e < pplane;
  }
  // the 16 channels / gradient rows this wave stages
  const int so0 = o0 + wave * 16, sc0 = c0 + wave * 16;
  const bool stage_full = so0 + 16 <= p.O && sc0 + 16 <= p.Cin;
  const long g_bytes = (long)HoWo * 4, in_bytes = (long)HW * 4;
  const char* gb = reinterpret_cast<const char*>(p.g + (size_t)so0 * HoWo);
  const char* ib = reinterpret_cast<const char*>(p.in + (size_t)sc0 * HW);
  // lane j holds the dropout scale of staged channel sc0 + j
  const float scv = has_scale ? p.in_scale[min(sc0 + (lane & 15), p.Cin - 1)] : 1.f;

  // Staging = two batches of 8 rows/channels per pixel tile.  Every batch issues ALL its global loads first
  // (unconditional, offsets clamped into the tensor -> no divergent branches, one latency exposure per
  // batch), then selects the zero fill and writes LDS.  Written for instruction count (a staging wave shares
  // its SIMD with a wave that saturates the matrix pipe's issue port): scalar row/channel bases + per-lane
  // byte offsets, activation flags resolved outside the loops.  (Prefetching the next tile into registers
  // during the MFMAs was measured slower: 144 accumulator + 48 staging registers spill, and scratch traffic
  // shares vmcnt with the prefetch.)
  auto batch = [&](int b, bool gok, unsigned gofs, const bool* pok, const unsigned* pofs, auto slope_c, auto scale_c, auto full_c) {
    constexpr bool SLOPE = decltype(slope_c)::value, SCALE = decltype(scale_c)::value, FULL = decltype(full_c)::value;
    float vg[8], vp[8][PM];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // rows / channels past the end re-read the last valid one (finite data) and are zeroed below
      const long jo = FULL ? b * 8 + j : b * 8 + j - max(0, so0 + b * 8 + j - (p.O - 1));
      const long jc = FULL ? b * 8 + j : b * 8 + j - max(0, sc0 + b * 8 + j - (p.Cin - 1));
      vg[j] = *reinterpret_cast<const float*>(gb + jo * g_bytes + gofs);
#pragma unroll
      for (int m = 0; m < PM; ++m) vp[j][m] = *reinterpret_cast<const float*>(ib + jc * in_bytes + pofs[m]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ook = FULL || so0 + b * 8 + j < p.O, cok = FULL || sc0 + b * 8 + j < p.Cin;
      float sc = 1.f;
      if (SCALE) sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, scv), b * 8 + j));
      gs[(wave * 16 + b * 8 + j) * WG_NTP + lane] = (gok && ook) ? vg[j] : 0.f;
#pragma unroll
      for (int m = 0; m < PM; ++m) {
        float v = vp[j][m];
        if (SLOPE) v = v > 0.f ? v : slope * v;
        if (SCALE) v *= sc;
        ps[(wave * 16 + b * 8 + j) * PP + lane + 64 * m] = (pok[m] && cok) ? v : 0.f;
      }
    }
  };
  auto stage_tile = [&](int t, auto slope_c, auto scale_c) {
    const int oy0 = (t / p.tilesX) * p.TH, ox0 = (t % p.tilesX) * p.TW;
    const int goy = oy0 + g_ty, gox = ox0 + g_tx;
    const bool gok = g_in && goy < p.Ho && gox < p.Wo;
    const unsigned gofs = gok ? (unsigned)(goy * p.Wo + gox) * 4u : 0u;
    bool pok[PM];
    unsigned pofs[PM];
#pragma unroll
    for (int m = 0; m < PM; ++m) {
      const int iy = oy0 - p.pad + ky0 + p_r[m], ix = ox0 - p.pad + p_c[m];
      pok[m] = p_in[m] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      pofs[m] = pok[m] ? (unsigned)(iy * p.W + ix) * 4u : 0u;
    }
    if (stage_full) {
      batch(0, gok, gofs, pok, pofs, slope_c, scale_c, std::true_type{});
      batch(1, gok, gofs, pok, pofs, slope_c, scale_c, std::true_type{});
    } else {
      batch(0, gok, gofs, pok, pofs, slope_c, scale_c, std::false_type{});
      batch(1, gok, gofs, pok, pofs, slope_c, scale_c, std::false_type{});
    }
  };

  const int nPix = p.tilesX * p.tilesY;
  for (int t = split; t < nPix; t += p.nSplit) {
    if (!(p.dbg & 4)) {
      if (has_slope) {
        if (has_scale) stage_tile(t, std::true_type{}, std::true_type{}); else stage_tile(t, std::true_type{}, std::false_type{});
      } else {
        if (has_scale) stage_tile(t, std::false_type{}, std::true_type{}); else stage_tile(t, std::false_type{}, std::false_type{});
      }
    }
    __syncthreads();
    if (wave_active && !(p.dbg & 2)) {
      // K = pixels: lane half h walks rows [h*halfrows, (h+1)*halfrows).  Along a row the KS taps of one
      // patch row are a sliding window: per pixel ONE new patch value per tap row (+ one gradient value)
      // is read from LDS for TYS*KS MFMAs; the gradient value of the next pixel is fetched a step ahead and
      // the MFMAs that only need window values already in registers are issued first.
      const float* ga = gs + (wo * 32 + li) * WG_NTP + h * halfrows * p.TW;
      const float* pb = ps + (wc * 32 + li) * PP + h * halfrows * PW;
      float a_cur = ga[0];
      for (int r = 0; r < halfrows; ++r) {
        const float* gar = ga + r * p.TW;
        const float* pbr = pb + r * PW;
        float w[TYS][KS];
#pragma unroll
        for (int ty = 0; ty < TYS; ++ty)
#pragma unroll
          for (int kx = 0; kx < KS - 1; ++kx) w[ty][kx] = pbr[ty * PW + kx];
        for (int x0 = 0; x0 < p.TW; x0 += KS) {
#pragma unroll
          for (int st = 0; st < KS; ++st) {
            if (x0 + st < p.TW) {
              const int x = x0 + st;
              const float a_next = gar[x + 1];   // next pixel (next row's first one at the row end)
#pragma unroll
              for (int ty = 0; ty < TYS; ++ty) w[ty][(st + KS - 1) % KS] = pbr[ty * PW + x + KS - 1];
#pragma unroll
              for (int ty = 0; ty < TYS; ++ty)
#pragma unroll
                for (int kx = 0; kx < KS - 1; ++kx)
                  acc[ty * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, w[ty][(st + kx) % KS], acc[ty * KS + kx], 0, 0, 0);
#pragma unroll
              for (int ty = 0; ty < TYS; ++ty)
                acc[ty * KS + KS - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, w[ty][(st + KS - 1) % KS], acc[ty * KS + KS - 1], 0, 0, 0);
              a_cur = a_next;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  // ---- epilogue: D col = lane&31 -> c (contiguous in the slab), row -> o
  if (wave_active && !(p.dbg & 1)) {
    const int c = c0 + wc * 32 + li;
    const size_t OC = (size_t)p.O * p.Cin;
    float* sl = p.slab + (size_t)split * KS * KS * OC;
#pragma unroll
    for (int ty = 0; ty < TYS; ++ty) {
      const int ky = ky0 + ty;
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o0 + wo * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (o < p.O && c < p.Cin && ky < KS) sl[(size_t)(ky * KS + kx) * OC + (size_t)o * p.Cin + c] = acc[ty * KS + kx][r];
        }
      }
    }
  }
}

// gw[o][c][tap] += sum_s slab[s][tap][o][c].  Block = 64 elements x 4 split groups (the slabs are the
// long dimension when the result is small), 4 loads in flight per thread.
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int nSplit, int taps, int OC, float* __restrict__ gw) {
  __shared__ float sh[4][64];
  const long total = (long)taps * OC;
  const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
  for (long t0 = (long)blockIdx.x * 64; t0 < total; t0 += (long)gridDim.x * 64) {
    const long t = t0 + tx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (t < total) {
      int s = g;
      for (; s + 12 < nSplit; s += 16) {
        a0 += slab[(size_t)s * total + t];
        a1 += slab[(size_t)(s + 4) * total + t];
        a2 += slab[(size_t)(s + 8) * total + t];
        a3 += slab[(size_t)(s + 12) * total + t];
      }
      for (; s < nSplit; s += 4) a0 += slab[(size_t)s * total + t];
    }
    sh[g][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && t < total) {
      const int tap = (int)(t / OC);
      const int oc = (int)(t - (long)tap * OC);
      gw[(size_t)oc * taps + tap] += (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
    }
    __syncthreads();
  }
}

// The same fold for a SMALL result under MANY slabs (first layer: 1728 elements x 512 slabs): 16 elements x 64 slab groups per
// block, so every thread has its (up to 8) loads in flight at once -- the kernel above walked 128 slabs per thread four at a
// time on 27 blocks (~12 us of the first layer's ~48).  Same fixed summation order on every run.
__global__ __launch_bounds__(1024) void wgrad_reduce_tall_kernel(const float* __restrict__ slab, int nSplit, int taps, int OC,
                                                                 float* __restrict__ gw, int nExtra, float* __restrict__ gbias,
                                                                 float* __restrict__ gslope) {
  __shared__ float sh[64][17];
  const int total = taps * OC + nExtra;   // per slab: [tap][o][c], then (fused first layer) O bias sums and the slope sum
  const int tx = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int t = blockIdx.x * 16 + tx;
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = 0.f;
  if (t < total) {
    for (int s0 = g; s0 < nSplit; s0 += 512) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int s = s0 + 64 * i;
        if (s < nSplit) a[i] += slab[(size_t)s * total + t];
      }
    }
  }
  sh[g][tx] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
#pragma unroll
  for (int w = 32; w > 0; w >>= 1) {
    if (g < w) sh[g][tx] += sh[g + w][tx];
    __syncthreads();
  }
  if (g == 0 && t < total) {
    if (t < taps * OC) {
      const int tap = t / OC, oc = t - tap * OC;
      gw[(size_t)oc * taps + tap] += sh[0][tx];
    } else if (t < total - 1) {
      if (gbias) gbias[t - taps * OC] += sh[0][tx];
    } else if (gslope) {
      *gslope += sh[0][tx];
    }
  }
}
static void wgrad_reduce_first(const float* slab, int nSplit, int OC, float* gw, hipStream_t s, int nExtra = 0, float* gbias = nullptr,
                               float* gslope = nullptr) {
  if (nSplit >= 128 || nExtra)
    hipLaunchKernelGGL(wgrad_reduce_tall_kernel, dim3(cdiv(9 * OC + nExtra, 16)), dim3(1024), 0, s, slab, nSplit, 9, OC, gw, nExtra, gbias,
                       gslope);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)std::min<long>(cdivl(9L * OC, 64), 4096)), dim3(256), 0, s, slab, nSplit, 9, OC, gw);
}

// ------------------------------------------------------------------------------------------
// weight gradient of the FIRST layer (Cin*k*k <= 32, O <= 64: 3 -> 64 @ 450x800).  With three input
// channels the (o, c) tiling above would leave 29 of 32 MFMA columns empty, so here the GEMM is
// M = o (two 32-row tiles), N = (c, ky, kx) = 27 columns of ONE 32x32 tile, K = pixels.  Every wave
// owns one row of a 4 x 64 pixel tile; the kernel is bound by streaming the gradient map once.
// ------------------------------------------------------------------------------------------
#define W1_TH 4
#define W1_TW 64
#define W1_GP 257   // odd pitch of gs[o][256]
#define W1_PATCH ((3 * (W1_TH + 2) * (W1_TW + 2) + 255) / 256 * 256)   // floats of the input patch (<= 3 channels), whole passes of 256
// FUSED (round 4): the gradient tile is not read but COMPUTED while it is staged -- the 2x2 max-pooling backward and the PReLU
// backward of the layer's own output (elem.hip act_backward_kernel<POOLED>): g = (the window's winner ? gpooled : 0) * prelu'(x)
// -- so the full-resolution gradient (92 MB for 64 x 450 x 800) is neither written nor read back.  The slope gradient leaves
// through one atomic per block; the bias gradient is column 27 of the product: a B column of ones sums g over the pixels.
struct WgradFirstFused {
  const float* gpool;            // [O][Hp][Wp] gradient of the pooled map
  const unsigned char* pidx;     // [O][Hp][Wp] winner code 2 * (y & 1) + (x & 1)
  const float* x;                // [O][Ho][Wo] pre-activation output of this convolution
  const float* slope;            // device scalar
  float* gslope;                 // device scalar, accumulated
  float* gbias;                  // [O], accumulated
  int Hp, Wp;
};
template <int KS, bool FUSED>
__global__ __launch_bounds__(256, 2) void conv_wgrad_first_kernel(WgradArgs p, WgradFirstFused q) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int PW = W1_TW + KS - 1, PH = W1_TH + KS - 1, PLANE = PH * PW;
  float* gs = smem;                   // [64][W1_GP]
  float* ps = smem + 64 * W1_GP;      // [Cin][PH][PW]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, li = lane & 31;
  const int ntap = p.Cin * KS * KS;
  // lane li is output column j = (c, ky, kx)
  const int jj = li < ntap ? li : 0;
  const int jc = jj / (KS * KS), jt = jj % (KS * KS);
  const int tapoff = jc * PLANE + (jt / KS) * PW + (jt % KS);
  const int HW = p.H * p.W, HoWo = p.Ho * p.Wo;
  const bool ones = FUSED && li == ntap;     // the extra B column of ones (bias gradient): reads a row of 1.0 behind the patch
  const float pa = FUSED ? *q.slope : 1.f;
  float sa = 0.f;                            // slope gradient: sum over x <= 0 of x * g
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int nPix = p.tilesX * p.tilesY;
  // Staging (round 4): thread = one 2x2 window of the tile (64 windows: 2 rows of 32) x 16 channels (its wave's), so a channel
  // costs two 8-byte loads of the full-resolution tensor -- plus, FUSED, one dword of the pooled gradient and one winner byte --
  // instead of four dword loads; and the loads of tile t + 1 are issued BEFORE the MFMA phase of tile t (registers: 64 / 96),
  // where the first version waited four times per tile for 16 loads each with eight waves a CU (latency-bound: 48 us for
  // 96 MB; fused 160 us).
  constexpr int NPATCH = (3 * PLANE + 255) / 256;
  const int wq = tid & 63, wy = wq >> 5, wx = wq & 31, o0 = wave * 16;
  const float* full = FUSED ? q.x : p.g;
  const int HpWp = FUSED ? q.Hp * q.Wp : 0;
  // buffer loads: one resource per tensor (SGPRs), the channel's plane as the scalar offset, ONE 32-bit lane offset per row --
  // with global loads the compiler kept a 64-bit address pair per load and spilled
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t full_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(full), 0, p.O * HoWo * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t gp_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(FUSED ? q.gpool : full), 0, FUSED ? p.O * HpWp * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ib_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(FUSED ? q.pidx : nullptr), 0, FUSED ? p.O * HpWp : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.Cin * HW * 4, 0x00020000);
  float2 r0[16], r1[16];
  float gp[FUSED ? 16 : 1];
  unsigned char ib[FUSED ? 16 : 1];
  float pv[NPATCH];
  bool shifted = false, okx0 = false, okx1 = false, oky0 = false, oky1 = false;
  auto load = [&](int t, bool live) {   // !live (no next tile): every offset outside its resource -- zeros, no traffic, no branch
    const int oy0 = (t / p.tilesX) * W1_TH, ox0 = (t % p.tilesX) * W1_TW;
    const int y0 = oy0 + 2 * wy, x0 = ox0 + 2 * wx;
    // the pair (x0, x0 + 1) is read from column min(x0, Wo - 2): at an odd width the last column arrives in .y
    // (the launchers take this kernel for Wo >= 2 only)
    const int xb = x0 < p.Wo - 1 ? x0 : p.Wo - 2;
    shifted = x0 == p.Wo - 1;
    okx0 = x0 < p.Wo; okx1 = x0 + 1 < p.Wo; oky0 = y0 < p.Ho; oky1 = y0 + 1 < p.Ho;
    const int ya = y0 < p.Ho ? y0 : p.Ho - 1, yb = y0 + 1 < p.Ho ? y0 + 1 : p.Ho - 1;
    // (BYTE offsets in 32 bits: base in SGPRs + zero-extended lane offset, not a 64-bit address per load)
    const unsigned ofa = live ? 4u * (ya * p.Wo + xb) : 0x7FFFFFFFu, ofb = live ? 4u * (yb * p.Wo + xb) : 0x7FFFFFFFu;
    unsigned pofs = 0;
    if (FUSED) {
      const int py = (y0 >> 1) < q.Hp ? (y0 >> 1) : q.Hp - 1, px = (x0 >> 1) < q.Wp ? (x0 >> 1) : q.Wp - 1;
      pofs = live ? py * q.Wp + px : 0x1FFFFFFFu;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int o = o0 + j, oo = o < p.O ? o : 0;
      const unsigned so = (unsigned)oo * (unsigned)HoWo * 4u;   // wave-uniform: the channel's plane rides in soffset
      const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(full_rsrc, ofa, so, 0);
      const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(full_rsrc, ofb, so, 0);
      r0[j].x = __uint_as_float(a.x); r0[j].y = __uint_as_float(a.y);
      r1[j].x = __uint_as_float(b.x); r1[j].y = __uint_as_float(b.y);
      if (FUSED) {
        const unsigned sp = (unsigned)oo * (unsigned)HpWp;
        gp[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gp_rsrc, 4u * pofs, 4u * sp, 0));
        ib[j] = __builtin_amdgcn_raw_buffer_load_b8(ib_rsrc, pofs, sp, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NPATCH; ++i) {
      const int e = tid + 256 * i;
      const int c = e / PLANE, r2 = e - c * PLANE;
      const int r = r2 / PW, col = r2 - r * PW;
      const int iy = oy0 - p.pad + r, ix = ox0 - p.pad + col;
      // (a position outside the image or the patch gets an offset outside the resource and arrives as zero: no branch)
      const bool okp = live & (e < p.Cin * PLANE) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
      pv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(in_rsrc, okp ? 4u * (unsigned)(c * HW + iy * p.W + ix) : 0x7FFFFFFFu, 0, 0));
    }
  };
  if (FUSED && tid < W1_TW) ps[NPATCH * 256 + tid] = 1.f;   // (visible after the first barrier; nothing else writes there)
  const bool dbg_dead = p.dbg & 8;
  load(blockIdx.x, (int)blockIdx.x < nPix && !dbg_dead);
  for (int t = blockIdx.x; t < nPix; t += gridDim.x) {
    // ---- gradient tile: 64 rows (o) x 256 pixels (4 rows of 64)
    if (!(p.dbg & 4)) {
      float* gw = gs + (2 * wy) * W1_TW + 2 * wx;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int o = o0 + j;
        const bool oko = o < p.O;
        float v[4] = {shifted ? r0[j].y : r0[j].x, r0[j].y, shifted ? r1[j].y : r1[j].x, r1[j].y};
        const bool ok[4] = {oko && oky0 && okx0, oko && oky0 && okx1, oko && oky1 && okx0, oko && oky1 && okx1};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float g;
          if (FUSED) {
            g = (ok[e] && ib[j] == e) ? gp[j] : 0.f;
            const bool pos = v[e] > 0.f;
            sa = fmaf(pos ? 0.f : v[e], g, sa);
            g = pos ? g : pa * g;
          } else {
            g = ok[e] ? v[e] : 0.f;
          }
          gw[o * W1_GP + (e >> 1) * W1_TW + (e & 1)] = g;
        }
        // one channel at a time: left alone, the scheduler does all 64 stores first and the slope sums last, with every
        // value, gradient and comparison mask of the tile alive in between (256 registers, masks spilled lane by lane)
        if (FUSED) asm volatile("" : "+v"(sa)::"memory");   // (and the slope sums HERE: they were moved behind the barrier)
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < NPATCH; ++i) ps[tid + 256 * i] = pv[i];   // (NPATCH * 256 floats are there; zeros behind the patch)
    }
    __syncthreads();
    load(t + gridDim.x, t + (int)gridDim.x < nPix && !dbg_dead);   // in flight under the MFMA phase
    if (!(p.dbg & 2)) {
      const float* ga = gs + li * W1_GP + wave * W1_TW + h;          // pixel pair (x, x+1): h selects
      const float* pb = ones ? ps + NPATCH * 256 : ps + tapoff + wave * PW + h;
#pragma unroll 4
      for (int x = 0; x < W1_TW; x += 2) {
        const float b = pb[x];
        const float a0 = ga[x], a1 = ga[32 * W1_GP + x];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- the four waves meet in LDS, then one slab slice [tap][o][c] per block
  float* red = smem;  // [4][64][32]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      red[(wave * 64 + o) * 32 + li] = acc[t][r];
    }
  __syncthreads();
  const int OC = p.O * p.Cin;
  // FUSED: the bias sums (column 27) and the slope sum ride behind the block's slab slice and are folded with it -- 512 blocks
  // adding to the same 65 addresses with atomics cost ~90 us (they serialise at the memory side)
  const int stride = KS * KS * OC + (FUSED ? p.O + 1 : 0);
  float* sl = p.slab + (size_t)blockIdx.x * stride;
  for (int e = tid; e < 64 * 32; e += 256) {
    const int o = e >> 5, j = e & 31;
    if (o < p.O && j < ntap) {
      const float v = red[e] + red[64 * 32 + e] + red[2 * 64 * 32 + e] + red[3 * 64 * 32 + e];
      sl[(size_t)(j % (KS * KS)) * OC + o * p.Cin + j / (KS * KS)] = v;
    } else if (FUSED && o < p.O && j == ntap) {
      sl[KS * KS * OC + o] = red[e] + red[64 * 32 + e] + red[2 * 64 * 32 + e] + red[3 * 64 * 32 + e];
    }
  }
  if (FUSED) {
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sa += __shfl_xor(sa, o);
    if (lane == 0) red[wave] = sa;
    __syncthreads();
    if (tid == 0) sl[KS * KS * OC + p.O] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

static void choose_wgrad_tile(int Ho, int Wo, int k, int tys, int* TH, int* TW) {
  long best = -1;
  int bth = 2, btw = 1;
  for (int th = 2; th <= 8; th += 2) {
    for (int tw = 1; tw <= std::min(Wo, WG_NT / th); ++tw) {
      if (tw < 8 && Wo >= 8) continue;
      if ((th + tys - 1) * (tw + k - 1) > 64 * WG_PM) continue;
      long tiles = (long)cdiv(Ho, th) * cdiv(Wo, tw);
      long cost = tiles * th * tw * (long)(k * tys) + tiles * 64;  // MFMA K-steps + per-tile staging/barrier overhead
      if (best < 0 || cost < best || (cost == best && tw > btw)) {
        best = cost; bth = th; btw = tw;
      }
    }
  }
  *TH = bth; *TW = btw;
}

static bool g_wgrad_first_ok = true;  // cleared by conv_wgrad when the input carries a fused activation
static bool wgrad_is_first(int Cin, int O, int k, int Wo) { return g_wgrad_first_ok && k == 3 && Cin * 9 <= 32 && O <= 64 && Wo >= 2; }

static void wgrad_plan(WgradArgs& a, int k) {
  if (wgrad_is_first(a.Cin, a.O, k, a.Wo)) {
    a.TH = W1_TH; a.TW = W1_TW;
    a.tilesX = cdiv(a.Wo, W1_TW); a.tilesY = cdiv(a.Ho, W1_TH);
    a.oTiles = a.cTiles = a.kyGroups = 1;
    a.nSplit = std::min(512, a.tilesX * a.tilesY);   // one slab slice per block
    a.dbg = 0;
    if (const char* e = getenv("FRCNN_WG_DBG")) a.dbg = atoi(e);
    if (const char* e = getenv("FRCNN_W1_SPLIT")) a.nSplit = std::min(atoi(e), a.tilesX * a.tilesY);
    return;
  }
  const int tys = k == 3 ? 3 : 1;
  choose_wgrad_tile(a.Ho, a.Wo, k, tys, &a.TH, &a.TW);
  a.tilesX = cdiv(a.Wo, a.TW); a.tilesY = cdiv(a.Ho, a.TH);
  a.oTiles = cdiv(a.O, 64); a.cTiles = cdiv(a.Cin, 64);
  a.kyGroups = k / tys;
  long base = (long)a.oTiles * a.cTiles * a.kyGroups;
  long npix = (long)a.tilesX * a.tilesY;
  // two blocks per CU: aim for ~512 blocks
  a.nSplit = (int)std::max<long>(1, std::min<long>(npix, (512 + base / 2) / base));
  a.dbg = 0;
  if (const char* e = getenv("FRCNN_WG_DBG")) a.dbg = atoi(e);
  if (const char* e = getenv("FRCNN_WG_NSPLIT")) a.nSplit = (int)std::max<long>(1, std::min<long>(npix, atoi(e)));
}

// the same with four consecutive (o, c) pairs of one tap per thread: 16-byte slab loads, four of them in flight (needs 4 | OC)
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* __restrict__ slab, int nSplit, int taps, int OC,
                                                            float* __restrict__ gw) {
  __shared__ float4 sh[4][64];
  const long total = (long)taps * OC;
  const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
  for (long t0 = (long)blockIdx.x * 256; t0 < total; t0 += (long)gridDim.x * 256) {
    const long t = t0 + 4 * tx;
    float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    auto add = [](float4& a, const float4 x) { a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; };
    if (t < total) {
      int s = g;
      for (; s + 12 < nSplit; s += 16) {   // (the order of round 2's scalar kernel: the sums are the same numbers)
        const float4 x0 = *reinterpret_cast<const float4*>(slab + (size_t)s * total + t);
        const float4 x1 = *reinterpret_cast<const float4*>(slab + (size_t)(s + 4) * total + t);
        const float4 x2 = *reinterpret_cast<const float4*>(slab + (size_t)(s + 8) * total + t);
        const float4 x3 = *reinterpret_cast<const float4*>(slab + (size_t)(s + 12) * total + t);
        add(a0, x0); add(a1, x1); add(a2, x2); add(a3, x3);
      }
      for (; s < nSplit; s += 4) add(a0, *reinterpret_cast<const float4*>(slab + (size_t)s * total + t));
    }
    sh[g][tx] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                            (a0.w + a1.w) + (a2.w + a3.w));
    __syncthreads();
    if (g == 0 && t < total) {
      const int tap = (int)(t / OC);
      const int oc = (int)(t - (long)tap * OC);
      const float4 p0 = sh[0][tx], p1 = sh[1][tx], p2 = sh[2][tx], p3 = sh[3][tx];
      float* d = gw + (size_t)oc * taps + tap;
      d[0] += (p0.x + p1.x) + (p2.x + p3.x);
      d[taps] += (p0.y + p1.y) + (p2.y + p3.y);
      d[2 * (size_t)taps] += (p0.z + p1.z) + (p2.z + p3.z);
      d[3 * (size_t)taps] += (p0.w + p1.w) + (p2.w + p3.w);
    }
    __syncthreads();
  }
}

int wgrad_reduce(const float* slab, int nSplit, int taps, int OC, float* gw, hipStream_t s) {
  const long total = (long)taps * OC;
  if (OC % 4 == 0 && ((uintptr_t)slab & 15) == 0) {
    const int rgrid = (int)std::min<long>(cdivl(total, 256), 4096);
    hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(rgrid), dim3(256), 0, s, slab, nSplit, taps, OC, gw);
    return FRCNN_OK;
  }
  const int rgrid = (int)std::min<long>(cdivl(total, 64), 4096);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rgrid), dim3(256), 0, s, slab, nSplit, taps, OC, gw);
  return FRCNN_OK;
}

// the fold through a WgradMap: wgrad_reduce4_kernel's walk (four consecutive (o', c') pairs of one tap per thread, 16-byte slab
// loads, four split groups per block folded through LDS, the same summation order), the four results scattered through the map
__global__ __launch_bounds__(256) void wgrad_reduce_map_kernel(const float* __restrict__ slab, int nSplit, int taps, int O, int C,
                                                               float* __restrict__ gw, WgradMap map) {
  __shared__ float4 sh[4][64];
  const long OC = (long)O * C, total = (long)taps * OC;
  const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
  for (long t0 = (long)blockIdx.x * 256; t0 < total; t0 += (long)gridDim.x * 256) {
    const long t = t0 + 4 * tx;
    float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    auto add = [](float4& a, const float4 x) { a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; };
    if (t < total) {
      int s = g;
      for (; s + 12 < nSplit; s += 16) {
        const float4 x0 = *reinterpret_cast<const float4*>(slab + (size_t)s * total + t);
        const float4 x1 = *reinterpret_cast<const float4*>(slab + (size_t)(s + 4) * total + t);
        const float4 x2 = *reinterpret_cast<const float4*>(slab + (size_t)(s + 8) * total + t);
        const float4 x3 = *reinterpret_cast<const float4*>(slab + (size_t)(s + 12) * total + t);
        add(a0, x0); add(a1, x1); add(a2, x2); add(a3, x3);
      }
      for (; s < nSplit; s += 4) add(a0, *reinterpret_cast<const float4*>(slab + (size_t)s * total + t));
    }
    sh[g][tx] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                            (a0.w + a1.w) + (a2.w + a3.w));
    __syncthreads();
    if (g == 0 && t < total) {
      const int tap = (int)(t / OC);
      const long oc = t - (long)tap * OC;
      const int o = (int)(oc / C), c = (int)(oc - (long)o * C);   // (C % 4 == 0: the four pairs share their filter)
      const int od = map.omap ? map.omap[o] : o;
      const float4 p0 = sh[0][tx], p1 = sh[1][tx], p2 = sh[2][tx], p3 = sh[3][tx];
      const float r[4] = {(p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                          (p0.w + p1.w) + (p2.w + p3.w)};
      if (od >= 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cd = map.cmap ? map.cmap[c + e] : c + e;
          if (cd >= 0) gw[((size_t)od * map.Cfull + cd) * taps + tap] += r[e];
        }
      }
    }
    __syncthreads();
  }
}
int wgrad_reduce_map(const float* slab, int nSplit, int taps, int O, int C, float* gw, const WgradMap& map, hipStream_t s) {
  FR_CHECK(map.Cfull > 0 && C % 4 == 0 && ((uintptr_t)slab & 15) == 0, "wgrad_reduce_map: the full tensor's channel count is missing, or C %% 4 != 0");
  const int rgrid = (int)std::min<long>(cdivl((long)taps * O * C, 256), 4096);
  hipLaunchKernelGGL(wgrad_reduce_map_kernel, dim3(rgrid), dim3(256), 0, s, slab, nSplit, taps, O, C, gw, map);
  return FRCNN_OK;
}

size_t conv_wgrad_workspace_bytes(int Cin, int H, int W, int O, int k, int pad) {
  if (k == 3 && Cin % 64 == 0 && O % 64 == 0) {   // either form may run (option split_bf16): room for both
    WgradArgs b;
    b.Cin = Cin; b.H = H; b.W = W; b.O = O; b.pad = pad; b.Ho = H + 2 * pad - k + 1; b.Wo = W + 2 * pad - k + 1;
    wgrad_plan(b, k);
    return std::max((size_t)b.nSplit * k * k * O * Cin * 4 + 256, conv_wgradx_workspace_bytes(Cin, H, W, O, pad));
  }
  WgradArgs a;
  a.Cin = Cin; a.H = H; a.W = W; a.O = O; a.pad = pad;
  a.Ho = H + 2 * pad - k + 1; a.Wo = W + 2 * pad - k + 1;
  wgrad_plan(a, k);
  // (first-layer shapes: room for the bias / slope sums that conv_wgrad_first_pooled keeps behind every slab slice)
  return (size_t)a.nSplit * (k * k * O * Cin + O + 1) * 4 + 256;
}

template <int KS, int TYS, int PM>
static int launch_wgrad_pm(WgradArgs& a, int klass, double flops, float* gw, hipStream_t s) {
  size_t lds = ((size_t)64 * WG_NTP + (size_t)64 * (PM == 2 ? 129 : WG_PP)) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<KS, TYS, PM>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  int grid = a.oTiles * a.cTiles * a.kyGroups * a.nSplit;
  double bytes = 4.0 * ((double)a.Cin * a.H * a.W + (double)a.O * a.Ho * a.Wo);
  if (frcnn::prof_enabled(klass)) frcnn::prof_before(klass, s);
  hipLaunchKernelGGL((conv_wgrad_kernel<KS, TYS, PM>), dim3(grid), dim3(256), lds, s, a);
  const int OC = a.O * a.Cin;
  long total = (long)KS * KS * OC;
  int rgrid = (int)std::min<long>(cdivl(total, 64), 4096);
  if (!(a.dbg & 1))
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rgrid), dim3(256), 0, s, (const float*)a.slab, a.nSplit, KS * KS, OC, gw);
  if (frcnn::prof_enabled(klass)) frcnn::prof_after(klass, flops, bytes, s);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

template <int KS, int TYS>
static int launch_wgrad(WgradArgs& a, int klass, double flops, float* gw, hipStream_t s) {
  const int pplane = (a.TH + TYS - 1) * (a.TW + KS - 1);   // patch slots: 2 or 3 per lane
  if (pplane <= 128) return launch_wgrad_pm<KS, TYS, 2>(a, klass, flops, gw, s);
  return launch_wgrad_pm<KS, TYS, 3>(a, klass, flops, gw, s);
}

int conv_wgrad(const float* in, int Cin, int H, int W, const float* in_slope, const float* in_scale,
               const float* g, int O, int k, int pad, float* gw, void* ws, size_t ws_bytes, hipStream_t s, float* gbias,
               const float* amax_in, const float* amax_g, const WgradMap* map) {
  WgradArgs a;
  a.in = in; a.in_slope = in_slope; a.in_scale = in_scale; a.g = g;
  a.Cin = Cin; a.H = H; a.W = W; a.O = O; a.pad = pad;
  a.Ho = H + 2 * pad - k + 1; a.Wo = W + 2 * pad - k + 1;
  FR_CHECK(k == 1 || k == 3 || k == 5 || k == 7, "conv_wgrad: unsupported kernel size %d", k);
  if (conv_wgradx_eligible(Cin, O, k)) return conv_wgradx(in, Cin, H, W, in_slope, in_scale, g, O, pad, gw, ws, ws_bytes, s, gbias, amax_in, amax_g, map);
  FR_CHECK(!map || (!map->omap && !map->cmap), "conv_wgrad: a gathered weight gradient needs a conv_wgradx shape");
  if (gbias) FR_TRY(channel_sum(g, O, (long)a.Ho * a.Wo, gbias, s));   // the fp32 kernels leave the bias half to a pass of its own
  FR_CHECK((long)Cin * H * W < (1L << 31) && (long)O * a.Ho * a.Wo < (1L << 31), "conv_wgrad: tensor too large for 32-bit offsets");
  g_wgrad_first_ok = (in_slope == nullptr && in_scale == nullptr);
  wgrad_plan(a, k);
  size_t need = (size_t)a.nSplit * k * k * O * Cin * 4 + 256;
  FR_CHECK(ws && ws_bytes >= need, "conv_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
  a.slab = (float*)(((uintptr_t)ws + 255) / 256 * 256);
  double flops = 2.0 * O * Cin * k * k * (double)a.Ho * a.Wo;
  int klass = k == 3 ? KC_CONV_WGRAD_K3 : KC_CONV_WGRAD_OTHER;
  if (wgrad_is_first(Cin, O, k, a.Wo)) {
    FR_CHECK((double)O * a.Ho * a.Wo * 4.0 < 4294967295.0 && (double)Cin * H * W * 4.0 < 4294967295.0,
             "conv_wgrad: first-layer tensors too large for 32-bit buffer resources");
    size_t lds = ((size_t)64 * W1_GP + W1_PATCH) * 4;
    static bool attr_set = false;
    if (!attr_set) {
      FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_first_kernel<3, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
    if (frcnn::prof_enabled(klass)) frcnn::prof_before(klass, s);
    hipLaunchKernelGGL((conv_wgrad_first_kernel<3, false>), dim3(a.nSplit), dim3(256), lds, s, a, WgradFirstFused{});
    wgrad_reduce_first(a.slab, a.nSplit, O * Cin, gw, s);
    if (frcnn::prof_enabled(klass)) frcnn::prof_after(klass, flops, 4.0 * ((double)Cin * H * W + (double)O * a.Ho * a.Wo), s);
    FR_LAUNCH_CHECK();
    return FRCNN_OK;
  }
  if (k == 3) return launch_wgrad<3, 3>(a, klass, flops, gw, s);
  if (k == 1) return launch_wgrad<1, 1>(a, klass, flops, gw, s);
  if (k == 5) return launch_wgrad<5, 1>(a, klass, flops, gw, s);
  return launch_wgrad<7, 1>(a, klass, flops, gw, s);
}

// accGradParameters of the first layer straight from the POOLED map's gradient (see WgradFirstFused): replaces
// maxpool_act_backward + conv_wgrad for a block of one convolution whose input gradient nobody needs (objective.lua:189).
bool conv_wgrad_first_pooled_eligible(int Cin, int O, int k, int Wo) { return k == 3 && Cin * 9 < 32 && O <= 64 && Wo >= 2; }
int conv_wgrad_first_pooled(const float* in, int Cin, int H, int W, const float* gpool, const unsigned char* pidx, const float* x,
                            const float* slope, int O, int pad, float* gw, float* gbias, float* gslope, void* ws, size_t ws_bytes,
                            hipStream_t s) {
  FR_CHECK(conv_wgrad_first_pooled_eligible(Cin, O, 3, W + 2 * pad - 2) && slope, "conv_wgrad_first_pooled: not a first-layer shape");
  FR_CHECK((double)O * (H + 2 * pad - 2) * (W + 2 * pad - 2) * 4.0 < 4294967295.0 && (double)Cin * H * W * 4.0 < 4294967295.0,
           "conv_wgrad_first_pooled: tensors too large for 32-bit buffer resources");
  WgradArgs a;
  a.in = in; a.in_slope = nullptr; a.in_scale = nullptr; a.g = nullptr;
  a.Cin = Cin; a.H = H; a.W = W; a.O = O; a.pad = pad;
  a.Ho = H + 2 * pad - 2; a.Wo = W + 2 * pad - 2;
  a.TH = W1_TH; a.TW = W1_TW;
  a.tilesX = cdiv(a.Wo, W1_TW); a.tilesY = cdiv(a.Ho, W1_TH);
  a.oTiles = a.cTiles = a.kyGroups = 1;
  a.nSplit = std::min(512, a.tilesX * a.tilesY);
  a.dbg = 0;
  if (const char* e = getenv("FRCNN_WG_DBG")) a.dbg = atoi(e);
  const size_t need = (size_t)a.nSplit * (9 * O * Cin + O + 1) * 4 + 256;
  FR_CHECK(ws && ws_bytes >= need, "conv_wgrad_first_pooled: workspace too small (%zu < %zu)", ws_bytes, need);
  a.slab = (float*)(((uintptr_t)ws + 255) / 256 * 256);
  WgradFirstFused q{gpool, pidx, x, slope, gslope, gbias, (a.Ho - 2 + 1) / 2 + 1, (a.Wo - 2 + 1) / 2 + 1};
  const size_t lds = ((size_t)64 * W1_GP + W1_PATCH + W1_TW) * 4;   // + the row of ones
  static bool attr_set = false;
  if (!attr_set) {
    FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_first_kernel<3, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const double flops = 2.0 * O * Cin * 9 * (double)a.Ho * a.Wo;
  if (frcnn::prof_enabled(KC_CONV_WGRAD_K3)) frcnn::prof_before(KC_CONV_WGRAD_K3, s);
  hipLaunchKernelGGL((conv_wgrad_first_kernel<3, true>), dim3(a.nSplit), dim3(256), lds, s, a, q);
  wgrad_reduce_first(a.slab, a.nSplit, O * Cin, gw, s, O + 1, gbias, gslope);
  if (frcnn::prof_enabled(KC_CONV_WGRAD_K3))
    frcnn::prof_after(KC_CONV_WGRAD_K3, flops, 4.0 * ((double)Cin * H * W + (double)O * a.Ho * a.Wo * 1.3125), s);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

}  // namespace frcnn
