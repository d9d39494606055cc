// convx.hip -- the 3x3 (and the anchor nets' 5x5 / 7x7) nn.SpatialConvolution forward / updateGradInput
// (models/model_utilities.lua:8,31 driven by objective.lua:71,189 and Detector.lua:33) on the gfx950 16-bit matrix cores
// WITHOUT giving up fp32 results: "split operands".  Two forms, one kernel body (template parameter NP = planes per operand):
//   NP = 2 (option x3_f16, the default since round 5): two fp16 planes of the operand scaled by a power of two, THREE exact
//          partial products per fp32 product -- described at split8h below; the scale comes from the tensor's magnitude record
//          (amax.h), kept by the launch that wrote the tensor; a ring of four 8 KB A stages, one barrier per two taps.
//   NP = 3: three bf16 planes, SIX partial products, the exact split described next; a ring of two 12 KB stages.
// Every fp32 value x is written as the exact sum of three bf16 numbers
//     x = h + m + l + eps,   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m),   |eps| <= 2^-26 |x|
// (round to nearest; x - h and x - h - m are exact in fp32), and a product x*y is formed from the six partial products
//     l*h' + h*l' + m*m' + m*h' + h*m' + h*h'                      (the dropped m*l', l*m', l*l' are <= 2^-25 |x y|)
// each of which is EXACT in fp32 (8 x 8 significand bits) and is accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
// The result carries the 24 significand bits of an fp32 product; against an fp64 reference its error is the same as that
// of the fp32 matrix-core kernel in conv.hip (tests/test_gpu_convx.py measures both), but six bf16 instructions of 32
// cycles replace eight fp32 instructions (v_mfma_f32_32x32x2_f32) of 64 cycles for the same 32 x 32 x 16 block of the
// product: 2.67x less matrix-pipe time.
//
// Tensors stay fp32 CHW in HBM exactly as in conv.hip.  Weights are split once per optimiser step by the pack kernel
// (three bf16 planes in MFMA fragment order, brought to LDS by DMA); the input patch is split while it is staged (global
// -> registers -> PReLU / dropout scale of the producing layer -> split -> LDS), once per block and 16-channel chunk, and
// re-used by the nine taps and the four waves.
//
// GEMM view: D[m = filter][n = pixel] = sum_{tap, c} W[m][c][tap] * X[c][pixel + tap];  block = 128 filters x (TH x TW <=
// 128) pixels, 2 x 2 waves of 64 x 64 (four 32x32 accumulators), K step = 16 channels of one tap (lane half h takes
// channels 8h..8h+7), stage = one tap of a 16-channel chunk: A ring of two 12 KB stages (LDS-DMA one stage ahead), one patch
// buffer per block.  44 KB of LDS (45 KB with NP = 2), <= 168 registers -> three blocks per CU.
#include <cstdlib>

#include "kernels.h"
#include "amax.h"
#include <type_traits>
#ifdef CX_TRACE
#include <cstdio>
#include <vector>
#endif

namespace frcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define CX_BM 128                       // filters per block
#define CX_CH 16                        // channels per chunk (one MFMA K step per tap)
#define CX_OCC 3                        // blocks per CU the kernel is laid out for
#define CX_PP 204                       // patch positions per (plane, half) in LDS (pitch) of the 3x3 kernels; patch plane <= CX_PP
// the 5x5 / 7x7 instantiations take the patch of an 8 x 16-pixel tile: 12 x 20 / 14 x 22 positions (two blocks per CU)
constexpr int cx_pp(int k) { return k == 3 ? CX_PP : k == 5 ? 240 : 308; }
#define CX_NA 2                         // A ring slots of the three-plane form
#ifndef CX_RING2
#define CX_RING2 4                       // ... of the two-plane form: one barrier per TWO taps (cx_ring)
#endif
constexpr int cx_ring(int np) { return np == 2 ? CX_RING2 : CX_NA; }
#define CX_NB 1                         // patch buffers
#define CX_NTMAX 128                    // pixels per block
#ifndef CX_PREFETCH_B
#define CX_PREFETCH_B 1                 // fp16 form: the next tap's patch fragments are read a product early (see the stage loop)
#endif
#define CX_ASTAGE (6 * CX_BM * 16)      // bytes of one A stage of a 128-filter block: [plane 3][half 2][128 filters][8 bf16]

static int g_splitbf16 = -1;   // -1: not decided yet (environment FRCNN_SPLIT_BF16, default on)
void set_split_bf16(int on) { g_splitbf16 = on ? 1 : 0; }
int get_split_bf16() {
  if (g_splitbf16 < 0) g_splitbf16 = getenv("FRCNN_SPLIT_BF16") ? (atoi(getenv("FRCNN_SPLIT_BF16")) != 0) : 1;
  return g_splitbf16;
}

static int g_x3_f16 = -1;   // option "x3_f16" (environment FRCNN_X3_F16): the split launches take the two-plane fp16 form
void set_x3_f16(int on) { g_x3_f16 = on ? 1 : 0; }
int get_x3_f16() {
  if (g_x3_f16 < 0) g_x3_f16 = getenv("FRCNN_X3_F16") ? (atoi(getenv("FRCNN_X3_F16")) != 0) : 1;   // default on since round 5
  return g_x3_f16;
}

bool conv_x3_eligible(int Cin, int M, int k) {
  return get_split_bf16() && (k == 3 || k == 5 || k == 7) && Cin % CX_CH == 0 && Cin >= CX_CH && M % (k == 3 ? 64 : 128) == 0;
}

static void x3_choose_tile(int Ho, int Wo, int k, int* TH, int* TW);
// filters per block: 128 (2 x 2 waves of 64 x 64), or 64 (1 x 4 waves of 64 filters x 32 pixels) when 128 does not divide M
// -- or when 128-filter blocks would need a K split to fill the chip while 64-filter blocks fill it whole (the 113 x 200
// layers of vgg_small: 354 -> 708 blocks): the smaller block multiplies 6 % slower, but the partial-sum slabs and the fold
// launch behind them (42 us each, on the dependent chain) go away -- 96 vs 114 us and 171 vs 182 us for the two layers.
// The weight pack depends on the choice, so the launch and the pack job both ask this function with the same shape.
int conv_x3_bm(int M, int Ho, int Wo, int k) {
  static const int force = getenv("FRCNN_X3_BM") ? atoi(getenv("FRCNN_X3_BM")) : 0;
  if (force == 64) return 64;
  if (M % 128 != 0) return 64;
  if (force == 128 || k != 3 || Ho <= 0) return 128;
  int TH, TW;
  x3_choose_tile(Ho, Wo, k, &TH, &TW);
  const long tiles = (long)cdiv(Ho, TH) * cdiv(Wo, TW);
  const long b128 = tiles * (M / 128), b64 = tiles * (M / 64);
  return (b128 < 512 && b64 >= 512 && b64 <= 256 * CX_OCC) ? 64 : 128;
}
size_t conv_x3_pack_bytes(int Kchan, int M, int k) { return (size_t)M * (Kchan / CX_CH) * k * k * 96; }   // 6 x 16 bytes per filter, chunk and tap

// ---- three-way bf16 split of 8 values -> three 16-byte plane entries
__device__ __forceinline__ unsigned cvt2(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);   // v_cvt_pk_bf16_f32: round to nearest even
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void split8(const float* v, uint4& H, uint4& Mi, uint4& L) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = v[2 * j], x1 = v[2 * j + 1];
    h[j] = cvt2(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h[j] << 16), r1 = x1 - __builtin_bit_cast(float, h[j] & 0xFFFF0000u);
    m[j] = cvt2(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m[j] << 16), s1 = r1 - __builtin_bit_cast(float, m[j] & 0xFFFF0000u);
    l[j] = cvt2(s0, s1);
  }
  H = make_uint4(h[0], h[1], h[2], h[3]);
  Mi = make_uint4(m[0], m[1], m[2], m[3]);
  L = make_uint4(l[0], l[1], l[2], l[3]);
}

// ---- the two-plane fp16 form (option x3_f16): x * 2^e = h + l + eps with h = f16(x 2^e), l = f16(x 2^e - h),
// |eps| <= 2^-23 |x 2^e| in the normal range; e puts the tensor's largest magnitude into [2^14, 2^15) (x16_exp), so that
// h never overflows and the residual of every element within 2^-18 of the maximum is a normal fp16 number.  A product is
// formed from THREE exact partial products h*h' + h*l' + l*h' (11 x 11 significand bits; the dropped l*l' is <= 2^-22).
__device__ __forceinline__ unsigned cvt2h(float a, float b) {
  f32x2 v = {a, b};
  f16x2 r = __builtin_convertvector(v, f16x2);   // round to nearest even
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void split8h(const float* v, uint4& H, uint4& L) {
  unsigned h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = v[2 * j], x1 = v[2 * j + 1];
    h[j] = cvt2h(x0, x1);
    const f16x2 hv = __builtin_bit_cast(f16x2, h[j]);
    l[j] = cvt2h(x0 - (float)hv[0], x1 - (float)hv[1]);
  }
  H = make_uint4(h[0], h[1], h[2], h[3]);
  L = make_uint4(l[0], l[1], l[2], l[3]);
}
// exponent e with amax * 2^e in [2^top, 2^(top+1)); 0 for an all-zero tensor
__device__ __forceinline__ int x16_exp(float amax, int top) {
  const int be = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xFFu);
  if (be == 0 || be == 255) return 0;   // all zero -- or an infinity somewhere: no scaling (the finite elements keep fp16's own range)
  const int e = top - (be - 127);
  return e < -100 ? -100 : e > 100 ? 100 : e;
}
__device__ __forceinline__ float x16_pow2(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }

// largest magnitude of a tensor in a pass of its own (operator-level calls, and the tensors whose producer keeps no record)
__device__ __forceinline__ float absmax_span(const float* __restrict__ x, long n, long first, long stride) {
  float m = 0.f;
  // scalar head up to the first 16-byte boundary, 16-byte body, scalar tail (the weight tensors start at any element)
  long head = (long)((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) >> 2;
  if (head > n) head = n;
  for (long i = first; i < head; i += stride) m = fmaxf(m, fabsf(x[i]));
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x + head);
  const long n4 = (n - head) / 4;
  // four 16-byte loads in flight per thread (one at a time left a 30 MB tensor latency-bound: 16 us for the classification net's input)
  long i = first;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
    const float ma = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
    const float mb = fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)));
    const float mc = fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w)));
    const float md = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)));
    m = fmaxf(m, fmaxf(fmaxf(ma, mb), fmaxf(mc, md)));
  }
  for (; i < n4; i += stride) {
    const float4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (long i = head + 4 * n4 + first; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
  return m;
}
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, float* __restrict__ rec) {
  amax_store_block(absmax_span(x, n, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x), rec);
}
// one launch for several segments of the flat parameter vector (the weight tensors of the split launches), AMAX_SPAN floats
// per block; segment j's record is jobs[j].out
#define AMAX_SPAN 8192
__global__ __launch_bounds__(256) void absmax_multi_kernel(const float* __restrict__ w, const AmaxJob* __restrict__ jobs, int njobs) {
  __shared__ float part[4];
  int jb = 0;
  while (jb + 1 < njobs && (int)blockIdx.x >= jobs[jb + 1].blk_begin) ++jb;
  const AmaxJob j = jobs[jb];
  const int blk = blockIdx.x - j.blk_begin;
  const long first = (long)blk * AMAX_SPAN;        // this block's span of the segment
  const long n = j.n - first < AMAX_SPAN ? j.n - first : AMAX_SPAN;
  float m = absmax_span(w + j.off + first, n, threadIdx.x, blockDim.x);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    j.out[1 + blk] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    if (blk == 0) j.out[0] = __builtin_bit_cast(float, (int)((j.n + AMAX_SPAN - 1) / AMAX_SPAN));
  }
}
int tensor_absmax_assign_blocks(AmaxJob* jobs, int njobs) {
  int b = 0;
  for (int i = 0; i < njobs; ++i) {
    jobs[i].blk_begin = b;
    const long nb = cdivl(jobs[i].n, AMAX_SPAN);
    if (nb > AMAX_MAX_BLOCKS) return -1;   // (a segment of more than 134 M values: its record would not hold the spans)
    b += (int)nb;
  }
  return b;
}
long tensor_absmax_record_floats(long n) { return 1 + cdivl(n, AMAX_SPAN); }
int tensor_absmax_multi(const float* w, const AmaxJob* jobs_dev, int njobs, int grid, hipStream_t s) {
  if (njobs <= 0 || grid <= 0) return FRCNN_OK;
  FR_LAUNCH(KC_ELEMWISE, 0, 0, s, absmax_multi_kernel, dim3(grid), dim3(256), 0, w, jobs_dev, njobs);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}
int tensor_absmax(const float* x, long n, float* rec, hipStream_t s) {
  const int grid = (int)std::max<long>(1, std::min<long>(cdivl(n, 1024), 1024));
  FR_LAUNCH(KC_ELEMWISE, 0, n * 4.0, s, absmax_kernel, dim3(grid), dim3(256), 0, x, n, rec);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------ weight packing
// dst, per (m tile, chunk, tap): one A stage [plane][half][128][8] -- exactly the LDS image.
//  mode 0 (fwd):   A[m][kc][tap] = W[m][kc][tap]            (W is [O][C][3][3]: M = O, K channels = C)
//  mode 1 (dgrad): A[m][kc][tap] = W[kc][m][8 - tap]        (M = C, K channels = O; the flipped filter)
// One block = one (M tile, chunk) pair: its 128 x 16 x k*k source floats are contiguous runs (mode 0: 16 k*k floats per
// filter row; mode 1: 128 k*k floats per K channel), read coalesced into LDS (ROWS filter rows at a time), split, and
// written as the k*k 12 KB stages.
// LDS floats of the pack kernels: 64 filter rows x (144 + 1) for k = 3 -- 37 KB, four blocks per CU: the ~540 (M tile, chunk)
// pairs of a vgg_small step are ONE round on the 1024 slots (with 128 rows / 74 KB / two per CU they were 1.05 rounds of 512:
// the launch took two block times, 50 us at the head of every step)
#define PX_FLOATS (64 * 145)
template <int KS, int ROWS, int NP = 3>
__device__ __forceinline__ void pack_x_job(const float* __restrict__ weights, const PackXJob& j, int blk, int nblk, float* tile) {
  constexpr int KK = KS * KS, PITCH = CX_CH * KK + 1;   // odd pitch: the stride-KK reads of a lane's 8 channels spread over the banks
  static_assert(ROWS * PITCH <= PX_FLOATS, "pack tile does not fit");
  const float* __restrict__ w = weights + j.w_off;
  const int M = j.mode == 0 ? j.O : j.C, KC = j.mode == 0 ? j.C : j.O;
  const int BM = j.bm, AST = 2 * NP * BM * 16;
  float wmul = 1.f;
  if (NP == 2) {   // the weight tensor's largest magnitude: from the record of the absmax launch; published as a scalar for the convolution
    const float wa = amax_load_block(j.amax);
    wmul = x16_pow2(x16_exp(wa, 14));
    if (blk == 0 && threadIdx.x == 0 && j.amax_w) *j.amax_w = wa;
  }
  const int nCh = KC / CX_CH, pairs = (M / BM) * nCh;
  const int tid = threadIdx.x;
  if (j.bias_dst && blk == 0)   // the gathered bias vector of a forward pack (PackXJob)
    for (int m = tid; m < j.O; m += 256) {
      const int o = j.oidx ? j.oidx[m] : m;
      j.bias_dst[m] = o >= 0 ? weights[j.bias_off + o] : 0.f;
    }
  const int groups = BM / ROWS;   // work unit = ROWS filter rows of one pair: equal units, one per block (conv_x3_pack_assign_blocks)
  for (int u = blk; u < pairs * groups; u += nblk) {
    const int pr = u / groups;
    const int mt = pr / nCh, chunk = pr - mt * nCh;
    char* base = reinterpret_cast<char*>(j.dst) + (size_t)pr * KK * AST;
    {
      const int r0 = (u - pr * groups) * ROWS;
      // tile[r][kc16 * KK + tap] (source tap order), r = filter row inside this group of ROWS rows
      if (j.oidx || j.cidx) {   // gathered filters / channels (PackXJob): one (row, channel) pair per thread and turn, its KK taps in a run
        const int Cs = j.Cs ? j.Cs : j.C;
        for (int e = tid; e < ROWS * CX_CH; e += 256) {
          const int r = e / CX_CH, kc = e - r * CX_CH;
          const int mrow = mt * BM + r0 + r, krow = chunk * CX_CH + kc;      // the pack's M row / K channel
          const int op = j.mode == 0 ? mrow : krow, cp = j.mode == 0 ? krow : mrow;   // ... as filter o' / channel c'
          const int o = j.oidx ? j.oidx[op] : op, c = j.cidx ? j.cidx[cp] : cp;
          float* dst = tile + r * PITCH + kc * KK;
          if (o >= 0 && c >= 0) {
            const float* src = w + ((size_t)o * Cs + c) * KK;
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) dst[tap] = src[tap];
          } else {
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) dst[tap] = 0.f;
          }
        }
      } else if (j.mode == 0) {
        for (int e = tid; e < ROWS * CX_CH * KK; e += 256) {
          const int r = e / (CX_CH * KK), q = e - r * (CX_CH * KK);
          tile[r * PITCH + q] = w[((size_t)(mt * BM + r0 + r) * j.C + chunk * CX_CH) * KK + q];
        }
      } else {
        for (int e = tid; e < CX_CH * ROWS * KK; e += 256) {
          const int kc = e / (ROWS * KK), q = e - kc * (ROWS * KK);   // q = r * KK + tap, contiguous in the source
          const int r = q / KK, tap = q - r * KK;
          tile[r * PITCH + kc * KK + tap] = w[((size_t)(chunk * CX_CH + kc) * j.C + mt * BM + r0) * KK + q];
        }
      }
      __syncthreads();
      for (int it = tid; it < KK * 2 * ROWS; it += 256) {
        const int r = it % ROWS, h = (it / ROWS) & 1, tap = it / (2 * ROWS);
        const int st = j.mode == 0 ? tap : KK - 1 - tap;   // source tap (mode 1: the flipped filter)
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = tile[r * PITCH + (8 * h + i) * KK + st];
        uint4 H, Mi, L;
        char* stage = base + (size_t)tap * AST;
        if (NP == 2) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] *= wmul;
          split8h(v, H, L);
          *reinterpret_cast<uint4*>(stage + ((0 * 2 + h) * BM + r0 + r) * 16) = H;
          *reinterpret_cast<uint4*>(stage + ((1 * 2 + h) * BM + r0 + r) * 16) = L;
        } else {
          split8(v, H, Mi, L);
          *reinterpret_cast<uint4*>(stage + ((0 * 2 + h) * BM + r0 + r) * 16) = H;
          *reinterpret_cast<uint4*>(stage + ((1 * 2 + h) * BM + r0 + r) * 16) = Mi;
          *reinterpret_cast<uint4*>(stage + ((2 * 2 + h) * BM + r0 + r) * 16) = L;
        }
      }
      __syncthreads();
    }
  }
}
__device__ __forceinline__ void pack_x3_job(const float* __restrict__ weights, const PackXJob& j, int blk, int nblk, float* tile) {
  if (j.k == 3 && j.amax) pack_x_job<3, 64, 2>(weights, j, blk, nblk, tile);
  else if (j.k == 3) pack_x_job<3, 64>(weights, j, blk, nblk, tile);
  else if (j.k == 5 && j.amax) pack_x_job<5, 16, 2>(weights, j, blk, nblk, tile);
  else if (j.k == 5) pack_x_job<5, 16>(weights, j, blk, nblk, tile);
  else if (j.amax) pack_x_job<7, 8, 2>(weights, j, blk, nblk, tile);
  else pack_x_job<7, 8>(weights, j, blk, nblk, tile);
}

__global__ __launch_bounds__(256) void pack_x3_multi_kernel(const float* __restrict__ weights, const PackXJob* __restrict__ jobs, int njobs) {
  __shared__ float tile[PX_FLOATS];
  int jb = 0;
  while (jb + 1 < njobs && (int)blockIdx.x >= jobs[jb + 1].blk_begin) ++jb;
  const PackXJob j = jobs[jb];
  pack_x3_job(weights, j, blockIdx.x - j.blk_begin, j.nblk, tile);
}

__global__ __launch_bounds__(256) void pack_x3_kernel(const float* __restrict__ weights, PackXJob j) {
  __shared__ float tile[PX_FLOATS];
  pack_x3_job(weights, j, blockIdx.x, gridDim.x, tile);
}

// Ho x Wo: the OUTPUT map of the launch the pack feeds (the input map of the layer for mode 1, the input-gradient pass)
PackXJob conv_x3_pack_job(long w_off, int O, int C, int k, int mode, void* dst, int Ho, int Wo) {
  PackXJob j;
  j.w_off = w_off; j.O = O; j.C = C; j.k = k; j.mode = mode; j.dst = dst;
  j.bm = conv_x3_bm(mode == 0 ? O : C, Ho, Wo, k);
  j.total = (long)O * C * k * k;
  j.blk_begin = 0; j.nblk = 1;
  j.amax = nullptr; j.amax_w = nullptr;
  return j;
}

int conv_x3_pack_assign_blocks(PackXJob* jobs, int njobs) {
  int b = 0;
  for (int i = 0; i < njobs; ++i) {
    jobs[i].blk_begin = b;
    const int M = jobs[i].mode == 0 ? jobs[i].O : jobs[i].C, KC = jobs[i].mode == 0 ? jobs[i].C : jobs[i].O;
    const int rows = jobs[i].k == 3 ? 64 : jobs[i].k == 5 ? 16 : 8;   // (pack_x3_job's ROWS)
    jobs[i].nblk = (M / jobs[i].bm) * (KC / CX_CH) * (jobs[i].bm / rows);   // one block per (M tile, chunk) pair and group of rows
    b += jobs[i].nblk;
  }
  return b;
}

int conv_x3_pack_multi(const float* weights, const PackXJob* jobs_dev, int njobs, int grid, hipStream_t s) {
  if (njobs <= 0) return FRCNN_OK;
  FR_LAUNCH(KC_ELEMWISE, 0, 0, s, pack_x3_multi_kernel, dim3(grid), dim3(256), 0, weights, jobs_dev, njobs);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// one pack by itself (op-level entry points and tests)
int conv_x3_pack(const float* w, int O, int C, int k, int mode, void* dst, hipStream_t s, int Ho, int Wo, const float* amax_rec_w, float* amax_w) {
  PackXJob j = conv_x3_pack_job(0, O, C, k, mode, dst, Ho, Wo);
  j.amax = amax_rec_w;   // non-null: the two-plane fp16 form, scaled by the weight tensor's largest magnitude (its record, amax.h)
  j.amax_w = amax_w;     // ... which the pack publishes as a scalar here
  int grid = conv_x3_pack_assign_blocks(&j, 1);
  FR_LAUNCH(KC_ELEMWISE, 0, 0, s, pack_x3_kernel, dim3(grid), dim3(256), 0, w, j);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------ the convolution
struct CxArgs {
  const float* in;
  const float* in_slope;  // device scalar or null
  const float* in_scale;  // device [Cin] or null
  const void* wp;         // packed stages [mTile][chunk][tap][CX_ASTAGE]
  const float* bias;      // [M] or null
  float* out;             // [M][Ho][Wo]
  int Cin, H, W, M, Ho, Wo, pad;
  int TH, TW, tilesX, tilesY, mTiles;
  int nChunks, splitK, chunksPerSplit;
  int out_mode;           // 0 store, 1 add, 3 split-K slab
#ifdef CX_TRACE
  unsigned long long* trace;   // [block][64] timestamps (s_memrealtime, 100 MHz) -- tools/x3_trace.py
#endif
  const float* amax_in;   // NP = 2 only: the input tensor's magnitude record (amax.h) and the weight tensor's largest magnitude (a scalar, from the pack)
  const float* amax_w;
  float* amax_out;        // non-null (storing launches): the record of what is stored (amax.h)
  int wide;               // 1: the epilogue goes through LDS and stores 16 bytes per lane (Wo % 4 == 0, TW % 4 == 0, aligned tensors)
  X3PostAct post;         // EPI = 1 only: the activation backward applied to the stored tile (kernels.h)
};

// WM = waves along the filter dimension: 2 -> block = 128 filters, 2 x 2 waves of 64 x 64 (four 32x32 accumulators each);
// 1 -> block = 64 filters, 1 x 4 waves of 64 filters x 32 pixels (two accumulators each) for layers whose filter count is
// not a multiple of 128 (the 64-channel input gradients).
// EPI = 1 (input-gradient launches that store, out_mode 0): the stored tile goes through the backward of the activation
// that follows in the chain (X3PostAct) -- one read of x per element instead of act_backward's read + read + write pass.
template <int KS, int WM, bool SLOPE, bool SCALE, int EPI = 0, int NP = 3>
__global__ __launch_bounds__(256, KS == 3 ? CX_OCC : 2) void conv_x3_kernel(CxArgs p) {
  constexpr int KK = KS * KS, PP = cx_pp(KS), NIT = (2 * PP + 255) / 256;   // staging items per thread
  constexpr int BMK = 64 * WM, NTW = WM, AST = 2 * NP * BMK * 16, NDMA = AST / 1024;   // filters per block, pixel tiles per wave, stage bytes
  using frag_t = typename std::conditional<NP == 2, f16x8, bf16x8>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CX_TRACE
  auto stamp = [&](int slot) {
    if (p.trace && tid == 0) {
      p.trace[(size_t)blockIdx.x * 64 + slot] = __builtin_amdgcn_s_memrealtime();
      p.trace[(size_t)blockIdx.x * 64 + 32 + slot] = __builtin_amdgcn_s_memtime();   // shader clock
    }
  };
  stamp(0);
  if (p.trace && tid == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p.trace[(size_t)blockIdx.x * 64 + 7] = ((unsigned long long)xcc << 32) | hwid;
  }
#else
  auto stamp = [&](int) {};
#endif
  const int wm = WM == 2 ? wave >> 1 : 0, wn = WM == 2 ? wave & 1 : wave;
  const int h = lane >> 5, li = lane & 31;

  // XCD-aware order (see conv.hip): consecutive virtual indices of one XCD are the M tiles of one pixel tile
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
  const int m0 = mt_id * BMK;
  const int PW = p.TW + KS - 1, plane = (p.TH + KS - 1) * PW;
  const int NT = p.TH * p.TW;
  const int HW = p.H * p.W;
  const size_t hw_bytes = (size_t)HW * 4;

  // ---- this thread's two staging items: (half g, patch position): 8 channels each
  unsigned gofs[NIT];    // byte offset inside the chunk's first channel plane (includes the 8 g channels)
  bool gok[NIT];
  unsigned sdst[NIT];    // LDS byte offset of the item's 16-byte entry inside plane 0 of a B buffer
  int gsel[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = tid + 256 * it;
    const int g = e >= plane ? 1 : 0;
    const int pos = e - g * plane;
    const int r = pos / PW, col = pos - r * PW;
    const int gy = ty0 - p.pad + r, gx = tx0 - p.pad + col;
    const bool valid = e < 2 * plane;
    gok[it] = valid && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    gofs[it] = (gok[it] ? (unsigned)(gy * p.W + gx) * 4u : 0u) + (unsigned)g * 8u * (unsigned)hw_bytes;
    sdst[it] = valid ? (unsigned)((g * PP + pos) * 16) : 0xFFFFFFFFu;
    gsel[it] = g;
  }
  const float slope = SLOPE ? *p.in_slope : 1.f;
  float in_mul = 1.f, out_mul = 1.f, out_mul2 = 1.f;   // (the inverse scale in two halves: 2^-(ei + ew) need not be an fp32 number)
  if (NP == 2) {   // (headroom of one binade for a dropout scale: its entries are <= 1 in both modes, see net.cpp)
    float ai = amax_load_block(p.amax_in);
    if (SLOPE) ai *= fmaxf(1.f, fabsf(slope));
    const int ei = x16_exp(ai, SCALE ? 13 : 14), ew = x16_exp(*p.amax_w, 14);
    in_mul = x16_pow2(ei);
    const int et = -(ei + ew), e1 = et / 2;
    out_mul = x16_pow2(e1); out_mul2 = x16_pow2(et - e1);
  }

  // ---- lane offsets of the MFMA operand reads
  const unsigned aoff = (unsigned)((h * BMK + wm * 64 + li) * 16);
  unsigned boff[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    int q = wn * (32 * NTW) + nt * 32 + li;
    q = q < NT ? q : NT - 1;
    const int ty = q / p.TW, tx = q - ty * p.TW;
    boff[nt] = (unsigned)((h * PP + ty * PW + tx) * 16);
  }

  f32x16 acc[2][NTW];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NTW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int cbeg = split * p.chunksPerSplit;
  const int cend = min(cbeg + p.chunksPerSplit, p.nChunks);

  char* const As = smem;
  constexpr int RING = cx_ring(NP);
  char* const Bs = smem + RING * AST;

  // A stage DMA: 12 KB (128-filter blocks) / 6 KB = wave instructions of 1 KB, dealt round-robin to the four waves.
  // (global_load_lds, not the buffer form: with BOTH the ring DMA and the patch loads below on buffer resources the
  // kernels that also fetch a dropout scale computed wrong results at full size -- each form alone is right, cause not
  // found; tools/x3_check.py is the test that caught it.)
  const char* const wsrc = reinterpret_cast<const char*>(p.wp) + ((size_t)mt_id * p.nChunks + cbeg) * KK * AST + lane * 16;
  auto dma_stage = [&](int stage, int buf) {
    const char* src = wsrc + (size_t)stage * AST;
    char* dst = As + buf * AST;
    if (NDMA % 4 == 0) {   // a wave's instructions cover consecutive kilobytes
#pragma unroll
      for (int i = 0; i < NDMA / 4; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (wave * (NDMA / 4) + i) * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + (wave * (NDMA / 4) + i) * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < (NDMA + 3) / 4; ++i)
        if (wave + 4 * i < NDMA)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (wave + 4 * i) * 1024),
                                           (__attribute__((address_space(3))) void*)(dst + (wave + 4 * i) * 1024), 16, 0, 0);
    }
  };

  float vb[NIT][8];
  int patch_chunk = 0;
  // patch loads: buffer form too (the channel plane is the scalar offset, the lane's position a 32-bit offset)
  const __amdgpu_buffer_rsrc_t in_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)((size_t)p.Cin * hw_bytes), 0x00020000);
  auto load_patch = [&](int chunk) {
    const unsigned so = (unsigned)chunk * CX_CH * (unsigned)hw_bytes;
    patch_chunk = chunk;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        vb[it][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, gofs[it], so + (unsigned)j * (unsigned)hw_bytes, 0));
  };
  auto store_patch = [&](char* Bb) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      float sc[8];
      if (SCALE) {   // (L1 / L2 hits; fetched here rather than held in registers since the patch was requested)
        const float4* sp = reinterpret_cast<const float4*>(p.in_scale + patch_chunk * CX_CH + 8 * gsel[it]);
        const float4 s0 = sp[0], s1 = sp[1];
        sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
      }
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = vb[it][j];
        // (scalar multiplies, by hand: left alone, LLVM's SLP pass packs the eight of them into v_pk_mul_f32 with op_sel
        // swizzles, and in the kernels that apply BOTH slope and scale that sequence produced wrong products for a quarter
        // wave at a time at full size -- timing dependent, never in the small shapes; tools/x3_check.py and
        // tests/test_gpu_convx.py::test_model_layer_shapes_full_size are the checks that caught it.  Cause not established
        // (the packed form beside in-flight VMEM returns is the suspect); packed fp32 is no faster beside MFMAs anyway.)
#ifdef CX_PLAIN_MUL   // (the round-5 fault's code shape, kept for tools/x3_isa_diff.sh: plain C multiplies the SLP pass may pack)
        if (SLOPE) t = t > 0.f ? t : slope * t;
        if (SCALE) t *= sc[j];
#else
        if (SLOPE) {
          float ts;
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ts) : "v"(slope), "v"(t));
          t = t > 0.f ? t : ts;
        }
        if (SCALE) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(sc[j]));
#endif
        x[j] = gok[it] ? t : 0.f;
      }
      uint4 H, Mi, L;
      if (NP == 2) {
#pragma unroll
#ifdef CX_PLAIN_MUL
        for (int j = 0; j < 8; ++j) x[j] *= in_mul;
#else
        for (int j = 0; j < 8; ++j) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[j]) : "v"(x[j]), "v"(in_mul));
#endif
        split8h(x, H, L);
        if (sdst[it] != 0xFFFFFFFFu) {
          char* d = Bb + sdst[it];
          *reinterpret_cast<uint4*>(d) = H;
          *reinterpret_cast<uint4*>(d + 2 * PP * 16) = L;
        }
      } else {
        split8(x, H, Mi, L);
        if (sdst[it] != 0xFFFFFFFFu) {
          char* d = Bb + sdst[it];
          *reinterpret_cast<uint4*>(d) = H;
          *reinterpret_cast<uint4*>(d + 2 * PP * 16) = Mi;
          *reinterpret_cast<uint4*>(d + 4 * PP * 16) = L;
        }
      }
    }
  };

  // ---- one stage = one tap of one chunk: 12 fragment reads (3 planes x (2 A + NTW B)) and 6 partial products of 2 x NTW
  // MFMAs each: (l,h) (h,l) (m,h) (m,m) (h,m) (h,h) -- the small ones first; plane index 0 = h, 1 = m, 2 = l.
  // Left to the compiler a stage reads 6 + 3 + 3 fragments with a full LDS round trip exposed in front of each group
  // (tools/x3_trace.sh: a block alone on its CU needs 1 690 cycles per stage for 768 cycles of MFMAs, and three resident
  // blocks only fill 71 % of the pipe).  So the order is fixed by hand (sched_barrier fences), software-pipelined inside the
  // wave: every group of reads is issued one or two products before its first use, the (h,h) product of a stage is carried
  // over the barrier into the next stage -- it needs registers only -- where it covers the first reads of the new stage and the
  // issue of the A-ring DMA.  Ten fragments are live at most (as in the compiler's own schedule).
  auto readA = [&](const char* Ab, int pl, frag_t (&f)[2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) f[mt] = *reinterpret_cast<const frag_t*>(Ab + aoff + (pl * 2 * BMK + mt * 32) * 16);
  };
  auto readB = [&](const char* Bb, int tapoff, int pl, frag_t (&f)[NTW]) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) f[nt] = *reinterpret_cast<const frag_t*>(Bb + boff[nt] + tapoff + pl * 2 * PP * 16);
  };
  auto mm = [&](const frag_t (&a)[2], const frag_t (&b)[NTW]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        if constexpr (NP == 2) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        else acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
      }
  };
#define CX_FENCE() __builtin_amdgcn_sched_barrier(0)
  // The barriers of the stage loop.  NOT __syncthreads(): its release fence makes the compiler wait for EVERY outstanding VMEM
  // operation in front of the barrier (s_waitcnt vmcnt(0): an LDS-DMA is an LDS write it must publish) -- but the ring stages
  // requested a few instructions before the patch-publishing barrier are meant to stay in flight across it, and so are the
  // patch loads at a leader tap's barrier: what has to have landed is waited for by hand (the s_waitcnt vmcnt(n) in front of
  // each leader's barrier).  Found in round 6 in the ISA: with __syncthreads() every chunk waited out a whole DMA round trip at
  // its first tap.  lgkmcnt(0): this wave's LDS reads of the slots about to be refilled have returned, its patch writes are done.
#ifndef CX_SYNC2
#define CX_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define CX_BARRIER() __syncthreads()
#endif

  const int nC = cend - cbeg;
  frag_t pAH[2], pBH[NTW];   // the (h,h) operands of the previous stage (zeros before the first: the product adds nothing)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) pAH[i][j] = 0;
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) pBH[i][j] = 0;

  // Three blocks per CU (44 KB of LDS, <= 168 registers): A ring of two stages -- stage s+1 is requested behind the barrier
  // of stage s, into the slot stage s-1 was read from -- and ONE patch buffer: at a chunk boundary every wave has passed
  // the barrier of the new chunk's first stage before the new patch (in registers since the old chunk's first tap) is split
  // and written, and a second barrier publishes it.
  load_patch(cbeg);
  dma_stage(0, 0);
  const int nStages = nC * KK;
  if constexpr (RING == 4) dma_stage(nStages > 1 ? 1 : 0, 1);
  store_patch(Bs);
  stamp(1);
  bool more = false;
  int stage = 0;
  // NP = 2: the patch fragments of tap t + 1 are read under the last product of tap t -- the patch is valid for the whole chunk, only
  // the A image waits for the stage's barrier -- so that behind a barrier a wave reads two A fragments, not four (FRCNN... see
  // EXPERIMENTS.md round 5: with three products a stage has half the MFMAs to hide its LDS round trips behind)
  frag_t nbH[NTW], nbL[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { nbH[i][j] = 0; nbL[i][j] = 0; }
  if constexpr (RING == 4) {
    // ---- two-plane form: ring of FOUR stages, taps in groups (0,1) (2,3) ... (KK-1): ONE barrier per group.  At a group's
    // barrier its stages have landed (requested at the previous group's barrier, a whole group of MFMAs ago) and every wave is done
    // with the stages before it, whose slots the next group's request takes.  A follower tap has no wait and no barrier: its A
    // fragments are read under the leader's products like the patch fragments.  (With three products a stage has 12 MFMAs per wave:
    // a barrier per stage cost as much as it protected.)
    for (int chunk = cbeg; chunk < cend; ++chunk) {
#ifdef CX_TRACE
      if (chunk - cbeg < 24) stamp(8 + chunk - cbeg);
#endif
#pragma unroll
      for (int tap = 0; tap < KK; ++tap, ++stage) {
        const int ky = tap / KS, kx = tap - ky * KS;
        const char* const Ab = As + (stage & 3) * AST;
        const int tapoff = (ky * PW + kx) * 16;
        frag_t aL[2], aH[2], bL[NTW], bH[NTW];
        constexpr int PL = 1;
        const bool leader = (tap & 1) == 0;   // (compile-time: the tap loop is unrolled)
        // the next group: taps (t+2, t+3), the single last tap, or the next chunk's (0, 1)
        const int g1 = stage + (tap + 2 <= KK - 1 ? 2 : (tap == KK - 1 ? 1 : 2));
        const bool two = tap == KK - 1 || tap + 3 <= KK - 1;
        if (leader) {
          if (tap == 2 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * NIT) : "memory");   // (the patch loads issued at tap 0 stay in flight)
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          CX_BARRIER();
        }
        if (tap == 0) {
          mm(pAH, pBH);
          CX_FENCE();
          if (chunk > cbeg) store_patch(Bs);
          dma_stage(min(g1, nStages - 1), g1 & 3);
          if (two) dma_stage(min(g1 + 1, nStages - 1), (g1 + 1) & 3);
          CX_BARRIER();
          more = chunk + 1 < cend;
          if (more) load_patch(chunk + 1);
          readA(Ab, PL, aL); readB(Bs, tapoff, 0, bH);
          readA(Ab, 0, aH); readB(Bs, tapoff, PL, bL);
        } else {
          readA(Ab, PL, aL);
#pragma unroll
          for (int i = 0; i < NTW; ++i) { bH[i] = nbH[i]; bL[i] = nbL[i]; }
          CX_FENCE();
          mm(pAH, pBH);
          CX_FENCE();
          readA(Ab, 0, aH);
          if (leader) {
            dma_stage(min(g1, nStages - 1), g1 & 3);
            if (two) dma_stage(min(g1 + 1, nStages - 1), (g1 + 1) & 3);
          }
        }
        CX_FENCE();
        mm(aL, bH);
        CX_FENCE();
        if (tap + 1 < KK) {
          const int nky = (tap + 1) / KS, nkx = (tap + 1) - nky * KS;
          readB(Bs, (nky * PW + nkx) * 16, 0, nbH); readB(Bs, (nky * PW + nkx) * 16, PL, nbL);
          CX_FENCE();
        }
        mm(aH, bL);
#pragma unroll
        for (int i = 0; i < 2; ++i) pAH[i] = aH[i];
#pragma unroll
        for (int i = 0; i < NTW; ++i) pBH[i] = bH[i];
      }
    }
  } else
  for (int chunk = cbeg; chunk < cend; ++chunk) {
    const int par = KK % 2 == 0 ? 0 : (chunk - cbeg) & 1;   // A ring slot of the chunk's first tap
#ifdef CX_TRACE
    if (chunk - cbeg < 24) stamp(8 + chunk - cbeg);
#endif
#pragma unroll
    for (int tap = 0; tap < KK; ++tap, ++stage) {
      const int ky = tap / KS, kx = tap - ky * KS;
      const char* const Ab = As + ((tap + par) & 1) * AST;
      const int tapoff = (ky * PW + kx) * 16;
      frag_t aL[2], aM[2], aH[2], bL[NTW], bM[NTW], bH[NTW];
      constexpr int PL = NP - 1;   // plane index of the smallest part
      // stage's A image (requested one stage ago) has landed in every wave's part; right after a chunk's first tap the
      // patch loads issued behind it may still be in flight
      if (tap == 1 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * NIT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      CX_BARRIER();
      // (no accumulator is touched inside a conditional: a branch around MFMAs makes the register allocator keep two copies
      // of the 64 accumulator registers.  The ring DMA is unconditional -- behind the last stage it re-requests that stage
      // into the idle slot -- so that a stage is one basic block and the LDS waits are counted ones.)
      const int nxt = stage + 1 < nStages ? stage + 1 : stage;
      if (tap == 0) {
        // a chunk's first tap: the carried product covers the split of the new patch, a second barrier publishes it
        mm(pAH, pBH);
        CX_FENCE();
        if (chunk > cbeg) store_patch(Bs);   // (the first chunk's patch was written before the loop)
        dma_stage(nxt, (tap + 1 + par) & 1);
        CX_BARRIER();
        more = chunk + 1 < cend;
        if (more) load_patch(chunk + 1);
        readA(Ab, PL, aL); readB(Bs, tapoff, 0, bH);
        readA(Ab, 0, aH); readB(Bs, tapoff, PL, bL);
      } else if constexpr (NP == 2 && CX_PREFETCH_B) {
        readA(Ab, PL, aL);
#pragma unroll
        for (int i = 0; i < NTW; ++i) { bH[i] = nbH[i]; bL[i] = nbL[i]; }
        CX_FENCE();
        mm(pAH, pBH);
        CX_FENCE();
        readA(Ab, 0, aH);
        dma_stage(nxt, (tap + 1 + par) & 1);
      } else {
        readA(Ab, PL, aL); readB(Bs, tapoff, 0, bH);
        CX_FENCE();
        mm(pAH, pBH);
        CX_FENCE();
        readA(Ab, 0, aH); readB(Bs, tapoff, PL, bL);
        dma_stage(nxt, (tap + 1 + par) & 1);
      }
      CX_FENCE();
      mm(aL, bH);
      CX_FENCE();
      if constexpr (NP == 2) {
        if constexpr (CX_PREFETCH_B) {
          if (tap + 1 < KK) {   // (compile-time: the tap loop is unrolled)
            const int nky = (tap + 1) / KS, nkx = (tap + 1) - nky * KS;
            readB(Bs, (nky * PW + nkx) * 16, 0, nbH); readB(Bs, (nky * PW + nkx) * 16, PL, nbL);
            CX_FENCE();
          }
        }
        mm(aH, bL);
      } else {
        readA(Ab, 1, aM);
        CX_FENCE();
        mm(aH, bL);
        CX_FENCE();
        readB(Bs, tapoff, 1, bM);
        CX_FENCE();
        mm(aM, bH);
        mm(aM, bM);
        mm(aH, bM);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) pAH[i] = aH[i];
#pragma unroll
      for (int i = 0; i < NTW; ++i) pBH[i] = bH[i];
    }
  }
  mm(pAH, pBH);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the idle slot's last request)
#undef CX_FENCE
#undef CX_BARRIER
  if (NP == 2) {   // undo the two tensors' scales
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < NTW; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = (acc[a][b][r] * out_mul) * out_mul2;
  }

  // ---- epilogue: D layout col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (filter).  Every filter row of the
  // block exists (M is a multiple of the block's filter count), so only the pixel is predicated.  The 32 bias values of the
  // lane's rows are loaded in one batch before the stores, and the accumulate mode reads the 16 old values of a tile in one
  // batch (pixel clamped, not branched around): with a test, a bias load and an old-value load PER ELEMENT the compiler
  // emitted load - wait - add - store 64 times in a row -- 64 serialized memory round trips at the end of every block.
  stamp(2);
  const long HoWo = (long)p.Ho * p.Wo;
  const bool add_bias = p.bias != nullptr && split == 0;
  const int mrow0 = m0 + wm * 64 + 4 * h;
  // ---- wide epilogue.  The accumulator layout gives a lane ONE pixel of 16 filter rows: 64 dword stores per lane, each
  // wave instruction two 128-byte pieces -- an issue-bound tail in a launch whose blocks all finish together.  Instead the
  // wave turns its tile over in LDS (the A ring and the patch are dead by now): per 32-pixel column group the 64 x 32 tile is
  // written as [filter][pixel] with ds_write_b32 (a wave instruction = two rows of 32 consecutive dwords: conflict-free) and
  // read back as 16-byte pieces of four consecutive pixels (lane = (row & 7, pixel group); with a 32-dword pitch the four
  // 16-lane groups of a ds_read_b128 touch 64 distinct banks), so a lane stores four pixels of one filter and a wave
  // instruction writes eight filter rows of 128 bytes: 16 dwordx4 stores per lane instead of 64 dword stores, and x / the
  // old values of the accumulate mode are read the same way.  Needs Wo % 4 == 0 and TW % 4 == 0 (a group of four pixels then
  // never straddles a tile row or the map's edge and is 16-byte aligned); the launcher sets `wide` when that holds.
  if (p.wide) {
    __syncthreads();   // every wave is done with the last stage's fragments
    float* const T = reinterpret_cast<float*>(smem) + wave * (64 * 32);
    const int rr = lane >> 3, pg = lane & 7;
    const float pa = EPI == 1 ? *p.post.slope : 0.f;
    float sa = 0.f, am = 0.f;
    float* const obase = p.out + (p.out_mode == 3 ? (size_t)split * p.M * HoWo : (size_t)0);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) T[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + li] = acc[mt][nt][r];
      const int q = wn * (32 * NTW) + nt * 32 + 4 * pg;
      const int ty = q / p.TW, tx = q - ty * p.TW;
      const int oy = ty0 + ty, ox = tx0 + tx;
      const bool ok = q < NT && oy < p.Ho && ox < p.Wo;
      const size_t colo = ok ? (size_t)oy * p.Wo + ox : (size_t)0;
      // two batches of four rows groups: 4 LDS reads + 4 global reads in flight, then 4 stores (16 + 16 registers)
      const size_t rowo = (size_t)(m0 + wm * 64 + rr) * HoWo + colo;
#pragma unroll
      for (int ib = 0; ib < 8; ib += 4) {
        float4 v[4], o[4];
        float bsc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = m0 + wm * 64 + (ib + i) * 8 + rr;
          const size_t off = rowo + (size_t)(ib + i) * 8 * HoWo;
          v[i] = *reinterpret_cast<const float4*>(T + ((ib + i) * 8 + rr) * 32 + 4 * pg);
          if (EPI == 1) {
            o[i] = *reinterpret_cast<const float4*>(p.post.x + off);
            bsc[i] = p.post.scale ? p.post.scale[f] : 1.f;
          } else {
            if (p.out_mode == 1) o[i] = *reinterpret_cast<const float4*>(obase + off);
            bsc[i] = add_bias ? p.bias[f] : 0.f;
          }
        }
        if (ok) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const size_t off = rowo + (size_t)(ib + i) * 8 * HoWo;
            float4 w = v[i];
            if (EPI == 1) {
              float g;
              g = w.x * bsc[i]; if (o[i].x > 0.f) w.x = g; else { sa += o[i].x * g; w.x = pa * g; }
              g = w.y * bsc[i]; if (o[i].y > 0.f) w.y = g; else { sa += o[i].y * g; w.y = pa * g; }
              g = w.z * bsc[i]; if (o[i].z > 0.f) w.z = g; else { sa += o[i].z * g; w.z = pa * g; }
              g = w.w * bsc[i]; if (o[i].w > 0.f) w.w = g; else { sa += o[i].w * g; w.w = pa * g; }
            } else {
              w.x += bsc[i]; w.y += bsc[i]; w.z += bsc[i]; w.w += bsc[i];
              if (p.out_mode == 1) { w.x += o[i].x; w.y += o[i].y; w.z += o[i].z; w.w += o[i].w; }
            }
            *reinterpret_cast<float4*>(obase + off) = w;
            am = fmaxf(fmaxf(am, fmaxf(fabsf(w.x), fabsf(w.y))), fmaxf(fabsf(w.z), fabsf(w.w)));
          }
        }
      }
    }
    if (p.amax_out) amax_store_block(am, p.amax_out);
    if (EPI == 1) {
#pragma unroll
      for (int o2 = 32; o2 > 0; o2 >>= 1) sa += __shfl_xor(sa, o2);
      if (lane == 0 && p.post.gslope) unsafeAtomicAdd(p.post.gslope, sa);
    }
#ifdef CX_TRACE
    stamp(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(4);
#endif
    return;
  }
  float bv[2][16];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[mt][r] = add_bias ? p.bias[mrow0 + mt * 32 + (r & 3) + 8 * (r >> 2)] : 0.f;
  float am = 0.f;
  auto store_tile = [&](auto mode_c) {
    constexpr int OM = decltype(mode_c)::value;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int q = wn * (32 * NTW) + nt * 32 + li;
      const int ty = q / p.TW, tx = q - ty * p.TW;
      const int oy = ty0 + ty, ox = tx0 + tx;
      const bool ok = q < NT && oy < p.Ho && ox < p.Wo;
      float* const col = p.out + (OM == 3 ? (size_t)split * p.M * HoWo : (size_t)0) + (ok ? (size_t)oy * p.Wo + ox : (size_t)0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float* const row = col + (size_t)(mrow0 + mt * 32) * HoWo;
        float old[16];
        if (OM == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) old[r] = row[(size_t)((r & 3) + 8 * (r >> 2)) * HoWo];
        }
        if (ok) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float val = acc[mt][nt][r] + bv[mt][r];
            const float st = OM == 1 ? old[r] + val : val;
            row[(size_t)((r & 3) + 8 * (r >> 2)) * HoWo] = st;
            am = fmaxf(am, fabsf(st));
          }
        }
      }
    }
  };
  if (EPI == 1) {
    const float pa = *p.post.slope;
    float sa = 0.f;   // this lane's share of the slope gradient: sum over x <= 0 of x * scale * g
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int q = wn * (32 * NTW) + nt * 32 + li;
      const int ty = q / p.TW, tx = q - ty * p.TW;
      const int oy = ty0 + ty, ox = tx0 + tx;
      const bool ok = q < NT && oy < p.Ho && ox < p.Wo;
      const size_t colo = ok ? (size_t)oy * p.Wo + ox : (size_t)0;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const size_t rowo = (size_t)(mrow0 + mt * 32) * HoWo + colo;
        float xv[16], sc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {   // one batch of loads (pixel clamped, not branched around)
          xv[r] = p.post.x[rowo + (size_t)((r & 3) + 8 * (r >> 2)) * HoWo];
          sc[r] = p.post.scale ? p.post.scale[mrow0 + mt * 32 + (r & 3) + 8 * (r >> 2)] : 1.f;
        }
        if (ok) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float g = acc[mt][nt][r] * sc[r];
            const bool pos = xv[r] > 0.f;
            if (!pos) sa += xv[r] * g;
            const float st = pos ? g : pa * g;
            p.out[rowo + (size_t)((r & 3) + 8 * (r >> 2)) * HoWo] = st;
            am = fmaxf(am, fabsf(st));
          }
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sa += __shfl_xor(sa, o);
    if (lane == 0 && p.post.gslope) unsafeAtomicAdd(p.post.gslope, sa);
    if (p.amax_out) amax_store_block(am, p.amax_out);
    return;
  }
  if (p.out_mode == 0) store_tile(std::integral_constant<int, 0>{});
  else if (p.out_mode == 1) store_tile(std::integral_constant<int, 1>{});
  else store_tile(std::integral_constant<int, 3>{});
  if (p.amax_out) amax_store_block(am, p.amax_out);
#ifdef CX_TRACE
  stamp(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp(4);
#endif
}

// out[m][p] (= | +=) bias[m] + sum_s slab[s][m][p]
// Four consecutive pixels of one filter per thread when the map size allows (16-byte loads, the slabs of up to eight splits
// in flight at once), in split order -- the sum is the same number whatever the vector width.
// post.x != null: the folded value goes through the activation backward of X3PostAct (see conv_x3_kernel, EPI = 1)
template <int VEC>
__global__ __launch_bounds__(256) void x3_splitk_reduce_kernel(const float* __restrict__ slab, int nSplit, int M, long hw,
                                                               const float* __restrict__ bias, float* __restrict__ out, int accumulate,
                                                               X3PostAct post, float* amax) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const long total = (long)M * hw, nvec = total / VEC;
  const float pa = post.x ? *post.slope : 1.f;
  float sa = 0.f, am = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const long t = i * VEC;
    const float b = bias ? bias[t / hw] : 0.f;   // (VEC divides hw: the vector stays inside one filter's map)
    vec_t v = b;
    int s = 0;
    for (; s + 8 <= nSplit; s += 8) {
      vec_t x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const vec_t*>(slab + (size_t)(s + j) * total + t);
#pragma unroll
      for (int j = 0; j < 8; ++j) v += x[j];
    }
    for (; s + 2 <= nSplit; s += 2) {
      const vec_t x0 = *reinterpret_cast<const vec_t*>(slab + (size_t)s * total + t);
      const vec_t x1 = *reinterpret_cast<const vec_t*>(slab + (size_t)(s + 1) * total + t);
      v += x0; v += x1;
    }
    if (s < nSplit) v += *reinterpret_cast<const vec_t*>(slab + (size_t)s * total + t);
    vec_t* o = reinterpret_cast<vec_t*>(out + t);
    if (post.x) {
      const vec_t xv = *reinterpret_cast<const vec_t*>(post.x + t);
      const float sc = post.scale ? post.scale[t / hw] : 1.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float g = v[j] * sc;
        const bool pos = xv[j] > 0.f;
        if (!pos) sa += xv[j] * g;
        v[j] = pos ? g : pa * g;
      }
    }
    if (accumulate) v += *o;
    *o = v;
#pragma unroll
    for (int j = 0; j < VEC; ++j) am = fmaxf(am, fabsf(v[j]));
  }
  if (amax) amax_store_block(am, amax);
  if (post.x && post.gslope) {
    __shared__ float part[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sa += __shfl_xor(sa, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sa;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(post.gslope, (part[0] + part[1]) + (part[2] + part[3]));
  }
}

static void* g_x3_ws[8] = {};
static size_t g_x3_ws_bytes[8] = {};
static int x3_workspace(size_t need, float** out, int slot) {
  if (need > g_x3_ws_bytes[slot]) {
    if (g_x3_ws[slot]) FR_HIP(hipFree(g_x3_ws[slot]));
    g_x3_ws[slot] = nullptr; g_x3_ws_bytes[slot] = 0;
    FR_HIP(hipMalloc(&g_x3_ws[slot], need));
    g_x3_ws_bytes[slot] = need;
  }
  *out = (float*)g_x3_ws[slot];
  return FRCNN_OK;
}

// output tile TH x TW <= 128 pixels with a patch plane <= CX_PP that wastes the least work
// FRCNN_X3_WIDE=0: the scalar epilogue everywhere (and tile widths of any size)
static int x3_wide_enabled() {
  static const int on = getenv("FRCNN_X3_WIDE") ? atoi(getenv("FRCNN_X3_WIDE")) : 1;
  return on;
}
static void x3_choose_tile(int Ho, int Wo, int k, int* TH, int* TW) {
  static int fth = 0, ftw = 0;   // FRCNN_X3_TILE=THxTW (experiments)
  static bool parsed = false;
  if (!parsed) {
    parsed = true;
    if (const char* e = getenv("FRCNN_X3_TILE")) sscanf(e, "%dx%d", &fth, &ftw);
  }
  if (fth > 0 && ftw > 0 && k == 3 && fth * ftw <= CX_NTMAX && (fth + k - 1) * (ftw + k - 1) <= cx_pp(k)) {
    *TH = std::min(fth, Ho); *TW = std::min(ftw, Wo);
    return;
  }
  // the wide epilogue stores groups of four pixels: tile widths that are multiples of 4 where the map's width is one
  const int mult = (x3_wide_enabled() && Wo % 4 == 0) ? 4 : 1;
  long best = -1;
  int bth = 1, btw = 1;
  for (int tw = 1; tw <= std::min(Wo, CX_NTMAX); ++tw) {
    if (tw < 8 && Wo >= 8) continue;
    if (tw % mult != 0) continue;
    int th = std::min(CX_NTMAX / tw, Ho);
    while (th > 1 && (th + k - 1) * (tw + k - 1) > cx_pp(k)) --th;
    if (th < 1 || (th + k - 1) * (tw + k - 1) > cx_pp(k)) continue;
    long tiles = (long)cdiv(Ho, th) * cdiv(Wo, tw);
    long cost = tiles * CX_NTMAX * 64 + tiles * (th + k - 1) * (tw + k - 1);  // MFMA slots + halo traffic
    if (best < 0 || cost < best || (cost == best && tw > btw)) { best = cost; bth = th; btw = tw; }
  }
  *TH = bth; *TW = btw;
}

template <int KS, int WM, bool SLOPE, bool SCALE, int EPI = 0, int NP = 3>
static int launch_x3(CxArgs& a, double flops, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_x3_kernel<KS, WM, SLOPE, SCALE, EPI, NP>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));   // (the kernel also has a few static words: amax.h)
    attr_set = true;
  }
  int grid = a.tilesX * a.tilesY * a.mTiles * a.splitK;
  if (grid > AMAX_MAX_BLOCKS) a.amax_out = nullptr;   // (more blocks than a record has entries: conv_x3 takes the magnitude in a pass of its own)
#ifdef CX_TRACE
  static unsigned long long* tbuf = nullptr;
  const char* tfile = getenv("FRCNN_X3_TRACE");
  if (tfile && !tbuf) FR_HIP(hipMalloc(&tbuf, 8192 * 512));
  a.trace = tfile && grid <= 8192 ? tbuf : nullptr;
  if (a.trace) FR_HIP(hipMemsetAsync(tbuf, 0, (size_t)grid * 512, s));
#endif
  double bytes = 4.0 * ((double)a.Cin * a.H * a.W + (double)a.M * a.Ho * a.Wo);
  // (the wide epilogue turns four 64 x 32 fp32 tiles over in LDS: 32 KB)
  static const size_t lds_min = getenv("FRCNN_X3_LDS_MIN") ? (size_t)atol(getenv("FRCNN_X3_LDS_MIN")) : 0;   // (experiments: fewer blocks per CU)
  const size_t lds = std::max(lds_min, std::max<size_t>((size_t)cx_ring(NP) * (2 * NP * 64 * WM * 16) + (size_t)CX_NB * (2 * NP * cx_pp(KS) * 16), a.wide ? 4 * 64 * 32 * 4 : 0));
  FR_LAUNCH(KC_CONV_X3, flops, bytes, s, (conv_x3_kernel<KS, WM, SLOPE, SCALE, EPI, NP>), dim3(grid), dim3(256), lds, a);
  FR_LAUNCH_CHECK();
#ifdef CX_TRACE
  if (a.trace) {   // the last launch's stamps, one line per block
    FR_HIP(hipStreamSynchronize(s));
    std::vector<unsigned long long> hst((size_t)grid * 64);
    FR_HIP(hipMemcpy(hst.data(), tbuf, hst.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(tfile, "w")) {
      fprintf(f, "# grid %d M %d Cin %d Ho %d Wo %d TH %d TW %d splitK %d wide %d WM %d EPI %d\n", grid, a.M, a.Cin, a.Ho, a.Wo, a.TH, a.TW, a.splitK, a.wide, WM, EPI);
      for (int b = 0; b < grid; ++b) {
        for (int j = 0; j < 64; ++j) fprintf(f, "%llu ", hst[(size_t)b * 64 + j]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
#endif
  return FRCNN_OK;
}

int conv_x3(const float* in, int Cin, int H, int W, const float* in_slope, const float* in_scale, const void* wp,
            const float* bias, int M, int k, int pad, float* out, int out_mode, double algo_flops, hipStream_t s, int ws_slot,
            const X3PostAct* post, const float* amax_in, const float* amax_w, float* amax_out) {
  FR_CHECK((k == 3 || k == 5 || k == 7) && Cin % CX_CH == 0 && M % 64 == 0,
           "conv_x3: %d channels -> %d filters, %dx%d is not a split-bf16 shape", Cin, M, k, k);
  FR_CHECK((double)Cin * H * W * 4.0 < 4294967295.0, "conv_x3: input tensor too large for 32-bit offsets");
  CxArgs a;
  a.in = in; a.in_slope = in_slope; a.in_scale = in_scale; a.wp = wp; a.bias = bias; a.out = out;
  a.amax_in = amax_in; a.amax_w = amax_w; a.amax_out = amax_out;
  const bool f16 = amax_in != nullptr;
  FR_CHECK(!f16 || amax_w, "conv_x3: the two-plane fp16 form needs the magnitudes of both tensors");
  a.Cin = Cin; a.H = H; a.W = W; a.M = M; a.pad = pad;
  a.Ho = H + 2 * pad - k + 1; a.Wo = W + 2 * pad - k + 1;
  FR_CHECK(a.Ho > 0 && a.Wo > 0, "conv_x3: empty output (%dx%d, k=%d, pad=%d)", H, W, k, pad);
  x3_choose_tile(a.Ho, a.Wo, k, &a.TH, &a.TW);
  a.tilesX = cdiv(a.Wo, a.TW); a.tilesY = cdiv(a.Ho, a.TH);
  const int bm = conv_x3_bm(M, a.Ho, a.Wo, k);
  a.mTiles = M / bm;
  a.nChunks = Cin / CX_CH;
  const long blocks = (long)a.tilesX * a.tilesY * a.mTiles;
  // split K until one round of blocks fills the CX_OCC x 256 resident slots, keeping >= 36 stages (4 chunks of a 3x3) per split
  const int min_chunks = cdiv(36, k * k);
  // (the 5x5 / 7x7 kernels hold two blocks per CU and have 25 / 49 stages per chunk: up to 24 splits of one chunk each)
  const long slots = 256 * (k == 3 ? CX_OCC : 2);
  int splitK = (int)std::min<long>(std::min<long>(std::max<long>(1, slots / blocks), k == 3 ? 16 : 24), std::max(1, a.nChunks / min_chunks));
  if (const char* e = getenv("FRCNN_X3_SPLITK")) splitK = std::max(1, std::min(a.nChunks, atoi(e)));
  a.chunksPerSplit = cdiv(a.nChunks, splitK);
  a.splitK = cdiv(a.nChunks, a.chunksPerSplit);
  a.out_mode = out_mode;
  a.post = post ? *post : X3PostAct{nullptr, nullptr, nullptr, nullptr};
  a.wide = 0;   // decided below, once the destination (tensor or slab) is known
  FR_CHECK(!post || (k == 3 && out_mode == OUT_STORE && !in_slope && !in_scale && !bias && post->x && post->slope),
           "conv_x3: the fused activation backward belongs to a storing 3x3 input-gradient launch");
  bool slab = false;
  if (a.splitK > 1) {
    // (the fold stays a launch of its own: folding inside the launch -- write-through slabs, a ticket per output tile, the last
    // arriver sums -- was built in round 5, bit-identical, and measured slower: b4c1 91 -> 112 us, the step 2.85 -> 2.985 ms;
    // EXPERIMENTS.md, commit 3b8a58d)
    float* ws = nullptr;
    FR_TRY(x3_workspace((size_t)a.splitK * M * a.Ho * a.Wo * 4, &ws, ws_slot & 7));
    a.out = ws; a.out_mode = 3; a.bias = nullptr; slab = true;
    a.amax_out = nullptr;   // (the fold sees the final values)
  }
  a.wide = x3_wide_enabled() && a.Wo % 4 == 0 && a.TW % 4 == 0 && ((uintptr_t)a.out & 15) == 0 &&
           (!post || ((uintptr_t)post->x & 15) == 0);
  if (algo_flops <= 0) algo_flops = 2.0 * M * Cin * k * k * (double)a.Ho * a.Wo;
  int rc;
  const int act = (in_slope ? 2 : 0) | (in_scale ? 1 : 0);
  if (f16 && k != 3) {
    FR_CHECK(act == 0 && bm == 128, "conv_x3: a %dx%d launch takes no fused input activation and 128-filter blocks", k, k);
    rc = k == 5 ? launch_x3<5, 2, false, false, 0, 2>(a, algo_flops, s) : launch_x3<7, 2, false, false, 0, 2>(a, algo_flops, s);
  } else if (f16) {
    if (post && a.splitK == 1)
      rc = bm == 128 ? launch_x3<3, 2, false, false, 1, 2>(a, algo_flops, s) : launch_x3<3, 1, false, false, 1, 2>(a, algo_flops, s);
    else if (bm == 128)
      rc = act == 3 ? launch_x3<3, 2, true, true, 0, 2>(a, algo_flops, s) : act == 2 ? launch_x3<3, 2, true, false, 0, 2>(a, algo_flops, s)
         : act == 1 ? launch_x3<3, 2, false, true, 0, 2>(a, algo_flops, s) : launch_x3<3, 2, false, false, 0, 2>(a, algo_flops, s);
    else
      rc = act == 3 ? launch_x3<3, 1, true, true, 0, 2>(a, algo_flops, s) : act == 2 ? launch_x3<3, 1, true, false, 0, 2>(a, algo_flops, s)
         : act == 1 ? launch_x3<3, 1, false, true, 0, 2>(a, algo_flops, s) : launch_x3<3, 1, false, false, 0, 2>(a, algo_flops, s);
  } else if (post && a.splitK == 1) {   // (with a K split the fold below applies it)
    rc = bm == 128 ? launch_x3<3, 2, false, false, 1>(a, algo_flops, s) : launch_x3<3, 1, false, false, 1>(a, algo_flops, s);
  } else if (k == 3 && bm == 128) {
    rc = act == 3 ? launch_x3<3, 2, true, true>(a, algo_flops, s) : act == 2 ? launch_x3<3, 2, true, false>(a, algo_flops, s)
       : act == 1 ? launch_x3<3, 2, false, true>(a, algo_flops, s) : launch_x3<3, 2, false, false>(a, algo_flops, s);
  } else if (k == 3) {
    rc = act == 3 ? launch_x3<3, 1, true, true>(a, algo_flops, s) : act == 2 ? launch_x3<3, 1, true, false>(a, algo_flops, s)
       : act == 1 ? launch_x3<3, 1, false, true>(a, algo_flops, s) : launch_x3<3, 1, false, false>(a, algo_flops, s);
  } else {   // anchor nets: their input is a pooled map (activation already applied by the pooling kernel)
    FR_CHECK(act == 0 && bm == 128, "conv_x3: a %dx%d launch takes no fused input activation and 128-filter blocks", k, k);
    rc = k == 5 ? launch_x3<5, 2, false, false>(a, algo_flops, s) : launch_x3<7, 2, false, false>(a, algo_flops, s);
  }
  FR_TRY(rc);
  if (amax_out && !slab && (long)a.tilesX * a.tilesY * a.mTiles > AMAX_MAX_BLOCKS)   // (see launch_x3: the launch kept no record)
    FR_TRY(tensor_absmax(out, (long)M * a.Ho * a.Wo, amax_out, s));
  if (slab) {
    long total = (long)M * a.Ho * a.Wo;
    int grid;
    const long hw = (long)a.Ho * a.Wo;
    const X3PostAct pa = post ? *post : X3PostAct{nullptr, nullptr, nullptr, nullptr};
    const bool al = ((uintptr_t)out & 15) == 0 && ((uintptr_t)a.out & 15) == 0 && (!post || ((uintptr_t)post->x & 15) == 0);
    const int vec = al && hw % 4 == 0 ? 4 : al && hw % 2 == 0 ? 2 : 1;
    grid = (int)std::min<long>(cdivl(total / vec, 256), 4096);
    if (vec == 4)
      FR_LAUNCH(KC_ELEMWISE, 0, total * 4.0 * (a.splitK + 1), s, x3_splitk_reduce_kernel<4>, dim3(grid), dim3(256), 0,
                (const float*)a.out, a.splitK, M, hw, bias, out, out_mode == OUT_ADD ? 1 : 0, pa, amax_out);
    else if (vec == 2)
      FR_LAUNCH(KC_ELEMWISE, 0, total * 4.0 * (a.splitK + 1), s, x3_splitk_reduce_kernel<2>, dim3(grid), dim3(256), 0,
                (const float*)a.out, a.splitK, M, hw, bias, out, out_mode == OUT_ADD ? 1 : 0, pa, amax_out);
    else
      FR_LAUNCH(KC_ELEMWISE, 0, total * 4.0 * (a.splitK + 1), s, x3_splitk_reduce_kernel<1>, dim3(grid), dim3(256), 0,
                (const float*)a.out, a.splitK, M, hw, bias, out, out_mode == OUT_ADD ? 1 : 0, pa, amax_out);
    FR_LAUNCH_CHECK();
  }
  return FRCNN_OK;
}

}  // namespace frcnn
