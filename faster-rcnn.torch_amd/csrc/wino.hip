// wino.hip -- 3x3 / pad 1 / stride 1 convolution (nn.SpatialConvolution :updateOutput and :updateGradInput of the backbone,
// models/model_utilities.lua:8) by the Winograd minimal-filtering form F(2x2, 3x3): every 2x2 output tile costs 16
// multiplications per (output channel, input channel) instead of 36 -- 2.25x fewer MFMA cycles for the same result.
//
//   Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A        g: 3x3 filter, d: 4x4 input tile, Y: 2x2 output tile
//
// The sum over channels at each of the 16 transform points is a GEMM, U_p[o][c] * V_p[c][tile]: 16 independent GEMMs that
// share one launch.  One block = 64 output channels x 64 tiles (8 x 8 tiles = 16 x 16 pixels) x all 16 points; 2 x 2 waves,
// each keeping its 32 x 32 tile of all 16 points in the accumulator file (256 registers, one wave per SIMD).  Per chunk of 8
// input channels: the input patch (18 x 18 per channel, zero fill outside the image by the buffer range check) and the
// filter chunk U[16][8][64] arrive by LDS-DMA into double buffers while the previous chunk computes; every thread
// transforms two (tile, channel) patches (the producing layer's PReLU / SpatialDropout scale applied on the way: the
// transform reads each value into a register anyway) into V[16][8][64]; then 64 MFMAs per wave.  The output transform runs
// on the accumulators in registers (all 16 points of one (channel, tile) sit in one lane) and writes 2 x 2 pixels.
//
// fp32 throughout; the transforms only add, subtract and halve, so the result differs from the direct form by rounding
// (tests: within the 1e-4 bar of SURVEY 8d against the fp64-accumulating oracle).
#include "kernels.h"

namespace frcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WinoArgs {
  const float* in;
  const float* in_slope;   // device scalar or null
  const float* in_scale;   // device [Cin] or null
  const float* U;          // [nChunks][16][8][Mpad]
  const float* bias;       // [M] or null
  float* out;              // [M][Ho][Wo]
  int Cin, H, W, M, Mpad, Ho, Wo;
  int tbX, tbY, mTiles, nChunks;
  int out_mode;            // 0 store, 1 add
};

#define WN_PP 384          // LDS stride of one patch channel (18 x 18 = 324 positions, 6 x 64 unconditional DMA lanes)
#define WN_UFL (16 * 8 * 64)

// ---- filter transform: U[chunk][p][kk][Mpad] = (G g G^T)[p] for g = W[o][c] (forward: m = o, k = c) or the flipped filter of
// the transposed convolution (input gradient: m = c, k = o, g[ky][kx] = W[o][c][2-ky][2-kx])
__global__ void wino_filter_kernel(const float* __restrict__ w, int O, int C, int transposed, int Mpad, int Kpad,
                                   float* __restrict__ U) {
  const int M = transposed ? C : O, K = transposed ? O : C;
  const long total = (long)Mpad * Kpad;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int m = (int)(t % Mpad), k = (int)(t / Mpad);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        float v = 0.f;
        if (m < M && k < K) v = transposed ? w[(((size_t)k * C + m) * 3 + (2 - a)) * 3 + (2 - b)] : w[(((size_t)m * C + k) * 3 + a) * 3 + b];
        g[a][b] = v;
      }
    // G g : rows [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2]
    float t4[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t4[0][b] = g[0][b];
      t4[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      t4[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      t4[3][b] = g[2][b];
    }
    float* dst = U + ((size_t)(k >> 3) * 16 * 8 + (k & 7)) * Mpad + m;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float u0 = t4[a][0], u1 = 0.5f * (t4[a][0] + t4[a][1] + t4[a][2]), u2 = 0.5f * (t4[a][0] - t4[a][1] + t4[a][2]), u3 = t4[a][2];
      dst[(size_t)(a * 4 + 0) * 8 * Mpad] = u0;
      dst[(size_t)(a * 4 + 1) * 8 * Mpad] = u1;
      dst[(size_t)(a * 4 + 2) * 8 * Mpad] = u2;
      dst[(size_t)(a * 4 + 3) * 8 * Mpad] = u3;
    }
  }
}

template <bool SLOPE, bool SCALE, int SKIP = 0>   // SKIP: ablation instantiations (1 no MFMA, 2 no input transform, 4 no DMA staging)
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(WinoArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Us = smem;                          // [2][16][8][64]
  float* Vs = smem + 2 * WN_UFL;             // [2][16][8][64]
  float* Ps = smem + 4 * WN_UFL;             // [2][8][WN_PP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, li = lane & 31;

  // XCD-aware order (as conv_igemm): one XCD's blocks are consecutive, the M tiles of one pixel block follow each other
  int v;
  {
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    v = xcd * q + min(xcd, r) + idx;
  }
  const int mt = v % p.mTiles;
  const int tb = v / p.mTiles;
  const int oy0 = (tb / p.tbX) * 16, ox0 = (tb % p.tbX) * 16;
  const int m0 = mt * 64;
  const int HW = p.H * p.W;

  // patch DMA offsets (bytes inside a channel plane; outside the image -> outside the buffer range -> zeros)
  unsigned dofs[6];
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int e = it * 64 + lane;
    const int r = e / 18, col = e - r * 18;
    const int gy = oy0 - 1 + r, gx = ox0 - 1 + col;
    const bool ok = e < 324 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    dofs[it] = ok ? (unsigned)(gy * p.W + gx) * 4u : 0x7FFFFFFFu;
  }
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.Cin * HW * 4, 0x00020000);
  const unsigned hw_bytes = (unsigned)HW * 4u;
  // filter DMA: one wave instruction = 4 rows of 64 floats; wave w moves rows [32w, 32w+32) of the chunk's 128
  const unsigned u_voff = ((unsigned)(lane >> 4) * p.Mpad + (lane & 15) * 4) * 4u;
  const size_t u_step = (size_t)4 * p.Mpad * 4;

  auto stage_u = [&](int chunk) {
    const char* srcU = reinterpret_cast<const char*>(p.U + ((size_t)chunk * 128 + wave * 32) * p.Mpad + m0);
    float* dstU = Us + (chunk & 1) * WN_UFL + wave * 32 * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcU + i * u_step + u_voff),
                                       (__attribute__((address_space(3))) void*)(dstU + i * 4 * 64), 16, 0, 0);
  };
  auto stage_p = [&](int chunk) {
    float* dstP = Ps + (chunk & 1) * 8 * WN_PP;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cc = wave * 2 + j;
#pragma unroll
      for (int it = 0; it < 6; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (__attribute__((address_space(3))) void*)(dstP + cc * WN_PP + it * 64), 4,
                                                 dofs[it], (unsigned)(chunk * 8 + cc) * hw_bytes, 0, 0);
    }
  };

  f32x16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  constexpr bool has_slope = SLOPE, has_scale = SCALE;
  const float slope = has_slope ? *p.in_slope : 1.f;
  // this thread's two (channel, tile) patches of the input transform: channels `wave` and `wave + 4`, tile = lane
  const int tty = lane >> 3, ttx = lane & 7;
  const int poff = (2 * tty) * 18 + 2 * ttx;

  // Input transform V = B^T d B of this thread's two patches (channels `wave` and `wave + 4` of a chunk, tile = lane) in
  // micro-steps that are placed between the MFMAs of the chunk before it: row a of a patch is read (+ activation) at step a,
  // output row a is formed and written at step 4 + a; patch 1 follows at steps 8..15.
  float x[4][4];
  float scn[2] = {1.f, 1.f};      // dropout scales of this thread's two channels of the chunk being transformed
  auto load_scales = [&](int chunk) {
    if (has_scale) {
      scn[0] = p.in_scale[min(chunk * 8 + wave, p.Cin - 1)];
      scn[1] = p.in_scale[min(chunk * 8 + wave + 4, p.Cin - 1)];
    }
  };
  auto tr_row_load = [&](int chunk, int j, int a) {
    const int ch = wave + 4 * j;
    const float* d = Ps + (chunk & 1) * 8 * WN_PP + ch * WN_PP + poff + a * 18;
    const float sc = scn[j];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float val = d[b];
      if (has_slope) val = val > 0.f ? val : slope * val;
      if (has_scale) val *= sc;
      x[a][b] = val;
    }
  };
  auto tr_row_store = [&](int chunk, int j, int a) {
    const int ch = wave + 4 * j;
    float t[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)        // row a of B^T d
      t[b] = a == 0 ? x[0][b] - x[2][b] : a == 1 ? x[1][b] + x[2][b] : a == 2 ? x[2][b] - x[1][b] : x[1][b] - x[3][b];
    float* Vd = Vs + (chunk & 1) * WN_UFL + ch * 64 + lane;
    Vd[(a * 4 + 0) * 8 * 64] = t[0] - t[2];      // (.) B
    Vd[(a * 4 + 1) * 8 * 64] = t[1] + t[2];
    Vd[(a * 4 + 2) * 8 * 64] = t[2] - t[1];
    Vd[(a * 4 + 3) * 8 * 64] = t[1] - t[3];
  };
  auto transform_all = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int a = 0; a < 4; ++a) tr_row_load(chunk, j, a);
#pragma unroll
      for (int a = 0; a < 4; ++a) tr_row_store(chunk, j, a);
    }
  };

  // prologue: chunk 0 arrives, is transformed; chunk 1's patch is on its way
  const int lastc = p.nChunks - 1;
  stage_p(0);
  stage_u(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  stage_p(min(1, lastc));
  load_scales(0);
  transform_all(0);

  for (int chunk = 0; chunk < p.nChunks; ++chunk) {
    const int cur = chunk & 1;
    load_scales(min(chunk + 1, lastc));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // patch(chunk+1) and U(chunk): this wave's share has landed
    __syncthreads();                                    // ... everybody's; V(chunk) is complete; MFMA(chunk-1) is done
    // (past the end the same chunk is staged / transformed again into buffers nobody reads any more: no branch in the loop body)
    if (!(SKIP & 4)) {
      const int c2 = min(chunk + 2, lastc), c1 = min(chunk + 1, lastc);
      float* dstP = Ps + ((chunk + 2) & 1) * 8 * WN_PP;   // the buffer whose patch was transformed during the previous chunk
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cc = wave * 2 + j;
#pragma unroll
        for (int it = 0; it < 6; ++it)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (__attribute__((address_space(3))) void*)(dstP + cc * WN_PP + it * 64), 4,
                                                   dofs[it], (unsigned)(c2 * 8 + cc) * hw_bytes, 0, 0);
      }
      const char* srcU = reinterpret_cast<const char*>(p.U + ((size_t)c1 * 128 + wave * 32) * p.Mpad + m0);
      float* dstU = Us + ((chunk + 1) & 1) * WN_UFL + wave * 32 * 64;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcU + i * u_step + u_voff),
                                         (__attribute__((address_space(3))) void*)(dstU + i * 4 * 64), 16, 0, 0);
    }
    // ---- 16 GEMMs: acc[q] += U_q[32 x 8] * V_q[8 x 32], with the input transform of the NEXT chunk placed between them (one
    // wave per SIMD: nothing else would keep the matrix pipe busy while this wave transforms).  Operands of point q+1 are read
    // before the MFMAs of point q are issued.
    const float* Ua = Us + cur * WN_UFL + h * 64 + wm * 32 + li;
    const float* Vb = Vs + cur * WN_UFL + h * 64 + wn * 32 + li;
    float fa[2][4], fb[2][4];
    auto frag = [&](int q, float* a, float* b) {
#pragma unroll
      for (int kp = 0; kp < 4; ++kp) {
        a[kp] = Ua[(q * 8 + 2 * kp) * 64];
        b[kp] = Vb[(q * 8 + 2 * kp) * 64];
      }
    };
    frag(0, fa[0], fb[0]);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (q + 1 < 16) frag(q + 1, fa[(q + 1) & 1], fb[(q + 1) & 1]);
      const int j = q >> 3, st = q & 7;
      if (!(SKIP & 2)) { if (st < 4) tr_row_load(chunk + 1, j, st); else tr_row_store(chunk + 1, j, st - 4); }
#pragma unroll
      for (int kp = 0; kp < 4; ++kp) { if (!(SKIP & 1)) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][kp], fb[q & 1][kp], acc[q], 0, 0, 0); else acc[q][kp] += fa[q & 1][kp] * fb[q & 1][kp]; }
      // issue order of this point: the LDS reads first (operands of the next point, a patch row), then the four MFMAs with the
      // transform's arithmetic and LDS writes in their shadow
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int kp = 0; kp < 4; ++kp) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
  }

  // ---- output transform on the accumulators: Y = A^T M A, 2 x 2 pixels per (channel, tile).  Lanes 0..7 of a group hold
  // eight horizontally adjacent tiles: with an even row pitch the two pixels of a tile row leave as one 8-byte store and a
  // lane group writes 64 contiguous bytes.
  const int t = wn * 32 + li;
  const int oy = oy0 + 2 * (t >> 3), ox = ox0 + 2 * (t & 7);
  const long HoWo = (long)p.Ho * p.Wo;
  const bool c0ok = ox < p.Wo, c1ok = ox + 1 < p.Wo, r0ok = oy < p.Ho, r1ok = oy + 1 < p.Ho;
  const bool pair = (p.Wo & 1) == 0 && ((uintptr_t)p.out & 7) == 0;   // (ox is even: the pair is 8-byte aligned)
  const int mrow = m0 + wm * 32 + 4 * h;
  float* const obase = p.out + (size_t)oy * p.Wo + ox;
  auto emit = [&](auto add_c, auto pair_c) {
    constexpr bool ADD = decltype(add_c)::value, PAIR = decltype(pair_c)::value;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mrow + (r & 3) + 8 * (r >> 2);
      float s[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {      // A^T M
        s[0][j] = acc[0 * 4 + j][r] + acc[1 * 4 + j][r] + acc[2 * 4 + j][r];
        s[1][j] = acc[1 * 4 + j][r] - acc[2 * 4 + j][r] - acc[3 * 4 + j][r];
      }
      const bool mok = m < p.M;
      const float bv = (p.bias && mok) ? p.bias[m] : 0.f;
      float* dst = obase + (size_t)(mok ? m : 0) * HoWo;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float y0 = s[i][0] + s[i][1] + s[i][2] + bv;
        float y1 = s[i][1] - s[i][2] - s[i][3] + bv;
        const bool rok = (i == 0 ? r0ok : r1ok) && mok;
        float* d = dst + i * p.Wo;
        if (PAIR) {
          if (rok && c0ok) {   // (an even row pitch: column ox + 1 exists whenever column ox does)
            float2* d2 = reinterpret_cast<float2*>(d);
            if (ADD) { const float2 o = *d2; y0 += o.x; y1 += o.y; }
            *d2 = make_float2(y0, y1);
          }
        } else {
          if (rok && c0ok) { if (ADD) y0 += d[0]; d[0] = y0; }
          if (rok && c1ok) { if (ADD) y1 += d[1]; d[1] = y1; }
        }
      }
    }
  };
  if (p.out_mode == 1) {
    if (pair) emit(std::true_type{}, std::true_type{}); else emit(std::true_type{}, std::false_type{});
  } else {
    if (pair) emit(std::false_type{}, std::true_type{}); else emit(std::false_type{}, std::false_type{});
  }
}

// table-driven variant: the filters of every Winograd launch of a model in ONE launch (jobs live in device memory)
__global__ void wino_filter_multi_kernel(const float* __restrict__ weights, const WinoFilterJob* __restrict__ jobs, int njobs) {
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].blk_begin) ++j;
  const WinoFilterJob jb = jobs[j];
  const float* w = weights + jb.w_off;
  const int O = jb.O, C = jb.C, transposed = jb.transposed, Mpad = jb.Mpad;
  const int M = transposed ? C : O, K = transposed ? O : C;
  const long total = (long)Mpad * jb.Kpad;
  for (long t = (long)(blockIdx.x - jb.blk_begin) * blockDim.x + threadIdx.x; t < total; t += (long)jb.nblk * blockDim.x) {
    const int m = (int)(t % Mpad), k = (int)(t / Mpad);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        float v = 0.f;
        if (m < M && k < K) v = transposed ? w[(((size_t)k * C + m) * 3 + (2 - a)) * 3 + (2 - b)] : w[(((size_t)m * C + k) * 3 + a) * 3 + b];
        g[a][b] = v;
      }
    float t4[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t4[0][b] = g[0][b];
      t4[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      t4[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      t4[3][b] = g[2][b];
    }
    float* dst = jb.dst + ((size_t)(k >> 3) * 16 * 8 + (k & 7)) * Mpad + m;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      dst[(size_t)(a * 4 + 0) * 8 * Mpad] = t4[a][0];
      dst[(size_t)(a * 4 + 1) * 8 * Mpad] = 0.5f * (t4[a][0] + t4[a][1] + t4[a][2]);
      dst[(size_t)(a * 4 + 2) * 8 * Mpad] = 0.5f * (t4[a][0] - t4[a][1] + t4[a][2]);
      dst[(size_t)(a * 4 + 3) * 8 * Mpad] = t4[a][2];
    }
  }
}
WinoFilterJob conv_wino_filter_job(long w_off, int O, int C, int transposed, float* dst) {
  WinoFilterJob j;
  const int M = transposed ? C : O, K = transposed ? O : C;
  j.w_off = w_off; j.dst = dst; j.O = O; j.C = C; j.transposed = transposed;
  j.Mpad = conv_mpad(M); j.Kpad = cdiv(K, 8) * 8; j.blk_begin = 0; j.nblk = 0;
  return j;
}
int conv_wino_filter_assign_blocks(WinoFilterJob* jobs, int njobs) {
  int b = 0;
  for (int i = 0; i < njobs; ++i) {
    jobs[i].blk_begin = b;
    jobs[i].nblk = (int)std::max<long>(1, std::min<long>(cdivl((long)jobs[i].Mpad * jobs[i].Kpad, 256), 512));
    b += jobs[i].nblk;
  }
  return b;
}
int conv_wino_filter_multi(const float* weights, const WinoFilterJob* jobs_dev, int njobs, int grid, hipStream_t s) {
  if (njobs <= 0) return FRCNN_OK;
  FR_LAUNCH(KC_ELEMWISE, 0, 0, s, wino_filter_multi_kernel, dim3(grid), dim3(256), 0, weights, jobs_dev, njobs);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

size_t conv_wino_filter_floats(int Kchan, int M) { return (size_t)cdiv(Kchan, 8) * 16 * 8 * conv_mpad(M); }

int conv_wino_filter(const float* w, int O, int C, int transposed, float* U, hipStream_t s) {
  const int M = transposed ? C : O, K = transposed ? O : C;
  const int Mpad = conv_mpad(M), Kpad = cdiv(K, 8) * 8;
  const long total = (long)Mpad * Kpad;
  int grid = (int)std::min<long>(cdivl(total, 256), 4096);
  FR_LAUNCH(KC_ELEMWISE, 0, total * 16 * 4.0 + (double)O * C * 36.0, s, wino_filter_kernel, dim3(grid), dim3(256), 0, w, O, C,
            transposed, Mpad, Kpad, U);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// option "winograd" (frcnn_set_option; environment FRCNN_WINO sets the default): off unless asked for -- see DESIGN.md for the
// measurements (single launches 6-25 % faster than the direct kernel, the training step +0.7 % on vgg_small, -1.7 % on vgg_large)
static int g_wino = getenv("FRCNN_WINO") ? atoi(getenv("FRCNN_WINO")) : 0;
void set_winograd(int on) { g_wino = on; }
int get_winograd() { return g_wino; }

bool conv_wino_eligible(int Cin, int H, int W, int M, int k, int pad) {
  const int on = g_wino;
  if (!on || k != 3 || pad != 1 || Cin < 8 || (double)Cin * H * W * 4.0 >= 2147483647.0) return false;
  const long blocks = (long)cdiv(H, 16) * cdiv(W, 16) * (conv_mpad(M) / 64);
  static const int minb = getenv("FRCNN_WINO_MINBLOCKS") ? atoi(getenv("FRCNN_WINO_MINBLOCKS")) : 200;
  return blocks >= minb;   // one block per CU: smaller problems stay with the direct kernel and its K split
}

// out[M][H][W] (=|+=) conv3x3_pad1(act(in)[Cin][H][W]) (+ bias) with U = conv_wino_filter(...)
int conv_wino(const float* in, int Cin, int H, int W, const float* in_slope, const float* in_scale, const float* U,
              const float* bias, int M, float* out, int out_mode, double algo_flops, hipStream_t s) {
  WinoArgs a;
  a.in = in; a.in_slope = in_slope; a.in_scale = in_scale; a.U = U; a.bias = bias; a.out = out;
  a.Cin = Cin; a.H = H; a.W = W; a.M = M; a.Mpad = conv_mpad(M); a.Ho = H; a.Wo = W;
  a.tbX = cdiv(W, 16); a.tbY = cdiv(H, 16); a.mTiles = a.Mpad / 64; a.nChunks = cdiv(Cin, 8);
  a.out_mode = out_mode == OUT_ADD ? 1 : 0;
  FR_CHECK(out_mode == OUT_STORE || out_mode == OUT_ADD, "conv_wino: unsupported output mode");
  const size_t lds = (size_t)(4 * WN_UFL + 2 * 8 * WN_PP) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  if (algo_flops <= 0) algo_flops = 2.0 * M * Cin * 9.0 * (double)H * W;
  const int grid = a.tbX * a.tbY * a.mTiles;
  const double bytes = 4.0 * ((double)Cin * H * W + (double)M * H * W);
  if (in_slope && in_scale) FR_LAUNCH(KC_CONV_IGEMM_K3, algo_flops, bytes, s, (conv_wino_kernel<true, true>), dim3(grid), dim3(256), lds, a);
  else if (in_slope) FR_LAUNCH(KC_CONV_IGEMM_K3, algo_flops, bytes, s, (conv_wino_kernel<true, false>), dim3(grid), dim3(256), lds, a);
  else if (in_scale) FR_LAUNCH(KC_CONV_IGEMM_K3, algo_flops, bytes, s, (conv_wino_kernel<false, true>), dim3(grid), dim3(256), lds, a);
  else FR_LAUNCH(KC_CONV_IGEMM_K3, algo_flops, bytes, s, (conv_wino_kernel<false, false>), dim3(grid), dim3(256), lds, a);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

}  // namespace frcnn
