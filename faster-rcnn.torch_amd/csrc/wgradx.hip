// wgradx.hip -- accGradParameters of the 3x3 nn.SpatialConvolution (models/model_utilities.lua:8 driven by
// objective.lua:189) in the split-bf16 operand form of convx.hip: fp32 tensors in and out, every product formed from six
// exact bf16 x bf16 partial products accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
//
//   gw[o][c][ky][kx] += sum_pix g[o][pix] * act(in)[c][pix + (ky,kx) - pad]
//
// GEMM view per tap: M = o, N = c, K = pixels -- BOTH operands are activations, so both are split while they are staged
// (global -> registers -> PReLU / dropout scale -> three bf16 planes -> LDS).  The reduction index of an MFMA operand
// lives inside the lane's 8-element vector, and a tap is a shift along that index: with pixel-contiguous operand rows
// two of three taps would start 2 or 4 bytes off a 16-byte boundary.  The planes are therefore stored PIXEL-major,
// [plane][8-channel group][position][8 channels] (the layout convx.hip stages), where a tap is a whole number of 16-byte
// entries, and the fragments are fetched with the gfx950 transpose read ds_read_b64_tr_b16: per 16-lane group a
// [4 pixels][16 channels] block comes back with lane = channel holding its 4 pixels (tools/tr_probe.hip prints the lane
// map this kernel relies on: result lane i, element j  <-  supplier lane 4j + (i>>2), element i&3).
//
// Block = 64 o x 64 c x 9 taps (2 x 2 waves, nine 32x32 accumulators = 144 registers each), walking its share of 4 x 16
// pixel tiles (K = 64 per tile: 216 MFMAs per wave).  ONE block per CU (<= 256 blocks), four waves of ~300 registers:
//   * the 48 global loads of the NEXT tile are issued right before the products of the current one and stay in flight
//     under its 216 MFMAs (with two 254-register blocks per CU -- rounds 1 and 2 -- there was no room to hold them: each
//     tile paid its memory round trip in front of its products, covered only by the CU's other block);
//   * the products walk the PATCH rows: the fragments of patch row r (three kx shifts x three planes) serve every
//     (K step ks, tap row ky) with ks + ky = r, so each is read once instead of up to three times -- 132 transposed reads
//     per wave and tile instead of 240 (ds_read_b64_tr_b16 moves ~50-64 B/clk: at 240 reads the LDS pipe was as busy as
//     the matrix pipe);
//   * a wave of this kernel leaves 200 registers per lane of its SIMD free, i.e. room for a wave of the main stream's
//     convolution kernels beside it: launches alone take what they took (b2c2 167 vs 153 us), the training step is 1.5 %
//     faster.
// Tried and measured slower (all parity-green): the same with double-buffered LDS images and the next
// tile's conversion interleaved into the products (189 us: the scheduler serialises conversion and MFMAs in one wave),
// twelve waves = quadrants x tap rows or tap columns with three accumulators each and three waves per SIMD (205 / 174 us).
// Partial sums go to slabs [split][tap][o][c] and are folded by wgrad_reduce_kernel (conv.hip) exactly as in the fp32
// matrix-core kernel.
#include <cstdlib>

#include "kernels.h"
#include <type_traits>

namespace frcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WX_TH 4
#define WX_TW 16
#define WX_PW 18                 // patch width  (TW + 2)
#define WX_P 108                 // patch positions (6 x 18)
#define WX_GP 68                 // LDS pitch (positions) of a gradient channel group: 68*16 B = 272 dwords = 16 mod 64 banks
#define WX_XP 116                // ... of a patch channel group: 116*16 B = 464 dwords = 16 mod 64 banks
#define WX_GBYTES (3 * 8 * WX_GP * 16)
#define WX_XBYTES (3 * 8 * WX_XP * 16)
#define WX_LDS (WX_GBYTES + WX_XBYTES)
#ifndef WX_TG
#define WX_TG 3                  // taps whose MFMAs are interleaved
#endif

bool conv_wgradx_eligible(int Cin, int O, int k) {
  return get_split_bf16() && k == 3 && Cin % 64 == 0 && O % 64 == 0;
}

__device__ __forceinline__ unsigned wx_cvt2(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void wx_split8(const float* v, uint4& H, uint4& Mi, uint4& L) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = v[2 * j], x1 = v[2 * j + 1];
    h[j] = wx_cvt2(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h[j] << 16), r1 = x1 - __builtin_bit_cast(float, h[j] & 0xFFFF0000u);
    m[j] = wx_cvt2(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m[j] << 16), s1 = r1 - __builtin_bit_cast(float, m[j] & 0xFFFF0000u);
    l[j] = wx_cvt2(s0, s1);
  }
  H = make_uint4(h[0], h[1], h[2], h[3]);
  Mi = make_uint4(m[0], m[1], m[2], m[3]);
  L = make_uint4(l[0], l[1], l[2], l[3]);
}

struct WgradXArgs {
  const float* in;
  const float* in_slope;
  const float* in_scale;
  const float* g;
  float* slab;   // [nSplit][9][O][Cin]
  int Cin, H, W, O, Ho, Wo, pad;
  int tilesX, tilesY, oTiles, cTiles, nSplit;
};

__device__ __forceinline__ bf16x8 wx_frag(const char* p0, const char* p1) {
  typedef bf16x4 __attribute__((address_space(3))) * lds4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)p0);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)p1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <bool SLOPE, bool SCALE>
__global__ __launch_bounds__(256, 1) void conv_wgradx_kernel(WgradXArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Gs = smem;                // [plane][og][WX_GP][8]
  char* const Xs = smem + WX_GBYTES;    // [plane][cg][WX_XP][8]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave >> 1, wc = wave & 1;
  const int h = lane >> 5, li = lane & 31;

  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (own L2 each); the virtual index gives every XCD a
  // contiguous range, in which the (o tile, c tile) blocks of one pixel split follow each other -- they walk the same
  // pixel tiles at the same time and now share them through ONE L2 instead of fetching them once per XCD
  int bid;
  {
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    bid = xcd * q + min(xcd, r) + idx;
  }
  const int ot = bid % p.oTiles; bid /= p.oTiles;
  const int ct = bid % p.cTiles;
  const int split = bid / p.cTiles;
  const int o0 = ot * 64, c0 = ct * 64;
  const int HW = p.H * p.W, HoWo = p.Ho * p.Wo;
  const size_t g_bytes = (size_t)HoWo * 4, in_bytes = (size_t)HW * 4;
  const float slope = SLOPE ? *p.in_slope : 1.f;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- staging geometry: wave w stages gradient groups w, w+4 (pixel = lane) and patch groups 2w, 2w+1 (positions lane,
  // 64 + lane) -- the channel group of every item is wave-uniform, so its 8 loads are scalar base + lane offset
  const int g_ty = lane >> 4, g_tx = lane & 15;
  int p_r[2], p_c[2];
  bool p_in[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int e = lane + 64 * m;
    p_r[m] = e / WX_PW; p_c[m] = e - p_r[m] * WX_PW;
    p_in[m] = e < WX_P;
  }

  // ---- fragment addressing (transpose reads): this lane SUPPLIES pixel row js, channels 4*qs.. of its 16-lane group
  const int js = (lane & 15) >> 2, qs = lane & 3, g16 = (lane >> 4) & 1;
  const int ol = wo * 32 + 16 * g16 + 4 * qs, cl = wc * 32 + 16 * g16 + 4 * qs;
  const char* const laneA = Gs + ((ol >> 3) * WX_GP + 8 * h + js) * 16 + (ol & 7) * 2;
  const char* const laneB = Xs + ((cl >> 3) * WX_XP + 8 * h + js) * 16 + (cl & 7) * 2;

  const int nPix = p.tilesX * p.tilesY;
  // ---- every global load of a tile is issued in one go (one memory round trip, 48 values in flight per thread: gradient
  // tile 2 items per thread, input patch 4 = 2 channel groups x 2 position slots) -- for the NEXT tile, right before the
  // products of the current one, so that the round trip runs under 216 MFMAs instead of in front of them; the splits and
  // LDS writes follow when the products are done.
  float vg[2][8], vp[2][2][8];
  bool pok[2], gok = false;
  auto load_tile = [&](int t) {
    const int oy0 = (t / p.tilesX) * WX_TH, ox0 = (t % p.tilesX) * WX_TW;
    const int goy = oy0 + g_ty, gox = ox0 + g_tx;
    gok = goy < p.Ho && gox < p.Wo;
    const unsigned gofs = gok ? (unsigned)(goy * p.Wo + gox) * 4u : 0u;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const char* gb = reinterpret_cast<const char*>(p.g) + (size_t)(o0 + 8 * (wave + 4 * it)) * g_bytes;
#pragma unroll
      for (int j = 0; j < 8; ++j) vg[it][j] = *reinterpret_cast<const float*>(gb + j * g_bytes + gofs);
    }
    unsigned pofs[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int iy = oy0 - p.pad + p_r[m], ix = ox0 - p.pad + p_c[m];
      pok[m] = p_in[m] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      pofs[m] = pok[m] ? (unsigned)(iy * p.W + ix) * 4u : 0u;
    }
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const char* ib = reinterpret_cast<const char*>(p.in) + (size_t)(c0 + 8 * (2 * wave + gi)) * in_bytes;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) vp[gi][m][j] = *reinterpret_cast<const float*>(ib + j * in_bytes + pofs[m]);
    }
  };
  auto stage_tile = [&]() {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = gok ? vg[it][j] : 0.f;
      uint4 Hh, Mi, L;
      wx_split8(x, Hh, Mi, L);
      char* d = Gs + ((wave + 4 * it) * WX_GP + lane) * 16;
      *reinterpret_cast<uint4*>(d) = Hh;
      *reinterpret_cast<uint4*>(d + 8 * WX_GP * 16) = Mi;
      *reinterpret_cast<uint4*>(d + 16 * WX_GP * 16) = L;
    }
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int cg = 2 * wave + gi;
      float sc[8];
      if (SCALE) {
        const float4* sp = reinterpret_cast<const float4*>(p.in_scale + c0 + 8 * cg);
        const float4 s0 = sp[0], s1 = sp[1];
        sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = vp[gi][m][j];
          if (SLOPE) v = v > 0.f ? v : slope * v;
          if (SCALE) v *= sc[j];
          x[j] = pok[m] ? v : 0.f;
        }
        uint4 Hh, Mi, L;
        wx_split8(x, Hh, Mi, L);
        if (p_in[m]) {
          char* d = Xs + (cg * WX_XP + lane + 64 * m) * 16;
          *reinterpret_cast<uint4*>(d) = Hh;
          *reinterpret_cast<uint4*>(d + 8 * WX_XP * 16) = Mi;
          *reinterpret_cast<uint4*>(d + 16 * WX_XP * 16) = L;
        }
      }
    }
  };
  if (split < nPix) load_tile(split);
  for (int t = split; t < nPix; t += p.nSplit) {
    stage_tile();
    __syncthreads();
    if (t + p.nSplit < nPix) load_tile(t + p.nSplit);
    // ---- K = the tile's 64 pixels: K-step ks = tile row ks (16 pixels), lane half h takes pixels 8h..8h+7 of it.  The loop
    // runs over PATCH rows: the fragments of patch row r (three kx shifts x three planes) serve every (ks, ky) with
    // ks + ky = r, so each is read from LDS once instead of up to three times -- 132 transposed reads per tile instead of
    // 240.  (ds_read_b64_tr_b16 moves 64 B/clk: at 240 reads the LDS pipe was busier than the matrix pipe.)  The gradient
    // fragments of the up to three K steps that meet a row stay in registers (a sliding window).
    bf16x8 a[WX_TH][3];
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // smallest partial products first; plane 0 = h, 1 = m, 2 = l
#pragma unroll
    for (int r = 0; r < WX_TH + 2; ++r) {
      if (r < WX_TH) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          a[r][pl] = wx_frag(laneA + (pl * 8 * WX_GP + 16 * r) * 16, laneA + (pl * 8 * WX_GP + 16 * r + 4) * 16);
      }
      bf16x8 b[3][3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          b[kx][pl] = wx_frag(laneB + (pl * 8 * WX_XP + r * WX_PW + kx) * 16, laneB + (pl * 8 * WX_XP + r * WX_PW + kx + 4) * 16);
      // the (ks, ky) pairs of this row, their 3 kx taps and 6 partial products interleaved: consecutive MFMAs write
      // different accumulators (a chain on ONE accumulator issues at its dependent latency, not at the pipe rate)
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int ks = r - ky;
          if (ks >= 0 && ks < WX_TH) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
              acc[3 * ky + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][PA[q]], b[kx][PB[q]], acc[3 * ky + kx], 0, 0, 0);
          }
        }
    }
    __syncthreads();
  }
  // ---- epilogue: D col = lane&31 -> c (contiguous in the slab), row -> o
  {
    const int c = c0 + wc * 32 + li;
    const size_t OC = (size_t)p.O * p.Cin;
    float* sl = p.slab + (size_t)split * 9 * OC;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + wo * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        sl[(size_t)tap * OC + (size_t)o * p.Cin + c] = acc[tap][r];
      }
  }
}

static void wgradx_plan(WgradXArgs& a) {
  a.tilesX = cdiv(a.Wo, WX_TW); a.tilesY = cdiv(a.Ho, WX_TH);
  a.oTiles = a.O / 64; a.cTiles = a.Cin / 64;
  const long base = (long)a.oTiles * a.cTiles, npix = (long)a.tilesX * a.tilesY;
  a.nSplit = (int)std::max<long>(1, std::min<long>(npix, 256 / base));   // one block per CU, one round: <= 256 blocks
  if (const char* e = getenv("FRCNN_WGX_NSPLIT")) a.nSplit = (int)std::max<long>(1, std::min<long>(npix, atoi(e)));
}

size_t conv_wgradx_workspace_bytes(int Cin, int H, int W, int O, int pad) {
  WgradXArgs a;
  a.Cin = Cin; a.H = H; a.W = W; a.O = O; a.pad = pad; a.Ho = H + 2 * pad - 2; a.Wo = W + 2 * pad - 2;
  wgradx_plan(a);
  return (size_t)a.nSplit * 9 * O * Cin * 4 + 256;
}

template <bool SLOPE, bool SCALE>
static int launch_wgradx(WgradXArgs& a, double flops, float* gw, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgradx_kernel<SLOPE, SCALE>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int grid = a.oTiles * a.cTiles * a.nSplit;
  const double bytes = 4.0 * ((double)a.Cin * a.H * a.W + (double)a.O * a.Ho * a.Wo);
  if (prof_enabled(KC_CONV_WGRADX)) prof_before(KC_CONV_WGRADX, s);
  const size_t lds = std::max<size_t>(WX_LDS, 84 * 1024);   // (> 80 KB: one block per CU, whatever the register count)
  hipLaunchKernelGGL((conv_wgradx_kernel<SLOPE, SCALE>), dim3(grid), dim3(256), lds, s, a);
  FR_TRY(wgrad_reduce(a.slab, a.nSplit, 9, a.O * a.Cin, gw, s));
  if (prof_enabled(KC_CONV_WGRADX)) prof_after(KC_CONV_WGRADX, flops, bytes, s);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

int conv_wgradx(const float* in, int Cin, int H, int W, const float* in_slope, const float* in_scale, const float* g, int O,
                int pad, float* gw, void* ws, size_t ws_bytes, hipStream_t s) {
  WgradXArgs a;
  a.in = in; a.in_slope = in_slope; a.in_scale = in_scale; a.g = g;
  a.Cin = Cin; a.H = H; a.W = W; a.O = O; a.pad = pad; a.Ho = H + 2 * pad - 2; a.Wo = W + 2 * pad - 2;
  FR_CHECK(Cin % 64 == 0 && O % 64 == 0, "conv_wgradx: %d channels x %d filters is not a split-bf16 shape", Cin, O);
  FR_CHECK((long)Cin * H * W < (1L << 30) && (long)O * a.Ho * a.Wo < (1L << 30), "conv_wgradx: tensor too large for 32-bit offsets");
  wgradx_plan(a);
  const size_t need = (size_t)a.nSplit * 9 * O * Cin * 4 + 256;
  FR_CHECK(ws && ws_bytes >= need, "conv_wgradx: workspace too small (%zu < %zu)", ws_bytes, need);
  a.slab = (float*)(((uintptr_t)ws + 255) / 256 * 256);
  const double flops = 2.0 * O * Cin * 9 * (double)a.Ho * a.Wo;
  if (in_slope) return in_scale ? launch_wgradx<true, true>(a, flops, gw, s) : launch_wgradx<true, false>(a, flops, gw, s);
  return in_scale ? launch_wgradx<false, true>(a, flops, gw, s) : launch_wgradx<false, false>(a, flops, gw, s);
}

}  // namespace frcnn
