// wgradx.hip -- accGradParameters of the 3x3 nn.SpatialConvolution (models/model_utilities.lua:8 driven by
// objective.lua:189) in the split operand forms of convx.hip: fp32 tensors in and out, every product formed from three exact
// fp16 x fp16 partial products of two-plane operands (template parameter NP = 2: option x3_f16, the default; both tensors scaled
// by powers of two from their magnitude records, amax.h) or from six exact bf16 x bf16 partial products (NP = 3), accumulated
// in fp32 by v_mfma_f32_32x32x16_f16 / _bf16.  The description below is written for three planes; with two, a tile has 108 MFMAs
// per wave instead of 216 and an LDS image 52 KB instead of 78.
//
//   gw[o][c][ky][kx] += sum_pix g[o][pix] * act(in)[c][pix + (ky,kx) - pad]          (+ optionally gbias[o] += sum_pix g[o][pix])
//
// GEMM view per tap: M = o, N = c, K = pixels -- BOTH operands are activations, so both are split while they are staged
// (global -> registers -> PReLU / dropout scale -> three bf16 planes -> LDS).  The reduction index of an MFMA operand lives
// inside the lane's 8-element vector and a tap is a shift along that index.  Round 4 layout: the planes are PIXEL-CONTIGUOUS --
// gradient [plane][o][4 rows x 16 px], patch [plane][c][6 rows][24 columns] -- read with plain 16-byte ds_read_b128, and the
// column taps are cut IN REGISTERS: the 16 elements a lane reads of a patch row hold its 8 pixels at all three taps (tap kx
// starts at element 3 + kx: one is the registers as they are, the other two are v_perm of neighbours).  (Rounds 2-3 kept the
// planes pixel-major and fetched fragments with the transposing ds_read_b64_tr_b16: 8 bytes per lane, half the LDS rate, five
// to nine LDS cycles per MFMA -- the kernel was LDS-bound at 0.37 of its peak.)
//
// Block = 64 o x 64 c x 9 taps (2 x 2 waves, nine 32x32 accumulators = 144 AGPRs each), walking its share of 4 x 16 pixel
// tiles (K = 64 per tile: 216 MFMAs per wave).  ONE block per CU, one wave per SIMD -- so nothing hides a wave's own
// latencies but the wave itself, and the whole tile is ONE basic block laid out by hand (see the kernel's comment): LDS reads
// of the next patch row, the tap cuts, the split of the NEXT tile into the other of two LDS images and the global loads of
// the tile after it all ride between the MFMAs as micro-steps of about five VALU instructions, fenced by sched_barrier and
// pinned by empty asm (the compiler's own order puts every read in front of its first use and every conversion behind the
// last product).  One barrier per tile.
//
// Measured (MI355X, alone, with the fold): b2c2 167 -> 141 us, the six launches of a vgg_small step 845 -> 654 us;
// SQ_LDS_IDX_ACTIVE 59 % -> 19 % of the launch, matrix pipe busy 43 % -> 61 %.  Tried on the way and dropped:
// per-XCD fp32 atomics in L2 instead of slabs (correct -- blockIdx & 7 IS the XCC id -- but 320 G atomics/s make it no
// faster than slab + fold), the scalar-base load form (7 % slower), sched_group_barrier (the greedy solver interleaves some
// rows only, the exact one does not terminate), fp32 LDS images split behind the fragment reads (106 KB instead of 160:
// 8 % slower alone, the step unchanged).
// Partial sums go to slabs [split][tap][o][c] and are folded by wgrad_reduce_kernel (conv.hip) exactly as in the fp32
// matrix-core kernel.
#include <cstdlib>

#include "kernels.h"
#include "amax.h"
#include <type_traits>

namespace frcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define WX_TH 4
#define WX_TW 16
#define WX_GROW 128                      // bytes of one filter's gradient tile in LDS: [4 rows][16 px] bf16, 16-byte chunks XOR-swizzled
#define WX_GPLANE (64 * WX_GROW)
#define WX_XROW 48                       // bytes of one patch row: 24 elements, element e <-> input column ox0 - pad - 3 + e
#define WX_XCH (6 * WX_XROW)             // bytes of one channel's patch (6 rows)
#define WX_XPLANE (64 * WX_XCH + 8 * 16)  // (+16 per 8 channels, see wx_xaddr)
#define WX_LDS (3 * WX_GPLANE + 3 * WX_XPLANE)    // one image of the three-plane form (the kernel's own constant: NP planes)

bool conv_wgradx_eligible(int Cin, int O, int k) {
  return get_split_bf16() && k == 3 && Cin % 64 == 0 && O % 64 == 0;
}

__device__ __forceinline__ unsigned wx_cvt2(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned, r);
}
// four fp32 values -> their three bf16 planes (h, m, l), 8 bytes each
__device__ __forceinline__ void wx_split4(const float* v, uint2& H, uint2& Mi, uint2& L) {
  unsigned h[2], m[2], l[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float x0 = v[2 * j], x1 = v[2 * j + 1];
    h[j] = wx_cvt2(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h[j] << 16), r1 = x1 - __builtin_bit_cast(float, h[j] & 0xFFFF0000u);
    m[j] = wx_cvt2(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m[j] << 16), s1 = r1 - __builtin_bit_cast(float, m[j] & 0xFFFF0000u);
    l[j] = wx_cvt2(s0, s1);
  }
  H = make_uint2(h[0], h[1]); Mi = make_uint2(m[0], m[1]); L = make_uint2(l[0], l[1]);
}

__device__ __forceinline__ unsigned wx_cvt2h(float a, float b) {
  f32x2 v = {a, b};
  f16x2 r = __builtin_convertvector(v, f16x2);
  return __builtin_bit_cast(unsigned, r);
}
// (convx.hip's x16_exp / x16_pow2: the power of two that puts a tensor's largest magnitude into [2^top, 2^(top+1)))
__device__ __forceinline__ int wx_exp(float amax, int top) {
  const int be = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xFFu);
  if (be == 0 || be == 255) return 0;   // all zero -- or an infinity somewhere: no scaling (the finite elements keep fp16's own range)
  const int e = top - (be - 127);
  return e < -100 ? -100 : e > 100 ? 100 : e;
}
__device__ __forceinline__ float wx_pow2(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }

struct WgradXArgs {
  const float* in;
  const float* in_slope;
  const float* in_scale;
  const float* g;
  float* slab;   // [nSplit][9][O][Cin]
  float* gbias;  // optional [O]: += sum over pixels of g (accGradParameters' bias half), taken from the staged gradient tiles
  const float* amax_in;   // NP = 2 (two-plane fp16 form): magnitude records of `in` and `g` (amax.h)
  const float* amax_g;
  const int* omap;        // optional (WgradMap): filter o of this launch is the tensor's filter omap[o] (negative: none) -- for gbias
  int Cin, H, W, O, Ho, Wo, pad;
  int tilesX, tilesY, oTiles, cTiles, nSplit;
};

// LDS byte offset of channel c's patch inside a plane: the 288-byte pitch is 8 banks (of 4 bytes) mod 64, so the 16-byte
// reads of 8 consecutive channels cover every second group of 4 banks; each further group of 8 channels starts one
// 16-byte chunk later and fills the gaps of the one before
__device__ __forceinline__ unsigned wx_xaddr(int c) { return (unsigned)(c * WX_XCH + (c >> 3) * 16); }
// chunk swizzle of filter row o: a bijection of (o >> 1) & 7 whose upper two bits differ between neighbours (conflict-free
// 16-byte fragment reads of 16 consecutive filters AND conflict-free 8-byte staging writes of 8 filters x 4 segments)
__device__ __forceinline__ unsigned wx_gswz(int o) { const unsigned u = (o >> 1) & 7; return ((u & 3) << 1) | (u >> 2); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte load the compiler may not assume aligned

// Pins a value at this point of the instruction stream: an empty volatile asm that "modifies" it.  sched_barrier fences the
// machine scheduler, but instruction selection is free to sink side-effect-free arithmetic down to its first use -- without
// the pins the split arithmetic of a tile gathers behind the last MFMA of its row instead of riding between the MFMAs.
#define WX_PIN(x) asm volatile("" : "+v"(x))

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// VEC = 1: every 4-pixel segment is one aligned 16-byte load that lies inside a row or outside the image as a whole (pad = 1,
// W % 4 = 0, 16-byte aligned tensors).  VEC = 2: one UNALIGNED 16-byte load per segment for any width (pad = 1): a segment may
// hang over the end of its row -- those elements are zeroed when the values are split -- and the last one of the tensor reads up
// to 12 bytes past its end, which the launcher only allows when the allocation is known to extend that far.  VEC = 0: four
// predicated dword loads per segment (any shape).
//
// One block per CU, one wave per SIMD, and ONE basic block per tile in which everything overlaps (the order is laid down
// with sched_group_barrier, the compiler on its own puts every LDS read right in front of its first use and every
// conversion behind the last product):
//   tile i, patch row r = 0..5:  fragment reads of row r + 1  |  the 18 / 36 / 54 MFMAs of row r, and between them
//                                * the v_perm / v_mov that cut the three column taps of row r + 1 out of the 16 elements read,
//                                * rows 1..3: the split of tile i + 1 (in registers since tile i - 1) and its LDS writes into
//                                  the OTHER LDS image,
//   after row 4: one barrier, the global loads of tile i + 2, and row 0 of tile i + 1 is read under the MFMAs of row 5.
// NP = 3: three bf16 planes per operand, six partial products (the exact split).  NP = 2: two fp16 planes of the operands scaled
// by powers of two (convx.hip, "two-plane fp16 form"), three partial products: half the MFMAs under the same staging work.
template <bool SLOPE, bool SCALE, int VEC, int NP = 3>
__global__ __launch_bounds__(256, 1) void conv_wgradx_kernel(WgradXArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // two images of LDSI bytes: [G planes][X planes]
  constexpr unsigned LDSI = NP * (WX_GPLANE + WX_XPLANE), XB = NP * WX_GPLANE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave >> 1, wc = wave & 1;
  const int h = lane >> 5, li = lane & 31;

  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (own L2 each); the virtual index gives every XCD a
  // contiguous range, in which the (o tile, c tile) blocks of one pixel split follow each other -- they walk the same
  // pixel tiles at the same time and share them through ONE L2 instead of fetching them once per XCD
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
  const float slope = SLOPE ? *p.in_slope : 1.f;
  float gmul = 1.f, xmul = 1.f, out_mul = 1.f, out_mul2 = 1.f;   // (the inverse scale in two halves: 2^-(eg + ex) need not be an fp32 number)
  if (NP == 2) {
    const float ag = amax_load_block(p.amax_g);
    float ax = amax_load_block(p.amax_in);
    if (SLOPE) ax *= fmaxf(1.f, fabsf(slope));
    const int eg = wx_exp(ag, 14), ex = wx_exp(ax, SCALE ? 13 : 14);   // (a dropout scale's entries are <= 1: one binade of headroom)
    gmul = wx_pow2(eg); xmul = wx_pow2(ex);
    const int et = -(eg + ex), e1 = et / 2;
    out_mul = wx_pow2(e1); out_mul2 = wx_pow2(et - e1);
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- staging roles: thread = (row of the operand: filter / channel tid >> 2, quarter tid & 3).  Gradient: the quarter is
  // the 4-pixel segment, items k = tile rows 0..3.  Patch: the 36 segments of a channel (6 rows x 6 segments of 4 columns)
  // are dealt j = quarter + 4 k, k = 0..8: the LDS offset of segment j is simply 8 j.
  const int srow = tid >> 2, sq = tid & 3;
  // per-thread 64-bit row bases + signed element offsets.  (The scalar-base form -- uniform tensor base + 32-bit byte offset per
  // lane -- costs this kernel 7 %: measured, b2c2 148 against 141 us.)
  const float* const gsrc = p.g + (size_t)(o0 + srow) * HoWo;
  const float* const xsrc = p.in + (size_t)(c0 + srow) * HW;
  auto gat = [&](int elem_off) { return gsrc + elem_off; };
  auto xat = [&](int elem_off) { return xsrc + elem_off; };
  const float xscale = SCALE ? p.in_scale[c0 + srow] : 1.f;
  const unsigned gdst = (unsigned)(srow * WX_GROW) + ((((unsigned)sq >> 1) ^ wx_gswz(srow)) << 4) + (sq & 1) * 8;   // ^ (k << 5) per row
  const unsigned xdst = XB + wx_xaddr(srow) + sq * 8;                                                  // + 32 k per item
  int xr[9], xs[9];   // patch row / segment of item k
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int j = sq + 4 * k;
    xr[k] = j / 6; xs[k] = j - 6 * xr[k];
  }

  // ---- fragment addressing
  const int oA = wo * 32 + li, cB = wc * 32 + li;
  const unsigned abase = (unsigned)(oA * WX_GROW) + (((unsigned)h ^ wx_gswz(oA)) << 4);   // ^ (ks << 5)
  const unsigned bbase = XB + wx_xaddr(cB) + h * 16;                           // + 48 r (+ 16)

  // ---- the tiles of this block: t_i = split + i nSplit
  const int nPix = p.tilesX * p.tilesY;
  const int nT = (nPix - split + p.nSplit - 1) / p.nSplit;
  // TWO sets of prefetch registers: the loads of a tile are issued two tiles ahead (row 4 of tile i loads tile i + 3 into the set
  // tile i + 1 has just been staged from), so that they have a whole tile -- 216 MFMAs, 4 us -- to arrive whatever the memory
  // system's mood (with one set and one tile of lead a slow box spent three times as long in s_waitcnt: SQ_WAIT_ANY).
  struct Pre {
    float vg[4][4], vx[9][4];
    bool ok[13];
    int oy0, ox0;   // origin of the tile the set holds
  };
#ifdef WX_ONE_SET   // (experiment, round 6: one prefetch set -- 52 registers fewer, a block then fits beside a conv_x3 block; one tile less lead)
  Pre pre[1];
#else
  Pre pre[2];
#endif
  // ---- the work that rides under the MFMAs, cut into MICRO-STEPS of about five VALU instructions.  Steps of item `it`
  // (0..3 gradient rows, 4..12 patch segments): load (one 16-byte segment, branch-free: a segment outside the image reads the
  // tensor's first elements and is zeroed later) | prepare (zero / activation) | split pair 0: level 1, levels 2 + 3 | split
  // pair 1: level 1, levels 2 + 3 | three 8-byte LDS writes.
  float sr[13][4];
  unsigned sh[13][2], sm[13][2], sl[13][2];
  float bsum = 0.f;        // this thread's share of the bias gradient of filter o0 + srow (its 4-pixel segments of every tile)
  bool bsum_on = true;     // off while the walk repeats the block's last tile
  auto up_lo = [](unsigned v) { return __builtin_bit_cast(float, v << 16); };
  auto up_hi = [](unsigned v) { return __builtin_bit_cast(float, v & 0xFFFF0000u); };
  // tile walk of the block: t_i = split + i nSplit as (row, column) of the tile grid, advanced without a division; past the
  // block's last tile the walk stays on it (its loads and its staging are harmless repeats nobody reads)
  const int adv_y = p.nSplit / p.tilesX, adv_x = p.nSplit % p.tilesX;
  int w_ty = split / p.tilesX, w_tx = split % p.tilesX, w_i = 0;   // the tile the NEXT load_origin() call selects
  auto load_origin = [&](Pre& P) {
    P.oy0 = w_ty * WX_TH; P.ox0 = w_tx * WX_TW;
    const int more = w_i + 1 < nT ? 1 : 0;   // (selects, not branches: the tile loop body stays one basic block)
    w_tx += more * adv_x; w_ty += more * adv_y;
    const int wrap = w_tx >= p.tilesX ? 1 : 0;
    w_tx -= wrap * p.tilesX; w_ty += wrap;
    ++w_i;
  };
  // branch-free: a segment outside the image reads the tensor's first elements; the VEC form remembers one predicate per
  // segment (a lane mask in scalar registers) for the moment the values are split, the per-element form recomputes them
  auto load_step = [&](auto itc, Pre& P) {
    constexpr int it = decltype(itc)::value;
    const int oy0 = P.oy0, ox0 = P.ox0;
    if constexpr (it < 4) {
      constexpr int k = it;
      const int oy = oy0 + k, ox = ox0 + 4 * sq;
      if (VEC) {
        P.ok[it] = oy < p.Ho && ox < p.Wo;
        const f32x4u v = *reinterpret_cast<const f32x4u*>(gat((oy * p.Wo + ox) & -(int)P.ok[it]));
        P.vg[k][0] = v[0]; P.vg[k][1] = v[1]; P.vg[k][2] = v[2]; P.vg[k][3] = v[3];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) P.vg[k][e] = *gat((oy * p.Wo + ox + e) & -(int)(oy < p.Ho && ox + e < p.Wo));
      }
    } else {
      constexpr int k = it - 4;
      const int iy = oy0 - p.pad + xr[k], ix = ox0 - p.pad - 3 + 4 * xs[k];
      if (VEC) {
        P.ok[it] = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;   // (pad = 1: ix is -4 or >= 0)
        const f32x4u v = *reinterpret_cast<const f32x4u*>(xat((iy * p.W + ix) & -(int)P.ok[it]));
        P.vx[k][0] = v[0]; P.vx[k][1] = v[1]; P.vx[k][2] = v[2]; P.vx[k][3] = v[3];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          P.vx[k][e] = *xat((iy * p.W + ix + e) & -(int)((unsigned)iy < (unsigned)p.H && (unsigned)(ix + e) < (unsigned)p.W));
      }
    }
  };
  // is element e of item `it` (as held in the registers) inside the image?
  auto inside = [&](auto itc, int e, Pre& P) -> bool {
    constexpr int it = decltype(itc)::value;
    if (VEC == 1) return P.ok[it];
    if (VEC == 2) {   // the segment's predicate from load time, and the element's own column
      if constexpr (it < 4) return P.ok[it] && P.ox0 + 4 * sq + e < p.Wo;
      else return P.ok[it] && P.ox0 - p.pad - 3 + 4 * xs[it < 4 ? 0 : it - 4] + e < p.W;
    }
    if constexpr (it < 4) return P.oy0 + it < p.Ho && P.ox0 + 4 * sq + e < p.Wo;
    else return (unsigned)(P.oy0 - p.pad + xr[it < 4 ? 0 : it - 4]) < (unsigned)p.H &&
                (unsigned)(P.ox0 - p.pad - 3 + 4 * xs[it < 4 ? 0 : it - 4] + e) < (unsigned)p.W;
  };
  // phases of an item, common numbering: 0..3 activation of a patch segment (one value each) | 4..7 split | 8 LDS writes.
  // Gradient items start at phase 4 (5 phases), patch items at 0 when there is an activation to apply (9), else at 4 (5).
  constexpr int XPH0 = (SLOPE || SCALE) ? 0 : 4;
  auto stage_step = [&](auto itc, auto phc, unsigned gwb, unsigned xwb, Pre& P) {
    constexpr int it = decltype(itc)::value, ph = decltype(phc)::value;
    if constexpr (ph < 4) {
      constexpr int k = it - 4;
      {
        constexpr int e = ph;
        float v = P.vx[k][e];
        if (SLOPE) v = v > 0.f ? v : slope * v;
        if (SCALE) v *= xscale;
        P.vx[k][e] = inside(itc, e, P) ? v : 0.f;
        WX_PIN(P.vx[k][e]);
      }
    } else if constexpr (ph == 4 || ph == 6) {
      constexpr int j = (ph - 4) / 2;
      float x0 = it < 4 ? P.vg[it < 4 ? it : 0][2 * j] : P.vx[it < 4 ? 0 : it - 4][2 * j];
      float x1 = it < 4 ? P.vg[it < 4 ? it : 0][2 * j + 1] : P.vx[it < 4 ? 0 : it - 4][2 * j + 1];
      if (it < 4 || XPH0 == 4) {   // (patch segments with an activation were zeroed when it was applied)
        x0 = inside(itc, 2 * j, P) ? x0 : 0.f;
        x1 = inside(itc, 2 * j + 1, P) ? x1 : 0.f;
      }
      if (it < 4) bsum += bsum_on ? x0 + x1 : 0.f;
      if constexpr (NP == 2) {
        const float mul = it < 4 ? gmul : xmul;
        x0 *= mul; x1 *= mul;
        sh[it][j] = wx_cvt2h(x0, x1);
        const f16x2 hv = __builtin_bit_cast(f16x2, sh[it][j]);
        sr[it][2 * j] = x0 - (float)hv[0];
        sr[it][2 * j + 1] = x1 - (float)hv[1];
      } else {
        sh[it][j] = wx_cvt2(x0, x1);
        sr[it][2 * j] = x0 - up_lo(sh[it][j]);
        sr[it][2 * j + 1] = x1 - up_hi(sh[it][j]);
      }
      WX_PIN(sh[it][j]); WX_PIN(sr[it][2 * j]); WX_PIN(sr[it][2 * j + 1]);
    } else if constexpr (ph == 5 || ph == 7) {
      constexpr int j = (ph - 5) / 2;
      if constexpr (NP == 2) {
        sl[it][j] = wx_cvt2h(sr[it][2 * j], sr[it][2 * j + 1]);
        WX_PIN(sl[it][j]);
      } else {
        sm[it][j] = wx_cvt2(sr[it][2 * j], sr[it][2 * j + 1]);
        const float s0 = sr[it][2 * j] - up_lo(sm[it][j]), s1 = sr[it][2 * j + 1] - up_hi(sm[it][j]);
        sl[it][j] = wx_cvt2(s0, s1);
        WX_PIN(sm[it][j]); WX_PIN(sl[it][j]);
      }
    } else {
      char* d = it < 4 ? smem + (gwb ^ ((unsigned)it << 5)) : smem + xwb + 32 * (it - 4);
      constexpr int PL = it < 4 ? WX_GPLANE : WX_XPLANE;
      *reinterpret_cast<uint2*>(d) = make_uint2(sh[it][0], sh[it][1]);
      if constexpr (NP == 2) {
        *reinterpret_cast<uint2*>(d + PL) = make_uint2(sl[it][0], sl[it][1]);
      } else {
        *reinterpret_cast<uint2*>(d + PL) = make_uint2(sm[it][0], sm[it][1]);
        *reinterpret_cast<uint2*>(d + 2 * PL) = make_uint2(sl[it][0], sl[it][1]);
      }
      asm volatile("" ::: "memory");
    }
  };
  // step s of the staging sequence of items [I0, I1): item-major, each item's phases in order
  auto stage_seq = [&](auto i0c, auto sc, unsigned gwb, unsigned xwb, Pre& P) {
    constexpr int I0 = decltype(i0c)::value, s = decltype(sc)::value;
    constexpr int NG = I0 < 4 ? 4 - I0 : 0;                     // gradient items at the head of the range (5 phases each)
    constexpr int NPX = 9 - XPH0;                               // phases of a patch item
    constexpr int it = s < 5 * NG ? I0 + s / 5 : I0 + NG + (s - 5 * NG) / NPX;
    constexpr int ph = s < 5 * NG ? 4 + s % 5 : XPH0 + (s - 5 * NG) % NPX;
    stage_step(std::integral_constant<int, it>{}, std::integral_constant<int, ph>{}, gwb, xwb, P);
  };

  // ---- K = the tile's 64 pixels: K step ks = tile row ks (16 pixels), lane half h takes pixels 8h..8h+7 of it.  The products
  // walk the PATCH rows: the 16 elements a lane reads of patch row r (two 16-byte reads per plane) hold its 8 pixels at all
  // three column taps -- tap kx starts at element 3 + kx: kx = 1 is registers 2..5 as they are, kx = 0 / 2 are v_perm of
  // neighbouring registers -- and serve every (ks, ky) with ks + ky = r.  The gradient fragments stay in registers.
  u32x4 a[WX_TH][NP];
  u32x4 b[3][NP], bn[3][NP];   // [kx][plane] of the row being multiplied / of the next one
  u32x4 raw[NP][2];
  constexpr int NQ = NP == 2 ? 3 : 6;   // partial products, smallest first; plane 0 = h, then m, l (three planes) / l (two)
  constexpr int PA[6] = {NP == 2 ? 1 : 2, 0, NP == 2 ? 0 : 1, 1, 0, 0}, PB[6] = {0, NP == 2 ? 1 : 2, NP == 2 ? 0 : 1, 0, 1, 0};
  auto read_row = [&](unsigned rb, int r) {
    if (r < WX_TH) {
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        a[r][pl] = *reinterpret_cast<const u32x4*>(smem + rb + pl * WX_GPLANE + (abase ^ ((unsigned)r << 5)));
    }
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      raw[pl][0] = *reinterpret_cast<const u32x4*>(smem + rb + bbase + pl * WX_XPLANE + r * WX_XROW);
      raw[pl][1] = *reinterpret_cast<const u32x4*>(smem + rb + bbase + pl * WX_XPLANE + r * WX_XROW + 16);
    }
  };
  // step 2 pl + half: elements 2 half, 2 half + 1 of the three tap fragments of plane pl
  auto cut_step = [&](auto sc) {
    constexpr int pl = decltype(sc)::value >> 1, hf = decltype(sc)::value & 1;
    const u32x4 d0 = raw[pl][0], d1 = raw[pl][1];
    const unsigned d[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
    // elements 0 and 7 are never used: keep their registers occupied until here all the same -- handed out early they become
    // scratch of the staging arithmetic while the read is still in flight, and that write has to wait for the whole LDS queue
    if (hf == 1) asm volatile("" ::"v"(d[0]), "v"(d[7]));
#pragma unroll
    for (int j = 2 * hf; j < 2 * hf + 2; ++j) {
      bn[0][pl][j] = __builtin_amdgcn_alignbit(d[j + 2], d[j + 1], 16);
      bn[1][pl][j] = d[j + 2];
      bn[2][pl][j] = __builtin_amdgcn_alignbit(d[j + 3], d[j + 2], 16);
      WX_PIN(bn[0][pl][j]); WX_PIN(bn[1][pl][j]); WX_PIN(bn[2][pl][j]);
    }
  };

  // ---- prologue: tile 0 into image 0, tile 1 into the registers, row 0 of tile 0 into fragments
  load_origin(pre[0]);
  static_for<13>([&](auto itc) { load_step(itc, pre[0]); });
  constexpr int NST = 4 * 5 + 9 * (9 - XPH0);   // staging steps of a whole tile
  static_for<NST>([&](auto sc) { stage_seq(std::integral_constant<int, 0>{}, sc, gdst, xdst, pre[0]); });
  __syncthreads();
#ifdef WX_ONE_SET
  load_origin(pre[0]);                                                // tile 1
  static_for<13>([&](auto itc) { load_step(itc, pre[0]); });
#else
  load_origin(pre[1]);                                                // tile 1
  static_for<13>([&](auto itc) { load_step(itc, pre[1]); });
  load_origin(pre[0]);                                                // tile 2
  static_for<13>([&](auto itc) { load_step(itc, pre[0]); });
#endif
  read_row(0, 0);
  static_for<2 * NP>([&](auto sc) { cut_step(sc); });
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) b[kx][pl] = bn[kx][pl];

  // tile i: products from image i & 1; tile i + 1 is staged from set P (= (i + 1) & 1), which then receives tile i + 3
  auto tile = [&](int i, Pre& P) {
    const unsigned rb = (i & 1) ? LDSI : 0, wb = LDSI - rb;
    const unsigned gwb = gdst + wb, xwb = xdst + wb;   // (LDSI is a multiple of 128: the row XOR of gdst still applies)
    bsum_on = i + 1 < nT;                                 // rows 1..3 stage tile i + 1
    auto row = [&](auto rc) {
      constexpr int r = decltype(rc)::value;
      constexpr int nky = r < 3 ? r + 1 : 6 - r;            // (ks, ky) pairs of this row: 1 2 3 3 2 1
      constexpr int NM = 3 * NQ * nky;
      // side work of the row: staging steps (rows 1..3), load steps (row 4), then the six tap-cut steps of the next row
      constexpr int NPX = 9 - XPH0;
      constexpr int NSTG = r == 1 ? 4 * 5 + NPX : (r == 2 || r == 3) ? 4 * NPX : r == 4 ? 13 : 0;
      constexpr int NS = NSTG + 2 * NP;
      __builtin_amdgcn_sched_barrier(0);
      if (r < 5) read_row(rb, r + 1); else read_row(wb, 0);
      __builtin_amdgcn_sched_barrier(0);
      static_for<NM>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        // the (ks, ky) pairs of this row, their 3 kx taps and 6 partial products interleaved: consecutive MFMAs write
        // different accumulators (a chain on ONE accumulator issues at its dependent latency, not at the pipe rate)
        constexpr int q = m / (3 * nky), kyi = (m / 3) % nky, kx = m % 3;
        constexpr int ky = (r < WX_TH ? 0 : r - (WX_TH - 1)) + kyi, ks = r - ky;
        static_assert(ks >= 0 && ks < WX_TH && ky < 3, "tap walk");
        if constexpr (NP == 2)
          acc[3 * ky + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ks][PA[q]]), __builtin_bit_cast(f16x8, b[kx][PB[q]]), acc[3 * ky + kx], 0, 0, 0);
        else
          acc[3 * ky + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ks][PA[q]]), __builtin_bit_cast(bf16x8, b[kx][PB[q]]), acc[3 * ky + kx], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // rows 0 and 5 (side work = the tap cuts only) leave the first MFMAs to cover the latency of the reads above
        constexpr int LEAD = NSTG == 0 ? 6 : 0;
        constexpr int s0 = m < LEAD ? 0 : (m - LEAD) * NS / (NM - LEAD), s1 = m < LEAD ? 0 : (m - LEAD + 1) * NS / (NM - LEAD);
        static_for<s1 - s0>([&](auto dc) {
          constexpr int s = s0 + decltype(dc)::value;
          if constexpr (s >= NSTG) {
            cut_step(std::integral_constant<int, s - NSTG>{});
          } else if constexpr (r >= 1 && r <= 3) {   // items 0..4 | 5..8 | 9..12
            stage_seq(std::integral_constant<int, r == 1 ? 0 : r == 2 ? 5 : 9>{}, std::integral_constant<int, s>{}, gwb, xwb, P);
          } else {
            if constexpr (s == 0) load_origin(P);   // tile i + 3: the staging of tile i + 1 out of this set is complete
            load_step(std::integral_constant<int, s>{}, P);
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) b[kx][pl] = bn[kx][pl];
      if (r == 4) __syncthreads();   // image `wb` is complete, image `rb` has been read for the last time
    };
    row(std::integral_constant<int, 0>{}); row(std::integral_constant<int, 1>{}); row(std::integral_constant<int, 2>{});
    row(std::integral_constant<int, 3>{}); row(std::integral_constant<int, 4>{}); row(std::integral_constant<int, 5>{});
  };
  {
    int i = 0;
#ifdef WX_ONE_SET
    for (; i < nT; ++i) tile(i, pre[0]);
#else
    for (; i + 1 < nT; i += 2) { tile(i, pre[1]); tile(i + 1, pre[0]); }
    if (i < nT) tile(i, pre[1]);
#endif
  }
  // ---- bias gradient: the blocks of channel tile 0 saw every gradient tile of their filters exactly once
  if (p.gbias && ct == 0) {
    float v = bsum;
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2);      // the four quarters of one filter row are neighbouring lanes
    const int od = p.omap ? p.omap[o0 + srow] : o0 + srow;
    if (sq == 0 && od >= 0) unsafeAtomicAdd(p.gbias + od, v);
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
        sl[(size_t)tap * OC + (size_t)o * p.Cin + c] = NP == 2 ? (acc[tap][r] * out_mul) * out_mul2 : acc[tap][r];
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

static const WgradMap* g_wx_map = nullptr;   // (the map of the call being launched: host-side, single-threaded launch path)

template <bool SLOPE, bool SCALE, int VEC, int NP = 3>
static int launch_wgradx_v(WgradXArgs& a, double flops, float* gw, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgradx_kernel<SLOPE, SCALE, VEC, NP>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, NP == 2 ? 120 * 1024 : 160 * 1024));   // (NP = 2 has static words: amax.h)
    attr_set = true;
  }
  const int grid = a.oTiles * a.cTiles * a.nSplit;
  const double bytes = 4.0 * ((double)a.Cin * a.H * a.W + (double)a.O * a.Ho * a.Wo);
  if (prof_enabled(KC_CONV_WGRADX)) prof_before(KC_CONV_WGRADX, s);
  const size_t lds = 2 * (size_t)NP * (WX_GPLANE + WX_XPLANE);   // two images (> 80 KB: one block per CU)
  hipLaunchKernelGGL((conv_wgradx_kernel<SLOPE, SCALE, VEC, NP>), dim3(grid), dim3(256), lds, s, a);
  if (g_wx_map) FR_TRY(wgrad_reduce_map(a.slab, a.nSplit, 9, a.O, a.Cin, gw, *g_wx_map, s));
  else FR_TRY(wgrad_reduce(a.slab, a.nSplit, 9, a.O * a.Cin, gw, s));
  if (prof_enabled(KC_CONV_WGRADX)) prof_after(KC_CONV_WGRADX, flops, bytes, s);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// does the allocation `p` lives in extend at least `slack` bytes past p + bytes?  Asked of the runtime at every launch that
// needs it (a host-side table look-up; only widths that are no multiple of 4 come here): a cached answer would outlive a
// free + smaller re-allocation at the same base and approve a read past the new end.
static bool readable_past_end(const void* p, size_t bytes, size_t slack) {
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return (uintptr_t)p + bytes + slack <= (uintptr_t)base + size;
}

template <bool SLOPE, bool SCALE, int NP = 3>
static int launch_wgradx(WgradXArgs& a, double flops, float* gw, hipStream_t s) {
  // one aligned 16-byte load per 4-pixel segment when every segment lies inside a row or outside the image as a whole; one
  // unaligned one when the width is arbitrary and the 12 bytes behind both tensors belong to their allocations
  if (a.pad == 1 && a.W % 4 == 0 && ((uintptr_t)a.in & 15) == 0 && ((uintptr_t)a.g & 15) == 0)
    return launch_wgradx_v<SLOPE, SCALE, 1, NP>(a, flops, gw, s);
  if (a.pad == 1 && readable_past_end(a.in, (size_t)a.Cin * a.H * a.W * 4, 12) && readable_past_end(a.g, (size_t)a.O * a.Ho * a.Wo * 4, 12))
    return launch_wgradx_v<SLOPE, SCALE, 2, NP>(a, flops, gw, s);
  return launch_wgradx_v<SLOPE, SCALE, 0, NP>(a, flops, gw, s);
}

int conv_wgradx(const float* in, int Cin, int H, int W, const float* in_slope, const float* in_scale, const float* g, int O,
                int pad, float* gw, void* ws, size_t ws_bytes, hipStream_t s, float* gbias, const float* amax_in, const float* amax_g,
                const WgradMap* map) {
  WgradXArgs a;
  a.in = in; a.in_slope = in_slope; a.in_scale = in_scale; a.g = g; a.gbias = gbias;
  a.amax_in = amax_in; a.amax_g = amax_g;
  a.omap = map ? map->omap : nullptr;
  struct MapScope { MapScope(const WgradMap* m) { g_wx_map = (m && (m->omap || m->cmap)) ? m : nullptr; } ~MapScope() { g_wx_map = nullptr; } } scope(map);
  FR_CHECK((amax_in != nullptr) == (amax_g != nullptr), "conv_wgradx: the fp16 form needs the magnitude records of both tensors");
  a.Cin = Cin; a.H = H; a.W = W; a.O = O; a.pad = pad; a.Ho = H + 2 * pad - 2; a.Wo = W + 2 * pad - 2;
  FR_CHECK(Cin % 64 == 0 && O % 64 == 0, "conv_wgradx: %d channels x %d filters is not a split-bf16 shape", Cin, O);
  FR_CHECK((long)Cin * H * W < (1L << 30) && (long)O * a.Ho * a.Wo < (1L << 30), "conv_wgradx: tensor too large for 32-bit offsets");
  wgradx_plan(a);
  const size_t need = (size_t)a.nSplit * 9 * O * Cin * 4 + 256;
  FR_CHECK(ws && ws_bytes >= need, "conv_wgradx: workspace too small (%zu < %zu)", ws_bytes, need);
  a.slab = (float*)(((uintptr_t)ws + 255) / 256 * 256);
  const double flops = 2.0 * O * Cin * 9 * (double)a.Ho * a.Wo;
  if (amax_in) {
    if (in_slope) return in_scale ? launch_wgradx<true, true, 2>(a, flops, gw, s) : launch_wgradx<true, false, 2>(a, flops, gw, s);
    return in_scale ? launch_wgradx<false, true, 2>(a, flops, gw, s) : launch_wgradx<false, false, 2>(a, flops, gw, s);
  }
  if (in_slope) return in_scale ? launch_wgradx<true, true>(a, flops, gw, s) : launch_wgradx<true, false>(a, flops, gw, s);
  return in_scale ? launch_wgradx<false, true>(a, flops, gw, s) : launch_wgradx<false, false>(a, flops, gw, s);
}

}  // namespace frcnn
