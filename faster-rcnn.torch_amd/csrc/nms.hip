// nms.hip -- nms(boxes, overlap, scores) of the reference (nms.lua:23-102; callers
// Detector.lua:82,133) as coalesced-HBM, wavefront-ballot kernels (no MFMA: this is compare /
// gather work).  Survivor ids are BIT-EXACT with the reference's fp32 CPU arithmetic:
//   area  = (x2 - x1 + 1) * (y2 - y1 + 1)                      nms.lua:35
//   w     = max(0, (xx2 + (-1)*xx1) + 1), h likewise           nms.lua:85-86
//   IoU   = (w*h) / ((area_j + area_i) - w*h), keep IoU <= t   nms.lua:89-96
// every operation individually rounded to fp32 -- so FMA contraction is switched off for this
// translation unit.  Sort key dispatch (nms.lua:37-43): column / 'area' / otherwise y2.
// Tie rule (TH's quicksort is unstable, tie order unpinned by the reference): ascending key,
// ties by ascending row id, picks taken from the end.
//
// Pipeline: (1) area+key, (2) O(n^2) rank sort (exact, stable, n <= ~64K), (3) 64x64 suppression
// bit-matrix, upper triangle only, one wave per tile row-block, (4) single-workgroup scan that
// resolves each 64-row group sequentially with readlane and ORs kept rows into the removed set.
#pragma clang fp contract(off)
#include "kernels.h"

namespace frcnn {

// n_dev (optional, every kernel): the row count lives in device memory (the match count a scan just produced, the
// number of candidates that passed the class test): n = min(*n_dev, n) -- the launch is sized for the host-side bound
// and the pipeline that feeds it never waits for a read-back (Detector.lua:39-85 without a host round trip).
__global__ void nms_prep_kernel(const float* __restrict__ boxes, int n, const int* __restrict__ n_dev, int ncols, int key_mode,
                                int key_col, float* __restrict__ area, float* __restrict__ key) {
  if (n_dev) n = min(*n_dev, n);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* b = boxes + (size_t)i * ncols;
  float dx = b[2] - b[0];
  float dy = b[3] - b[1];
  dx = dx + 1.0f;
  dy = dy + 1.0f;
  float a = dx * dy;
  area[i] = a;
  key[i] = key_mode == 2 ? b[key_col - 1] : (key_mode == 1 ? a : b[3]);
}

// rank[i] = #{ j : key[j] < key[i]  or (key[j] == key[i] and j < i) }; sorted[n-1-rank] = i.
// 2-D grid: block (x, y) counts, for its 256 keys i, the 256 keys j of slice y (one LDS image, one barrier, 64 broadcast reads of
// four keys); partial counts meet in an integer atomic (exact, order-independent), then nms_scatter_kernel writes the permutation.
// Off the diagonal the tie rule is a constant (j < i for every pair of a block below the diagonal, never above it).  Round 6:
// slices of 256 instead of 1024 keys -- a thread's 1 024 dependent compare steps were the whole 30 us of the launch at any n,
// with one block per compute unit.
#define NMS_RANK_SLICE 256
__global__ __launch_bounds__(256) void nms_rank_kernel(const float* __restrict__ key, int n, const int* __restrict__ n_dev, int* __restrict__ rank) {
  __shared__ __attribute__((aligned(16))) float sk[NMS_RANK_SLICE];
  if (n_dev) n = min(*n_dev, n);
  const int ib = blockIdx.x, jb = blockIdx.y, t = threadIdx.x;
  if (ib * 256 >= n || jb * NMS_RANK_SLICE >= n) return;   // (uniform per block)
  const int i = ib * 256 + t, j = jb * NMS_RANK_SLICE + t;
  const float ki = i < n ? key[i] : 0.f;
  sk[t] = j < n ? key[j] : __builtin_nanf("");   // (past the end: compares false both ways)
  __syncthreads();
  int r = 0;
  const float4* s4 = reinterpret_cast<const float4*>(sk);
  if (jb < ib) {
#pragma unroll 8
    for (int q = 0; q < NMS_RANK_SLICE / 4; ++q) {
      const float4 k = s4[q];
      r += (k.x <= ki) + (k.y <= ki) + (k.z <= ki) + (k.w <= ki);
    }
  } else if (jb > ib) {
#pragma unroll 8
    for (int q = 0; q < NMS_RANK_SLICE / 4; ++q) {
      const float4 k = s4[q];
      r += (k.x < ki) + (k.y < ki) + (k.z < ki) + (k.w < ki);
    }
  } else {
    for (int q = 0; q < NMS_RANK_SLICE; ++q) {
      const float kj = sk[q];
      r += (kj < ki || (kj == ki && q < t)) ? 1 : 0;
    }
  }
  if (i < n && r) atomicAdd(rank + i, r);
}
__global__ void nms_scatter_kernel(const int* __restrict__ rank, int n, const int* __restrict__ n_dev, int* __restrict__ sorted) {
  if (n_dev) n = min(*n_dev, n);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sorted[n - 1 - rank[i]] = i;
}

// mask[a][w] bit b: box at sorted position a suppresses box at sorted position w*64+b (b > a)
// cls (optional): rows only suppress rows of the same class -- the per-class NMS problems of Detector.lua:125-136 in one pass
__global__ void nms_mask_kernel(const float* __restrict__ boxes, int ncols, const float* __restrict__ area,
                                const int* __restrict__ sorted, int n, const int* __restrict__ n_dev, int nw, float thr,
                                const int* __restrict__ cls, unsigned long long* __restrict__ mask,
                                unsigned long long* __restrict__ diagT) {
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  if (n_dev) n = min(*n_dev, n);
  if (cb * 64 >= n) return;   // (rows of the device-side count only; nw stays the pitch of the host-side bound)
  __shared__ float cx1[64], cy1[64], cx2[64], cy2[64], car[64];
  __shared__ int ccl[64];
  const int t = threadIdx.x;  // 64 threads = one wave
  const int cpos = cb * 64 + t;
  if (cpos < n) {
    int j = sorted[cpos];
    const float* b = boxes + (size_t)j * ncols;
    cx1[t] = b[0]; cy1[t] = b[1]; cx2[t] = b[2]; cy2[t] = b[3]; car[t] = area[j];
    ccl[t] = cls ? cls[j] : 0;
  }
  __syncthreads();
  const int rpos = rb * 64 + t;
  if (rpos >= n) return;   // (the ballots of a diagonal tile's transpose below see these lanes as empty rows)
  const int i = sorted[rpos];
  const float* bi = boxes + (size_t)i * ncols;
  const float ix1 = bi[0], iy1 = bi[1], ix2 = bi[2], iy2 = bi[3], iar = area[i];
  const int icl = cls ? cls[i] : 0;
  unsigned long long bits = 0ull;
  const int lim = min(64, n - cb * 64);
  for (int c = 0; c < lim; ++c) {
    if (cb * 64 + c <= rpos) continue;
    float xx1 = cx1[c] > ix1 ? cx1[c] : ix1;  // cmax, nms.lua:78
    float yy1 = cy1[c] > iy1 ? cy1[c] : iy1;
    float xx2 = cx2[c] < ix2 ? cx2[c] : ix2;  // cmin, nms.lua:80
    float yy2 = cy2[c] < iy2 ? cy2[c] : iy2;
    float w = xx2 + (-1.0f) * xx1;
    w = w + 1.0f;
    w = w > 0.0f ? w : 0.0f;
    float h = yy2 + (-1.0f) * yy1;
    h = h + 1.0f;
    h = h > 0.0f ? h : 0.0f;
    float inter = w * h;
    float denom = car[c] + iar;
    denom = denom - inter;
    float iou = inter / denom;
    if (!(iou <= thr) && ccl[c] == icl) bits |= 1ull << c;
  }
  mask[(size_t)rpos * nw + cb] = bits;
  // diagonal tile: the block transposed as well -- diagT[pos] bit u: the box at sorted position 64 rb + u (u before pos in its
  // group) suppresses the box at pos.  The scan resolves a group from these columns in a few wave-wide rounds (nms_reduce_kernel).
  if (rb == cb) {
    unsigned long long col = 0ull;
    for (int c = 0; c < 64; ++c) {
      const unsigned long long b = __ballot((int)((bits >> c) & 1ull));
      if (t == c) col = b;
    }
    diagT[rpos] = col;
  }
}

// OR of a 64-bit value over the 64 lanes of a wave, returned wave-uniform: data-parallel-primitive moves inside the vector ALU
// (row shifts, then the two row broadcasts), twelve instructions and two readlanes -- a shuffle through the LDS crossbar costs a
// round trip per step.
__device__ __forceinline__ unsigned nms_wave_or32(unsigned v) {
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8  -> lane 15 of each row: the row's OR
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1, 3
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2, 3 -> lane 63: all
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long nms_wave_or64(unsigned long long v) {
  return ((unsigned long long)nms_wave_or32((unsigned)(v >> 32)) << 32) | nms_wave_or32((unsigned)v);
}

// Greedy scan in sorted order, 64 boxes per step, ONE barrier per step; a step is as long as the instruction stream of its
// longest wave (a wave issues one instruction every four cycles at best), so round 6 splits the step's work over three roles and
// takes every memory round trip out of it:
//   wave 0          resolves the 64 x 64 diagonal block of group g in a few wave-wide rounds (from the block's TRANSPOSE, which the
//                   mask kernel writes beside it), ORs word g + 1 of the rows it keeps and publishes the kept set of the group;
//   waves 1..NMS_NU ("helpers", one step behind): wave k ORs word (g - 1) + 1 + k of the rows kept in group g - 1 -- final one
//                   barrier later, i.e. at step g + 1 <= the step that reads that word;
//   wave NMS_NU + 1 ("emitter", one step behind, stores only): writes the sorted POSITIONS of the kept rows to pick[]; when the scan
//                   is over the whole block turns positions into box ids (pick[i] = sorted[pick[i]] + 1).  Kept apart because on
//                   gfx950 loads and stores share the vmcnt counter and may complete out of order: a wave with both in flight
//                   makes the compiler wait for everything (vmcnt(0)) -- in the scan wave that was a full memory round trip;
//   the other waves ("background") OR the words > h + 1 + NMS_NU of the rows kept in group h: requested in step h + 1 (a wave takes
//                   whole rows, its lanes 64 consecutive words: no per-thread index arithmetic), ORed NMS_LEAD - 1 steps later.
// Everything wave 0 and the helpers read from memory -- the diagonal word, the words behind it, the box ids of a group's 64 rows --
// does not depend on any decision: it is requested NMS_LEAD - 1 steps ahead, branch-free (clamped addresses, masked at use) so that
// the compiler's wait counts stay exact, and sits in registers when its step starts.  The step barrier is a bare s_barrier.
// (Round 5: one wave did the scan AND the next word's ORs through the compiler's per-lane atomic loop, ~560 instructions = 1.3 us a
// step whatever the block size: 170 us of a Detector:detect frame's 8 000-box scan.)
#ifndef NMS_RED_THREADS
#define NMS_RED_THREADS 1024
#endif
#ifndef NMS_LEAD
#define NMS_LEAD 4
#endif
//      NMS_LEAD:              // register sets: operands are requested NMS_LEAD - 1 steps before they are used
#define NMS_NU NMS_LEAD         // helper waves = words behind word g + 1 that are ORed one step behind the scan
#ifndef NMS_BG_ITEMS
#define NMS_BG_ITEMS 4          // mask words a background lane requests per step (64-word pieces of the rows its wave takes)
#endif
#define NMS_EMIT_WAVE (NMS_NU + 1)
#define NMS_BG_WAVES (NMS_RED_THREADS / 64 - 2 - NMS_NU)
// the step barrier: a bare s_barrier behind lgkmcnt(0) (this wave's LDS atomics and list writes are done).  __syncthreads() also
// carries a release fence for GLOBAL memory, which on gfx950 is `s_waitcnt vmcnt(0)`: it would wait for the pick store of the step
// and, with it, for every operand requested ahead.  Nothing in global memory is exchanged between the waves of this kernel.
#define NMS_STEP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__global__ __launch_bounds__(NMS_RED_THREADS) void nms_reduce_kernel(const unsigned long long* __restrict__ mask,
                                                                     const unsigned long long* __restrict__ diagT,
                                                                     const int* __restrict__ sorted, int n,
                                                                     const int* __restrict__ n_dev, int nwp,
                                                                     long long* __restrict__ pick, int* __restrict__ count) {
  extern __shared__ unsigned long long removed[];  // [nw + NMS_NU + 1]
  if (n_dev) n = min(*n_dev, n);
  const int nw = (n + 63) >> 6;   // words of this run; nwp = pitch of the mask rows (the host-side bound)
  __shared__ int kept_list[2][64];
  __shared__ int nkept[2];
  __shared__ unsigned long long keptw[2];
  __shared__ int cntw[2], total;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int w = tid; w < nw + NMS_NU + 1; w += blockDim.x) removed[w] = 0ull;
  if (tid < 2) { nkept[tid] = 0; keptw[tid] = 0ull; }
  if (n <= 0) { if (tid == 0) *count = 0; return; }
  __syncthreads();
  if (wave == 0) {
    // ---- the scan.  Register set d: group g's diagonal word and word g + 1 of this lane's row
    unsigned long long dg[NMS_LEAD], u0[NMS_LEAD];
    auto request = [&](int d, int g) {
      const int gc = min(g, nw - 1);
      const int row = min(gc * 64 + lane, n - 1);   // (past the end: harmless repeats, masked when used)
      dg[d] = diagT[row];                                           // the group's diagonal block, transposed (column of this row)
      u0[d] = mask[(size_t)row * nwp + min(gc + 1, nw - 1)];
    };
#pragma unroll
    for (int d = 0; d < NMS_LEAD; ++d) { dg[d] = 0ull; u0[d] = 0ull; }
#pragma unroll
    for (int d = 0; d < NMS_LEAD - 1; ++d) request(d, d);
    int cnt = 0;               // picks so far (uniform)
    for (int g0 = 0; g0 < nw; g0 += NMS_LEAD) {
#pragma unroll
      for (int d = 0; d < NMS_LEAD; ++d) {
        const int g = g0 + d;
        if (g >= nw) break;    // (uniform)
        const int row = g * 64 + lane;
        const bool rok = row < n;
        const unsigned long long diag = rok ? dg[d] : 0ull;
        const unsigned long long nxt = (rok && g + 1 < nw) ? u0[d] : 0ull;
        request((d + NMS_LEAD - 1) % NMS_LEAD, g + NMS_LEAD - 1);   // (into the set consumed in the previous step)
        // Greedy order without a serial walk: box t is kept iff it is not removed on entry and no KEPT box before it in the group
        // suppresses it -- decidable as soon as every box of its column (its potential suppressors, `diag`) is decided.  Each
        // round decides all such boxes at once with two ballots; the first undecided box always qualifies, and the number of
        // rounds is the depth of the suppression chains among the survivors (2-5 here), not their count.  (Round 5 walked the
        // kept boxes one by one with readlane: 52 ns each, 64 us of the 8 000-box scan.)
        const unsigned long long w0 = removed[g];
        unsigned long long done = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(w0 >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)w0);   // removed on entry: decided, not kept
        const int lim = min(64, n - g * 64);
        if (lim < 64) done |= ~0ull << lim;          // positions past the last box count as removed
        unsigned long long kept = 0ull;
        bool mine = (done >> lane) & 1ull;           // this lane's box is decided
        while (~done) {
          const bool ready = !mine && (diag & ~done) == 0ull;
          const unsigned long long nd = __ballot(ready);
          const unsigned long long nk = __ballot(ready && (diag & kept) == 0ull);
          mine = mine || ready;
          done |= nd; kept |= nk;
        }
        // word g + 1 of the kept rows, ORed across the wave
        unsigned long long acc = ((kept >> lane) & 1ull) ? nxt : 0ull;
        acc = nms_wave_or64(acc);
        if (lane == 0) {
          if (acc) atomicOr(&removed[g + 1], acc);   // (the background waves OR into the same words)
          keptw[g & 1] = kept;
          nkept[g & 1] = __popcll(kept);
          cntw[g & 1] = cnt;
        }
        if ((kept >> lane) & 1ull)     // the kept rows' sorted positions, in order: for the background waves and the emitter
          kept_list[g & 1][__popcll(kept & ((1ull << lane) - 1ull))] = row;
        cnt += __popcll(kept);
        NMS_STEP_BARRIER();
      }
    }
    if (lane == 0) total = cnt;
  } else if (wave <= NMS_NU) {
    // ---- helper k = wave: in step g, word (g - 1) + 1 + k = g + k of the rows kept in group g - 1
    const int k = wave;
    unsigned long long un[NMS_LEAD];
    auto request = [&](int d, int g) {     // operands of step g
      const int gc = min(max(g - 1, 0), nw - 1);
      const int row = min(gc * 64 + lane, n - 1);
      un[d] = mask[(size_t)row * nwp + min(gc + 1 + k, nw - 1)];
    };
#pragma unroll
    for (int d = 0; d < NMS_LEAD; ++d) un[d] = 0ull;
#pragma unroll
    for (int d = 0; d < NMS_LEAD - 1; ++d) request(d, d);
    for (int g0 = 0; g0 < nw; g0 += NMS_LEAD) {
#pragma unroll
      for (int d = 0; d < NMS_LEAD; ++d) {
        const int g = g0 + d;
        if (g >= nw) break;    // (uniform)
        const unsigned long long v = un[d];
        request((d + NMS_LEAD - 1) % NMS_LEAD, g + NMS_LEAD - 1);
        if (g > 0 && g + k < nw) {
          const unsigned long long kept = keptw[(g - 1) & 1];      // (published by the barrier that ended step g - 1)
          if (((kept >> lane) & 1ull) && v) atomicOr(&removed[g + k], v);   // (a row of the kept set exists: < n)
        }
        NMS_STEP_BARRIER();
      }
    }
  } else if (wave == NMS_EMIT_WAVE) {
    // ---- emitter: in step g the picks of group g - 1 (stores only; the last group's after the loop)
    for (int g = 0; g < nw; ++g) {
      if (g > 0 && lane < nkept[(g - 1) & 1]) pick[cntw[(g - 1) & 1] + lane] = (long long)kept_list[(g - 1) & 1][lane];
      NMS_STEP_BARRIER();
    }
    if (lane < nkept[(nw - 1) & 1]) pick[cntw[(nw - 1) & 1] + lane] = (long long)kept_list[(nw - 1) & 1][lane];
  } else {
    // ---- background: words in flight across barriers, one slot per step of lead
    const int bwave = wave - 2 - NMS_NU;
    unsigned long long bv[NMS_LEAD][NMS_BG_ITEMS];
    int bw[NMS_LEAD][NMS_BG_ITEMS];      // the word each one belongs to (-1: none)
#pragma unroll
    for (int d = 0; d < NMS_LEAD; ++d)
#pragma unroll
      for (int j = 0; j < NMS_BG_ITEMS; ++j) { bv[d][j] = 0ull; bw[d][j] = -1; }
    for (int g0 = 0; g0 < nw; g0 += NMS_LEAD) {
#pragma unroll
      for (int d = 0; d < NMS_LEAD; ++d) {
        const int g = g0 + d;
        if (g >= nw) break;    // (uniform)
        // (a) the words requested NMS_LEAD - 1 steps ago (rows kept in group g - NMS_LEAD): slot d + 1
        const int dc = (d + 1) % NMS_LEAD;   // (compile-time: the loop is unrolled)
#pragma unroll
        for (int j = 0; j < NMS_BG_ITEMS; ++j) {
          if (bw[dc][j] >= 0 && bv[dc][j]) atomicOr(&removed[bw[dc][j]], bv[dc][j]);
          bw[dc][j] = -1;
        }
        // (b) request words > (g - 1) + 1 + NMS_NU of the rows kept in group g - 1 into slot d: ORed in step g + NMS_LEAD - 1,
        //     final from step g + NMS_LEAD = (g - 1) + 1 + NMS_NU + 1 on, the first step that reads such a word.  A wave takes
        //     whole rows (kept row q = wave, wave + NMS_BG_WAVES, ...), its lanes 64 consecutive words of the row.  Branch-free:
        //     a lane without a word reads the first word of the matrix and drops it.
        {
          const int wfirst = g + 1 + NMS_NU;
          const int nk = __builtin_amdgcn_readfirstlane(g > 0 ? nkept[(g - 1) & 1] : 0);
          const int W = nw - wfirst, nchunks = (W + 63) >> 6;
          const int* kl = kept_list[(g - 1) & 1];
          int q = W > 0 ? bwave : nk, c = 0;      // (q >= nk: nothing to do; q, c wave-uniform: scalar registers)
          // three passes, so that the slots' LDS reads (the kept rows' numbers) are in flight together: one LDS round trip a step
          int qs[NMS_BG_ITEMS], cs[NMS_BG_ITEMS], rs[NMS_BG_ITEMS];
#pragma unroll
          for (int j = 0; j < NMS_BG_ITEMS; ++j) {
            qs[j] = q; cs[j] = c;
            if (++c >= nchunks) { c = 0; q += NMS_BG_WAVES; }
          }
#pragma unroll
          for (int j = 0; j < NMS_BG_ITEMS; ++j) rs[j] = qs[j] < nk ? kl[qs[j]] : 0;   // (no row: row 0 -- the list holds nothing defined there)
#pragma unroll
          for (int j = 0; j < NMS_BG_ITEMS; ++j) {
            // the row's base is a scalar (one LDS word, broadcast), the lane adds its word: base + offset addressing
            const unsigned long long* rowp = mask + (size_t)__builtin_amdgcn_readfirstlane(rs[j]) * nwp;
            const int w = wfirst + cs[j] * 64 + lane;
            const bool has = qs[j] < nk && w < nw;
            bv[d][j] = rowp[has ? w : 0];         // (unconditional load: the number of loads of a step is fixed, the waits counted)
            bw[d][j] = has ? w : -1;
          }
          while (q < nk) {   // more rows / longer rows than the registers hold: at once
            const int w = wfirst + c * 64 + lane;
            if (w < nw) {
              const unsigned long long v = mask[(size_t)kl[q] * nwp + w];
              if (v) atomicOr(&removed[w], v);
            }
            if (++c >= nchunks) { c = 0; q += NMS_BG_WAVES; }
          }
        }
        NMS_STEP_BARRIER();
      }
    }
  }
  // ---- every wave: positions -> box ids, 1-based like the Lua surface (the emitter's stores are visible behind the fence)
  __syncthreads();
  const int cnt_all = total;
  for (int i = tid; i < cnt_all; i += blockDim.x) pick[i] = (long long)sorted[(int)pick[i]] + 1;
  if (tid == 0) *count = cnt_all;
}
#undef NMS_STEP_BARRIER

size_t nms_workspace_bytes(int n) {
  size_t nw = (size_t)cdiv(n, 64);
  size_t b = 0;
  b += (size_t)n * 4 * 4;          // area, key, sorted, rank
  b = (b + 255) / 256 * 256;
  b += (size_t)n * nw * 8;         // mask
  b = (b + 255) / 256 * 256;
  b += (size_t)n * 8;              // the diagonal blocks transposed
  return b + 256;
}

int nms_device(const float* boxes, int n, int ncols, float overlap, int key_mode, int key_col,
               long long* pick, int* count, void* ws, size_t ws_bytes, hipStream_t s, const int* cls, const int* n_dev) {
  if (n <= 0) {  // nms.lua:26-28
    FR_HIP(hipMemsetAsync(count, 0, sizeof(int), s));
    return FRCNN_OK;
  }
  FR_CHECK(ncols >= 4, "nms: boxes need >= 4 columns (got %d)", ncols);
  FR_CHECK(key_mode >= 0 && key_mode <= 2, "nms: bad key_mode %d", key_mode);
  FR_CHECK(key_mode != 2 || (key_col >= 1 && key_col <= ncols), "nms: key column %d out of range", key_col);
  FR_CHECK(ws_bytes >= nms_workspace_bytes(n), "nms: workspace too small (%zu < %zu)", ws_bytes,
           nms_workspace_bytes(n));
  const int nw = cdiv(n, 64);
  FR_CHECK((size_t)(nw + NMS_NU + 1) * 8 <= 64 * 1024, "nms: n=%d too large (max 523968)", n);
  char* base = (char*)(((uintptr_t)ws + 255) / 256 * 256);
  float* area = (float*)base;
  float* key = area + n;
  int* sorted = (int*)(key + n);
  int* rank = sorted + n;
  size_t off = ((size_t)n * 16 + 255) / 256 * 256;
  unsigned long long* mask = (unsigned long long*)(base + off);
  unsigned long long* diagT = (unsigned long long*)(base + (off + (size_t)n * nw * 8 + 255) / 256 * 256);
  double pair_bytes = 20.0 * n;
  FR_LAUNCH(KC_NMS, 0, pair_bytes, s, nms_prep_kernel, dim3(cdiv(n, 256)), dim3(256), 0, boxes, n, n_dev, ncols,
            key_mode, key_col, area, key);
  FR_HIP(hipMemsetAsync(rank, 0, (size_t)n * 4, s));
  FR_LAUNCH(KC_NMS, 0, 8.0 * n, s, nms_rank_kernel, dim3(cdiv(n, 256), cdiv(n, NMS_RANK_SLICE)), dim3(256), 0, key, n, n_dev, rank);
  FR_LAUNCH(KC_NMS, 0, 8.0 * n, s, nms_scatter_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (const int*)rank, n, n_dev, sorted);
  FR_LAUNCH(KC_NMS, 3.5 * n * (double)n, 8.0 * n * nw / 2, s, nms_mask_kernel, dim3(nw, nw), dim3(64), 0,
            boxes, ncols, area, sorted, n, n_dev, nw, overlap, cls, mask, diagT);
  FR_LAUNCH(KC_NMS, 0, 8.0 * n * nw / 2, s, nms_reduce_kernel, dim3(1), dim3(NMS_RED_THREADS), (size_t)(nw + NMS_NU + 1) * 8, mask,
            (const unsigned long long*)diagT, sorted, n, n_dev, nw, pick, count);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

}  // namespace frcnn
