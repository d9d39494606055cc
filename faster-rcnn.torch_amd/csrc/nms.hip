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
// 2-D grid: block (x, y) counts, for its 256 keys i, the keys j of slice y; partial counts meet in an integer
// atomic (exact, order-independent), then nms_scatter_kernel writes the permutation.
#define NMS_RANK_SLICE 1024
__global__ void nms_rank_kernel(const float* __restrict__ key, int n, const int* __restrict__ n_dev, int* __restrict__ rank) {
  __shared__ float sk[256];
  if (n_dev) n = min(*n_dev, n);
  if ((int)(blockIdx.x * blockDim.x) >= n || (int)(blockIdx.y * NMS_RANK_SLICE) >= n) return;   // (uniform per block)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float ki = i < n ? key[i] : 0.f;
  const int jbeg = blockIdx.y * NMS_RANK_SLICE, jend = min(jbeg + NMS_RANK_SLICE, n);
  int r = 0;
  for (int j0 = jbeg; j0 < jend; j0 += 256) {
    int j = j0 + threadIdx.x;
    sk[threadIdx.x] = j < jend ? key[j] : 0.f;
    __syncthreads();
    const int lim = min(256, jend - j0);
    for (int t = 0; t < lim; ++t) {
      const float kj = sk[t];
      r += (kj < ki || (kj == ki && (j0 + t) < i)) ? 1 : 0;
    }
    __syncthreads();
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
                                const int* __restrict__ cls, unsigned long long* __restrict__ mask) {
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
  if (rpos >= n) return;
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
}

// Greedy scan in sorted order, 64 boxes per step, as a two-stage pipeline with ONE barrier per step.  In step g wave 0 resolves
// the 64 x 64 diagonal block of group g serially in registers (the next step's diagonal words are already in flight), emits the
// picks, and ORs word g + 1 of the rows it kept into `removed` -- the only word step g + 1 needs from them.  The other fifteen
// waves meanwhile OR the REST of the mask rows (words >= g + 1) of the boxes kept in step g - 1.  So at the start of step g + 1 word
// g + 1 holds every contribution: group g's by wave 0, group g - 1's by the background waves, older groups' by earlier steps.
// (Round 4's scan had the two phases one after the other behind a barrier each: 1.5 us a step, 229 us a frame of Detector:detect.)
#define NMS_RED_THREADS 1024
__global__ __launch_bounds__(NMS_RED_THREADS) void nms_reduce_kernel(const unsigned long long* __restrict__ mask,
                                                                     const int* __restrict__ sorted, int n,
                                                                     const int* __restrict__ n_dev, int nwp,
                                                                     long long* __restrict__ pick, int* __restrict__ count) {
  extern __shared__ unsigned long long removed[];  // [nw]
  if (n_dev) n = min(*n_dev, n);
  const int nw = (n + 63) >> 6;   // words of this run; nwp = pitch of the mask rows (the host-side bound)
  __shared__ int kept_list[2][64];
  __shared__ int nkept[2];
  __shared__ int cnt;
  for (int w = threadIdx.x; w < nw; w += blockDim.x) removed[w] = 0ull;
  if (threadIdx.x == 0) { cnt = 0; nkept[0] = nkept[1] = 0; }
  unsigned long long diag_next = 0ull;
  if (threadIdx.x < 64 && (int)threadIdx.x < n) diag_next = mask[(size_t)threadIdx.x * nwp];
  __syncthreads();
  for (int g = 0; g < nw; ++g) {
    if (threadIdx.x < 64) {  // wave 0: group g
      const int row = g * 64 + threadIdx.x;
      const unsigned long long diag = diag_next;
      const int nrow = row + 64;
      if (g + 1 < nw) diag_next = nrow < n ? mask[(size_t)nrow * nwp + g + 1] : 0ull;
      // this row's word g + 1, requested before the serial part (needed only if the row is kept)
      const unsigned long long urgent = (g + 1 < nw && row < n) ? mask[(size_t)row * nwp + g + 1] : 0ull;
      // wave-uniform bit sets in scalar registers; one step per KEPT box (first clear bit), not per box
      const unsigned long long w0 = removed[g];
      unsigned long long word = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(w0 >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)w0);
      const int lim = min(64, n - g * 64);
      if (lim < 64) word |= ~0ull << lim;          // positions past the last box count as removed
      unsigned long long kept = 0ull;
      const int dlo = (int)diag, dhi = (int)(diag >> 32);
      while (~word) {
        const int t = __builtin_amdgcn_readfirstlane(__ffsll((long long)~word) - 1);
        kept |= 1ull << t;
        const unsigned long long dt = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(dhi, t) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane(dlo, t);
        word |= dt | (1ull << t);
      }
      // emit picks in order; the kept rows' word g + 1 goes into `removed` now, the rest of their rows in the next step
      const int base = cnt;
      if ((kept >> threadIdx.x) & 1ull) {
        const int before = __popcll(kept & ((1ull << threadIdx.x) - 1ull));
        pick[base + before] = (long long)sorted[row] + 1;  // 1-based like the Lua surface
        kept_list[g & 1][before] = row;
        if (urgent) atomicOr(&removed[g + 1], urgent);
      }
      if (threadIdx.x == 0) {
        nkept[g & 1] = __popcll(kept);
        cnt = base + __popcll(kept);
      }
    } else if (g > 0) {      // waves 1..15: the rest of the rows kept in step g - 1 (their word g went in during that step)
      const int nk = nkept[(g - 1) & 1], W = nw - (g + 1);
      const int items = nk * W;
      const int* kl = kept_list[(g - 1) & 1];
      for (int it = (int)threadIdx.x - 64; it < items; it += NMS_RED_THREADS - 64) {
        const int q = it / W, w = g + 1 + (it - q * W);
        const unsigned long long v = mask[(size_t)kl[q] * nwp + w];
        if (v) atomicOr(&removed[w], v);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = cnt;
}

size_t nms_workspace_bytes(int n) {
  size_t nw = (size_t)cdiv(n, 64);
  size_t b = 0;
  b += (size_t)n * 4 * 4;          // area, key, sorted, rank
  b = (b + 255) / 256 * 256;
  b += (size_t)n * nw * 8;         // mask
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
  FR_CHECK((size_t)nw * 8 <= 64 * 1024, "nms: n=%d too large (max 524288)", n);
  char* base = (char*)(((uintptr_t)ws + 255) / 256 * 256);
  float* area = (float*)base;
  float* key = area + n;
  int* sorted = (int*)(key + n);
  int* rank = sorted + n;
  size_t off = ((size_t)n * 16 + 255) / 256 * 256;
  unsigned long long* mask = (unsigned long long*)(base + off);
  double pair_bytes = 20.0 * n;
  FR_LAUNCH(KC_NMS, 0, pair_bytes, s, nms_prep_kernel, dim3(cdiv(n, 256)), dim3(256), 0, boxes, n, n_dev, ncols,
            key_mode, key_col, area, key);
  FR_HIP(hipMemsetAsync(rank, 0, (size_t)n * 4, s));
  FR_LAUNCH(KC_NMS, 0, 8.0 * n, s, nms_rank_kernel, dim3(cdiv(n, 256), cdiv(n, NMS_RANK_SLICE)), dim3(256), 0, key, n, n_dev, rank);
  FR_LAUNCH(KC_NMS, 0, 8.0 * n, s, nms_scatter_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (const int*)rank, n, n_dev, sorted);
  FR_LAUNCH(KC_NMS, 3.5 * n * (double)n, 8.0 * n * nw / 2, s, nms_mask_kernel, dim3(nw, nw), dim3(64), 0,
            boxes, ncols, area, sorted, n, n_dev, nw, overlap, cls, mask);
  FR_LAUNCH(KC_NMS, 0, 8.0 * n * nw / 2, s, nms_reduce_kernel, dim3(1), dim3(NMS_RED_THREADS), (size_t)nw * 8, mask,
            sorted, n, n_dev, nw, pick, count);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

}  // namespace frcnn
