// Replay ring: push (scatter k rows into the ring) and gather (logical index -> dense
// per-column batches).  Replaces the cat+slice FIFO and fancy-index gather of
// torchrl/utils/storage.py:7-16, 27-37, 44-45 (callers: examples/dqn.py:27-28, 40-41;
// examples/ddpg.py:25-26, 40-41).
//
// HBM layout: one byte buffer of `capacity` records, `stride` bytes each; the columns of
// a transition sit side by side inside its record, so a narrow transition (36-44 B for
// Pendulum / CartPole) is ONE contiguous read instead of five 4-16 B reads in five
// arrays, and an Atari-shaped record (2 x 28,224 B frames + 12 B) is a 56 KB stream.
// Two gather kernels:
//   k_gather_narrow  one lane per dword of a record region; a block takes 64 rows and
//                    loads their 64 indices with one coalesced wave-wide load.
//   k_gather_wide    a persistent, software-pipelined grid (4 workgroups per CU) over work items
//                    (row, 16 KiB chunk of a wide column): 4 x 16 B non-temporal loads per lane, the
//                    next item's loads issued before the current item's stores; 4 B -> float4 per
//                    lane when a u8 column is cast to f32.  The record's narrow columns ride in the
//                    same launch (its first B / 64 workgroups run the narrow body).
// Both are pure HBM/L2 byte movers: no LDS tiling is needed because nothing is reused.
#include <stdlib.h>

#include "trl_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // types the nontemporal builtins take
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ColArgs {
  int32_t ncols;
  int32_t off_dw[TRL_MAX_COLS];   // first dword of the column inside a record
  int32_t w_bytes[TRL_MAX_COLS];  // column width in bytes
  int32_t sdt[TRL_MAX_COLS];      // storage dtype
  int32_t odt[TRL_MAX_COLS];      // dtype on the dense side (gather output / push source)
  int32_t pre_dw[TRL_MAX_COLS + 1];  // gather: prefix sum of the listed columns' dwords
  void *ptr[TRL_MAX_COLS];        // dense per-column arrays
};

// which column holds dword d of a record (-1: padding)
struct Hit {
  int c;
  int off_dw;
  int w_bytes;
  int sdt;
  int odt;
  char *ptr;
};

__device__ __forceinline__ Hit find_col(const ColArgs &a, int d) {
  Hit h{-1, 0, 0, 0, 0, nullptr};
#pragma unroll
  for (int i = 0; i < TRL_MAX_COLS; ++i) {
    const int w_dw = (a.w_bytes[i] + 3) >> 2;
    if (i < a.ncols && d >= a.off_dw[i] && d < a.off_dw[i] + w_dw) {
      h.c = i;
      h.off_dw = a.off_dw[i];
      h.w_bytes = a.w_bytes[i];
      h.sdt = a.sdt[i];
      h.odt = a.odt[i];
      h.ptr = (char *)a.ptr[i];
    }
  }
  return h;
}

// k-th dword of the listed columns taken back to back (padding and unlisted columns skipped)
__device__ __forceinline__ Hit find_packed(const ColArgs &a, int k, int *d) {
  Hit h{-1, 0, 0, 0, 0, nullptr};
#pragma unroll
  for (int i = 0; i < TRL_MAX_COLS; ++i) {
    if (i < a.ncols && k >= a.pre_dw[i] && k < a.pre_dw[i + 1]) {
      h.c = i;
      h.off_dw = a.off_dw[i];
      h.w_bytes = a.w_bytes[i];
      h.sdt = a.sdt[i];
      h.odt = a.odt[i];
      h.ptr = (char *)a.ptr[i];
      *d = a.off_dw[i] + (k - a.pre_dw[i]);
    }
  }
  return h;
}

__device__ __forceinline__ int64_t logical_to_record(int64_t i, int64_t head, int64_t count,
                                                     int64_t capacity) {
  if (i < 0) i += count;           // torch-style negative index
  i = i < 0 ? 0 : (i >= count ? count - 1 : i);  // never fault on a bad index
  int64_t p = head + i;
  if (p >= capacity) p -= capacity;
  return p;
}

// Store dword d (value v) of a record into row `orow` of the dense array of the column that holds it.
__device__ __forceinline__ void store_col_dword(const Hit &h, int d, uint32_t v, int64_t orow) {
  const int j = d - h.off_dw;
  if (h.sdt == TRL_F32) {
    ((uint32_t *)h.ptr)[orow * (h.w_bytes >> 2) + j] = v;
  } else if (h.odt == TRL_F32) {  // u8 storage, f32 batch (exact)
    float *o = (float *)h.ptr + orow * h.w_bytes + 4 * j;
    const int nb = h.w_bytes - 4 * j;
    if (nb >= 4 && (h.w_bytes & 3) == 0) {
      *(float4 *)o = make_float4((float)(v & 255u), (float)((v >> 8) & 255u),
                                 (float)((v >> 16) & 255u), (float)(v >> 24));
    } else {
      for (int b = 0; b < (nb < 4 ? nb : 4); ++b) o[b] = (float)((v >> (8 * b)) & 255u);
    }
  } else {  // u8 -> u8
    uint8_t *o = (uint8_t *)h.ptr + orow * h.w_bytes + 4 * j;
    const int nb = h.w_bytes - 4 * j;
    if (nb >= 4 && (h.w_bytes & 3) == 0) {
      *(uint32_t *)o = v;
    } else {
      for (int b = 0; b < (nb < 4 ? nb : 4); ++b) o[b] = (uint8_t)(v >> (8 * b));
    }
  }
}

constexpr int NARROW_ROWS = 64;  // rows per block = one wave-wide index load

// 64 rows of the narrow columns: rows [blk * 64, blk * 64 + 64) of the batch.
template <bool RANGE>
__device__ __forceinline__ void narrow_block(int64_t blk, const uint32_t *__restrict__ ring, int64_t capacity,
                                             int64_t stride_dw, int64_t head, int64_t count,
                                             const int64_t *__restrict__ idx, int64_t start, int64_t B,
                                             const ColArgs &a, int nd, uint32_t magic, int64_t *s_rec) {
  const int64_t row0 = blk * NARROW_ROWS;
  const int nrows = (int)((B - row0) < NARROW_ROWS ? (B - row0) : NARROW_ROWS);
  if (threadIdx.x < NARROW_ROWS) {  // wave 0: one coalesced 64 x 8 B index load
    int64_t p = 0;
    if ((int)threadIdx.x < nrows) {
      const int64_t i = RANGE ? start + row0 + threadIdx.x : idx[row0 + threadIdx.x];
      p = logical_to_record(i, head, count, capacity);
    }
    s_rec[threadIdx.x] = p * stride_dw;
  }
  __syncthreads();
  const int total = nrows * nd;   // nd = dwords of the listed columns per row
#pragma unroll 2
  for (int e = threadIdx.x; e < total; e += 256) {
    const int r = (int)(((uint64_t)(uint32_t)e * magic) >> 32);  // e / nd
    int d = 0;
    const Hit h = find_packed(a, e - r * nd, &d);
    if (h.c < 0) continue;
    store_col_dword(h, d, ring[s_rec[r] + d], row0 + r);
  }
}

template <bool RANGE>
__global__ __launch_bounds__(256) void k_gather_narrow(
    const uint32_t *__restrict__ ring, int64_t capacity, int64_t stride_dw, int64_t head,
    int64_t count, const int64_t *__restrict__ idx, int64_t start, int64_t B, ColArgs a, int nd,
    uint32_t magic) {
  __shared__ int64_t s_rec[NARROW_ROWS];
  narrow_block<RANGE>(blockIdx.x, ring, capacity, stride_dw, head, count, idx, start, B, a, nd, magic, s_rec);
}

// ---- wide columns -------------------------------------------------------------------
constexpr int WIDE_COPY_UNIT = 4096;   // a copy item is UN of these (UN x 16 B per lane), UN = 2 or 4
constexpr int WIDE_CAST_CHUNK = 4096;  // source bytes one block casts (4 x 4 B per lane -> 4 x float4)

struct WideArgs {
  int32_t ncols;
  int32_t off_b[TRL_MAX_COLS];        // byte offset in the record (multiple of 16)
  int32_t w_bytes[TRL_MAX_COLS];      // multiple of 16
  int32_t cast[TRL_MAX_COLS];         // 1: u8 -> f32
  int32_t chunk_start[TRL_MAX_COLS + 1];  // prefix sum of chunks per column
  void *ptr[TRL_MAX_COLS];
};

// One work item = (row b, chunk ch of a wide column): 8 KiB copied as 2 x 16 B per lane, or 4 KiB of u8
// cast to 16 KiB of f32 (4 x 4 B -> 4 x float4 per lane).  The grid is persistent (a few workgroups per
// CU striding over the items) and software-pipelined: the loads of a workgroup's next item are issued
// before the stores of the current one, and the row index of the next item -- a block-uniform scalar
// load the data loads depend on -- is fetched an item ahead too.  The column table is read into scalar
// registers once per workgroup and an item is decoded with selects (no dependent scalar loads).
// The narrow columns of a record (the 12 B action / reward / done tail of an Atari-shaped transition)
// ride in the same launch: the first `narrow_blocks` workgroups run the narrow kernel's body, so a batch
// is ONE launch, not a wide and a narrow one with a kernel boundary between them.
struct WideCols {
  int off_b[TRL_MAX_COLS], w[TRL_MAX_COLS], cast[TRL_MAX_COLS], cs[TRL_MAX_COLS + 1];
  char *ptr[TRL_MAX_COLS];
};

struct WideItem {
  const char *src;
  char *dst;
  int base, w, cast;
};

template <bool RANGE, int UN>
__device__ __forceinline__ WideItem wide_item(uint32_t item, uint32_t chunks, const char *__restrict__ ring,
                                              int64_t capacity, int64_t stride, int64_t head, int64_t count,
                                              const int64_t *__restrict__ idx, int64_t start, const WideCols &c) {
  const uint32_t b = item / chunks;
  const int ch = (int)(item - b * chunks);
  int off_b = c.off_b[0], w = c.w[0], cast = c.cast[0], c0 = 0;
  char *dst = c.ptr[0];
#pragma unroll
  for (int i = 1; i < TRL_MAX_COLS; ++i) {   // unused columns have an empty chunk range
    const bool hit = ch >= c.cs[i];
    off_b = hit ? c.off_b[i] : off_b;
    w = hit ? c.w[i] : w;
    cast = hit ? c.cast[i] : cast;
    c0 = hit ? c.cs[i] : c0;
    dst = hit ? c.ptr[i] : dst;
  }
  const int64_t li = RANGE ? start + (int64_t)b : idx[b];  // block-uniform: a scalar load
  const int64_t rec = logical_to_record(li, head, count, capacity);
  WideItem it;
  it.src = ring + rec * stride + off_b;
  it.w = w;
  it.cast = cast;
  const int lc = ch - c0;
  it.base = lc * (cast ? WIDE_CAST_CHUNK : UN * WIDE_COPY_UNIT);
  it.dst = dst + (int64_t)b * w * (cast ? 4 : 1);
  return it;
}

template <int UN>
__device__ __forceinline__ void wide_load(const WideItem &it, u32x4 (&v)[UN]) {
  if (!it.cast) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int pos = it.base + (u * 256 + (int)threadIdx.x) * 16;
      if (pos < it.w) v[u] = __builtin_nontemporal_load((const u32x4 *)(it.src + pos));
    }
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pos = it.base + (u * 256 + (int)threadIdx.x) * 4;
      if (pos < it.w) v[0][u] = __builtin_nontemporal_load((const uint32_t *)(it.src + pos));
    }
  }
}

template <int UN>
__device__ __forceinline__ void wide_store(const WideItem &it, const u32x4 (&v)[UN]) {
  if (!it.cast) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int pos = it.base + (u * 256 + (int)threadIdx.x) * 16;
      if (pos < it.w) __builtin_nontemporal_store(v[u], (u32x4 *)(it.dst + pos));
    }
  } else {
    float *o = (float *)it.dst;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pos = it.base + (u * 256 + (int)threadIdx.x) * 4;
      if (pos < it.w) {
        const uint32_t x = v[0][u];
        const f32x4 f = {(float)(x & 255u), (float)((x >> 8) & 255u), (float)((x >> 16) & 255u), (float)(x >> 24)};
        __builtin_nontemporal_store(f, (f32x4 *)(o + pos));
      }
    }
  }
}

template <bool RANGE, int UN>
__global__ __launch_bounds__(256) void k_gather_wide(
    const char *__restrict__ ring, int64_t capacity, int64_t stride, int64_t head, int64_t count,
    const int64_t *__restrict__ idx, int64_t start, int64_t B, WideArgs a, ColArgs na, int nd, uint32_t magic,
    uint32_t narrow_blocks) {
  __shared__ int64_t s_rec[NARROW_ROWS];
  if (blockIdx.x < narrow_blocks) {
    narrow_block<RANGE>(blockIdx.x, (const uint32_t *)ring, capacity, stride / 4, head, count, idx, start, B, na, nd,
                        magic, s_rec);
    return;
  }
  WideCols c;
#pragma unroll
  for (int i = 0; i < TRL_MAX_COLS; ++i) {
    c.off_b[i] = a.off_b[i];
    c.w[i] = a.w_bytes[i];
    c.cast[i] = a.cast[i];
    c.cs[i] = a.chunk_start[i];
    c.ptr[i] = (char *)a.ptr[i];
  }
  const uint32_t chunks = (uint32_t)a.chunk_start[TRL_MAX_COLS];
  const uint32_t total = (uint32_t)B * chunks;
  const uint32_t step = gridDim.x - narrow_blocks;
  uint32_t item = blockIdx.x - narrow_blocks;
  if (item >= total) return;
  WideItem cur = wide_item<RANGE, UN>(item, chunks, ring, capacity, stride, head, count, idx, start, c);
  u32x4 vc[UN], vn[UN];
  wide_load<UN>(cur, vc);
  for (;;) {
    const uint32_t next = item + step;
    const bool more = next < total && next > item;
    WideItem nxt = cur;
    if (more) {
      nxt = wide_item<RANGE, UN>(next, chunks, ring, capacity, stride, head, count, idx, start, c);
      wide_load<UN>(nxt, vn);
    }
    wide_store<UN>(cur, vc);
    if (!more) break;
    cur = nxt;
#pragma unroll
    for (int u = 0; u < UN; ++u) vc[u] = vn[u];
    item = next;
  }
}

// ---- rows in place ------------------------------------------------------------------
// The prologue of a learner step whose first layer reads its frames straight out of the ring (trl_dqn_learn_fused_u8_ring):
// the physical record of every sampled row, and the transition's three one-float columns gathered densely (12 B per
// row; the two 28 KB frames of a row are NOT copied: the first layer's operand fetch reads them where they lie).
__global__ __launch_bounds__(256) void k_ring_rows(const char *__restrict__ ring, int64_t capacity, int64_t stride,
                                                   int64_t head, int64_t count, const int64_t *__restrict__ idx, int64_t B,
                                                   int off_a, int off_r, int off_d, int32_t *__restrict__ rec,
                                                   float *__restrict__ action, float *__restrict__ reward,
                                                   float *__restrict__ done) {
  for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (int64_t)gridDim.x * 256) {
    const int64_t p = logical_to_record(idx[b], head, count, capacity);
    const char *r = ring + p * stride;
    rec[b] = (int32_t)p;
    action[b] = *(const float *)(r + off_a);
    reward[b] = *(const float *)(r + off_r);
    done[b] = *(const float *)(r + off_d);
  }
}

// ---- push ---------------------------------------------------------------------------
// src columns are dense [k, w_bytes].  Value of dword d of new row r (0 for padding).
__device__ __forceinline__ uint32_t push_dword(const ColArgs &a, int64_t r, int d) {
  const Hit h = find_col(a, d);
  uint32_t v = 0;
  if (h.c >= 0) {
    const int j = d - h.off_dw;
    if ((h.w_bytes & 3) == 0) {
      v = ((const uint32_t *)h.ptr)[r * (h.w_bytes >> 2) + j];
    } else {  // ragged u8 column: assemble the dword byte by byte
      const uint8_t *s = (const uint8_t *)h.ptr + r * h.w_bytes + 4 * j;
      const int nb = h.w_bytes - 4 * j;
      for (int b = 0; b < (nb < 4 ? nb : 4); ++b) v |= (uint32_t)s[b] << (8 * b);
    }
  }
  return v;
}

// One lane per dword of the k new records.
__global__ __launch_bounds__(256) void k_ring_push(uint32_t *__restrict__ ring, int64_t capacity,
                                                   int64_t stride_dw, int64_t tail, int64_t k,
                                                   ColArgs a) {
  const int64_t total = k * stride_dw;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / stride_dw;
    const int d = (int)(e - r * stride_dw);
    int64_t p = tail + r;
    if (p >= capacity) p -= capacity;
    ring[p * stride_dw + d] = push_dword(a, r, d);
  }
}

// One lane per 16 B of the k new records (stride % 16 == 0): a quad that lies inside a
// 16 B-shaped column is one uint4 copy, any other quad is assembled dword by dword.
__global__ __launch_bounds__(256) void k_ring_push16(uint4 *__restrict__ ring, int64_t capacity,
                                                     int64_t stride_q, int64_t tail, int64_t k,
                                                     ColArgs a) {
  const int64_t total = k * stride_q;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / stride_q;
    const int q = (int)(e - r * stride_q);
    int64_t p = tail + r;
    if (p >= capacity) p -= capacity;
    const Hit h = find_col(a, q * 4);
    uint4 v;
    if (h.c >= 0 && (h.off_dw & 3) == 0 && (h.w_bytes & 15) == 0 && ((uintptr_t)h.ptr & 15) == 0) {
      v = ((const uint4 *)h.ptr)[r * (h.w_bytes >> 4) + (q - (h.off_dw >> 2))];
    } else {
      v.x = push_dword(a, r, q * 4 + 0);
      v.y = push_dword(a, r, q * 4 + 1);
      v.z = push_dword(a, r, q * 4 + 2);
      v.w = push_dword(a, r, q * 4 + 3);
    }
    ring[p * stride_q + q] = v;
  }
}

int check_layout(const char *fn, const trl_ring_layout *l, int64_t stride) {
  TRL_REQUIRE(l != nullptr, "%s: null layout", fn);
  TRL_REQUIRE(l->ncols >= 1 && l->ncols <= TRL_MAX_COLS, "%s: ncols=%d out of [1,%d]", fn, l->ncols,
              TRL_MAX_COLS);
  TRL_REQUIRE(stride > 0 && stride % 4 == 0, "%s: stride=%lld must be a positive multiple of 4", fn,
              (long long)stride);
  int64_t prev_end = 0;
  for (int c = 0; c < l->ncols; ++c) {
    TRL_REQUIRE(l->col_dtype[c] == TRL_F32 || l->col_dtype[c] == TRL_U8, "%s: column %d: bad dtype %d",
                fn, c, l->col_dtype[c]);
    TRL_REQUIRE(l->col_bytes[c] > 0, "%s: column %d: empty", fn, c);
    TRL_REQUIRE(l->col_dtype[c] != TRL_F32 || l->col_bytes[c] % 4 == 0,
                "%s: column %d: f32 column of %d bytes", fn, c, l->col_bytes[c]);
    TRL_REQUIRE(l->col_off[c] % 4 == 0 && l->col_off[c] >= prev_end,
                "%s: column %d: offset %d must be a multiple of 4 and columns must ascend without overlap",
                fn, c, l->col_off[c]);
    prev_end = (int64_t)l->col_off[c] + ((l->col_bytes[c] + 3) / 4) * 4;
    TRL_REQUIRE(prev_end <= stride, "%s: column %d ends at %lld beyond stride %lld", fn, c,
                (long long)prev_end, (long long)stride);
  }
  return TRL_OK;
}

constexpr int WIDE_MIN_BYTES = 1024;  // columns at least this wide (and 16 B-shaped) take the wide kernel

bool is_wide(const trl_ring_layout *l, int c, int64_t stride) {
  return l->col_bytes[c] >= WIDE_MIN_BYTES && l->col_bytes[c] % 16 == 0 && l->col_off[c] % 16 == 0 &&
         stride % 16 == 0;
}

int gather_impl(const char *fn, bool range, const void *ring, int64_t capacity, int64_t stride,
                const trl_ring_layout *layout, int64_t head, int64_t count, const int64_t *idx,
                int64_t start, int64_t B, void *const *dst, const int32_t *out_dtype, void *stream) {
  int rc = check_layout(fn, layout, stride);
  if (rc) return rc;
  TRL_REQUIRE(ring && dst && out_dtype, "%s: null pointer", fn);
  TRL_REQUIRE(capacity > 0 && count > 0 && count <= capacity && head >= 0 && head < capacity,
              "%s: need 0 < count <= capacity and 0 <= head < capacity (capacity=%lld head=%lld count=%lld)",
              fn, (long long)capacity, (long long)head, (long long)count);
  TRL_REQUIRE(B >= 0 && B < (1ll << 31), "%s: B=%lld out of range", fn, (long long)B);
  if (range)
    TRL_REQUIRE(start >= 0 && start + B <= count, "%s: range [%lld, %lld) outside [0, %lld)", fn,
                (long long)start, (long long)(start + B), (long long)count);
  else
    TRL_REQUIRE(idx != nullptr || B == 0, "%s: null index array", fn);
  if (B == 0) return TRL_OK;
  hipStream_t st = (hipStream_t)stream;

  ColArgs na{};
  WideArgs wa{};
  int wide_un = 4;                                          // 16 KiB copy items (2: 8 KiB)
#ifdef TRL_TUNING                                         // diagnostic build: tools/gather_sweep.py
  if (const char *t = getenv("TRL_GATHER_UN")) wide_un = atoi(t) == 2 ? 2 : 4;
#endif
  for (int c = 0; c < layout->ncols; ++c) {
    TRL_REQUIRE(dst[c] != nullptr, "%s: dst[%d] is null", fn, c);
    const int od = out_dtype[c];
    TRL_REQUIRE(od == layout->col_dtype[c] || (od == TRL_F32 && layout->col_dtype[c] == TRL_U8),
                "%s: column %d: cannot gather dtype %d as %d", fn, c, layout->col_dtype[c], od);
    const bool cast = od != layout->col_dtype[c];
    if (is_wide(layout, c, stride) && ((uintptr_t)dst[c] % 16 == 0) && ((uintptr_t)ring % 16 == 0)) {
      const int i = wa.ncols++;
      wa.off_b[i] = layout->col_off[c];
      wa.w_bytes[i] = layout->col_bytes[c];
      wa.cast[i] = cast;
      wa.ptr[i] = dst[c];
      const int chunk = cast ? WIDE_CAST_CHUNK : wide_un * WIDE_COPY_UNIT;
      wa.chunk_start[i + 1] = wa.chunk_start[i] + (int)trl_cdiv(layout->col_bytes[c], chunk);
    } else {
      TRL_REQUIRE((uintptr_t)dst[c] % 16 == 0, "%s: dst[%d] must be 16-byte aligned", fn, c);
      const int i = na.ncols++;
      na.off_dw[i] = layout->col_off[c] / 4;
      na.w_bytes[i] = layout->col_bytes[c];
      na.sdt[i] = layout->col_dtype[c];
      na.odt[i] = od;
      na.ptr[i] = dst[c];
      na.pre_dw[i + 1] = na.pre_dw[i] + (layout->col_bytes[c] + 3) / 4;
    }
  }
  const int nd = na.ncols > 0 ? na.pre_dw[na.ncols] : 0;
  uint32_t magic = 0;
  unsigned narrow_blocks = 0;
  if (na.ncols > 0) {
    TRL_REQUIRE((int64_t)nd * NARROW_ROWS < (1 << 22), "%s: %d narrow dwords per row is too wide for the narrow kernel",
                fn, nd);
    magic = (uint32_t)(((1ull << 32) + nd - 1) / nd);
    narrow_blocks = (unsigned)trl_cdiv(B, NARROW_ROWS);
  }
  if (na.ncols > 0 && wa.ncols == 0) {
    if (range)
      hipLaunchKernelGGL(k_gather_narrow<true>, dim3(narrow_blocks), dim3(256), 0, st, (const uint32_t *)ring,
                         capacity, stride / 4, head, count, idx, start, B, na, nd, magic);
    else
      hipLaunchKernelGGL(k_gather_narrow<false>, dim3(narrow_blocks), dim3(256), 0, st, (const uint32_t *)ring,
                         capacity, stride / 4, head, count, idx, start, B, na, nd, magic);
    TRL_CHECK_LAUNCH(fn);
  }
  if (wa.ncols > 0) {
    // one launch for the whole batch: the first narrow_blocks workgroups copy the narrow columns
    for (int i = wa.ncols; i < TRL_MAX_COLS; ++i) wa.chunk_start[i + 1] = wa.chunk_start[wa.ncols];   // empty ranges
    const int64_t items = B * (int64_t)wa.chunk_start[wa.ncols];
    TRL_REQUIRE(items < (1ll << 31), "%s: %lld chunks exceed the limit; split the batch", fn, (long long)items);
    // persistent grid of 4 workgroups per CU over 16 KiB items (measured best on MI355X among 1-32 per CU
    // and 8 / 16 KiB items: tools/gather_sweep.py); a small batch is one item per workgroup
    int64_t cap_blocks = 1024;
#ifdef TRL_TUNING                                         // diagnostic build: tools/gather_sweep.py
    if (const char *t = getenv("TRL_GATHER_WG_PER_CU"))
      if (atoi(t) > 0) cap_blocks = (int64_t)256 * atoi(t);
#endif
    const unsigned grid = (unsigned)(items < cap_blocks ? items : cap_blocks) + narrow_blocks;
#define TRL_WIDE_LAUNCH(R, U)                                                                               \
  hipLaunchKernelGGL((k_gather_wide<R, U>), dim3(grid), dim3(256), 0, st, (const char *)ring, capacity, stride, \
                     head, count, idx, start, B, wa, na, nd, magic, narrow_blocks)
    if (range) {
      if (wide_un == 4) TRL_WIDE_LAUNCH(true, 4); else TRL_WIDE_LAUNCH(true, 2);
    } else {
      if (wide_un == 4) TRL_WIDE_LAUNCH(false, 4); else TRL_WIDE_LAUNCH(false, 2);
    }
#undef TRL_WIDE_LAUNCH
    TRL_CHECK_LAUNCH(fn);
  }
  return TRL_OK;
}

}  // namespace

// Indices drawn on the host reach the device through this copy kernel, which reads the caller's pinned
// (device-mapped) host buffer directly.  A hipMemcpyAsync of the same 2 KB is a DMA-engine job behind a
// cross-queue dependency on both sides: measured 175 us per learner step on MI355X in the middle of a
// stream that is otherwise ~180 us of kernels.
__global__ __launch_bounds__(256) void k_stage_indices(const int64_t *__restrict__ src, int64_t *__restrict__ dst,
                                                       int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

extern "C" int trl_stage_indices(const int64_t *src_pinned, int64_t *dst, int64_t n, void *stream) {
  const char *fn = "trl_stage_indices";
  TRL_REQUIRE(n >= 0, "%s: n < 0", fn);
  if (n == 0) return TRL_OK;
  TRL_REQUIRE(src_pinned && dst, "%s: null pointer", fn);
  TRL_REQUIRE(((uintptr_t)src_pinned % 8 == 0) && ((uintptr_t)dst % 8 == 0), "%s: pointers must be 8 B aligned", fn);
  const int64_t blocks = trl_cdiv(n, 256);
  hipLaunchKernelGGL(k_stage_indices, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0,
                     (hipStream_t)stream, src_pinned, dst, n);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

extern "C" int trl_gather_rows(const void *ring, int64_t capacity, int64_t stride,
                               const trl_ring_layout *layout, int64_t head, int64_t count,
                               const int64_t *idx, int64_t B, void *const *dst,
                               const int32_t *out_dtype, void *stream) {
  return gather_impl("trl_gather_rows", false, ring, capacity, stride, layout, head, count, idx, 0, B, dst,
                     out_dtype, stream);
}

extern "C" int trl_gather_range(const void *ring, int64_t capacity, int64_t stride,
                                const trl_ring_layout *layout, int64_t head, int64_t count,
                                int64_t start, int64_t B, void *const *dst, const int32_t *out_dtype,
                                void *stream) {
  return gather_impl("trl_gather_range", true, ring, capacity, stride, layout, head, count, nullptr, start,
                     B, dst, out_dtype, stream);
}

int trl_ring_rows_launch(const char *fn, const void *ring, int64_t capacity, int64_t stride, int64_t head, int64_t count,
                         const int64_t *idx, int64_t B, const int32_t *f32_col_off, int32_t *rec, float *action,
                         float *reward, float *done, hipStream_t st) {
  TRL_REQUIRE(ring && idx && rec && action && reward && done && f32_col_off, "%s: null pointer", fn);
  TRL_REQUIRE(capacity > 0 && capacity < (1ll << 31) && stride > 0 && stride % 4 == 0 && count > 0 && count <= capacity &&
              head >= 0 && head < capacity && B > 0, "%s: bad ring geometry", fn);
  for (int i = 0; i < 3; ++i)
    TRL_REQUIRE(f32_col_off[i] >= 0 && f32_col_off[i] % 4 == 0 && f32_col_off[i] + 4 <= stride, "%s: bad column offset", fn);
  const unsigned grid = (unsigned)(trl_cdiv(B, 256) < 64 ? trl_cdiv(B, 256) : 64);
  hipLaunchKernelGGL(k_ring_rows, dim3(grid), dim3(256), 0, st, (const char *)ring, capacity, stride, head, count, idx, B,
                     f32_col_off[0], f32_col_off[1], f32_col_off[2], rec, action, reward, done);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

extern "C" int trl_ring_push(void *ring, int64_t capacity, int64_t stride, const trl_ring_layout *layout,
                             int64_t tail, const void *const *src, int64_t k, void *stream) {
  const char *fn = "trl_ring_push";
  int rc = check_layout(fn, layout, stride);
  if (rc) return rc;
  TRL_REQUIRE(ring && src, "%s: null pointer", fn);
  TRL_REQUIRE(capacity > 0 && tail >= 0 && tail < capacity, "%s: need 0 <= tail < capacity", fn);
  TRL_REQUIRE(k >= 0 && k <= capacity, "%s: k=%lld must be in [0, capacity=%lld]", fn, (long long)k,
              (long long)capacity);
  if (k == 0) return TRL_OK;
  ColArgs a{};
  a.ncols = layout->ncols;
  const bool all16 = stride % 16 == 0 && (uintptr_t)ring % 16 == 0 && stride >= 256;
  for (int c = 0; c < layout->ncols; ++c) {
    TRL_REQUIRE(src[c] != nullptr, "%s: src[%d] is null", fn, c);
    TRL_REQUIRE(layout->col_bytes[c] % 4 != 0 || (uintptr_t)src[c] % 4 == 0,
                "%s: src[%d] must be 4-byte aligned (its rows are read as dwords)", fn, c);
    a.off_dw[c] = layout->col_off[c] / 4;
    a.w_bytes[c] = layout->col_bytes[c];
    a.sdt[c] = a.odt[c] = layout->col_dtype[c];
    a.ptr[c] = (void *)src[c];
  }
  hipStream_t st = (hipStream_t)stream;
  if (all16) {
    const int64_t total = k * (stride / 16);
    const unsigned grid = (unsigned)(trl_cdiv(total, 256) < 8192 ? trl_cdiv(total, 256) : 8192);
    hipLaunchKernelGGL(k_ring_push16, dim3(grid), dim3(256), 0, st, (uint4 *)ring, capacity, stride / 16,
                       tail, k, a);
  } else {
    const int64_t total = k * (stride / 4);
    const unsigned grid = (unsigned)(trl_cdiv(total, 256) < 8192 ? trl_cdiv(total, 256) : 8192);
    hipLaunchKernelGGL(k_ring_push, dim3(grid), dim3(256), 0, st, (uint32_t *)ring, capacity, stride / 4,
                       tail, k, a);
  }
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

// ------------------------------------------------------------------------------ device sampler
// B distinct indices uniform over [0, n), drawn on the device: the throughput-mode counterpart
// of `torch.randperm(n)[:B]` (examples/dqn.py:40).  NOT index-identical to torch (that is
// trl_randperm_prefix, on the host, O(n) per call): a counter-based generator (Philox-4x32-10
// keyed by the seed, counter = (call number, sample position, attempt)) gives every position its
// own stream; duplicates are resolved in rounds through an LDS hash set in which the earlier
// round, then the lower position, keeps an index and the other redraws -- so the result is a
// function of (seed, call number, n, B) alone, whatever the scheduling.  One workgroup.
namespace {

constexpr int SAMP_MAX_B = 4096;
constexpr int SAMP_SLOTS = 2 * SAMP_MAX_B;     // open addressing, load factor <= 0.5
constexpr uint32_t SAMP_EMPTY = 0xffffffffu;

__global__ __launch_bounds__(1024) void k_sample_distinct(int64_t n, int B, uint64_t seed,
                                                          unsigned long long *__restrict__ call_counter,
                                                          int64_t *__restrict__ out) {
  __shared__ uint32_t s_key[SAMP_SLOTS];     // the index held by a slot
  __shared__ uint32_t s_own[SAMP_SLOTS];     // (round << 12 | position) of its best claimant
  __shared__ int s_pending;
  const uint64_t call = *call_counter;
  for (int i = threadIdx.x; i < SAMP_SLOTS; i += 1024) { s_key[i] = SAMP_EMPTY; s_own[i] = SAMP_EMPTY; }
  __syncthreads();
  // thread t owns positions t, t + 1024, ... (at most 4)
  uint32_t cand[4], slot[4], attempt[4];
  bool pend[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { pend[j] = (int)threadIdx.x + 1024 * j < B; attempt[j] = 0; cand[j] = slot[j] = 0; }
  for (uint32_t round = 0; round < (1u << 19); ++round) {
    if (threadIdx.x == 0) s_pending = 0;
    __syncthreads();
    // claim: find (or make) the slot of the candidate, leave the best claimant in s_own
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!pend[j]) continue;
      const uint32_t pos = threadIdx.x + 1024 * j;
      const uint64_t u = philox64(seed, call, pos, attempt[j]);
      cand[j] = (uint32_t)__umul64hi(u, (uint64_t)n);           // uniform over [0, n), bias < n / 2^64
      uint32_t s = (cand[j] * 0x9E3779B1u) >> 19;                  // 13 bits -> SAMP_SLOTS
      for (;;) {
        const uint32_t old = atomicCAS(&s_key[s], SAMP_EMPTY, cand[j]);
        if (old == SAMP_EMPTY || old == cand[j]) break;
        s = (s + 1) & (SAMP_SLOTS - 1);
      }
      slot[j] = s;
      atomicMin(&s_own[s], (round << 12) | pos);
    }
    __syncthreads();
    // settle: the owner keeps the index, everyone else redraws next round
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!pend[j]) continue;
      const uint32_t pos = threadIdx.x + 1024 * j;
      if (s_own[slot[j]] == ((round << 12) | pos)) {
        out[pos] = (int64_t)cand[j];
        pend[j] = false;
      } else {
        ++attempt[j];
        atomicAdd(&s_pending, 1);
      }
    }
    __syncthreads();
    if (s_pending == 0) break;
    __syncthreads();
  }
  if (threadIdx.x == 0) *call_counter = call + 1;
}

}  // namespace

extern "C" int trl_sample_indices(int64_t n, int64_t B, uint64_t seed, uint64_t *call_counter_d, int64_t *idx_d,
                                  void *stream) {
  const char *fn = "trl_sample_indices";
  TRL_REQUIRE(call_counter_d && idx_d, "%s: null pointer", fn);
  TRL_REQUIRE((uintptr_t)call_counter_d % 8 == 0, "%s: the call counter must be 8-byte aligned", fn);
  TRL_REQUIRE(n >= 1 && n < (1ll << 32) - 1, "%s: n=%lld out of range", fn, (long long)n);
  TRL_REQUIRE(B >= 0 && B <= SAMP_MAX_B && B <= n, "%s: need 0 <= B <= min(n, %d), got B=%lld n=%lld", fn, SAMP_MAX_B,
              (long long)B, (long long)n);
  if (B == 0) return TRL_OK;
  hipLaunchKernelGGL(k_sample_distinct, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, (int)B, seed,
                     (unsigned long long *)call_counter_d, idx_d);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}
