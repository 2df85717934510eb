// GEMM kernels of the feature path for gfx950 (MI355X), all on matrix cores with fused bias +
// activation epilogue and deterministic column sums:
//   * k_gemm_f32: 128-row tiles, weight K tiles double-buffered in LDS (any K, row-mapped output);
//   * k_gemm_resident: the whole weight matrix in LDS, persistent waves, optional message-summing
//     last segment, float32 or bf16 operands;
//   * k_gemm_split: float32 products from bf16 matrix cores with split operands.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <type_traits>

#include "gemm.h"

namespace gunet {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kThreads = 256;
// A lane offset beyond every buffer: the launcher keeps operands below 2 GiB, and adding a tile's
// chunk and K offsets (a few KiB) to it cannot wrap around to an address inside one.
constexpr uint32_t kOutOfBuffer = 0x80000000u;


__device__ inline float activate(float v, int relu) { return relu ? fmaxf(v, 0.0f) : v; }

// The A operand goes straight from global memory into MFMA registers: lane (li, lh) of wave w
// owns row 32w+li of the tile and reads floats 8c+4lh..8c+4lh+3 of every 8-float K chunk of that
// row, so one 128-bit load feeds four v_mfma_f32_32x32x2_f32 (the two half-waves supply the two
// k of each MFMA; the k order inside a chunk is a fixed permutation, the same for the weights).
// Only the weight tile goes through LDS (double-buffered, one barrier per K tile); the next A
// tile is prefetched in registers.  Epilogue: bias, activation, optional row scatter, and the
// column sums: per wave the float32 sum of its 32 rows (16 rows per lane in a fixed order, then the
// two half-waves), added in float64 over the workgroup's row tiles and then over its four waves.
// A workgroup walks the row tiles wg * tpw .. wg * tpw + tpw - 1 of problem `g` (tpw > 1 only to
// keep the number of column-sum partial rows small) and writes partial row `wg`.  `ct` shifts the
// launch's columns by 32 * ct: small launches are cut into NT = 1 column slices (grid z) so that
// four times as many waves share the matrix work.
// ANY_ACT = true: the activation is g.act, any of the GUNET_ACT_* codes (an instantiation of its own,
// so that the transcendental functions cost the linear / relu form neither registers nor code).
template <int NT, bool ANY_ACT = false>
__device__ __forceinline__ void gemm_f32_tiles(const GemmArgs& g, const int wg, const int tpw, const int ct) {
  constexpr int BN = 32 * NT;
  __shared__ __attribute__((aligned(16))) float Bs[2][kBK * BN];
  __shared__ __attribute__((aligned(16))) float Stage[4][16 * 32];   // epilogue: one per wave
  __shared__ double Csum[4][BN];   // per wave: column sums over the workgroup's row tiles (kept out of the registers:
                                   // three workgroups per compute unit need the kernel below 170 of them)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // a scalar, like all that follows from it
  const int li = lane & 31, lh = lane >> 5;
  const int64_t b = blockIdx.y;
  const int col0 = g.out_col0 + 32 * ct;

  int n_tiles = 0;
  for (int s = 0; s < g.n_seg; ++s) n_tiles += (g.seg[s].k + kBK - 1) / kBK;

  float bias_v[NT];   // loaded up front: a load inside the epilogue would drain its stores
  const bool want_cs = g.colsum != nullptr;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if (want_cs && lh == 0) Csum[wave][32 * n + li] = 0.0;
    // NT > 1 launches are never sliced (ct = 0): constant indices into the argument arrays
    const int tile = NT == 1 ? ct : n;
    if (g.bias_by_tile) {
      const float* bt = NT > 1 ? g.bias_tile[n]
                               : (tile == 0 ? g.bias_tile[0] : (tile == 1 ? g.bias_tile[1] : (tile == 2 ? g.bias_tile[2] : g.bias_tile[3])));
      bias_v[n] = bt ? bt[li] : 0.0f;
    } else {
      bias_v[n] = g.bias ? g.bias[b * g.bias_stride + 32 * tile + li] : 0.0f;
    }
  }

  const int row_tiles = (g.n_rows + kBM - 1) / kBM;
  const int bx_end = min((wg + 1) * tpw, row_tiles);
  for (int bx = wg * tpw; bx < bx_end; ++bx) {
    const int row0 = bx * kBM;
    const int my_row = row0 + 32 * wave + li;
    const bool row_ok = my_row < g.n_rows;

    float4 b_reg[NT];
    // this lane's operand row in each segment (a gathered segment looks it up once per row tile)
    int a_row[3];   // within the sample
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      a_row[s] = my_row;
      if (s < g.n_seg && g.seg[s].row_idx != nullptr)
        a_row[s] = row_ok ? g.seg[s].row_idx[static_cast<int64_t>(my_row) * g.seg[s].idx_stride] : 0;
    }
    // Request cursor: the K tile being fetched is columns ck0 .. ck0+31 of segment cseg.  The
    // segment's description is read from the argument block once, when the cursor enters it; from
    // then on a tile's addresses are a scalar base, a scalar offset that advances with ck0, and one
    // per-lane offset.  Rows past the end of the sample or of the weight matrix sit beyond the
    // buffers' sizes and read as zeros.
    int cseg = -1, ck0 = 0, seg_k = 0;
    uint32_t a_lane = 0, w_lane = 0, w_step = 0;
    const float* a_base = nullptr;   // (plain scalars; the buffer resources are formed from them at each request)
    const float* w_base[NT];
    uint32_t a_bytes = 0, w_bytes = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) w_base[j] = nullptr;
    auto enter = [&](int sx) {
      const GemmSeg& sg = g.seg[sx];
      cseg = sx;
      ck0 = 0;
      seg_k = sg.k;
      const int64_t sample_rows = sg.shared ? 0 : (sg.row_idx != nullptr ? sg.a_rows : g.n_rows);
      const int64_t rows_here = sg.row_idx != nullptr ? sg.a_rows : g.n_rows;
      a_base = sg.a + b * sample_rows * static_cast<int64_t>(sg.lda);
      a_bytes = static_cast<uint32_t>(rows_here) * static_cast<uint32_t>(sg.lda) * 4u;
      const int arow = sx == 0 ? a_row[0] : (sx == 1 ? a_row[1] : a_row[2]);
      a_lane = row_ok ? (static_cast<uint32_t>(arow) * static_cast<uint32_t>(sg.lda) + 4u * lh) * 4u : kOutOfBuffer;
      w_step = static_cast<uint32_t>(sg.ldw) * 4u;
      w_lane = static_cast<uint32_t>(tid >> 3) * w_step + 16u * (tid & 7);
      w_bytes = static_cast<uint32_t>(sg.k - 1) * w_step + 128u;
#pragma unroll
      for (int j = 0; j < NT; ++j)
        w_base[j] = NT > 1 ? sg.w[j] : (ct == 0 ? sg.w[0] : (ct == 1 ? sg.w[1] : (ct == 2 ? sg.w[2] : sg.w[3])));
    };
    auto advance = [&]() {   // to the next tile; the caller stops after n_tiles
      ck0 += kBK;
      if (ck0 >= seg_k && cseg + 1 < g.n_seg) enter(cseg + 1);
    };
    auto load_a = [&](float4 (&a)[4]) {
      const BufferRsrc a_rs = buffer_of(a_base, a_bytes);
      if (ck0 + kBK <= seg_k) {   // uniform
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = buffer_load_f4(a_rs, a_lane + 32u * c, 4u * ck0);
      } else {   // the segment ends inside this tile (k is a multiple of 4)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          a[c] = buffer_load_f4(a_rs, ck0 + 8 * c + 4 * lh < seg_k ? a_lane + 32u * c : kOutOfBuffer, 4u * ck0);
      }
    };
    // thread (kr = tid / 8, c4 = tid % 8) stages row ck0+kr, floats 4c4..4c4+3 of every column tile
    auto load_b = [&]() {
#pragma unroll
      for (int j = 0; j < NT; ++j)
        // the row offset rides in the range-checked lane offset (a scalar offset is not checked):
        // weight rows past k read as zeros whatever lies behind them in memory
        b_reg[j] = buffer_load_f4(buffer_of(w_base[j], w_bytes), w_lane + static_cast<uint32_t>(ck0) * w_step, 0u);
    };
    auto store_b = [&](int buf) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
        *reinterpret_cast<float4*>(&Bs[buf][(tid >> 3) * BN + 32 * j + 4 * (tid & 7)]) = b_reg[j];
    };

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

    float4 a_cur[4], a_nxt[4];
    enter(0);
    load_a(a_cur);
    load_b();
    store_b(0);
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
      const bool more = t + 1 < n_tiles;
      if (more) {
        advance();
        load_b();
        load_a(a_nxt);
      }
      const float* bs = Bs[t & 1];
#pragma unroll
      for (int c = 0; c < kBK / 8; ++c) {
        const float a4[4] = {a_cur[c].x, a_cur[c].y, a_cur[c].z, a_cur[c].w};
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float bv = bs[(8 * c + 4 * lh + j) * BN + 32 * n + li];
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], bv, acc[n], 0, 0, 0);
          }
        }
      }
      if (more) store_b((t + 1) & 1);
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 4; ++c) a_cur[c] = a_nxt[c];
    }

    // accumulator register r of lane (li, lh) is row (r&3) + 8*(r>>2) + 4*lh of the wave, column li
    float cs[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) cs[n] = 0.0f;
    const int wrow0 = row0 + 32 * wave;
    {
      // 16 rows x 32 columns at a time through the wave's own LDS, picked up row-wise: a store
      // instruction then writes eight full 128-byte lines instead of two.  With a row map the
      // pick-up lane looks its output row up (fetched here, ahead of the stores).
      float* stage = Stage[wave];
      const int srow = lane >> 3, scol = 4 * (lane & 7);
      int orow[4];   // output rows of this lane's four pick-up rows: 16h + 8i + srow
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wrow0 + 8 * j + srow;
        orow[j] = row;
        if (g.row_map != nullptr && row < g.n_rows)
          orow[j] = g.row_map[static_cast<int64_t>(row) * g.row_map_stride] - g.row_map_sub;
      }
      float* obase = g.out ? g.out + b * g.out_rows * static_cast<int64_t>(g.ldo) + col0 + scol : nullptr;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int r8 = 0; r8 < 8; ++r8) {
            const int r = 8 * h + r8;
            const int rl = (r & 3) + 8 * (r8 >> 2) + 4 * lh;
            const float v = ANY_ACT ? activate_any(acc[n][r] + bias_v[n], g.act) : activate(acc[n][r] + bias_v[n], g.relu);
            stage[rl * 32 + li] = v;
            if (wrow0 + 16 * h + rl < g.n_rows) cs[n] += v;
          }
          if (obase) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const float4 v = *reinterpret_cast<const float4*>(&stage[(8 * i + srow) * 32 + scol]);
              if (wrow0 + 16 * h + 8 * i + srow < g.n_rows)
                *reinterpret_cast<float4*>(obase + static_cast<int64_t>(orow[2 * h + i]) * g.ldo + 32 * n) = v;
            }
          }
        }
      }
    }
    if (want_cs) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        cs[n] += __shfl_xor(cs[n], 32);
        if (lh == 0) Csum[wave][32 * n + li] += static_cast<double>(cs[n]);   // only this lane touches the entry
      }
    }
  }
  if (want_cs) {   // one partial row per workgroup: the four waves' sums added in order
    __syncthreads();
    if (tid < BN) {
      const double s = ((Csum[0][tid] + Csum[1][tid]) + Csum[2][tid]) + Csum[3][tid];
      g.colsum[(b * g.colsum_tiles + g.colsum_tile0 + wg) * g.colsum_ld + col0 + tid] = s;
    }
  }
}

template <int NT, bool ANY_ACT = false>
__global__ __launch_bounds__(kThreads, 3) void k_gemm_f32(GemmArgs g, int tpw) {
  gemm_f32_tiles<NT, ANY_ACT>(g, blockIdx.x, tpw, blockIdx.z);
}

// Up to four independent problems of the same column-tile count in one grid (the four quadrant
// GEMMs of the upsampling layer): consecutive block ranges serve the problems.
constexpr int kGemmGroup = 4;
struct GemmGroup {
  GemmArgs p[kGemmGroup];
  unsigned block_end[kGemmGroup];   // running total of the problems' workgroups
  int tpw[kGemmGroup];              // row tiles per workgroup of each problem
};

template <int NT>
__global__ __launch_bounds__(kThreads, 3) void k_gemm_f32_group(GemmGroup gg) {
  int i = 0;
  while (i < kGemmGroup - 1 && blockIdx.x >= gg.block_end[i]) ++i;
  gemm_f32_tiles<NT>(gg.p[i], static_cast<int>(blockIdx.x - (i > 0 ? gg.block_end[i - 1] : 0u)), gg.tpw[i], blockIdx.z);
}

// Weights-resident variant (the weight matrix must fit LDS: 32 * NKT * 32 * NT floats): a workgroup
// of twelve waves (eight for the 128-column gathering form) copies the whole weight matrix into
// LDS once, then its waves work on their own, with no barrier between the first and the one
// that closes the kernel: every sample's 32-row groups are dealt out in runs of `gpw` consecutive
// groups, one run per wave, `bps` workgroups per sample (a workgroup never straddles two samples).  The A tile of step t+1 -- the next group's first tile at a
// group's last step -- is requested before the MFMAs of step t, into the other of two register
// buffers, so a wave always has one tile in flight behind its matrix work and the waves of a SIMD
// drift apart instead of stalling together.  Rows past the end of a sample are clamped for the
// loads and masked in the epilogue.  Column sums: one partial row per workgroup (its waves' runs,
// added in wave order in float64 after the closing barrier).
constexpr int kResWaves = 12;
constexpr int kResThreads = 64 * kResWaves;
constexpr int kResGatherWaves = 8;   // the 128-column gathering variant needs more registers per wave
constexpr int kResGatherThreads = 64 * kResGatherWaves;
constexpr size_t kResMaxLds = 160 * 1024;   // most LDS a workgroup may ask for on gfx950; see device_lds_limit()

// BF16 = true: operands are rounded to bf16 on their way into v_mfma_f32_32x32x16_bf16 (float32
// accumulate and storage); a lane then owns, per 16-wide k step, 8 consecutive floats of its row
// (k = 16s + 8 * lh + 0..7), and the LDS weight image holds one bf16x8 per lane, k step and column
// tile.  The matrix work shrinks 16-fold, so this form runs at the speed of its memory traffic.
// RAGGED = true (one operand segment, float32): k is not a whole number of K tiles; the weight rows
// past k are zero in the LDS image and the last tile's loads past k are redirected to the row's
// first piece of that tile and zeroed.
// Where each K tile read from the operand segments starts (sample 0, row 0) and its row stride.
constexpr int kResMaxTiles = 12;
struct ResidentTiles {
  const float* a[kResMaxTiles];
  int lda[kResMaxTiles];
  uint32_t gather_bytes[4];   // size of each gathered source tensor (all samples)
};

// A16 = true (with BF16): the operand segments and the gathered source tensors are stored as bf16; a
// lane's 8 consecutive k of a 16-wide k step are then one 16-byte load that is the MFMA operand as
// it stands.  O16 = true: the output is stored as bf16 (rounded once, from the float32 result).
template <int NT, int NKT, bool GATHER, bool BF16, bool RAGGED = false, bool A16 = false, bool O16 = false>
__global__ __launch_bounds__((GATHER && NT == 4) ? kResGatherThreads : kResThreads) void k_gemm_resident(
    GemmArgs g, ResidentTiles tl, int groups_per_sample, int bps, int gpw) {
  static_assert(!GATHER || (NKT >= 6 && NKT % 2 == 0), "the gathered segment is the last four K tiles");
  static_assert(!RAGGED || (!GATHER && !A16), "the ragged form takes one float32 operand segment");
  static_assert(!A16 || BF16, "bf16 operands go to the bf16 matrix instruction");
  constexpr int kWaves = (GATHER && NT == 4) ? kResGatherWaves : kResWaves;
  constexpr int NX = GATHER ? NKT - 4 : NKT;   // K tiles read from the operand segments
  constexpr int BN = 32 * NT;
  extern __shared__ __attribute__((aligned(16))) float res_smem[];
  // Weight image: for K tile kt, 8-row chunk c and column tile n, one 1 KiB piece in lane order:
  // lane (lh, li) finds rows 32kt + 8c + 4lh + 0..3 of column 32n + li as one float4.
  float* Ws = res_smem;   // [NKT][4][NT][64][4]
  const int tid = threadIdx.x;
  // the wave number as a scalar: everything derived from it (the run of groups, the loop around
  // them, base addresses) then stays in scalar registers
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;

  // Paired launch: blocks i and i + 8 -- same XCD under the round-robin dispatch -- take the two
  // 32*NT-column halves of the same rows (when the pair count is no multiple of 8, the halves are
  // simply the two halves of the grid).
  int block = blockIdx.x, half = 0;
  if (g.paired) {
    const int pairs = gridDim.x >> 1;
    if ((pairs & 7) == 0) {
      half = (block >> 3) & 1;
      block = ((block >> 4) << 3) | (block & 7);
    } else {
      half = block >= pairs;
      block -= half * pairs;
    }
  }
  const int col_shift = BN * half;

  {
    int kbase = 0;
    for (int s = 0; s < g.n_seg; ++s) {
      const GemmSeg& sg = g.seg[s];
      const float wscale = (GATHER && g.gather_fold && s == g.n_seg - 1) ? g.gather_scale : 1.0f;   // exact: a power of two
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float* w = (NT == 2 && half) ? sg.w[2 + t] : sg.w[t];
        if constexpr (BF16) {
          // thread -> (group of 8 k rows, column): 8 strided loads, one 16-byte LDS store
          uint4* img = reinterpret_cast<uint4*>(Ws);
          const int k_img = RAGGED ? kBK * NKT : sg.k;   // ragged: the image's rows past k are zeros
          for (int i = tid; i < k_img * 4; i += kWaves * 64) {
            const int kg = i >> 5, col = i & 31;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j)
              v[j] = static_cast<__bf16>((!RAGGED || 8 * kg + j < sg.k) ? wscale * w[static_cast<int64_t>(8 * kg + j) * sg.ldw + col]
                                                                         : 0.0f);
            const int k0 = kbase + 8 * kg;   // = 16 * kstep + 8 * half
            img[((k0 >> 4) * NT + t) * 64 + ((k0 >> 3) & 1) * 32 + col] = __builtin_bit_cast(uint4, v);
          }
        } else {
          const int k_fill = RAGGED ? kBK * NKT : sg.k;
          for (int i = tid; i < k_fill * 8; i += kWaves * 64) {
            const int row = kbase + (i >> 3), c4 = i & 7;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!RAGGED || (i >> 3) < sg.k)
              v = *reinterpret_cast<const float4*>(w + static_cast<int64_t>(i >> 3) * sg.ldw + 4 * c4);
            // row = 8 * chunk + 4 * half + j
            float* dst = Ws + (((row >> 3) * NT + t) * 64 + ((row >> 2) & 1) * 32 + 4 * c4) * 4 + (row & 3);
            dst[0] = wscale * v.x;
            dst[4] = wscale * v.y;
            dst[8] = wscale * v.z;
            dst[12] = wscale * v.w;
          }
        }
      }
      kbase += sg.k;
    }
  }
  __syncthreads();

  const int b = block / bps;
  const int blk = block - b * bps;
  int grp = (blk * kWaves + wave) * gpw;
  const int grp_end = min(grp + gpw, groups_per_sample);
  const bool has_work = grp < grp_end;   // the last workgroup of a sample may have idle waves

  // this lane's (clamped) row of group grp_, counted over all samples
  auto lane_row = [&](int grp_) {
    int row = grp_ * kRowGroup + li;
    row = row < g.n_rows ? row : g.n_rows - 1;
    return static_cast<uint32_t>(b) * static_cast<uint32_t>(g.n_rows) + static_cast<uint32_t>(row);
  };
  // The four 16-byte pieces a lane reads of a 32-float K tile: floats 8c + 4lh + 0..3 (c = 0..3)
  // for the float32 MFMA, floats 16s + 8lh + 4i + 0..3 (s, i = 0, 1) for the bf16 one.
  constexpr uint32_t kLaneFloats = BF16 ? 8u : 4u;
  auto chunk_bytes = [](int c) -> uint32_t { return BF16 ? 64u * (c >> 1) + 16u * (c & 1) : 32u * c; };
  // uniform base + 32-bit per-lane byte offset (the host checks that every operand segment and the
  // output are smaller than 4 GiB)
  auto load_tile = [&](uint32_t arow, int kt, float4 (&a)[4]) {
    // kt is a constant after unrolling: the tile's operand address and row stride are plain fields
    // of the argument block
    const BufferRsrc rs = buffer_of(tl.a[kt]);
    if constexpr (A16) {   // 32 bf16 of the row: this lane's 8 k of either k step are 16 bytes
      const uint32_t voff16 = arow * static_cast<uint32_t>(tl.lda[kt]) * 2u + 16u * static_cast<uint32_t>(lh);
      const int at = 2 * (kt & 1);   // a buffer holds two K tiles (see load_step)
      a[at] = buffer_load_f4(rs, voff16, 0u);
      a[at + 1] = buffer_load_f4(rs, voff16 + 32u, 0u);
      return;
    }
    const uint32_t voff = (arow * static_cast<uint32_t>(tl.lda[kt]) + kLaneFloats * static_cast<uint32_t>(lh)) * 4u;
    if (RAGGED && kt == NKT - 1) {
      const int left = g.seg[0].k - kBK * (NKT - 1) - static_cast<int>(kLaneFloats) * lh;   // floats of this lane's k phase left in the tile
      const uint32_t first = arow * static_cast<uint32_t>(tl.lda[kt]) * 4u;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // piece c starts 8c floats into the lane's k phase (float32 MFMA) or 16 (c / 2) + 4 (c % 2) (bf16)
        const bool ok = (BF16 ? 16 * (c >> 1) + 4 * (c & 1) : 8 * c) < left;
        const float4 v = buffer_load_f4(rs, ok ? voff + chunk_bytes(c) : first, 0u);
        a[c] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
      }
      return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = buffer_load_f4(rs, voff + chunk_bytes(c), 0u);
  };

  f32x16 acc[NT];
  const float4* wl = reinterpret_cast<const float4*>(Ws) + lane;
  auto multiply = [&](int kt, const float4 (&a)[4], int at = 0) {
    (void)at;
    if constexpr (BF16) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 ap;
        if constexpr (A16) {
          ap = __builtin_bit_cast(bf16x8, a[at + s2]);
        } else {
          const float av[8] = {a[2 * s2].x,     a[2 * s2].y,     a[2 * s2].z,     a[2 * s2].w,
                               a[2 * s2 + 1].x, a[2 * s2 + 1].y, a[2 * s2 + 1].z, a[2 * s2 + 1].w};
#pragma unroll
          for (int j = 0; j < 8; ++j) ap[j] = static_cast<__bf16>(av[j]);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const bf16x8 w = __builtin_bit_cast(bf16x8, wl[((2 * kt + s2) * NT + n) * 64]);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap, w, acc[n], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // ragged form: 8-float pieces of the last tile that lie wholly past k are zero -- skip them
        if (RAGGED && kt == NKT - 1 && 8 * c >= g.seg[0].k - kBK * (NKT - 1)) continue;
        float4 w[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) w[n] = wl[((kt * 4 + c) * NT + n) * 64];
        // the NT accumulators take turns, so no MFMA waits for the one before it
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].x, w[n].x, acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].y, w[n].y, acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].z, w[n].z, acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].w, w[n].w, acc[n], 0, 0, 0);
      }
    }
  };
  // Epilogue: bias, activation, column sums, and the tile's way to memory.  A lane holds one column
  // of 16 rows; storing that directly would take 64 one-dword store instructions per group.  The
  // wave instead passes 16 rows x 32 columns at a time through its own 2 KiB of LDS and picks them
  // up row-wise, so a store instruction writes eight full 128-byte lines (four floats per lane).
  const bool has_out = g.out != nullptr;
  const uint32_t ldo = static_cast<uint32_t>(g.ldo);
  const BufferRsrc out_rs = buffer_of(g.out);
  float* stage = res_smem + (BF16 ? 16 : 32) * NKT * BN + wave * (16 * 32);
  // float32 output: a lane picks up 4 columns of rows srow and 8 + srow of the staged 16; bf16
  // output: 8 columns of one row (16 bytes either way)
  const int srow = O16 ? lane >> 2 : lane >> 3, scol = O16 ? 8 * (lane & 3) : 4 * (lane & 7);
  constexpr uint32_t kOutElem = O16 ? 2u : 4u;
  // this lane's pick-up position in the sample's first group, as a byte offset into the output
  const uint32_t out_lane_off =
      ((static_cast<uint32_t>(b) * static_cast<uint32_t>(g.out_rows) + static_cast<uint32_t>(srow)) * ldo +
       static_cast<uint32_t>(g.out_col0 + col_shift + scol)) * kOutElem;
  // Column sums: the float32 sum of each group's 32 rows (16 rows per lane in a fixed order, then
  // the two half-waves), added over the wave's run of groups in float64.
  float bias_v[NT];
  double cs[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if (g.bias_by_tile)
      bias_v[n] = g.bias_tile[n] ? g.bias_tile[n][li] : 0.0f;
    else
      bias_v[n] = g.bias ? g.bias[static_cast<int64_t>(b) * g.bias_stride + col_shift + 32 * n + li] : 0.0f;
    cs[n] = 0.0;
  }
  const float bias_one = lh == 0 ? 1.0f : 0.0f;
  float bias_row[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) bias_row[n] = lh == 0 ? bias_v[n] : 0.0f;
  const bool want_cs = g.colsum != nullptr;
  // activation: one bare maximum against 0 (ReLU), or nothing at all (linear: a NaN stays a NaN)
  auto epilogue = [&](auto full_tag, auto relu_tag, int row0) {
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr bool RELU = decltype(relu_tag)::value;
    const uint32_t group_off = static_cast<uint32_t>(row0) * ldo * kOutElem;   // uniform
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      float gs = 0.0f;   // this lane's 16 rows of column 32n + li
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
          const int r = 8 * h + r8;
          const int rl = (r & 3) + 8 * (r8 >> 2) + 4 * lh;   // row within this half of the tile
          const float v = RELU ? max_bare(acc[n][r], 0.0f) : acc[n][r];
          stage[rl * 32 + li] = v;
          if (FULL || row0 + 16 * h + rl < g.n_rows) gs += v;
        }
        if (has_out) {
          if constexpr (O16) {
            const float4 v0 = lds_load_f4(&stage[srow * 32 + scol]);
            const float4 v1 = lds_load_f4(&stage[srow * 32 + scol + 4]);
            bf16x8 pk;
            pk[0] = static_cast<__bf16>(v0.x); pk[1] = static_cast<__bf16>(v0.y);
            pk[2] = static_cast<__bf16>(v0.z); pk[3] = static_cast<__bf16>(v0.w);
            pk[4] = static_cast<__bf16>(v1.x); pk[5] = static_cast<__bf16>(v1.y);
            pk[6] = static_cast<__bf16>(v1.z); pk[7] = static_cast<__bf16>(v1.w);
            if (FULL || row0 + 16 * h + srow < g.n_rows)
              buffer_store_f4(out_rs, out_lane_off, group_off + static_cast<uint32_t>(16 * h) * ldo * 2u + 64u * n,
                              __builtin_bit_cast(float4, pk));
          } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const float4 v = *reinterpret_cast<const float4*>(&stage[(8 * i + srow) * 32 + scol]);
              if (FULL || row0 + 16 * h + 8 * i + srow < g.n_rows)
                buffer_store_f4(out_rs, out_lane_off, group_off + static_cast<uint32_t>(16 * h + 8 * i) * ldo * 4u + 128u * n, v);
            }
          }
        }
      }
      if (want_cs) {
        gs += __shfl_xor(gs, 32);
        cs[n] += static_cast<double>(gs);
      }
    }
  };

  // Gathered tiles: the node's table row (12 edge positions) is fetched at the top of its group; the
  // (up to) three source rows of quadrant q are requested one K tile ahead -- the first two into
  // `gbuf`, the third into the operand buffer a0, which is idle during these tiles -- and summed
  // into a1 right before the MFMAs of that tile.
  int4 tab[3];
  float4 gbuf[2][4];
  const BufferRsrc tab_rs = buffer_of(g.gather_table);
  auto load_table = [&](int grp_) {
    int row = grp_ * kRowGroup + li;
    row = row < g.n_rows ? row : g.n_rows - 1;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      tab[i] = __builtin_bit_cast(int4, buffer_load_f4(tab_rs, static_cast<uint32_t>(row) * 48u + 16u * i, 0u));
  };
  auto tab_at = [&](int i) {   // i = 3 * q + j, compile-time after unrolling
    const int4 t = tab[i >> 2];
    return (i & 3) == 0 ? t.x : ((i & 3) == 1 ? t.y : ((i & 3) == 2 ? t.z : t.w));
  };
  // A missing table entry (-1) is read at the end of the source tensor, which the buffer's range
  // check answers with zeros: the sum below then needs no per-entry selection.
  auto gather_issue = [&](int q, float4 (&third)[4]) {
    const uint32_t bytes = tl.gather_bytes[q];
    const BufferRsrc rs = buffer_of(g.gather_src[q], bytes);
    const uint32_t base = static_cast<uint32_t>(b) * static_cast<uint32_t>(g.gather_rows[q]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int id = tab_at(3 * q + j);
      if constexpr (A16) {   // a source row is 32 bf16: 16 bytes per k step and lane
        const uint32_t voff = id >= 0 ? (base + static_cast<uint32_t>(id)) * 64u + 16u * static_cast<uint32_t>(lh) : bytes;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float4 v = buffer_load_f4(rs, voff + 32u * c, 0u);
          if (j < 2)
            gbuf[j][c] = v;
          else
            third[c] = v;
        }
      } else {
        const uint32_t voff = id >= 0 ? ((base + static_cast<uint32_t>(id)) * 32u + kLaneFloats * static_cast<uint32_t>(lh)) * 4u : bytes;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 v = buffer_load_f4(rs, voff + chunk_bytes(c), 0u);
          if (j < 2)
            gbuf[j][c] = v;
          else
            third[c] = v;
        }
      }
    }
  };
  // (m0 + m1) + m2 in table order; the factor gather_scale is already in the weights when it is a
  // power of two (gather_fold: scaling by it commutes exactly with every product)
  auto gather_reduce = [&](int q, const float4 (&third)[4], float4 (&a)[4]) {
    (void)q;
    if constexpr (A16) {   // bf16 rows: widen, add in float32 in table order, round the sum once
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const u32x4 m0 = __builtin_bit_cast(u32x4, gbuf[0][c]), m1 = __builtin_bit_cast(u32x4, gbuf[1][c]),
                    m2 = __builtin_bit_cast(u32x4, third[c]);
        bf16x8 pk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float lo = (__builtin_bit_cast(float, m0[i] << 16) + __builtin_bit_cast(float, m1[i] << 16)) +
                     __builtin_bit_cast(float, m2[i] << 16);
          float hi = (__builtin_bit_cast(float, m0[i] & 0xFFFF0000u) + __builtin_bit_cast(float, m1[i] & 0xFFFF0000u)) +
                     __builtin_bit_cast(float, m2[i] & 0xFFFF0000u);
          if (!g.gather_fold) {
            lo *= g.gather_scale;
            hi *= g.gather_scale;
          }
          pk[2 * i] = static_cast<__bf16>(lo);
          pk[2 * i + 1] = static_cast<__bf16>(hi);
        }
        a[c] = __builtin_bit_cast(float4, pk);
      }
      return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 s = gbuf[0][c];
      s.x += gbuf[1][c].x; s.y += gbuf[1][c].y; s.z += gbuf[1][c].z; s.w += gbuf[1][c].w;
      s.x += third[c].x; s.y += third[c].y; s.z += third[c].z; s.w += third[c].w;
      a[c] = s;
    }
    if (!g.gather_fold) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        a[c] = make_float4(a[c].x * g.gather_scale, a[c].y * g.gather_scale, a[c].z * g.gather_scale, a[c].w * g.gather_scale);
    }
  };

  // A step of the K loop is one K tile, or two with bf16 operands: their tiles are half the bytes and
  // their matrix work a sixteenth, so a wave must keep twice the K range in flight to cover the
  // memory latency (the kernel then runs at the speed of its loads).
  constexpr int TPB = A16 ? 2 : 1;
  auto load_step = [&](uint32_t arow, int st, float4 (&a)[4]) {
#pragma unroll
    for (int j = 0; j < TPB; ++j) load_tile(arow, st * TPB + j, a);
  };
  auto multiply_step = [&](int st, const float4 (&a)[4]) {
#pragma unroll
    for (int j = 0; j < TPB; ++j) multiply(st * TPB + j, a, 2 * j);
  };
  constexpr int NXS = NX / TPB, NKS = NKT / TPB;   // steps over the operand segments / over all of K
  static_assert(NX % (2 * TPB) == 0, "the K loop walks the operand tiles two steps at a time");

  uint32_t cur = lane_row(has_work ? grp : 0);
  float4 a0[4], a1[4];
  if (has_work) load_step(cur, 0, a0);
  while (has_work) {
    const bool has_next = grp + 1 < grp_end;
    const uint32_t nxt = lane_row(has_next ? grp + 1 : grp);   // a finished wave re-reads its own rows: harmless
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;   // (an inline constant of the first MFMAs, no instructions)
    if constexpr (GATHER) {
      load_table(grp);
#pragma unroll
      for (int st = 0; st < NXS; st += 2) {
        load_step(cur, st + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
        multiply_step(st, a0);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 2 < NXS)
          load_step(cur, st + 2, a0);
        else
          gather_issue(0, a0);
        __builtin_amdgcn_sched_barrier(0);
        multiply_step(st + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        gather_reduce(q, a0, a1);
        __builtin_amdgcn_sched_barrier(0);
        if (q < 3)
          gather_issue(q + 1, a0);
        else
          load_step(nxt, 0, a0);
        __builtin_amdgcn_sched_barrier(0);
        multiply(NX + q, a1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int st = 0; st < NKS; st += 2) {
        load_step(cur, st + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
        multiply_step(st, a0);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 2 < NKS)
          load_step(cur, st + 2, a0);
        else
          load_step(nxt, 0, a0);
        __builtin_amdgcn_sched_barrier(0);
        multiply_step(st + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // Take delivery of the next group's first tile now, while only loads are outstanding: a wait
    // placed after the stores below would have to cover them as well (loads and stores share one
    // counter), and the wave would sit out its own write latency at the top of every group.
    asm volatile("" : "+v"(a0[0].x), "+v"(a0[1].x), "+v"(a0[2].x), "+v"(a0[3].x));
    // The bias, added last as in the other GEMM kernels, by the matrix cores: one more product step
    // whose operands are a column of ones and the bias row (k = 0; zeros at k = 1), which adds
    // 1 * bias to every row of the tile in one rounding -- 4 instructions instead of 64 additions.
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_one, bias_row[n], acc[n], 0, 0, 0);
    const int row0 = grp * kRowGroup;
    if (g.relu) {
      if (row0 + kRowGroup <= g.n_rows)
        epilogue(std::true_type{}, std::true_type{}, row0);
      else
        epilogue(std::false_type{}, std::true_type{}, row0);
    } else {
      if (row0 + kRowGroup <= g.n_rows)
        epilogue(std::true_type{}, std::false_type{}, row0);
      else
        epilogue(std::false_type{}, std::false_type{}, row0);
    }
    if (!has_next) break;
    ++grp;
    cur = nxt;
  }
  if (g.colsum != nullptr) {   // uniform over the grid
    // every wave leaves its run's sums in its own (now idle) staging area; thread c then adds the
    // waves' values of column c in wave order
    double* red = reinterpret_cast<double*>(res_smem + (BF16 ? 16 : 32) * NKT * BN);
#pragma unroll
    for (int n = 0; n < NT; ++n)
      if (lh == 0) red[wave * 256 + 32 * n + li] = cs[n];
    __syncthreads();
    if (tid < BN) {
      double s = red[tid];
      for (int w = 1; w < kWaves; ++w) s += red[w * 256 + tid];
      g.colsum[(static_cast<int64_t>(b) * g.colsum_tiles + g.colsum_tile0 + blk) * g.colsum_ld + g.out_col0 + col_shift + tid] = s;
    }
  }
}

// Any shape, one thread per output element (used only when k or n_out break the tile rules).
__global__ void k_dense_naive(const float* __restrict__ x, int64_t rows, int k,
                              const float* __restrict__ w, const float* __restrict__ bias, int n_out,
                              int act, float* __restrict__ out) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= rows * n_out) return;
  const int64_t r = idx / n_out;
  const int c = static_cast<int>(idx % n_out);
  float s = 0.0f;
  for (int j = 0; j < k; ++j) s = fmaf(x[r * k + j], w[static_cast<int64_t>(j) * n_out + c], s);
  if (bias) s += bias[c];
  out[idx] = activate_any(s, act);
}

// ---------------------------------------------------------------- GEMM on bf16 MFMA with split operands

// float32 products from bf16 matrix cores: every operand is written as a sum of SPL bf16 pieces
// (x = x0 + x1 + x2, each piece the bf16 rounding of what the previous ones left), and the
// product is the sum of the piece products whose weight is not below 2^-24 of the leading one:
//   SPL 3: x0w0 + x0w1 + x1w0 + x0w2 + x1w1 + x2w0   (float32-class accuracy, 6 MFMAs)
//   SPL 2: x0w0 + x0w1 + x1w0                        (about 2^-17 relative, 3 MFMAs)
//   SPL 1: x0w0                                      (plain bf16 inputs, 1 MFMA)
// all accumulated in float32 by v_mfma_f32_32x32x16_bf16, which runs 16x the rate of the float32
// MFMA, so even the 6-term form needs 2.7x fewer matrix-core cycles than k_gemm_f32.
// The weight pieces are prepared once per launch by k_pack_weights in the exact image the
// workgroups copy into LDS: [k tile][piece][k step][half][column][8 consecutive k] bf16.

template <int SPL>
struct Pieces {
  bf16x8 p[SPL];
};

template <int SPL>
__device__ inline Pieces<SPL> split_bf16(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                         float v7) {
  Pieces<SPL> out;
  const float v[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float r = v[j];
    const __bf16 h0 = static_cast<__bf16>(r);
    out.p[0][j] = h0;
    if constexpr (SPL > 1) {
      r -= static_cast<float>(h0);
      const __bf16 h1 = static_cast<__bf16>(r);
      out.p[1][j] = h1;
      if constexpr (SPL > 2) {
        r -= static_cast<float>(h1);
        out.p[2][j] = static_cast<__bf16>(r);
      }
    }
  }
  return out;
}

struct PackArgs {
  GemmSeg seg[3];
  int n_seg;
  int nt;        // column tiles of 32
  uint4* out;    // [n_ktiles][SPL][2][2][32 * nt] groups of 8 bf16
};

template <int SPL>
__global__ void k_pack_weights(PackArgs p) {
  const int bn = 32 * p.nt;
  int idx = blockIdx.x * blockDim.x + threadIdx.x;   // (tile, kstep, half, column)
  const int col = idx % bn;
  idx /= bn;
  const int half = idx & 1, kstep = (idx >> 1) & 1;
  int tile = idx >> 2;
  const int tile_id = tile;
  int s = 0;
  while (s < p.n_seg) {
    const int nt = (p.seg[s].k + kBK - 1) / kBK;
    if (tile < nt) break;
    tile -= nt;
    ++s;
  }
  if (s >= p.n_seg) return;
  const GemmSeg& sg = p.seg[s];
  const int k0 = tile * kBK + 16 * kstep + 8 * half;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    v[j] = (k0 + j < sg.k) ? sg.w[col >> 5][static_cast<int64_t>(k0 + j) * sg.ldw + (col & 31)] : 0.0f;
  const Pieces<SPL> parts = split_bf16<SPL>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
  for (int t = 0; t < SPL; ++t)
    p.out[((static_cast<int64_t>(tile_id) * SPL + t) * 4 + kstep * 2 + half) * bn + col] =
        __builtin_bit_cast(uint4, parts.p[t]);
}

template <int NT, int SPL>
__global__ __launch_bounds__(kThreads, 2) void k_gemm_split(GemmArgs g, const uint4* __restrict__ wpack) {
  constexpr int BN = 32 * NT;
  constexpr int kTileVec = SPL * 4 * BN;                  // uint4 per K tile of the weight image
  constexpr int kStage = (kTileVec + kThreads - 1) / kThreads;
  __shared__ uint4 Bs[2][kTileVec];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t b = blockIdx.y;
  const int row0 = blockIdx.x * kBM;
  const int my_row = row0 + 32 * wave + li;
  const bool row_ok = my_row < g.n_rows;

  int n_tiles = 0;
  for (int s = 0; s < g.n_seg; ++s) n_tiles += (g.seg[s].k + kBK - 1) / kBK;

  auto locate = [&](int t, int& s, int& k0) {
    s = 0;
    while (true) {
      const int nt = (g.seg[s].k + kBK - 1) / kBK;
      if (t < nt) break;
      t -= nt;
      ++s;
    }
    k0 = t * kBK;
  };
  // lane (li, lh) owns row 32*wave+li and, per 16-wide k step s, floats 16s+8lh .. 16s+8lh+7
  auto load_a = [&](int t, float4 (&a)[4]) {
    int s, k0;
    locate(t, s, k0);
    const GemmSeg& sg = g.seg[s];
    const float* p = sg.a + (b * g.n_rows + my_row) * static_cast<int64_t>(sg.lda) + k0 + 8 * lh;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int off = 16 * (c >> 1) + 4 * (c & 1);
      if (row_ok && k0 + 8 * lh + off < sg.k)
        a[c] = *reinterpret_cast<const float4*>(p + off);
      else
        a[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // the weight image is already in LDS order: a straight 16-byte-per-lane copy, global -> LDS
  // without passing through registers (lane-linear destination, as global_load_lds requires)
  auto stage_b = [&](int t, int buf) {
    const uint4* src = wpack + static_cast<int64_t>(t) * kTileVec;
#pragma unroll
    for (int j = 0; j < kStage; ++j) {
      const int i = tid + kThreads * j;
      if (kTileVec % kThreads == 0 || i < kTileVec)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + i),
            (__attribute__((address_space(3))) void*)(&Bs[buf][i]), 16, 0, 0);
    }
  };

  f32x16 acc[NT];
  float bias_v[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if (g.bias_by_tile)
      bias_v[n] = g.bias_tile[n] ? g.bias_tile[n][li] : 0.0f;
    else
      bias_v[n] = g.bias ? g.bias[b * g.bias_stride + 32 * n + li] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
  }

  float4 a_cur[4], a_nxt[4];
  load_a(0, a_cur);
  stage_b(0, 0);
  __syncthreads();
  for (int t = 0; t < n_tiles; ++t) {
    const bool more = t + 1 < n_tiles;
    if (more) {
      stage_b(t + 1, (t + 1) & 1);
      load_a(t + 1, a_nxt);
    }
    const uint4* bs = Bs[t & 1];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const Pieces<SPL> apieces =
          split_bf16<SPL>(a_cur[2 * s].x, a_cur[2 * s].y, a_cur[2 * s].z, a_cur[2 * s].w, a_cur[2 * s + 1].x,
                          a_cur[2 * s + 1].y, a_cur[2 * s + 1].z, a_cur[2 * s + 1].w);
      const bf16x8* ap = apieces.p;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        bf16x8 wp[SPL];
#pragma unroll
        for (int p = 0; p < SPL; ++p)
          wp[p] = __builtin_bit_cast(bf16x8, bs[(p * 4 + s * 2 + lh) * BN + 32 * n + li]);
        // smallest terms first
        if constexpr (SPL == 3) {
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], wp[0], acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], wp[1], acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], wp[2], acc[n], 0, 0, 0);
        }
        if constexpr (SPL >= 2) {
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], wp[0], acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], wp[1], acc[n], 0, 0, 0);
        }
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], wp[0], acc[n], 0, 0, 0);
      }
    }
    __syncthreads();   // also retires the direct-to-LDS copy of the next weight tile
#pragma unroll
    for (int c = 0; c < 4; ++c) a_cur[c] = a_nxt[c];
  }

  float* obase = g.out ? g.out + b * g.out_rows * static_cast<int64_t>(g.ldo) + g.out_col0 + li : nullptr;
  float cs[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) cs[n] = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (row < g.n_rows) {
      const int64_t orow = g.row_map ? g.row_map[static_cast<int64_t>(row) * g.row_map_stride] - g.row_map_sub : row;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float v = activate(acc[n][r] + bias_v[n], g.relu);
        if (obase) obase[orow * g.ldo + 32 * n] = v;
        cs[n] += v;
      }
    }
  }
  if (g.colsum != nullptr) {   // one partial row per workgroup: the four waves' 32-row sums added in float64
    __shared__ double red[4 * BN];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      cs[n] += __shfl_xor(cs[n], 32);
      if (lh == 0) red[wave * BN + 32 * n + li] = static_cast<double>(cs[n]);
    }
    __syncthreads();
    if (tid < BN) {
      const double s = ((red[tid] + red[BN + tid]) + red[2 * BN + tid]) + red[3 * BN + tid];
      g.colsum[(b * g.colsum_tiles + g.colsum_tile0 + blockIdx.x) * g.colsum_ld + g.out_col0 + tid] = s;
    }
  }
}

}  // namespace

// Matmul arithmetic of the feature kernels (gunet_set_matmul_mode) and the GEMM kernel choice
// (gunet_set_kernel_choice).
// Process-wide settings, read at launch time by any host thread.
std::atomic<int> g_matmul_mode{GUNET_MATMUL_F32};
std::atomic<int> g_kernel_choice{GUNET_KERNELS_AUTO};

inline int mode_pieces(int mode) {
  return mode == GUNET_MATMUL_SPLIT3 ? 3 : mode == GUNET_MATMUL_SPLIT2 ? 2 : mode == GUNET_MATMUL_BF16 ? 1 : 0;
}



template <int SPL>
int launch_gemm_split(const GemmArgs& g, int nt, dim3 grid, uint4* wpack, hipStream_t stream) {
  PackArgs p;
  for (int s = 0; s < 3; ++s) p.seg[s] = g.seg[s];
  p.n_seg = g.n_seg;
  p.nt = nt;
  p.out = wpack;
  const int work = gemm_ktiles(g) * 4 * 32 * nt;
  k_pack_weights<SPL><<<static_cast<unsigned>(ceil_div(work, kThreads)), kThreads, 0, stream>>>(p);
  GUNET_LAUNCH_CHECK();
  switch (nt) {
    case 1: k_gemm_split<1, SPL><<<grid, kThreads, 0, stream>>>(g, wpack); break;
    case 2: k_gemm_split<2, SPL><<<grid, kThreads, 0, stream>>>(g, wpack); break;
    case 3: k_gemm_split<3, SPL><<<grid, kThreads, 0, stream>>>(g, wpack); break;
    case 4: k_gemm_split<4, SPL><<<grid, kThreads, 0, stream>>>(g, wpack); break;
    default: set_error("launch_gemm: bad column tile count %d", nt); return GUNET_ERR_INVALID;
  }
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

int device_cu_count() {
  static std::atomic<int> n_cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = n_cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    n_cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// Largest LDS allocation one workgroup may make on the current device (queried once per device);
// a weight image that does not fit sends the launch to the tiled kernel.
size_t device_lds_limit() {
  static std::atomic<int> limit[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 64 * 1024;
  int n = limit[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || n <= 0) n = 64 * 1024;
    limit[dev].store(n, std::memory_order_relaxed);
  }
  return std::min(static_cast<size_t>(n), kResMaxLds);
}

// The weights-resident kernels take launches whose operand segments are whole K tiles, with
// identity row mapping, when the weight image (float32, or bf16 in GUNET_MATMUL_BF16 mode) fits LDS
// next to 2 KiB of store staging per wave.
bool bf16_mode() { return g_matmul_mode == GUNET_MATMUL_BF16; }

inline size_t resident_lds_bytes(int nt, int nkt, bool bf16) {
  return (bf16 ? 2 : 4) * static_cast<size_t>(32 * nkt) * 32 * nt + sizeof(float) * kResWaves * 16 * 32;
}

inline bool resident_ragged(const GemmArgs& g) {
  return g.n_seg == 1 && g.gather_table == nullptr && g.seg[0].k % kBK != 0 && g.seg[0].k % 4 == 0 &&
         g.seg[0].k > 3 * kBK && g.seg[0].k < 4 * kBK;
}

inline bool gathers_rows(const GemmArgs& g) {   // operand addressing only the tiled float32 kernel knows
  for (int s = 0; s < g.n_seg; ++s)
    if (g.seg[s].row_idx != nullptr || g.seg[s].shared) return true;
  return false;
}

// Waves per workgroup of the resident kernel a launch gets.
inline int resident_block_waves(const GemmArgs& g, int nt) {
  return (g.gather_table != nullptr && nt == 4) ? kResGatherWaves : kResWaves;
}

inline bool resident_ok(const GemmArgs& g, int nt, int64_t batch) {
  if (g.row_map != nullptr || gathers_rows(g) || g.act > GUNET_ACT_RELU) return false;
  if (g.a16 || g.o16) return true;   // bf16 storage: these kernels only (launch_gemm_resident16 knows the shapes)
  if (mode_pieces(g_matmul_mode) > 1) return false;   // split-bf16 modes: tiled kernel only
  const bool bf16 = bf16_mode();
  const bool ragged = resident_ragged(g);   // one float32 segment of 97..127 columns, 128 output columns
  for (int s = 0; s < g.n_seg; ++s)
    if (g.seg[s].k % kBK != 0 && !(ragged && !bf16 && nt == 4)) return false;
  const int kt = gemm_ktiles(g);
  if (bf16) {   // 96 or 128 columns, 4 / 8 / 12 K tiles; gathering form: 128 columns, 8 or 12 K tiles
    if (g.gather_table != nullptr ? !(nt == 4 && (kt == 8 || kt == 12)) : !((nt == 3 || nt == 4) && (kt == 4 || kt == 8)))
      return false;
  } else if (g.gather_table != nullptr) {   // 128 columns x 8 K tiles, or 64 columns x 12 K tiles
    if (!((nt == 4 && kt == 8) || (nt == 2 && kt == 12))) return false;
  } else if ((nt != 3 && nt != 4) || (kt != 4 && kt != 8)) {
    return false;
  }
  if (resident_lds_bytes(nt, kt, bf16) > device_lds_limit()) return false;
  for (int s = 0; s < g.n_seg; ++s)   // 32-bit byte offsets into every operand segment and the output
    if (batch * g.n_rows * g.seg[s].lda * 4 >= (int64_t{1} << 32)) return false;
  if (batch * (g.out_rows ? g.out_rows : g.n_rows) * g.ldo * 4 >= (int64_t{1} << 32)) return false;
  // Below about two row groups per wave the one-off copy of the weights into every workgroup's LDS
  // costs more than the tiled kernel's barriers (measured crossover near 200 k rows); the gathering
  // form also saves the aggregation pass, which moves its break-even down to half a group per wave.
  if (g_kernel_choice == GUNET_KERNELS_TILED) return false;
  if (g_kernel_choice == GUNET_KERNELS_AUTO) {
    const int64_t groups = batch * ceil_div(g.n_rows, kRowGroup);
    const int64_t waves = static_cast<int64_t>(device_cu_count()) * resident_block_waves(g, nt);
    if (g.gather_table == nullptr ? groups < 2 * waves : 2 * groups < waves) return false;
  }
  if (g.gather_table != nullptr) {   // the last segment is the gathered one; 32-bit offsets into the sources
    if (g.n_seg < 2 || g.seg[g.n_seg - 1].k != 128) return false;
    for (int q = 0; q < 4; ++q)
      if (batch * g.gather_rows[q] * 128 >= (int64_t{1} << 32) - 256) return false;   // + the offsets inside a row
  }
  return batch * ceil_div(g.n_rows, kRowGroup) < (1 << 30);
}

// Work split of the resident kernel: every sample's row groups in runs of gpw, one run per wave,
// bps workgroups per sample (about one workgroup per compute unit in all).
struct ResidentSplit {
  int groups, bps, gpw;
};

inline ResidentSplit resident_split(int n_rows, int64_t batch, int block_waves, bool paired) {
  ResidentSplit r;
  r.groups = static_cast<int>(ceil_div(n_rows, kRowGroup));
  // workgroups per sample (a paired launch runs two per row range: both in the same round of workgroups)
  const int64_t want = std::max<int64_t>(1, device_cu_count() / (batch * (paired ? 2 : 1)));
  const int64_t bps = std::min<int64_t>(want, ceil_div(r.groups, block_waves));
  r.gpw = static_cast<int>(ceil_div(r.groups, bps * block_waves));
  r.bps = static_cast<int>(ceil_div(r.groups, static_cast<int64_t>(r.gpw) * block_waves));   // no empty workgroups
  return r;
}

// Which kernel a launch gets: a weights-resident one where it applies (float32, or bf16 operands in
// GUNET_MATMUL_BF16 mode); else, in the bf16 / split-bf16 modes and with a weight-image buffer at
// hand, the tiled split kernel; else the tiled float32 one.
bool takes_resident(const GemmArgs& g, int nt, int64_t batch, const void* wpack) {
  (void)wpack;
  return resident_ok(g, nt, batch);
}
inline bool takes_split(const void* wpack) { return mode_pieces(g_matmul_mode) > 0 && wpack != nullptr; }

// Row tiles (of 128 rows) one workgroup of the tiled float32 kernel walks.  More than one only when
// the launch writes column sums, so that a sample leaves at most about 2 * CUs / batch (32 .. 256)
// partial rows for the per-sample kernel that adds them up.
inline int tiled_tiles_per_wg(const GemmArgs& g, int64_t batch, bool with_colsum) {
  if (!with_colsum) return 1;
  const int64_t tiles = ceil_div(g.n_rows, kBM);
  const int64_t rows = std::min<int64_t>(256, std::max<int64_t>(32, 2 * device_cu_count() / std::max<int64_t>(1, batch)));
  return static_cast<int>(std::max<int64_t>(1, ceil_div(tiles, rows)));
}

// Rows of column-sum partials per sample that a launch writes (from colsum_tile0 on).
int gemm_colsum_rows(const GemmArgs& g, int nt, int64_t batch, const void* wpack) {
  if (g.n_rows == 0) return 0;
  if (takes_resident(g, nt, batch, wpack)) return resident_split(g.n_rows, batch, resident_block_waves(g, nt), g.paired != 0).bps;
  const int64_t tiles = ceil_div(g.n_rows, kBM);
  if (takes_split(wpack) && !gathers_rows(g)) return static_cast<int>(tiles);
  return static_cast<int>(ceil_div(tiles, tiled_tiles_per_wg(g, batch, true)));
}

// Small launches of the tiled kernel are cut into 32-column slices: the matrix work of a 128-row
// tile is then shared by four workgroups instead of sitting in one wave per 32 rows.
inline bool tiled_slices(int64_t row_tiles, int nt) {   // row_tiles: 128-row tiles of the whole launch, all samples
  return nt > 1 && row_tiles < 2 * static_cast<int64_t>(device_cu_count());
}

// The tiled float32 kernel addresses every sample's operand segments with 32-bit byte offsets.
inline bool tiled_offsets_fit(const GemmArgs& g) {
  for (int s = 0; s < g.n_seg; ++s) {
    const int64_t rows = g.seg[s].row_idx != nullptr ? g.seg[s].a_rows : g.n_rows;
    if (rows * g.seg[s].lda * 4 >= static_cast<int64_t>(kOutOfBuffer)) return false;
  }
  return true;
}

int launch_tiled(int nt, bool group, dim3 grid, hipStream_t stream, const GemmArgs* g, int tpw, const GemmGroup* gg) {
  (void)group;
  if (g != nullptr && g->act > GUNET_ACT_RELU) {   // an activation beyond relu: the instantiations that know them all
    switch (nt) {
      case 1: k_gemm_f32<1, true><<<grid, kThreads, 0, stream>>>(*g, tpw); break;
      case 2: k_gemm_f32<2, true><<<grid, kThreads, 0, stream>>>(*g, tpw); break;
      case 3: k_gemm_f32<3, true><<<grid, kThreads, 0, stream>>>(*g, tpw); break;
      case 4: k_gemm_f32<4, true><<<grid, kThreads, 0, stream>>>(*g, tpw); break;
      default: set_error("launch_gemm: bad column tile count %d", nt); return GUNET_ERR_INVALID;
    }
    GUNET_LAUNCH_CHECK();
    return GUNET_OK;
  }
  if (g != nullptr) {
    switch (nt) {
      case 1: k_gemm_f32<1><<<grid, kThreads, 0, stream>>>(*g, tpw); break;
      case 2: k_gemm_f32<2><<<grid, kThreads, 0, stream>>>(*g, tpw); break;
      case 3: k_gemm_f32<3><<<grid, kThreads, 0, stream>>>(*g, tpw); break;
      case 4: k_gemm_f32<4><<<grid, kThreads, 0, stream>>>(*g, tpw); break;
      default: set_error("launch_gemm: bad column tile count %d", nt); return GUNET_ERR_INVALID;
    }
  } else {
    switch (nt) {
      case 1: k_gemm_f32_group<1><<<grid, kThreads, 0, stream>>>(*gg); break;
      case 2: k_gemm_f32_group<2><<<grid, kThreads, 0, stream>>>(*gg); break;
      case 3: k_gemm_f32_group<3><<<grid, kThreads, 0, stream>>>(*gg); break;
      case 4: k_gemm_f32_group<4><<<grid, kThreads, 0, stream>>>(*gg); break;
      default: set_error("launch_gemm_group: bad column tile count %d", nt); return GUNET_ERR_INVALID;
    }
  }
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

template <int NT, int NKT, bool GATHER, bool BF16, bool RAGGED = false, bool A16 = false, bool O16 = false>
int launch_resident_inst(const GemmArgs& g, int64_t batch, hipStream_t stream) {
  static std::atomic<bool> raised[64] = {};   // per device: the kernel's dynamic-LDS limit is raised
  const size_t lds = resident_lds_bytes(NT, NKT, BF16);
  int dev = 0;
  GUNET_HIP_TRY(hipGetDevice(&dev));
  GUNET_REQUIRE(dev >= 0 && dev < 64, "launch_gemm: device index %d", dev);
  if (!raised[dev].load(std::memory_order_acquire)) {   // two threads may both raise it: harmless
    GUNET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_resident<NT, NKT, GATHER, BF16, RAGGED, A16, O16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(device_lds_limit())));
    raised[dev].store(true, std::memory_order_release);
  }
  constexpr int kWaves = (GATHER && NT == 4) ? kResGatherWaves : kResWaves;
  const ResidentSplit sp = resident_split(g.n_rows, batch, kWaves, g.paired != 0);
  const int blocks = static_cast<int>(batch * sp.bps) * (g.paired ? 2 : 1);
  static_assert(NKT <= kResMaxTiles, "tile table");
  ResidentTiles tl = {};
  {
    int t = 0;
    const int n_plain = g.n_seg - (GATHER ? 1 : 0);   // the gathered segment is not read through the table
    for (int s = 0; s < n_plain; ++s)
      for (int k0 = 0; k0 < g.seg[s].k && t < kResMaxTiles; k0 += kBK, ++t) {
        tl.a[t] = A16 ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(g.seg[s].a) + k0) : g.seg[s].a + k0;
        tl.lda[t] = g.seg[s].lda;
      }
    GUNET_REQUIRE(t == (GATHER ? NKT - 4 : NKT), "launch_gemm: %d operand K tiles for a %d-tile kernel", t, NKT);
  }
  GemmArgs ga = g;
  if (GATHER) {
    for (int q = 0; q < 4; ++q) tl.gather_bytes[q] = static_cast<uint32_t>(batch * g.gather_rows[q] * (A16 ? 64 : 128));
    int exponent = 0;
    ga.gather_fold = g.gather_scale > 0.0f && std::frexp(g.gather_scale, &exponent) == 0.5f;
  }
  k_gemm_resident<NT, NKT, GATHER, BF16, RAGGED, A16, O16><<<blocks, 64 * kWaves, lds, stream>>>(ga, tl, sp.groups, sp.bps, sp.gpw);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

// The launches of the bf16-storage forward: bf16 operands and / or bf16 output.
int launch_gemm_resident16(const GemmArgs& g, int nt, int64_t batch, hipStream_t stream) {
  const int kt = gemm_ktiles(g);
  if (!g.a16) {   // float32 operand (rounded to bf16 on its way into the matrix cores), bf16 result: the model's in_layer
    if (resident_ragged(g) && nt == 4) return launch_resident_inst<4, 4, false, true, true, false, true>(g, batch, stream);
    if (nt == 4 && kt == 4) return launch_resident_inst<4, 4, false, true, false, false, true>(g, batch, stream);
  } else if (g.gather_table != nullptr) {
    if (nt == 4 && kt == 8) return launch_resident_inst<4, 8, true, true, false, true, true>(g, batch, stream);
    if (nt == 4 && kt == 12) return launch_resident_inst<4, 12, true, true, false, true, true>(g, batch, stream);
  } else if (g.o16 || g.out == nullptr) {
    if (nt == 4 && kt == 4) return launch_resident_inst<4, 4, false, true, false, true, true>(g, batch, stream);
    if (nt == 4 && kt == 8) return launch_resident_inst<4, 8, false, true, false, true, true>(g, batch, stream);
  } else {        // bf16 operand, float32 result: the model's out layer
    if (nt == 3 && kt == 4) return launch_resident_inst<3, 4, false, true, false, true, false>(g, batch, stream);
    if (nt == 4 && kt == 4) return launch_resident_inst<4, 4, false, true, false, true, false>(g, batch, stream);
  }
  set_error("launch_gemm: no bf16-storage kernel for %d column tiles, %d K tiles (a16 %d, o16 %d, gather %d)", nt, kt,
            g.a16, g.o16, g.gather_table != nullptr);
  return GUNET_ERR_UNSUPPORTED;
}

int launch_gemm_resident(const GemmArgs& g, int nt, int64_t batch, hipStream_t stream) {
  const int kt = gemm_ktiles(g);
  if (g.a16 || g.o16) return launch_gemm_resident16(g, nt, batch, stream);
  if (bf16_mode()) {
    if (g.gather_table != nullptr)
      return kt == 8 ? launch_resident_inst<4, 8, true, true>(g, batch, stream)
                     : launch_resident_inst<4, 12, true, true>(g, batch, stream);
    if (nt == 4)
      return kt == 4 ? launch_resident_inst<4, 4, false, true>(g, batch, stream)
                     : launch_resident_inst<4, 8, false, true>(g, batch, stream);
    return kt == 4 ? launch_resident_inst<3, 4, false, true>(g, batch, stream)
                   : launch_resident_inst<3, 8, false, true>(g, batch, stream);
  }
  if (g.gather_table != nullptr)
    return nt == 4 ? launch_resident_inst<4, 8, true, false>(g, batch, stream)
                   : launch_resident_inst<2, 12, true, false>(g, batch, stream);
  if (resident_ragged(g)) return launch_resident_inst<4, 4, false, false, true>(g, batch, stream);
  if (nt == 4)
    return kt == 4 ? launch_resident_inst<4, 4, false, false>(g, batch, stream)
                   : launch_resident_inst<4, 8, false, false>(g, batch, stream);
  return kt == 4 ? launch_resident_inst<3, 4, false, false>(g, batch, stream)
                 : launch_resident_inst<3, 8, false, false>(g, batch, stream);
}

// wpack: scratch of at least wpack_bytes(gemm_ktiles(g), nt) for the split modes (may be null in
// GUNET_MATMUL_F32 mode).
int launch_gemm(GemmArgs g, int nt, int64_t batch, void* wpack, hipStream_t stream) {
  if (g.n_rows == 0 || batch == 0) return GUNET_OK;
  if (g.out_rows == 0) g.out_rows = g.n_rows;
  dim3 grid(static_cast<unsigned>(ceil_div(g.n_rows, kBM)), static_cast<unsigned>(batch));
  if (g.colsum != nullptr && g.colsum_tiles == 0) g.colsum_tiles = gemm_colsum_rows(g, nt, batch, wpack);
  if (g.colsum_ld == 0) {
    g.colsum_ld = 32 * nt;
    if (g.colsum != nullptr && g.out_col0 != 0) {
      set_error("launch_gemm: column sums of a column block need colsum_ld");
      return GUNET_ERR_INVALID;
    }
  }
  // (an activation beyond relu takes the tiled float32 kernel in every matmul mode)
  const bool resident = resident_ok(g, nt, batch),
             split = !resident && takes_split(wpack) && !gathers_rows(g) && g.act <= GUNET_ACT_RELU;
  GUNET_REQUIRE(!g.paired || (resident && nt == 2), "launch_gemm: the paired form belongs to the weights-resident kernel");
  double flops = 0.0;
  for (int s = 0; s < g.n_seg; ++s) flops += 2.0 * batch * g.n_rows * g.seg[s].k * (32.0 * nt * (g.paired ? 2 : 1));
  const int kt_all = gemm_ktiles(g);
  ProfScope prof(split ? kProfGemmSplit
                       : (!resident ? kProfGemmTiled
                                    : (bf16_mode() ? kProfGemmResBf16
                                                   : (kt_all == 4 ? kProfGemmRes128
                                                                  : (kt_all == 8 ? kProfGemmRes256 : kProfGemmRes384)))),
                 stream, flops);
  if (split) {
    const int pieces = mode_pieces(g_matmul_mode);
    uint4* wp = static_cast<uint4*>(wpack);
    if (pieces == 3) return launch_gemm_split<3>(g, nt, grid, wp, stream);
    if (pieces == 2) return launch_gemm_split<2>(g, nt, grid, wp, stream);
    return launch_gemm_split<1>(g, nt, grid, wp, stream);
  }
  if (resident) return launch_gemm_resident(g, nt, batch, stream);
  if (!tiled_offsets_fit(g)) {
    // one flat sample without row maps or column sums (the dense entry point) goes in row chunks
    GUNET_REQUIRE(batch == 1 && g.colsum == nullptr && g.row_map == nullptr && !gathers_rows(g) && g.out_rows == g.n_rows,
                  "launch_gemm: an operand of %d rows per sample is beyond the 2 GiB the tiled kernel addresses", g.n_rows);
    int64_t widest = 1;
    for (int s = 0; s < g.n_seg; ++s) widest = std::max<int64_t>(widest, g.seg[s].lda);
    const int chunk = static_cast<int>((int64_t{1} << 30) / (widest * 4) / kBM * kBM);
    for (int r0 = 0; r0 < g.n_rows; r0 += chunk) {
      GemmArgs gc = g;
      gc.n_rows = gc.out_rows = std::min(chunk, g.n_rows - r0);
      for (int s = 0; s < g.n_seg; ++s) gc.seg[s].a = g.seg[s].a + static_cast<int64_t>(r0) * g.seg[s].lda;
      if (g.out != nullptr) gc.out = g.out + static_cast<int64_t>(r0) * g.ldo;
      GUNET_TRY(launch_gemm(gc, nt, batch, wpack, stream));
    }
    return GUNET_OK;
  }
  const int tpw = tiled_tiles_per_wg(g, batch, g.colsum != nullptr);
  grid.x = static_cast<unsigned>(ceil_div(ceil_div(g.n_rows, kBM), tpw));
  if (tiled_slices(ceil_div(g.n_rows, kBM) * batch, nt)) {
    grid.z = static_cast<unsigned>(nt);
    nt = 1;
  }
  return launch_tiled(nt, false, grid, stream, &g, tpw, nullptr);
}

// Several problems with the same column-tile count and batch as one launch of the tiled float32
// kernel (the bf16 / split modes, and single problems, go through launch_gemm one by one).  The
// caller has set colsum_tiles / colsum_tile0 / colsum_ld / out_rows.
int launch_gemm_group(const GemmArgs* gs, int n, int nt, int64_t batch, void* wpack, hipStream_t stream) {
  int live[kGemmGroup], n_live = 0;
  for (int i = 0; i < n; ++i)
    if (gs[i].n_rows > 0) live[n_live++] = i;
  bool tiled = n_live > 1 && n_live <= kGemmGroup && batch > 0 && !takes_split(wpack);
  for (int i = 0; i < n_live && tiled; ++i)   // (the grouped kernel knows linear and relu only)
    tiled = !resident_ok(gs[live[i]], nt, batch) && gs[live[i]].act <= GUNET_ACT_RELU;
  if (!tiled) {
    for (int i = 0; i < n_live; ++i) GUNET_TRY(launch_gemm(gs[live[i]], nt, batch, wpack, stream));
    return GUNET_OK;
  }
  GemmGroup gg = {};
  unsigned blocks = 0;
  int64_t row_tiles = 0;
  double flops = 0.0;
  for (int i = 0; i < kGemmGroup; ++i) {
    if (i < n_live) {
      gg.p[i] = gs[live[i]];
      GemmArgs& g = gg.p[i];
      GUNET_REQUIRE(tiled_offsets_fit(g), "launch_gemm_group: an operand of %d rows per sample is beyond the 2 GiB the "
                    "tiled kernel addresses", g.n_rows);
      if (g.out_rows == 0) g.out_rows = g.n_rows;
      if (g.colsum_ld == 0) g.colsum_ld = 32 * nt;
      gg.tpw[i] = tiled_tiles_per_wg(g, batch, g.colsum != nullptr);
      row_tiles += ceil_div(g.n_rows, kBM);
      blocks += static_cast<unsigned>(ceil_div(ceil_div(g.n_rows, kBM), gg.tpw[i]));
      for (int s = 0; s < g.n_seg; ++s) flops += 2.0 * batch * g.n_rows * g.seg[s].k * (32.0 * nt);
    } else {
      gg.tpw[i] = 1;
    }
    gg.block_end[i] = blocks;
  }
  dim3 grid(blocks, static_cast<unsigned>(batch));
  ProfScope prof(kProfGemmTiled, stream, flops);
  if (tiled_slices(row_tiles * batch, nt)) {
    grid.z = static_cast<unsigned>(nt);
    nt = 1;
  }
  return launch_tiled(nt, true, grid, stream, nullptr, 0, &gg);
}

}  // namespace gunet

using namespace gunet;

extern "C" {

size_t gunet_dense_workspace_bytes(int32_t k, int32_t n_out) {
  if (k <= 0 || n_out <= 0) return 0;
  return wpack_bytes(static_cast<int>(ceil_div(k, kBK)), 4) + 256;
}

int gunet_dense_forward_f32(const float* x, int64_t rows, int32_t k, const float* kernel,
                            const float* bias, int32_t n_out, int32_t activation, float* out,
                            void* workspace, size_t workspace_bytes, gunet_stream_t stream_) {
  StreamDeviceGuard device_guard(as_stream(stream_));
  return dense_run(x, rows, k, kernel, bias, n_out, activation, out, workspace, workspace_bytes, 0, 0, as_stream(stream_));
}

}  // extern "C"

namespace gunet {

int dense_run(const float* x, int64_t rows, int32_t k, const float* kernel, const float* bias, int32_t n_out,
              int32_t activation, float* out, void* workspace, size_t workspace_bytes, int a16, int o16,
              hipStream_t stream) {
  GUNET_REQUIRE(rows >= 0 && k > 0 && n_out > 0, "dense_forward: bad shape");
  GUNET_REQUIRE(act_valid(activation), "dense_forward: unknown activation %d", activation);
  GUNET_REQUIRE(act_tuned(activation) || !(a16 || o16), "dense_forward: bf16 storage serves linear and relu only");
  if (rows == 0) return GUNET_OK;
  GUNET_REQUIRE(x && kernel && out, "dense_forward: null pointer");
  GUNET_REQUIRE(rows < INT32_MAX, "dense_forward: too many rows");
  const bool tiled = k % 4 == 0 && n_out % 32 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(kernel) % 16 == 0;
  GUNET_REQUIRE(tiled || !(a16 || o16), "dense_forward: bf16 storage needs k %% 4 == 0, n_out %% 32 == 0 and 16-byte alignment");
  if (!tiled) {
    const unsigned grid = static_cast<unsigned>(ceil_div(rows * n_out, kThreads));
    ProfScope prof(kProfDenseNaive, stream);
    k_dense_naive<<<grid, kThreads, 0, stream>>>(x, rows, k, kernel, bias, n_out, activation, out);
    GUNET_LAUNCH_CHECK();
    return GUNET_OK;
  }
  if (mode_pieces(g_matmul_mode) > 0) {
    if (workspace == nullptr || workspace_bytes < gunet_dense_workspace_bytes(k, n_out) ||
        reinterpret_cast<uintptr_t>(workspace) % 16 != 0) {
      set_error("dense_forward: the split-bf16 matmul modes need a 16-byte aligned workspace of %zu bytes",
                gunet_dense_workspace_bytes(k, n_out));
      return GUNET_ERR_WORKSPACE;
    }
  } else {
    workspace = nullptr;
  }
  for (int col0 = 0; col0 < n_out; col0 += 128) {
    const int nt = (n_out - col0 >= 128) ? 4 : (n_out - col0) / 32;
    GemmArgs g = {};
    g.n_seg = 1;
    g.seg[0].a = x;
    g.seg[0].lda = k;
    g.seg[0].k = k;
    g.seg[0].ldw = n_out;
    for (int t = 0; t < nt; ++t) g.seg[0].w[t] = kernel + col0 + 32 * t;
    g.n_rows = static_cast<int>(rows);
    g.bias = bias ? bias + col0 : nullptr;
    g.out = out;
    g.ldo = n_out;
    g.out_col0 = col0;
    g.relu = activation == GUNET_ACT_RELU;
    g.act = activation;
    g.a16 = a16;
    g.o16 = o16;
    GUNET_TRY(launch_gemm(g, nt, 1, workspace, stream));
  }
  return GUNET_OK;
}

}  // namespace gunet

extern "C" {

int gunet_set_matmul_mode(int mode) {
  GUNET_REQUIRE(mode == GUNET_MATMUL_F32 || mode == GUNET_MATMUL_SPLIT3 || mode == GUNET_MATMUL_SPLIT2 ||
                    mode == GUNET_MATMUL_BF16,
                "set_matmul_mode: unknown mode %d", mode);
  g_matmul_mode = mode;
  return GUNET_OK;
}

int gunet_get_matmul_mode(void) { return g_matmul_mode; }

int gunet_set_kernel_choice(int choice) {
  GUNET_REQUIRE(choice == GUNET_KERNELS_AUTO || choice == GUNET_KERNELS_RESIDENT || choice == GUNET_KERNELS_TILED,
                "set_kernel_choice: unknown choice %d", choice);
  g_kernel_choice = choice;
  return GUNET_OK;
}
}  // extern "C"
