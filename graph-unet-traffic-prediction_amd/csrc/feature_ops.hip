// Float32 feature kernels of the Graph U-Net forward path for gfx950 (MI355X):
//   * row gathers and CSR segment max / in-order segment sum (HBM-bound),
//   * a 128-row tiled GEMM on v_mfma_f32_32x32x2_f32 with LDS-staged operand tiles,
//     fused bias + ReLU epilogue and deterministic per-block column sums,
//   * the GeoQuadrantGCN layer: four-direction node-embedding GEMM, gathered edge MLP on MFMA,
//     receiver-CSR aggregation, node MLP GEMM and the small global-feature updates.
// All reductions are atomic-free and run in a fixed order, so outputs are bitwise reproducible.
#include <algorithm>
#include <cfloat>
#include <type_traits>

#include "common.h"

namespace gunet {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kThreads = 256;

__device__ inline float activate(float v, int relu) { return relu ? fmaxf(v, 0.0f) : v; }

// ---------------------------------------------------------------- gathers / segment reduces

template <int V>
__global__ void k_gather_rows(const float* __restrict__ x, int width, const int32_t* __restrict__ ei,
                              int64_t n_edges, float* __restrict__ senders,
                              float* __restrict__ receivers) {
  const int wv = width / V;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= n_edges * wv) return;
  const int64_t e = idx / wv;
  const int c = static_cast<int>(idx % wv) * V;
  const int2 sr = reinterpret_cast<const int2*>(ei)[e];
  if (V == 4) {
    *reinterpret_cast<float4*>(senders + e * width + c) =
        *reinterpret_cast<const float4*>(x + static_cast<int64_t>(sr.x) * width + c);
    *reinterpret_cast<float4*>(receivers + e * width + c) =
        *reinterpret_cast<const float4*>(x + static_cast<int64_t>(sr.y) * width + c);
  } else {
    senders[e * width + c] = x[static_cast<int64_t>(sr.x) * width + c];
    receivers[e * width + c] = x[static_cast<int64_t>(sr.y) * width + c];
  }
}

// One thread owns V consecutive columns of one output row and walks the segment front to back.
template <int V, bool IS_MAX>
__global__ void k_segment_reduce(const float* __restrict__ src, int64_t n_src, int width,
                                 const int32_t* __restrict__ rowptr,
                                 const int32_t* __restrict__ perm, int64_t n_segments, float scale,
                                 float* __restrict__ out) {
  const int wv = width / V;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= n_segments * wv) return;
  const int64_t seg = idx / wv;
  const int c = static_cast<int>(idx % wv) * V;
  const int64_t b = blockIdx.y;
  const float* s = src + b * n_src * width + c;
  const int beg = rowptr[seg];
  int end = rowptr[seg + 1];
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = IS_MAX ? -FLT_MAX : 0.0f;
  if (IS_MAX && V == 4) {   // four member rows in flight (a repeated member does not change a maximum)
    for (int p = beg; p < end; p += 4) {
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        v[i] = *reinterpret_cast<const float4*>(s + static_cast<int64_t>(perm[p + i < end ? p + i : beg]) * width);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[0] = fmaxf(acc[0], v[i].x);
        acc[1] = fmaxf(acc[1], v[i].y);
        acc[2] = fmaxf(acc[2], v[i].z);
        acc[3] = fmaxf(acc[3], v[i].w);
      }
    }
    end = beg;   // nothing left for the generic walk below
  }
  for (int p = beg; p < end; ++p) {
    const float* row = s + static_cast<int64_t>(perm[p]) * width;
    if (V == 4) {
      const float4 v = *reinterpret_cast<const float4*>(row);
      if (IS_MAX) {
        acc[0] = fmaxf(acc[0], v.x);
        acc[1] = fmaxf(acc[1], v.y);
        acc[2] = fmaxf(acc[2], v.z);
        acc[3] = fmaxf(acc[3], v.w);
      } else {
        acc[0] += v.x;
        acc[1] += v.y;
        acc[2] += v.z;
        acc[3] += v.w;
      }
    } else {
      acc[0] = IS_MAX ? fmaxf(acc[0], row[0]) : acc[0] + row[0];
    }
  }
  float* o = out + (b * n_segments + seg) * width + c;
  if (!IS_MAX) {
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] *= scale;
  }
  if (V == 4) {
    *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  } else {
    o[0] = acc[0];
  }
}

template <bool IS_MAX>
int launch_segment_reduce(const float* src, int64_t n_src, int32_t width, const int32_t* rowptr,
                          const int32_t* perm, int64_t n_segments, int32_t batch, float scale,
                          float* out, hipStream_t stream) {
  if (n_segments == 0 || batch == 0 || width == 0) return GUNET_OK;
  const bool vec = (width % 4 == 0) && (reinterpret_cast<uintptr_t>(src) % 16 == 0) &&
                   (reinterpret_cast<uintptr_t>(out) % 16 == 0);
  const int64_t work = n_segments * (vec ? width / 4 : width);
  dim3 grid(static_cast<unsigned>(ceil_div(work, kThreads)), static_cast<unsigned>(batch));
  ProfScope prof(kProfSegment, stream,
                 static_cast<double>(batch) * (4.0 * width * (n_src + n_segments) + 4.0 * (n_src + n_segments)));
  if (vec)
    k_segment_reduce<4, IS_MAX><<<grid, kThreads, 0, stream>>>(src, n_src, width, rowptr, perm,
                                                               n_segments, scale, out);
  else
    k_segment_reduce<1, IS_MAX><<<grid, kThreads, 0, stream>>>(src, n_src, width, rowptr, perm,
                                                               n_segments, scale, out);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

// The five segment maxima of one 2x2 pooling step (node features and the four edge-feature sets,
// layers/hybrid_gnn.py:427, :401) as one launch: consecutive block ranges serve the jobs, each
// thread does exactly what k_segment_reduce<4, true> does for its element.
constexpr int kPoolJobs = 5;
struct PoolJobs {
  const float* src[kPoolJobs];
  float* out[kPoolJobs];
  const int32_t* rowptr[kPoolJobs];
  const int32_t* perm[kPoolJobs];
  int64_t n_src[kPoolJobs];
  int64_t n_seg[kPoolJobs];
  int width[kPoolJobs];
  unsigned block_end[kPoolJobs];   // running total of the jobs' block counts
};

__global__ void k_pool_max(PoolJobs j) {
  int job = 0;
  while (job < kPoolJobs - 1 && blockIdx.x >= j.block_end[job]) ++job;
  const unsigned block = blockIdx.x - (job > 0 ? j.block_end[job - 1] : 0u);
  const int width = j.width[job], wv = width / 4;
  const int64_t n_seg = j.n_seg[job];
  const int64_t idx = static_cast<int64_t>(block) * blockDim.x + threadIdx.x;
  if (idx >= n_seg * wv) return;
  const int64_t seg = idx / wv;
  const int c = static_cast<int>(idx % wv) * 4;
  const int64_t b = blockIdx.y;
  const float* s = j.src[job] + b * j.n_src[job] * width + c;
  const int32_t* rowptr = j.rowptr[job];
  const int32_t* perm = j.perm[job];
  const int beg = rowptr[seg], end = rowptr[seg + 1];
  float4 acc = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
  for (int p = beg; p < end; p += 4) {   // a 2x2 cell has at most four members: all rows in flight at once
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pi = p + i < end ? p + i : beg;   // a repeated member does not change a maximum
      v[i] = *reinterpret_cast<const float4*>(s + static_cast<int64_t>(perm[pi]) * width);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc.x = fmaxf(acc.x, v[i].x);
      acc.y = fmaxf(acc.y, v[i].y);
      acc.z = fmaxf(acc.z, v[i].z);
      acc.w = fmaxf(acc.w, v[i].w);
    }
  }
  *reinterpret_cast<float4*>(j.out[job] + (b * n_seg + seg) * width + c) = acc;
}

// ---------------------------------------------------------------- GEMM on fp32 MFMA

// out[b, map(r), :] = act( sum_s A_s[b, r, :] @ W_s + bias[b] ),  r < n_rows, for up to three operand
// segments that are concatenated along K in the reference (x | aggregated messages, or
// upsampled x | skip x | aggregated messages).  One workgroup = 128 rows x (32*NT) columns,
// four waves of 32 rows each; K is consumed in tiles of 32.
struct GemmSeg {
  const float* a;      // sample 0, row 0
  int lda;             // row stride (floats); sample stride is n_rows * lda
  int k;               // width of this segment (multiple of 4)
  const float* w[4];   // weights of column tile t: element [k][j] at w[t][k * ldw + j], j < 32
  int ldw;
};

struct GemmArgs {
  GemmSeg seg[3];
  int n_seg;
  int n_rows;                 // A rows per sample
  const float* bias;          // [.., 32*NT] or null
  int bias_stride;            // floats between samples (0 = shared)
  float* out;                 // sample 0; null = compute the column sums only
  int ldo;
  int out_col0;               // first output column of this launch
  int out_rows;               // rows per sample of the output tensor
  const int32_t* row_map;     // A row r -> output row row_map[r * row_map_stride] - row_map_sub
  int row_map_stride;         //   (null = identity)
  int row_map_sub;
  float* colsum;              // partial column sums of the activated output, or null:
  int colsum_tiles;           //   [B][colsum_tiles][colsum_ld], this launch fills rows colsum_tile0 ..
  int colsum_tile0;           //   colsum_tile0 + gemm_colsum_rows() - 1 (one per workgroup or per wave)
  int colsum_ld;              //   floats (0 = 32*NT), this launch's columns start at out_col0
  int relu;
  // Gathered last segment (weights-resident kernel only): when gather_table is set, the last
  // segment (k = 128) is not read from seg[].a; row v's column block q is formed while loading as
  // gather_scale * (sum over j of row gather_table[(v * 4 + q) * 3 + j] (>= 0, in that order) of
  // gather_src[q], a [B][gather_rows[q]][32] tensor).
  const int32_t* gather_table;
  const float* gather_src[4];
  int gather_rows[4];
  float gather_scale;
};

constexpr int kBM = 128;
constexpr int kBK = 32;
constexpr int kRowGroup = 32;   // rows of one wave = rows of one column-sum partial

// The A operand goes straight from global memory into MFMA registers: lane (li, lh) of wave w
// owns row 32w+li of the tile and reads floats 8c+4lh..8c+4lh+3 of every 8-float K chunk of that
// row, so one 128-bit load feeds four v_mfma_f32_32x32x2_f32 (the two half-waves supply the two
// k of each MFMA; the k order inside a chunk is a fixed permutation, the same for the weights).
// Only the weight tile goes through LDS (double-buffered, one barrier per K tile); the next A
// tile is prefetched in registers.  Epilogue: bias, activation, optional row scatter, and the
// column sums of every wave's 32 rows (fixed order: 16 rows per lane, then the two half-waves).
// `bx`: this workgroup's row tile within the problem `g`.
template <int NT>
__device__ __forceinline__ void gemm_f32_tile(const GemmArgs& g, const int bx) {
  constexpr int BN = 32 * NT;
  __shared__ __attribute__((aligned(16))) float Bs[2][kBK * BN];
  __shared__ __attribute__((aligned(16))) float Stage[4][16 * 32];   // epilogue: one per wave

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t b = blockIdx.y;
  const int row0 = bx * kBM;
  const int my_row = row0 + 32 * wave + li;
  const bool row_ok = my_row < g.n_rows;

  int n_tiles = 0;
  for (int s = 0; s < g.n_seg; ++s) n_tiles += (g.seg[s].k + kBK - 1) / kBK;

  float4 b_reg[NT];
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
  auto load_a = [&](int t, float4 (&a)[4]) {
    int s, k0;
    locate(t, s, k0);
    const GemmSeg& sg = g.seg[s];
    const float* p = sg.a + (b * g.n_rows + my_row) * static_cast<int64_t>(sg.lda) + k0 + 4 * lh;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (row_ok && k0 + 8 * c + 4 * lh < sg.k)
        a[c] = *reinterpret_cast<const float4*>(p + 8 * c);
      else
        a[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // thread (kr = tid / 8, c4 = tid % 8) stages row k0+kr, floats 4c4..4c4+3 of every column tile
  auto load_b = [&](int t) {
    int s, k0;
    locate(t, s, k0);
    const GemmSeg& sg = g.seg[s];
    const int k = k0 + (tid >> 3);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (k < sg.k)
        b_reg[j] = *reinterpret_cast<const float4*>(sg.w[j] + static_cast<int64_t>(k) * sg.ldw + 4 * (tid & 7));
      else
        b_reg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
      *reinterpret_cast<float4*>(&Bs[buf][(tid >> 3) * BN + 32 * j + 4 * (tid & 7)]) = b_reg[j];
  };

  f32x16 acc[NT];
  float bias_v[NT];   // loaded up front: a load inside the epilogue would drain its stores
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    bias_v[n] = g.bias ? g.bias[b * g.bias_stride + 32 * n + li] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
  }

  float4 a_cur[4], a_nxt[4];
  load_a(0, a_cur);
  load_b(0);
  store_b(0);
  __syncthreads();
  for (int t = 0; t < n_tiles; ++t) {
    const bool more = t + 1 < n_tiles;
    if (more) {
      load_b(t + 1);
      load_a(t + 1, a_nxt);
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
  if (g.row_map == nullptr) {
    // 16 rows x 32 columns at a time through the wave's own LDS, picked up row-wise: a store
    // instruction then writes eight full 128-byte lines instead of two
    float* stage = Stage[wave];
    const int srow = lane >> 3, scol = 4 * (lane & 7);
    float* obase = g.out ? g.out + (b * g.out_rows + wrow0 + srow) * static_cast<int64_t>(g.ldo) + g.out_col0 + scol
                         : nullptr;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
          const int r = 8 * h + r8;
          const int rl = (r & 3) + 8 * (r8 >> 2) + 4 * lh;
          const float v = activate(acc[n][r] + bias_v[n], g.relu);
          stage[rl * 32 + li] = v;
          if (wrow0 + 16 * h + rl < g.n_rows) cs[n] += v;
        }
        if (obase) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(&stage[(8 * i + srow) * 32 + scol]);
            if (wrow0 + 16 * h + 8 * i + srow < g.n_rows)
              *reinterpret_cast<float4*>(obase + static_cast<int64_t>(16 * h + 8 * i) * g.ldo + 32 * n) = v;
          }
        }
      }
    }
  } else {
    float* obase = g.out ? g.out + b * g.out_rows * static_cast<int64_t>(g.ldo) + g.out_col0 + li : nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < g.n_rows) {
        const int64_t orow = g.row_map[static_cast<int64_t>(row) * g.row_map_stride] - g.row_map_sub;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float v = activate(acc[n][r] + bias_v[n], g.relu);
          if (obase) obase[orow * g.ldo + 32 * n] = v;
          cs[n] += v;
        }
      }
    }
  }
  if (g.colsum != nullptr) {   // one partial row per workgroup: the four waves' sums added in order
    float* red = Stage[0];     // 4 * BN <= 512 floats; the staging tiles are no longer in use
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      cs[n] += __shfl_xor(cs[n], 32);
      if (lh == 0) red[wave * BN + 32 * n + li] = cs[n];
    }
    __syncthreads();
    if (tid < BN) {
      const float s = ((red[tid] + red[BN + tid]) + red[2 * BN + tid]) + red[3 * BN + tid];
      g.colsum[(b * g.colsum_tiles + g.colsum_tile0 + bx) * g.colsum_ld + g.out_col0 + tid] = s;
    }
  }
}

template <int NT>
__global__ __launch_bounds__(kThreads, 2) void k_gemm_f32(GemmArgs g) {
  gemm_f32_tile<NT>(g, blockIdx.x);
}

// Up to four independent problems of the same column-tile count in one grid (the four quadrant
// GEMMs of the upsampling layer): consecutive block ranges serve the problems.
constexpr int kGemmGroup = 4;
struct GemmGroup {
  GemmArgs p[kGemmGroup];
  unsigned block_end[kGemmGroup];   // running total of the problems' row tiles
};

template <int NT>
__global__ __launch_bounds__(kThreads, 2) void k_gemm_f32_group(GemmGroup gg) {
  int i = 0;
  while (i < kGemmGroup - 1 && blockIdx.x >= gg.block_end[i]) ++i;
  gemm_f32_tile<NT>(gg.p[i], static_cast<int>(blockIdx.x - (i > 0 ? gg.block_end[i - 1] : 0u)));
}

// Weights-resident variant (the weight matrix must fit LDS: 32 * NKT * 32 * NT floats): a workgroup
// of twelve waves (eight for the 128-column gathering form) copies the whole weight matrix into
// LDS once, then its waves work on their own, with no barrier after the
// first: every sample's 32-row groups are dealt out in runs of `gpw` consecutive groups, one run
// per wave (`wps` waves per sample).  The A tile of step t+1 -- the next group's first tile at a
// group's last step -- is requested before the MFMAs of step t, into the other of two register
// buffers, so a wave always has one tile in flight behind its matrix work and the waves of a SIMD
// drift apart instead of stalling together.  Rows past the end of a sample are clamped for the
// loads and masked in the epilogue.  Column sums: one partial row per wave (its whole run).
constexpr int kResWaves = 12;
constexpr int kResThreads = 64 * kResWaves;
constexpr int kResGatherWaves = 8;   // the 128-column gathering variant needs more registers per wave
constexpr int kResGatherThreads = 64 * kResGatherWaves;
constexpr size_t kResMaxLds = 160 * 1024;

// BF16 = true: operands are rounded to bf16 on their way into v_mfma_f32_32x32x16_bf16 (float32
// accumulate and storage); a lane then owns, per 16-wide k step, 8 consecutive floats of its row
// (k = 16s + 8 * lh + 0..7), and the LDS weight image holds one bf16x8 per lane, k step and column
// tile.  The matrix work shrinks 16-fold, so this form runs at the speed of its memory traffic.
// RAGGED = true (one operand segment, float32): k is not a whole number of K tiles; the weight rows
// past k are zero in the LDS image and the last tile's loads past k are redirected to the row's
// first piece of that tile and zeroed.
template <int NT, int NKT, bool GATHER, bool BF16, bool RAGGED = false>
__global__ __launch_bounds__((GATHER && NT == 4) ? kResGatherThreads : kResThreads) void k_gemm_resident(
    GemmArgs g, int groups_per_sample, int wps, int gpw, int batch) {
  static_assert(!GATHER || (NKT >= 6 && NKT % 2 == 0), "the gathered segment is the last four K tiles");
  static_assert(!RAGGED || (!GATHER && !BF16), "the ragged form is float32 without gathering");
  constexpr int kWaves = (GATHER && NT == 4) ? kResGatherWaves : kResWaves;
  constexpr int NX = GATHER ? NKT - 4 : NKT;   // K tiles read from the operand segments
  constexpr int BN = 32 * NT;
  extern __shared__ __attribute__((aligned(16))) float res_smem[];
  // Weight image: for K tile kt, 8-row chunk c and column tile n, one 1 KiB piece in lane order:
  // lane (lh, li) finds rows 32kt + 8c + 4lh + 0..3 of column 32n + li as one float4.
  float* Ws = res_smem;   // [NKT][4][NT][64][4]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int t0 = g.seg[0].k / kBK, t1 = g.n_seg > 1 ? g.seg[1].k / kBK : 0;   // K tiles of segments 0, 1

  {
    int kbase = 0;
    for (int s = 0; s < g.n_seg; ++s) {
      const GemmSeg& sg = g.seg[s];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float* w = sg.w[t];
        if constexpr (BF16) {
          // thread -> (group of 8 k rows, column): 8 strided loads, one 16-byte LDS store
          uint4* img = reinterpret_cast<uint4*>(Ws);
          for (int i = tid; i < sg.k * 4; i += kWaves * 64) {
            const int kg = i >> 5, col = i & 31;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = static_cast<__bf16>(w[static_cast<int64_t>(8 * kg + j) * sg.ldw + col]);
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
            dst[0] = v.x;
            dst[4] = v.y;
            dst[8] = v.z;
            dst[12] = v.w;
          }
        }
      }
      kbase += sg.k;
    }
  }
  __syncthreads();

  const int wid = blockIdx.x * kWaves + wave;
  const int b = wid / wps;
  if (b >= batch) return;
  const int slot = wid - b * wps;
  int grp = slot * gpw;
  const int grp_end = min(grp + gpw, groups_per_sample);

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
  // uniform base pointer + 32-bit per-lane byte offset (the host checks that every operand
  // segment and the output are smaller than 4 GiB)
  auto load_tile = [&](uint32_t arow, int kt, float4 (&a)[4]) {
    const int s = RAGGED ? 0 : (kt < t0 ? 0 : (kt < t0 + t1 ? 1 : 2));
    const int k0 = kBK * (kt - (s == 0 ? 0 : (s == 1 ? t0 : t0 + t1)));
    const float* base = s == 0 ? g.seg[0].a : (s == 1 ? g.seg[1].a : g.seg[2].a);
    const uint32_t lda = static_cast<uint32_t>(s == 0 ? g.seg[0].lda : (s == 1 ? g.seg[1].lda : g.seg[2].lda));
    const char* sb = reinterpret_cast<const char*>(base + k0);
    const uint32_t voff = (arow * lda + kLaneFloats * static_cast<uint32_t>(lh)) * 4u;
    if (RAGGED && kt == NKT - 1) {
      const int left = g.seg[0].k - kBK * (NKT - 1) - 4 * lh;   // floats of this lane's k phase left in the tile
      const uint32_t first = static_cast<uint32_t>(arow * lda) * 4u;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool ok = 8 * c < left;
        const float4 v = *reinterpret_cast<const float4*>(sb + (ok ? voff + chunk_bytes(c) : first));
        a[c] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
      }
      return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = *reinterpret_cast<const float4*>(sb + voff + chunk_bytes(c));
  };

  f32x16 acc[NT];
  const float4* wl = reinterpret_cast<const float4*>(Ws) + lane;
  auto multiply = [&](int kt, const float4 (&a)[4]) {
    if constexpr (BF16) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 ap;
        const float av[8] = {a[2 * s2].x,     a[2 * s2].y,     a[2 * s2].z,     a[2 * s2].w,
                             a[2 * s2 + 1].x, a[2 * s2 + 1].y, a[2 * s2 + 1].z, a[2 * s2 + 1].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) ap[j] = static_cast<__bf16>(av[j]);
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
  float* stage = res_smem + (BF16 ? 16 : 32) * NKT * BN + wave * (16 * 32);
  const int srow = lane >> 3, scol = 4 * (lane & 7);
  float bias_v[NT], cs[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    bias_v[n] = g.bias ? g.bias[static_cast<int64_t>(b) * g.bias_stride + 32 * n + li] : 0.0f;
    cs[n] = 0.0f;
  }
  auto epilogue = [&](auto full_tag, int row0) {
    constexpr bool FULL = decltype(full_tag)::value;
    // this lane's first pick-up row as a 32-bit byte offset; the steps between rows are uniform
    const uint32_t voff =
        ((static_cast<uint32_t>(b) * static_cast<uint32_t>(g.out_rows) + static_cast<uint32_t>(row0 + srow)) * ldo +
         static_cast<uint32_t>(g.out_col0 + scol)) * 4u;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
          const int r = 8 * h + r8;
          const int rl = (r & 3) + 8 * (r8 >> 2) + 4 * lh;   // row within this half of the tile
          const float v = activate(acc[n][r] + bias_v[n], g.relu);
          stage[rl * 32 + li] = v;
          if (FULL || row0 + 16 * h + rl < g.n_rows) cs[n] += v;
        }
        if (has_out) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(&stage[(8 * i + srow) * 32 + scol]);
            char* orow = reinterpret_cast<char*>(g.out + static_cast<uint32_t>(16 * h + 8 * i) * ldo);
            if (FULL || row0 + 16 * h + 8 * i + srow < g.n_rows)
              *reinterpret_cast<float4*>(orow + voff + 128u * n) = v;
          }
        }
      }
    }
  };

  // Gathered tiles: the node's table row (12 edge positions) is fetched at the top of its group; the
  // (up to) three source rows of quadrant q are requested one K tile ahead -- the first two into
  // `gbuf`, the third into the operand buffer a0, which is idle during these tiles -- and summed
  // into a1 right before the MFMAs of that tile.
  int4 tab[3];
  float4 gbuf[2][4];
  auto load_table = [&](int grp_) {
    int row = grp_ * kRowGroup + li;
    row = row < g.n_rows ? row : g.n_rows - 1;
    const int4* tp = reinterpret_cast<const int4*>(g.gather_table) + static_cast<int64_t>(row) * 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) tab[i] = tp[i];
  };
  auto tab_at = [&](int i) {   // i = 3 * q + j, compile-time after unrolling
    const int4 t = tab[i >> 2];
    return (i & 3) == 0 ? t.x : ((i & 3) == 1 ? t.y : ((i & 3) == 2 ? t.z : t.w));
  };
  auto gather_issue = [&](int q, float4 (&third)[4]) {
    const char* sb = reinterpret_cast<const char*>(g.gather_src[q]);
    const uint32_t base = static_cast<uint32_t>(b) * static_cast<uint32_t>(g.gather_rows[q]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int id = tab_at(3 * q + j);
      const uint32_t voff = ((base + static_cast<uint32_t>(id > 0 ? id : 0)) * 32u + kLaneFloats * static_cast<uint32_t>(lh)) * 4u;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(sb + voff + chunk_bytes(c));
        if (j < 2)
          gbuf[j][c] = v;
        else
          third[c] = v;
      }
    }
  };
  auto gather_reduce = [&](int q, const float4 (&third)[4], float4 (&a)[4]) {
    const bool ok0 = tab_at(3 * q) >= 0, ok1 = tab_at(3 * q + 1) >= 0, ok2 = tab_at(3 * q + 2) >= 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok0) s = gbuf[0][c];
      if (ok1) { s.x += gbuf[1][c].x; s.y += gbuf[1][c].y; s.z += gbuf[1][c].z; s.w += gbuf[1][c].w; }
      if (ok2) { s.x += third[c].x; s.y += third[c].y; s.z += third[c].z; s.w += third[c].w; }
      a[c] = make_float4(s.x * g.gather_scale, s.y * g.gather_scale, s.z * g.gather_scale, s.w * g.gather_scale);
    }
  };

  uint32_t cur = lane_row(grp);
  float4 a0[4], a1[4];
  load_tile(cur, 0, a0);
  while (true) {
    const bool has_next = grp + 1 < grp_end;
    const uint32_t nxt = lane_row(has_next ? grp + 1 : grp);   // a finished wave re-reads its own rows: harmless
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    if constexpr (GATHER) {
      load_table(grp);
#pragma unroll
      for (int kt = 0; kt < NX; kt += 2) {
        load_tile(cur, kt + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
        multiply(kt, a0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < NX)
          load_tile(cur, kt + 2, a0);
        else
          gather_issue(0, a0);
        __builtin_amdgcn_sched_barrier(0);
        multiply(kt + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        gather_reduce(q, a0, a1);
        __builtin_amdgcn_sched_barrier(0);
        if (q < 3)
          gather_issue(q + 1, a0);
        else
          load_tile(nxt, 0, a0);
        __builtin_amdgcn_sched_barrier(0);
        multiply(NX + q, a1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < NKT; kt += 2) {
        load_tile(cur, kt + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
        multiply(kt, a0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < NKT)
          load_tile(cur, kt + 2, a0);
        else
          load_tile(nxt, 0, a0);
        __builtin_amdgcn_sched_barrier(0);
        multiply(kt + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // Take delivery of the next group's first tile now, while only loads are outstanding: a wait
    // placed after the stores below would have to cover them as well (loads and stores share one
    // counter), and the wave would sit out its own write latency at the top of every group.
    asm volatile("" : "+v"(a0[0].x), "+v"(a0[1].x), "+v"(a0[2].x), "+v"(a0[3].x));
    const int row0 = grp * kRowGroup;
    if (row0 + kRowGroup <= g.n_rows)
      epilogue(std::true_type{}, row0);
    else
      epilogue(std::false_type{}, row0);
    if (!has_next) break;
    ++grp;
    cur = nxt;
  }
  if (g.colsum != nullptr) {
    float* dst = g.colsum + (static_cast<int64_t>(b) * g.colsum_tiles + g.colsum_tile0 + slot) * g.colsum_ld +
                 g.out_col0 + li;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      cs[n] += __shfl_xor(cs[n], 32);
      if (lh == 0) dst[32 * n] = cs[n];
    }
  }
}

// Any shape, one thread per output element (used only when k or n_out break the tile rules).
__global__ void k_dense_naive(const float* __restrict__ x, int64_t rows, int k,
                              const float* __restrict__ w, const float* __restrict__ bias, int n_out,
                              int relu, float* __restrict__ out) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= rows * n_out) return;
  const int64_t r = idx / n_out;
  const int c = static_cast<int>(idx % n_out);
  float s = 0.0f;
  for (int j = 0; j < k; ++j) s = fmaf(x[r * k + j], w[static_cast<int64_t>(j) * n_out + c], s);
  if (bias) s += bias[c];
  out[idx] = activate(s, relu);
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
  if (g.colsum != nullptr) {   // one partial row per workgroup: the four waves' sums added in order
    __shared__ float red[4 * BN];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      cs[n] += __shfl_xor(cs[n], 32);
      if (lh == 0) red[wave * BN + 32 * n + li] = cs[n];
    }
    __syncthreads();
    if (tid < BN) {
      const float s = ((red[tid] + red[BN + tid]) + red[2 * BN + tid]) + red[3 * BN + tid];
      g.colsum[(b * g.colsum_tiles + g.colsum_tile0 + blockIdx.x) * g.colsum_ld + g.out_col0 + tid] = s;
    }
  }
}

// Matmul arithmetic of the feature kernels (gunet_set_matmul_mode) and the GEMM kernel choice
// (gunet_set_kernel_choice).
int g_matmul_mode = GUNET_MATMUL_F32;
int g_kernel_choice = GUNET_KERNELS_AUTO;

inline int mode_pieces(int mode) {
  return mode == GUNET_MATMUL_SPLIT3 ? 3 : mode == GUNET_MATMUL_SPLIT2 ? 2 : mode == GUNET_MATMUL_BF16 ? 1 : 0;
}

inline int gemm_ktiles(const GemmArgs& g) {
  int n = 0;
  for (int s = 0; s < g.n_seg; ++s) n += (g.seg[s].k + kBK - 1) / kBK;
  return n;
}

// Bytes of weight image a launch of `ktiles` K tiles and `nt` column tiles needs in this mode.
inline size_t wpack_bytes(int ktiles, int nt) {
  return static_cast<size_t>(ktiles) * 3 * 4 * 32 * nt * sizeof(uint4);
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

inline int device_cu_count() {
  static int n_cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (n_cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    n_cus[dev] = n;
  }
  return n_cus[dev];
}

// The weights-resident kernels take launches whose operand segments are whole K tiles, with
// identity row mapping, when the weight image (float32, or bf16 in GUNET_MATMUL_BF16 mode) fits LDS
// next to 2 KiB of store staging per wave.
inline bool bf16_mode() { return g_matmul_mode == GUNET_MATMUL_BF16; }

inline size_t resident_lds_bytes(int nt, int nkt, bool bf16) {
  return (bf16 ? 2 : 4) * static_cast<size_t>(32 * nkt) * 32 * nt + sizeof(float) * kResWaves * 16 * 32;
}

inline bool resident_ragged(const GemmArgs& g) {
  return g.n_seg == 1 && g.gather_table == nullptr && g.seg[0].k % kBK != 0 && g.seg[0].k % 4 == 0 &&
         g.seg[0].k > 3 * kBK && g.seg[0].k < 4 * kBK;
}

inline bool resident_ok(const GemmArgs& g, int nt, int64_t batch) {
  if (g.row_map != nullptr) return false;
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
  if (resident_lds_bytes(nt, kt, bf16) > kResMaxLds) return false;
  for (int s = 0; s < g.n_seg; ++s)   // 32-bit byte offsets into every operand segment and the output
    if (batch * g.n_rows * g.seg[s].lda * 4 >= (int64_t{1} << 32)) return false;
  if (batch * (g.out_rows ? g.out_rows : g.n_rows) * g.ldo * 4 >= (int64_t{1} << 32)) return false;
  // Below about two row groups per wave the one-off copy of the weights into every workgroup's LDS
  // costs more than the tiled kernel's barriers (measured crossover near 200 k rows); the gathering
  // form also saves the aggregation pass, which moves its break-even down to half a group per wave.
  if (g_kernel_choice == GUNET_KERNELS_TILED) return false;
  if (g_kernel_choice == GUNET_KERNELS_AUTO) {
    const int64_t groups = batch * ceil_div(g.n_rows, kRowGroup);
    const int64_t waves = static_cast<int64_t>(device_cu_count()) * kResWaves;
    if (g.gather_table == nullptr ? groups < 2 * waves : 2 * groups < waves) return false;
  }
  if (g.gather_table != nullptr) {   // the last segment is the gathered one; 32-bit offsets into the sources
    if (g.n_seg < 2 || g.seg[g.n_seg - 1].k != 128) return false;
    for (int q = 0; q < 4; ++q)
      if (batch * g.gather_rows[q] * 128 >= (int64_t{1} << 32)) return false;
  }
  return batch * ceil_div(g.n_rows, kRowGroup) < (1 << 30);
}

// Waves per workgroup of the resident kernel a launch gets.
inline int resident_block_waves(const GemmArgs& g, int nt) {
  return (g.gather_table != nullptr && nt == 4) ? kResGatherWaves : kResWaves;
}

// Work split of the resident kernel: every sample's row groups in runs of gpw, wps waves per sample.
struct ResidentSplit {
  int groups, wps, gpw;
};

inline ResidentSplit resident_split(int n_rows, int64_t batch, int block_waves) {
  ResidentSplit r;
  r.groups = static_cast<int>(ceil_div(n_rows, kRowGroup));
  const int64_t waves = static_cast<int64_t>(device_cu_count()) * block_waves;
  const int per_sample = static_cast<int>(std::max<int64_t>(1, waves / batch));
  r.gpw = static_cast<int>(ceil_div(r.groups, std::min(per_sample, r.groups)));
  r.wps = static_cast<int>(ceil_div(r.groups, r.gpw));
  return r;
}

// Which kernel a launch gets: a weights-resident one where it applies (float32, or bf16 operands in
// GUNET_MATMUL_BF16 mode); else, in the bf16 / split-bf16 modes and with a weight-image buffer at
// hand, the tiled split kernel; else the tiled float32 one.
inline bool takes_resident(const GemmArgs& g, int nt, int64_t batch, const void* wpack) {
  (void)wpack;
  return resident_ok(g, nt, batch);
}
inline bool takes_split(const void* wpack) { return mode_pieces(g_matmul_mode) > 0 && wpack != nullptr; }

// Rows of column-sum partials per sample that a launch writes (from colsum_tile0 on).
inline int gemm_colsum_rows(const GemmArgs& g, int nt, int64_t batch, const void* wpack) {
  if (g.n_rows == 0) return 0;
  if (takes_resident(g, nt, batch, wpack)) return resident_split(g.n_rows, batch, resident_block_waves(g, nt)).wps;
  return static_cast<int>(ceil_div(g.n_rows, kBM));
}

template <int NT, int NKT, bool GATHER, bool BF16, bool RAGGED = false>
int launch_resident_inst(const GemmArgs& g, int64_t batch, hipStream_t stream) {
  static bool raised[64] = {};   // per device: the kernel's dynamic-LDS limit is raised
  const size_t lds = resident_lds_bytes(NT, NKT, BF16);
  int dev = 0;
  GUNET_HIP_TRY(hipGetDevice(&dev));
  GUNET_REQUIRE(dev >= 0 && dev < 64, "launch_gemm: device index %d", dev);
  if (!raised[dev]) {
    GUNET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_resident<NT, NKT, GATHER, BF16, RAGGED>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kResMaxLds)));
    raised[dev] = true;
  }
  constexpr int kWaves = (GATHER && NT == 4) ? kResGatherWaves : kResWaves;
  const ResidentSplit sp = resident_split(g.n_rows, batch, kWaves);
  const int blocks = static_cast<int>(ceil_div(batch * sp.wps, kWaves));
  k_gemm_resident<NT, NKT, GATHER, BF16, RAGGED><<<blocks, 64 * kWaves, lds, stream>>>(g, sp.groups, sp.wps, sp.gpw,
                                                                                     static_cast<int>(batch));
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

int launch_gemm_resident(const GemmArgs& g, int nt, int64_t batch, hipStream_t stream) {
  const int kt = gemm_ktiles(g);
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
  const bool resident = resident_ok(g, nt, batch), split = !resident && takes_split(wpack);
  double flops = 0.0;
  for (int s = 0; s < g.n_seg; ++s) flops += 2.0 * batch * g.n_rows * g.seg[s].k * (32.0 * nt);
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
  switch (nt) {
    case 1: k_gemm_f32<1><<<grid, kThreads, 0, stream>>>(g); break;
    case 2: k_gemm_f32<2><<<grid, kThreads, 0, stream>>>(g); break;
    case 3: k_gemm_f32<3><<<grid, kThreads, 0, stream>>>(g); break;
    case 4: k_gemm_f32<4><<<grid, kThreads, 0, stream>>>(g); break;
    default: set_error("launch_gemm: bad column tile count %d", nt); return GUNET_ERR_INVALID;
  }
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

// Several problems with the same column-tile count and batch as one launch of the tiled float32
// kernel (the bf16 / split modes, and single problems, go through launch_gemm one by one).  The
// caller has set colsum_tiles / colsum_tile0 / colsum_ld / out_rows.
int launch_gemm_group(const GemmArgs* gs, int n, int nt, int64_t batch, void* wpack, hipStream_t stream) {
  int live[kGemmGroup], n_live = 0;
  for (int i = 0; i < n; ++i)
    if (gs[i].n_rows > 0) live[n_live++] = i;
  bool tiled = n_live > 1 && n_live <= kGemmGroup && batch > 0 && !takes_split(wpack);
  for (int i = 0; i < n_live && tiled; ++i) tiled = !resident_ok(gs[live[i]], nt, batch);
  if (!tiled) {
    for (int i = 0; i < n_live; ++i) GUNET_TRY(launch_gemm(gs[live[i]], nt, batch, wpack, stream));
    return GUNET_OK;
  }
  GemmGroup gg = {};
  unsigned blocks = 0;
  double flops = 0.0;
  for (int i = 0; i < kGemmGroup; ++i) {
    if (i < n_live) {
      gg.p[i] = gs[live[i]];
      GemmArgs& g = gg.p[i];
      if (g.out_rows == 0) g.out_rows = g.n_rows;
      if (g.colsum_ld == 0) g.colsum_ld = 32 * nt;
      blocks += static_cast<unsigned>(ceil_div(g.n_rows, kBM));
      for (int s = 0; s < g.n_seg; ++s) flops += 2.0 * batch * g.n_rows * g.seg[s].k * (32.0 * nt);
    }
    gg.block_end[i] = blocks;
  }
  dim3 grid(blocks, static_cast<unsigned>(batch));
  ProfScope prof(kProfGemmTiled, stream, flops);
  switch (nt) {
    case 1: k_gemm_f32_group<1><<<grid, kThreads, 0, stream>>>(gg); break;
    case 2: k_gemm_f32_group<2><<<grid, kThreads, 0, stream>>>(gg); break;
    case 3: k_gemm_f32_group<3><<<grid, kThreads, 0, stream>>>(gg); break;
    case 4: k_gemm_f32_group<4><<<grid, kThreads, 0, stream>>>(gg); break;
    default: set_error("launch_gemm_group: bad column tile count %d", nt); return GUNET_ERR_INVALID;
  }
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

// ---------------------------------------------------------------- GeoQuadrantGCN pieces
//
// Kernels of one GN layer, templated on UT = units / 128: U = 128*UT output columns per node,
// u = 32*UT per direction.  UT = 1 is the shipped checkpoint's width, UT = 2 the 256-wide variant.

constexpr int kEdgeGroups = 8;   // 32-edge groups per wave
constexpr int kEdgesPerBlock = 4 * 32 * kEdgeGroups;
constexpr int kMaxCin = 1024;    // widest layer input the workspace is sized for
constexpr int kMaxUT = 2;

struct LayerPtrs {
  const float* node_emb_b[4];
  const float* edge_w[4];
  const float* edge_b[4];
  const float* glob_w[4];
  const float* glob_b[4];
  const float* node_w;
  const float* node_b;
  const float* globt_w;
  const float* globt_b;
  int c_in, f_e, g;
  float gsn, gse;
};

constexpr int kSmallThreads = 1024;   // the two per-sample kernels

// y[4*cq .. 4*cq+3] = sum over rows r = sub, sub + S, ... < n_rows of v[r] * W[r][4*cq ..] for W
// row-major with U columns, S = 1024 / (U/4) row subsets; the caller adds the S subsets in
// order.  Four rows are in flight per thread.
template <int U>
__device__ inline float4 matvec_partial(const float* v, const float* w, int n_rows, int sub, int cq, int first = 0,
                                        float4 s = make_float4(0.f, 0.f, 0.f, 0.f)) {
  constexpr int QC = U / 4, S = kSmallThreads / QC;
  const float4* w4 = reinterpret_cast<const float4*>(w) + cq;
  int r = sub + first * S;   // `s` already holds this thread's rows sub, sub + S, ... below r
  for (; r + 3 * S < n_rows; r += 4 * S) {
    const float4 a0 = w4[static_cast<int64_t>(r) * QC];
    const float4 a1 = w4[static_cast<int64_t>(r + S) * QC];
    const float4 a2 = w4[static_cast<int64_t>(r + 2 * S) * QC];
    const float4 a3 = w4[static_cast<int64_t>(r + 3 * S) * QC];
    const float v0 = v[r], v1 = v[r + S], v2 = v[r + 2 * S], v3 = v[r + 3 * S];
    s.x = fmaf(v3, a3.x, fmaf(v2, a2.x, fmaf(v1, a1.x, fmaf(v0, a0.x, s.x))));
    s.y = fmaf(v3, a3.y, fmaf(v2, a2.y, fmaf(v1, a1.y, fmaf(v0, a0.y, s.y))));
    s.z = fmaf(v3, a3.z, fmaf(v2, a2.z, fmaf(v1, a1.z, fmaf(v0, a0.z, s.z))));
    s.w = fmaf(v3, a3.w, fmaf(v2, a2.w, fmaf(v1, a1.w, fmaf(v0, a0.w, s.w))));
  }
  for (; r < n_rows; r += S) {
    const float4 a0 = w4[static_cast<int64_t>(r) * QC];
    const float v0 = v[r];
    s.x = fmaf(v0, a0.x, s.x);
    s.y = fmaf(v0, a0.y, s.y);
    s.z = fmaf(v0, a0.z, s.z);
    s.w = fmaf(v0, a0.w, s.w);
  }
  return s;
}

// The first NPRE of a thread's rows of that product, fetched ahead of the vector: matvec_head_load
// when the weights are known, matvec_head_apply (the same fused multiply-adds in the same order as
// matvec_partial would do) once the vector is there; matvec_partial(.., first = NPRE, s) finishes.
template <int U, int NPRE>
__device__ inline void matvec_head_load(const float* w, int n_rows, int sub, int cq, float4 (&pre)[NPRE]) {
  constexpr int QC = U / 4, S = kSmallThreads / QC;
  const float4* w4 = reinterpret_cast<const float4*>(w) + cq;
#pragma unroll
  for (int j = 0; j < NPRE; ++j) {
    const int r = sub + j * S;
    pre[j] = r < n_rows ? w4[static_cast<int64_t>(r) * QC] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <int U, int NPRE>
__device__ inline float4 matvec_head_apply(const float* v, int n_rows, int sub, const float4 (&pre)[NPRE]) {
  constexpr int S = kSmallThreads / (U / 4);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < NPRE; ++j) {
    const int r = sub + j * S;
    if (r < n_rows) {
      const float x = v[r];
      s.x = fmaf(x, pre[j].x, s.x);
      s.y = fmaf(x, pre[j].y, s.y);
      s.z = fmaf(x, pre[j].z, s.z);
      s.w = fmaf(x, pre[j].w, s.w);
    }
  }
  return s;
}

// Adds `n` partials of column `col` (row stride `ld`) in order.
__device__ inline float combine_rows(const float* part, int n, int ld, int col) {
  float s = part[col];
  for (int g = 1; g < n; ++g) s += part[g * ld + col];
  return s;
}

// Per sample: the chained global transforms (hybrid_gnn.py:98-99), folded into the bias each
// direction's edge MLP sees; the node MLP's global columns folded into its bias (:122-125);
// and the four node-embedding biases laid side by side.  Everything that does not depend on the
// chain (the node-MLP bias, the constant receiver term of the upsampling form, and the weights the
// later chain steps need) is fetched first, so the chain itself only waits on LDS.
// Split in two so that a caller with other work to do first (k_global_bridge) can have the weight
// loads in flight behind it: prologue_prefetch fetches what does not depend on the global vector
// (DEEP: also the first rows of the two products that start from it), prologue_compute does the rest.
constexpr int kPreNode = 4;    // rows of the node-MLP bias product fetched ahead, per thread
constexpr int kPreChain = 4;   // rows of the first chain step fetched ahead, per thread

template <int UT>
struct ProloguePre {
  static constexpr int u = 32 * UT, U = 128 * UT;
  static constexpr int CG = kSmallThreads / u, CR = u / CG;   // chain: CG row groups, CR rows each (u -> u steps)
  static constexpr int EG = kSmallThreads / U, ER = u / EG;   // edge bias: EG row groups, ER rows each
  float wq[3][CR];
  float we[ER];
  float e_const;
  float4 node_w[kPreNode];
  float chain_w[kPreChain];
};

template <int UT, bool DEEP>
__device__ __forceinline__ void prologue_prefetch(const LayerPtrs& p, int relu, int fold_receiver, int64_t b,
                                                  float* __restrict__ nbias, ProloguePre<UT>& pre) {
  constexpr int U = 128 * UT, u = 32 * UT;
  constexpr int QC = U / 4;
  constexpr int CG = ProloguePre<UT>::CG, CR = ProloguePre<UT>::CR, EG = ProloguePre<UT>::EG, ER = ProloguePre<UT>::ER;
  const int tid = threadIdx.x;
  const int cgrp = tid / u, ccol = tid % u;
#pragma unroll
  for (int q = 1; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < CR; ++i) pre.wq[q - 1][i] = p.glob_w[q][(cgrp + CG * i) * u + ccol];
  const int egrp = tid / U, ecol = tid % U;
  const int eq = ecol / u, ej = ecol % u;
  {
    const float* w = p.edge_w[eq] + static_cast<int64_t>(p.f_e + 2 * u) * u;
#pragma unroll
    for (int i = 0; i < ER; ++i) pre.we[i] = w[(egrp + EG * i) * u + ej];
  }
  if (DEEP) {
    matvec_head_load<U, kPreNode>(p.node_w + static_cast<int64_t>(p.c_in + U) * U, p.g, tid / QC, tid % QC, pre.node_w);
#pragma unroll
    for (int i = 0; i < kPreChain; ++i) {
      const int k = cgrp + CG * i;
      pre.chain_w[i] = k < p.g ? p.glob_w[0][k * u + ccol] : 0.0f;
    }
  }
  pre.e_const = 0.0f;
  if (tid < U) {
    if (p.edge_b[eq]) pre.e_const = p.edge_b[eq][ej];
    if (fold_receiver) {
      // every receiver is a zero input row: its embedding is act(bias), a constant of the layer
      const float* wr = p.edge_w[eq] + static_cast<int64_t>(p.f_e + u) * u;
      float r = 0.0f;
      for (int k = 0; k < u; ++k)
        r = fmaf(activate(p.node_emb_b[eq] ? p.node_emb_b[eq][k] : 0.0f, relu), wr[k * u + ej], r);
      pre.e_const += r;
    }
    if (b == 0) nbias[tid] = p.node_emb_b[eq] ? p.node_emb_b[eq][ej] : 0.0f;
  }
}

// `gl` (LDS, [g]) holds the sample's global vector; it may still be in the making when this starts
// (the first barrier below covers it).  `part` is 4096 floats of LDS scratch, `gq` [4][u].
template <int UT, bool DEEP>
__device__ __forceinline__ void prologue_compute(const LayerPtrs& p, const float* gl, int relu, int64_t b, float* part,
                                                 float* gq, float* __restrict__ ebias, float* __restrict__ gbias,
                                                 const ProloguePre<UT>& pre) {
  constexpr int U = 128 * UT, u = 32 * UT;
  constexpr int QC = U / 4, S = kSmallThreads / QC;     // matvec: column quads x row subsets
  constexpr int CG = ProloguePre<UT>::CG, CR = ProloguePre<UT>::CR, EG = ProloguePre<UT>::EG, ER = ProloguePre<UT>::ER;
  const int tid = threadIdx.x;
  const int cgrp = tid / u, ccol = tid % u;
  const int egrp = tid / U, ecol = tid % U;
  const int eq = ecol / u;
  __syncthreads();

  {  // node-MLP bias: global . W[c_in + U .., :]
    const int sub = tid / QC, cq = tid % QC;
    const float* w = p.node_w + static_cast<int64_t>(p.c_in + U) * U;
    const float4 s = DEEP ? matvec_partial<U>(gl, w, p.g, sub, cq, kPreNode,
                                              matvec_head_apply<U, kPreNode>(gl, p.g, sub, pre.node_w))
                          : matvec_partial<U>(gl, w, p.g, sub, cq);
    *reinterpret_cast<float4*>(&part[sub * U + 4 * cq]) = s;
  }
  __syncthreads();
  if (tid < U) {
    float s = combine_rows(part, S, U, tid);
    if (p.node_b) s += p.node_b[tid];
    gbias[b * U + tid] = s;
  }
  __syncthreads();

  {  // chain step 0: [g] -> u
    float s = 0.0f;
    int k = cgrp;
    if (DEEP) {
#pragma unroll
      for (int i = 0; i < kPreChain; ++i, k += CG)
        if (k < p.g) s = fmaf(gl[k], pre.chain_w[i], s);
    }
    for (; k < p.g; k += CG) s = fmaf(gl[k], p.glob_w[0][k * u + ccol], s);
    part[cgrp * u + ccol] = s;
  }
  __syncthreads();
  if (tid < u) {
    float s = combine_rows(part, CG, u, tid);
    if (p.glob_b[0]) s += p.glob_b[0][tid];
    gq[tid] = activate(s, relu);
  }
  __syncthreads();
  for (int q = 1; q < 4; ++q) {   // chain steps 1..3: u -> u with the preloaded weights
    {
      const float* src = gq + (q - 1) * u;
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < CR; ++i) s = fmaf(src[cgrp + CG * i], pre.wq[q - 1][i], s);
      part[cgrp * u + ccol] = s;
    }
    __syncthreads();
    if (tid < u) {
      float s = combine_rows(part, CG, u, tid);
      if (p.glob_b[q]) s += p.glob_b[q][tid];
      gq[q * u + tid] = activate(s, relu);
    }
    __syncthreads();
  }
  {  // edge bias of direction eq: chained global of that direction . W_global + constants
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < ER; ++i) s = fmaf(gq[eq * u + egrp + EG * i], pre.we[i], s);
    part[egrp * U + ecol] = s;
  }
  __syncthreads();
  if (tid < U) ebias[b * U + tid] = combine_rows(part, EG, U, tid) + pre.e_const;
}

template <int UT>
__global__ __launch_bounds__(kSmallThreads) void k_layer_prologue(LayerPtrs p,
                                                                  const float* __restrict__ global_in,
                                                                  int relu, int fold_receiver,
                                                                  float* __restrict__ nbias,
                                                                  float* __restrict__ ebias,
                                                                  float* __restrict__ gbias) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  float* part = sh;                      // 4096 floats, reused by every reduction
  float* gl = part + 4096;               // [g]
  float* gq = gl + p.g;                  // [4][u]
  const int64_t b = blockIdx.x;
  for (int k = threadIdx.x; k < p.g; k += blockDim.x) gl[k] = global_in[b * p.g + k];
  ProloguePre<UT> pre;
  prologue_prefetch<UT, false>(p, relu, fold_receiver, b, nbias, pre);
  prologue_compute<UT, false>(p, gl, relu, b, part, gq, ebias, gbias, pre);
}

struct EdgeArgs {
  const int32_t* edge_index[4];
  const float* efeat[4];    // [B, E_q, f_e] or null
  const float* edge_w[4];   // [f_e + 3u, u]
  float* out[4];            // [B, E_q, u]
  int n_edges[4];
  int n_nodes;
  int f_e;
  const float* xq;          // [B, N, U] node embeddings, direction q in columns u*q .. u*q+u-1
  const float* ebias;       // [B, 4, u]
  float* partial;           // [B, 4, gridDim.x, u] column sums of the messages
  int relu;
  int sender_only;          // receivers contribute through the bias (graph upsampling)
  float out_scale;          // stored message = out_scale * message (the column sums stay unscaled)
};

// Edge MLP, 32 edges per wave step: the A operand row of edge e is
// [edge features | embedding of the sender | embedding of the receiver], gathered straight into
// MFMA registers as float4 pieces (half-wave h takes floats 8c+4h..8c+4h+3 of every 8-float
// chunk).  The weight rows (zero-padded to whole chunks) sit in LDS for the workgroup.  Each wave
// walks kEdgeGroups groups; for UT = 1 the indices of group g+2 and the rows of group g+1 are
// requested before the MFMAs of group g, and the stores of group g are issued after those
// requests, so a wave always has gathers in flight behind its matrix work.
template <int FE_CHUNKS, bool SENDER_ONLY, int UT>
struct EdgeRows {
  static constexpr int CH = FE_CHUNKS + (SENDER_ONLY ? 4 : 8) * UT;
  float4 v[CH];
};

template <int FE_CHUNKS, bool FE_VEC, bool SENDER_ONLY, int UT>
__device__ inline void gather_edge_rows(const float* __restrict__ xq, const float* __restrict__ ef, int f_e,
                                        int64_t e, int2 sr, int lh, EdgeRows<FE_CHUNKS, SENDER_ONLY, UT>& rows) {
  constexpr int U = 128 * UT, SC = 4 * UT;   // sender / receiver chunks of 8 floats
  if (FE_CHUNKS > 0) {
    if (FE_VEC) {
      const float4* pf = reinterpret_cast<const float4*>(ef + e * f_e) + lh;
#pragma unroll
      for (int c = 0; c < FE_CHUNKS; ++c) rows.v[c] = pf[2 * c];
    } else {
#pragma unroll
      for (int c = 0; c < FE_CHUNKS; ++c) {
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = 8 * c + 4 * lh + j;
          t[j] = (ef != nullptr && k < f_e) ? ef[e * f_e + k] : 0.0f;
        }
        rows.v[c] = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
  }
  const float4* ps = reinterpret_cast<const float4*>(xq + static_cast<int64_t>(sr.x) * U) + lh;
#pragma unroll
  for (int c = 0; c < SC; ++c) rows.v[FE_CHUNKS + c] = ps[2 * c];
  if (!SENDER_ONLY) {
    const float4* pr = reinterpret_cast<const float4*>(xq + static_cast<int64_t>(sr.y) * U) + lh;
#pragma unroll
    for (int c = 0; c < SC; ++c) rows.v[FE_CHUNKS + SC + c] = pr[2 * c];
  }
}

template <int FE_CHUNKS, bool FE_VEC, bool SENDER_ONLY, int UT>
__global__ __launch_bounds__(kThreads) void k_edge_mlp(EdgeArgs a) {
  using Rows = EdgeRows<FE_CHUNKS, SENDER_ONLY, UT>;
  constexpr int CH = Rows::CH;
  constexpr int U = 128 * UT, u = 32 * UT;
  constexpr bool PIPE = UT == 1;   // two row buffers fit the register file only at u = 32
  __shared__ float Ws[CH * 8 * u];
  __shared__ float red[4 * u];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int q = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int n_edges = a.n_edges[q];
  const int64_t block_base = static_cast<int64_t>(blockIdx.x) * kEdgesPerBlock;
  float colsum[UT];
#pragma unroll
  for (int t = 0; t < UT; ++t) colsum[t] = 0.0f;

  if (block_base < n_edges) {   // uniform over the workgroup
    const float* w = a.edge_w[q];
    for (int i = tid; i < CH * 8 * u; i += kThreads) {
      const int row = i / u, col = i % u;
      float v;
      if (row < 8 * FE_CHUNKS)
        v = row < a.f_e ? w[row * u + col] : 0.0f;
      else
        v = w[(a.f_e + row - 8 * FE_CHUNKS) * u + col];
      Ws[i] = v;
    }
    __syncthreads();
    float bias[UT];
#pragma unroll
    for (int t = 0; t < UT; ++t) bias[t] = a.ebias[(b * 4 + q) * u + 32 * t + li];
    const int2* ei = reinterpret_cast<const int2*>(a.edge_index[q]);
    const float* xq = a.xq + b * a.n_nodes * static_cast<int64_t>(U) + u * q;
    const float* ef = a.efeat[q] ? a.efeat[q] + b * n_edges * static_cast<int64_t>(a.f_e) : nullptr;
    float* out = a.out[q] + b * n_edges * static_cast<int64_t>(u);
    const int64_t wave_base = block_base + static_cast<int64_t>(wave) * kEdgeGroups * 32;
    auto edge_of = [&](int grp) {
      const int64_t e = wave_base + grp * 32 + li;
      return e < n_edges ? e : static_cast<int64_t>(n_edges) - 1;   // clamped lanes are masked at the store
    };
    // the matrix work and the stores of one group whose rows are in `cur`
    auto multiply = [&](auto full_tag, int64_t e0, const Rows& cur) {
      constexpr bool FULL = decltype(full_tag)::value;
      f32x16 acc[UT];
#pragma unroll
      for (int t = 0; t < UT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const float a4[4] = {cur.v[c].x, cur.v[c].y, cur.v[c].z, cur.v[c].w};
#pragma unroll
        for (int t = 0; t < UT; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], Ws[(8 * c + 4 * lh + j) * u + 32 * t + li],
                                                          acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = e0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (FULL || row < n_edges) {
#pragma unroll
          for (int t = 0; t < UT; ++t) {
            const float v = activate(acc[t][r] + bias[t], a.relu);
            out[row * u + 32 * t + li] = v * a.out_scale;
            colsum[t] += v;
          }
        }
      }
    };
    if constexpr (PIPE) {
      // One group: request the rows of group grp+1 into `nxt` (its indices were fetched a step
      // earlier) and the indices of group grp+2, then multiply the rows of group grp held in
      // `cur`.  The two row buffers swap roles every step (no register copies, which would have
      // to wait for the loads they copy).  FULL = every edge of the wave's groups exists: no
      // clamps, no masks and no branches, so the whole sweep is one basic block and the requests
      // stay where they are written, ahead of the matrix work of the previous group.
      auto step = [&](auto full_tag, int grp, Rows& cur, Rows& nxt, int2& sr_nxt, int2& sr_nxt2) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int64_t e0 = wave_base + grp * 32;
        if (grp + 1 < kEdgeGroups && (FULL || e0 + 32 < n_edges)) {
          if (grp + 2 < kEdgeGroups) sr_nxt2 = ei[FULL ? e0 + 64 + li : edge_of(grp + 2)];
          gather_edge_rows<FE_CHUNKS, FE_VEC, SENDER_ONLY, UT>(xq, ef, a.f_e, FULL ? e0 + 32 + li : edge_of(grp + 1),
                                                               sr_nxt, lh, nxt);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the requests above the matrix work
        multiply(full_tag, e0, cur);
        __builtin_amdgcn_sched_barrier(0);
      };
      Rows r0, r1;
      int2 sr_a, sr_b;
      if (wave_base + kEdgeGroups * 32 <= n_edges) {
        sr_a = ei[wave_base + 32 + li];
        gather_edge_rows<FE_CHUNKS, FE_VEC, SENDER_ONLY, UT>(xq, ef, a.f_e, wave_base + li, ei[wave_base + li], lh, r0);
#pragma unroll
        for (int grp = 0; grp < kEdgeGroups; grp += 2) {
          step(std::true_type{}, grp, r0, r1, sr_a, sr_b);
          step(std::true_type{}, grp + 1, r1, r0, sr_b, sr_a);
        }
      } else if (wave_base < n_edges) {
        sr_a = ei[edge_of(1)];
        gather_edge_rows<FE_CHUNKS, FE_VEC, SENDER_ONLY, UT>(xq, ef, a.f_e, edge_of(0), ei[edge_of(0)], lh, r0);
        for (int grp = 0; grp < kEdgeGroups && wave_base + grp * 32 < n_edges; grp += 2) {
          step(std::false_type{}, grp, r0, r1, sr_a, sr_b);
          if (wave_base + (grp + 1) * 32 < n_edges) step(std::false_type{}, grp + 1, r1, r0, sr_b, sr_a);
        }
      }
    } else {
      for (int grp = 0; grp < kEdgeGroups && wave_base + grp * 32 < n_edges; ++grp) {
        Rows cur;
        gather_edge_rows<FE_CHUNKS, FE_VEC, SENDER_ONLY, UT>(xq, ef, a.f_e, edge_of(grp), ei[edge_of(grp)], lh, cur);
        multiply(std::false_type{}, wave_base + grp * 32, cur);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < UT; ++t) {
    colsum[t] += __shfl_xor(colsum[t], 32);
    if (lh == 0) red[wave * u + 32 * t + li] = colsum[t];
  }
  __syncthreads();
  if (tid < u) {
    const float s = ((red[tid] + red[u + tid]) + red[2 * u + tid]) + red[3 * u + tid];
    a.partial[((b * 4 + q) * gridDim.x + blockIdx.x) * u + tid] = s;
  }
}

struct AggArgs {
  const int32_t* rowptr[4];
  const int32_t* perm[4];
  const float* msg[4];   // [B, E_q, u]
  int n_edges[4];
  int n_nodes;
  float scale;
  float* agg;            // [B, N, U]
};

// Receiver-CSR aggregation (unsorted_segment_sum, hybrid_gnn.py:110-113): 32*UT lanes own one
// receiver's U output floats, 8*UT lanes x float4 per direction; messages are added in the
// reference's edge order, starting from zero, then scaled.
template <int UT>
__global__ __launch_bounds__(kThreads) void k_aggregate(AggArgs a) {
  constexpr int U = 128 * UT, u = 32 * UT, LPN = 32 * UT, LPQ = 8 * UT;   // lanes per node / direction
  const int tid = threadIdx.x;
  const int64_t v = static_cast<int64_t>(blockIdx.x) * (kThreads / LPN) + tid / LPN;
  if (v >= a.n_nodes) return;
  const int64_t b = blockIdx.y;
  const int sub = tid % LPN, q = sub / LPQ, c4 = sub % LPQ;
  const int32_t* rowptr = q == 0 ? a.rowptr[0] : q == 1 ? a.rowptr[1] : q == 2 ? a.rowptr[2] : a.rowptr[3];
  const int32_t* perm = q == 0 ? a.perm[0] : q == 1 ? a.perm[1] : q == 2 ? a.perm[2] : a.perm[3];
  const float* msg = q == 0 ? a.msg[0] : q == 1 ? a.msg[1] : q == 2 ? a.msg[2] : a.msg[3];
  const int ne = q == 0 ? a.n_edges[0] : q == 1 ? a.n_edges[1] : q == 2 ? a.n_edges[2] : a.n_edges[3];
  msg += b * ne * static_cast<int64_t>(u) + 4 * c4;
  const int beg = rowptr[v], end = rowptr[v + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = beg; s < end; ++s) {
    const float4 m = *reinterpret_cast<const float4*>(msg + static_cast<int64_t>(perm[s]) * u);
    acc.x += m.x;
    acc.y += m.y;
    acc.z += m.z;
    acc.w += m.w;
  }
  acc.x *= a.scale;
  acc.y *= a.scale;
  acc.z *= a.scale;
  acc.w *= a.scale;
  *reinterpret_cast<float4*>(a.agg + (b * a.n_nodes + v) * U + u * q + 4 * c4) = acc;
}

// Per sample: finish the node / edge column sums (fp64 accumulation over the per-tile partials:
// the thread subsets take tiles sub, sub + S, ... with float4 loads, the subsets are then added in
// order), scale them and apply the global transform (hybrid_gnn.py:130-135).
// LDS: dpart 4096 doubles, part 4096 floats, gin [g + U + u] whose first g entries the caller
// fills with the layer's input global (the first barrier below covers that).  The new global goes
// to global_out and, when `keep` is not null, also to keep[0 .. width of the new global) in LDS.
constexpr int kPreGlobT = 8;   // rows of the glob_transform product fetched ahead, per thread (head != nullptr)

template <int UT>
__device__ __forceinline__ void global_update_body(const LayerPtrs& p, const float* __restrict__ npart, int n_nblk,
                                                   const float* __restrict__ epart, int n_eblk, int relu, int64_t b,
                                                   double* dpart, float* part, float* gin,
                                                   float* __restrict__ global_out, float* keep,
                                                   const float4 (*head)[kPreGlobT] = nullptr) {
  constexpr int U = 128 * UT, u = 32 * UT;
  constexpr int QC = U / 4, S = kSmallThreads / QC;
  constexpr int EG = kSmallThreads / U;
  const int tid = threadIdx.x;
  const int sub = tid / QC, cq = tid % QC;
  {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const float4* src = reinterpret_cast<const float4*>(npart + b * n_nblk * static_cast<int64_t>(U)) + cq;
    int i = sub;
    for (; i + 7 * S < n_nblk; i += 8 * S) {   // eight partial rows in flight, added in order
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[static_cast<int64_t>(i + j * S) * QC];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s0 += static_cast<double>(v[j].x);
        s1 += static_cast<double>(v[j].y);
        s2 += static_cast<double>(v[j].z);
        s3 += static_cast<double>(v[j].w);
      }
    }
    for (; i < n_nblk; i += S) {
      const float4 a = src[static_cast<int64_t>(i) * QC];
      s0 += static_cast<double>(a.x);
      s1 += static_cast<double>(a.y);
      s2 += static_cast<double>(a.z);
      s3 += static_cast<double>(a.w);
    }
    double* d = dpart + sub * U + 4 * cq;
    d[0] = s0;
    d[1] = s1;
    d[2] = s2;
    d[3] = s3;
  }
  __syncthreads();
  if (tid < U) {
    double s = dpart[tid];
    for (int g = 1; g < S; ++g) s += dpart[g * U + tid];
    const float denom = p.gsn * 300.0f * 300.0f;
    gin[p.g + tid] = (static_cast<float>(s) * 1.0f) / denom;
  }
  __syncthreads();
  {
    // message sums: [4 directions][n_eblk][u]; thread (grp, col) with col = u*q + c
    const int grp = tid / U, col = tid % U;
    const int q = col / u, c = col % u;
    double s = 0.0;
    const float* src = epart + (b * 4 + q) * n_eblk * static_cast<int64_t>(u) + c;
    int i = grp;
    for (; i + 7 * EG < n_eblk; i += 8 * EG) {   // eight partial rows in flight, added in order
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[static_cast<int64_t>(i + j * EG) * u];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += static_cast<double>(v[j]);
    }
    for (; i < n_eblk; i += EG) s += static_cast<double>(src[static_cast<int64_t>(i) * u]);
    dpart[grp * U + col] = s;
  }
  __syncthreads();
  if (tid < u) {
    float total = 0.0f;
    for (int q = 0; q < 4; ++q) {
      double s = dpart[q * u + tid];
      for (int g = 1; g < EG; ++g) s += dpart[g * U + q * u + tid];
      total += static_cast<float>(s);
    }
    const float denom = p.gse * 300.0f * 300.0f * 8.0f;
    gin[p.g + U + tid] = (total * 1.0f) / denom;
  }
  __syncthreads();
  const int kin = p.g + U + u;
  if (p.globt_w != nullptr) {
    const float4 s = head ? matvec_partial<U>(gin, p.globt_w, kin, sub, cq, kPreGlobT,
                                              matvec_head_apply<U, kPreGlobT>(gin, kin, sub, *head))
                          : matvec_partial<U>(gin, p.globt_w, kin, sub, cq);
    *reinterpret_cast<float4*>(&part[sub * U + 4 * cq]) = s;
    __syncthreads();
    if (tid < U) {
      float t = combine_rows(part, S, U, tid);
      if (p.globt_b) t += p.globt_b[tid];
      t = activate(t, relu);
      global_out[b * U + tid] = t;
      if (keep) keep[tid] = t;
    }
  } else {
    for (int k = tid; k < kin; k += blockDim.x) {
      global_out[b * kin + k] = gin[k];
      if (keep) keep[k] = gin[k];
    }
  }
}

template <int UT>
__global__ __launch_bounds__(kSmallThreads) void k_global_update(LayerPtrs p,
                                                                 const float* __restrict__ global_in,
                                                                 const float* __restrict__ npart, int n_nblk,
                                                                 const float* __restrict__ epart, int n_eblk,
                                                                 int relu, float* __restrict__ global_out) {
  extern __shared__ __attribute__((aligned(16))) float gsh[];
  double* dpart = reinterpret_cast<double*>(gsh);          // 4096 doubles
  float* part = reinterpret_cast<float*>(dpart + 4096);    // 4096 floats
  float* gin = part + 4096;                                // [g + U + u]
  const int64_t b = blockIdx.x;
  for (int k = threadIdx.x; k < p.g; k += blockDim.x) gin[k] = global_in[b * p.g + k];
  global_update_body<UT>(p, npart, n_nblk, epart, n_eblk, relu, b, dpart, part, gin, global_out, nullptr);
}

// global update of one layer and prologue of the next in one launch (the new global stays in LDS):
// between two consecutive GN layers of a model nothing else depends on either.  `extra` (may be
// null) is appended to the new global to form the next layer's input global -- the
// [global | target global] of the upsampling layer (hybrid_gnn.py:375) -- which then also goes to
// cat_out.
template <int UTC, int UTN>
__global__ __launch_bounds__(kSmallThreads) void k_global_bridge(LayerPtrs pc, const float* __restrict__ global_in,
                                                                 const float* __restrict__ npart, int n_nblk,
                                                                 const float* __restrict__ epart, int n_eblk, int relu,
                                                                 float* __restrict__ global_out,
                                                                 const float* __restrict__ extra, int extra_width,
                                                                 float* __restrict__ cat_out, LayerPtrs pn,
                                                                 int fold_receiver, float* __restrict__ nbias,
                                                                 float* __restrict__ ebias, float* __restrict__ gbias) {
  extern __shared__ __attribute__((aligned(16))) float bsh[];
  double* dpart = reinterpret_cast<double*>(bsh);          // 4096 doubles
  float* part = reinterpret_cast<float*>(dpart + 4096);    // 4096 floats
  float* gin = part + 4096;                                // [g + U + u] of the finished layer
  float* gl = gin + pc.g + 160 * UTC;                      // [g] of the next layer
  float* gq = gl + pn.g;                                   // [4][u] of the next layer
  const int64_t b = blockIdx.x;
  for (int k = threadIdx.x; k < pc.g; k += blockDim.x) gin[k] = global_in[b * pc.g + k];
  // every weight the two stages will want first is requested now, behind the column sums
  ProloguePre<UTN> pre;
  prologue_prefetch<UTN, true>(pn, relu, fold_receiver, b, nbias, pre);
  float4 head[kPreGlobT];
  if (pc.globt_w != nullptr)
    matvec_head_load<128 * UTC, kPreGlobT>(pc.globt_w, pc.g + 160 * UTC, threadIdx.x / (32 * UTC),
                                           threadIdx.x % (32 * UTC), head);
  global_update_body<UTC>(pc, npart, n_nblk, epart, n_eblk, relu, b, dpart, part, gin, global_out, gl,
                          pc.globt_w != nullptr ? &head : nullptr);
  if (extra != nullptr) {
    const int w0 = pn.g - extra_width;   // width of the new global
    for (int k = threadIdx.x; k < extra_width; k += blockDim.x) gl[w0 + k] = extra[b * extra_width + k];
    __syncthreads();
    for (int k = threadIdx.x; k < pn.g; k += blockDim.x) cat_out[b * pn.g + k] = gl[k];
  }
  // the barrier that opens prologue_compute frees `part` and completes gl
  prologue_compute<UTN, true>(pn, gl, relu, b, part, gq, ebias, gbias, pre);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct LayerWorkspace {
  float* xq;      // [B, N, U]            node embeddings (generic) / coarse rows only (upsample)
  float* agg;     // [B, N, U]            aggregated messages (generic) / messages [B, N_fine, u]
  float* nbias;   // [U]
  float* ebias;   // [B, U]
  float* gbias;   // [B, U]
  float* npart;   // [B, n_nblk + 5, U]   column sums per 32-row group
  float* epart;   // [B, 4, n_eblk, u]
  void* wpack;    // weight image of the GEMM in flight (split-bf16 matmul modes)
  int n_nblk, n_eblk;
  size_t bytes;
};

// One layout serves both entry points: the upsampling form needs at most 4 row groups more than
// the generic one (one partly filled group per quadrant) and less of everything else.
LayerWorkspace carve_workspace(const gunet_quadgraph* g, int32_t batch, int units, void* base) {
  LayerWorkspace ws;
  const size_t n = static_cast<size_t>(g->n_nodes), bsz = static_cast<size_t>(batch), U = units, u = units / 4;
  int max_e = 0;
  for (int q = 0; q < 4; ++q) max_e = g->n_edges[q] > max_e ? g->n_edges[q] : max_e;
  ws.n_nblk = static_cast<int>(ceil_div(g->n_nodes, kRowGroup));
  ws.n_eblk = static_cast<int>(ceil_div(max_e > 0 ? max_e : 1, kEdgesPerBlock));
  size_t off = 0;
  auto take = [&](size_t floats) {
    float* p = base ? reinterpret_cast<float*>(static_cast<char*>(base) + off) : nullptr;
    off = align_up(off + floats * sizeof(float), 256);
    return p;
  };
  // the folded biases come first, at offsets that depend on the batch only: the kernel that finishes
  // one layer's global and prepares the next layer's biases works on two layouts at once
  ws.nbias = take(128 * kMaxUT);
  ws.ebias = take(bsz * 128 * kMaxUT);
  ws.gbias = take(bsz * 128 * kMaxUT);
  ws.xq = take(bsz * n * U);
  ws.agg = take(bsz * n * U);
  ws.npart = take(bsz * (ws.n_nblk + 5) * U);
  ws.epart = take(bsz * 4 * ws.n_eblk * u);
  ws.wpack = take(wpack_bytes((kMaxCin + 128 * kMaxUT) / kBK + 2, 4) / sizeof(float));
  ws.bytes = off;
  return ws;
}

int check_layer_args(const char* fn, const gunet_quadgraph* graph, const gunet_gn_weights* w, int32_t batch,
                     int32_t activation, const void* workspace, size_t workspace_bytes, LayerWorkspace* ws) {
  GUNET_REQUIRE(graph && w, "%s: null graph or weights", fn);
  GUNET_REQUIRE(batch > 0, "%s: batch must be positive", fn);
  GUNET_REQUIRE(activation == GUNET_ACT_LINEAR || activation == GUNET_ACT_RELU, "%s: unknown activation %d",
                fn, activation);
  if (w->units != 128 && w->units != 256) {
    set_error("%s: units = %d is not implemented (128 and 256 are)", fn, w->units);
    return GUNET_ERR_UNSUPPORTED;
  }
  GUNET_REQUIRE(w->c_in > 0 && w->c_in % 4 == 0 && w->c_in <= kMaxCin,
                "%s: c_in = %d must be a positive multiple of 4, at most %d", fn, w->c_in, kMaxCin);
  GUNET_REQUIRE(w->g > 0 && w->g <= 2048, "%s: global width %d outside 1..2048", fn, w->g);
  if (w->f_e < 0 || w->f_e > 64) {
    set_error("%s: edge feature width %d outside 0..64 is not implemented", fn, w->f_e);
    return GUNET_ERR_UNSUPPORTED;
  }
  GUNET_REQUIRE(graph->n_nodes >= 0, "%s: negative node count", fn);
  GUNET_REQUIRE(w->node_w, "%s: node_transform kernel is null", fn);
  for (int q = 0; q < 4; ++q) {
    GUNET_REQUIRE(graph->n_edges[q] >= 0, "%s: negative edge count", fn);
    GUNET_REQUIRE(w->node_emb_w[q] && w->edge_w[q] && w->glob_w[q], "%s: null kernel for direction %d", fn, q);
    if (graph->n_edges[q] > 0)
      GUNET_REQUIRE(graph->edge_index[q] && graph->perm[q], "%s: null edge list for direction %d", fn, q);
    GUNET_REQUIRE(graph->n_nodes == 0 || graph->rowptr[q], "%s: null rowptr for direction %d", fn, q);
  }
  *ws = carve_workspace(graph, batch, w->units, const_cast<void*>(workspace));
  if (workspace == nullptr || workspace_bytes < ws->bytes) {
    set_error("%s: workspace of %zu bytes is smaller than the %zu required", fn, workspace_bytes, ws->bytes);
    return GUNET_ERR_WORKSPACE;
  }
  GUNET_REQUIRE(reinterpret_cast<uintptr_t>(workspace) % 16 == 0, "%s: workspace must be 16-byte aligned", fn);
  return GUNET_OK;
}

LayerPtrs layer_ptrs(const gunet_gn_weights* w) {
  LayerPtrs p;
  for (int q = 0; q < 4; ++q) {
    p.node_emb_b[q] = w->node_emb_b[q];
    p.edge_w[q] = w->edge_w[q];
    p.edge_b[q] = w->edge_b[q];
    p.glob_w[q] = w->glob_w[q];
    p.glob_b[q] = w->glob_b[q];
  }
  p.node_w = w->node_w;
  p.node_b = w->node_b;
  p.globt_w = w->globt_w;
  p.globt_b = w->globt_b;
  p.c_in = w->c_in;
  p.f_e = w->f_e;
  p.g = w->g;
  p.gsn = w->global_scale_nodes;
  p.gse = w->global_scale_edges;
  return p;
}

int launch_prologue(const LayerPtrs& p, int ut, const float* global_in, int relu, int fold_receiver,
                    int32_t batch, const LayerWorkspace& ws, hipStream_t stream) {
  ProfScope prof(kProfPrologue, stream);
  const size_t lds = sizeof(float) * (4096 + p.g + 4 * 32 * ut);
  if (ut == 1)
    k_layer_prologue<1><<<batch, kSmallThreads, lds, stream>>>(p, global_in, relu, fold_receiver, ws.nbias,
                                                                 ws.ebias, ws.gbias);
  else
    k_layer_prologue<2><<<batch, kSmallThreads, lds, stream>>>(p, global_in, relu, fold_receiver, ws.nbias,
                                                                 ws.ebias, ws.gbias);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

int launch_global_update(const LayerPtrs& p, int ut, const float* global_in, int n_nblk, int relu,
                         int32_t batch, const LayerWorkspace& ws, float* global_out, hipStream_t stream) {
  ProfScope prof(kProfGlobalUpdate, stream);
  const size_t lds = sizeof(double) * 4096 + sizeof(float) * (4096 + p.g + 160 * ut);
  if (ut == 1)
    k_global_update<1><<<batch, kSmallThreads, lds, stream>>>(p, global_in, ws.npart, n_nblk, ws.epart,
                                                                ws.n_eblk, relu, global_out);
  else
    k_global_update<2><<<batch, kSmallThreads, lds, stream>>>(p, global_in, ws.npart, n_nblk, ws.epart,
                                                                ws.n_eblk, relu, global_out);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

template <int UTC>
int launch_bridge_utc(const LayerPtrs& pc, const LayerPtrs& pn, int utn, size_t lds, int32_t batch, const float* global_in,
                      int n_nblk, const LayerWorkspace& wc, int relu, float* global_out, const float* extra,
                      int extra_width, float* cat_out, int fold_receiver, const LayerWorkspace& wn, hipStream_t stream) {
  if (utn == 1)
    k_global_bridge<UTC, 1><<<batch, kSmallThreads, lds, stream>>>(pc, global_in, wc.npart, n_nblk, wc.epart, wc.n_eblk,
                                                                    relu, global_out, extra, extra_width, cat_out, pn,
                                                                    fold_receiver, wn.nbias, wn.ebias, wn.gbias);
  else
    k_global_bridge<UTC, 2><<<batch, kSmallThreads, lds, stream>>>(pc, global_in, wc.npart, n_nblk, wc.epart, wc.n_eblk,
                                                                    relu, global_out, extra, extra_width, cat_out, pn,
                                                                    fold_receiver, wn.nbias, wn.ebias, wn.gbias);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

template <int UT>
int launch_edge_mlp_ut(EdgeArgs& a, int chunks, bool vec, dim3 grid, hipStream_t stream) {
  if (a.sender_only) {
    k_edge_mlp<0, false, true, UT><<<grid, kThreads, 0, stream>>>(a);
  } else if (chunks == 0) {
    k_edge_mlp<0, false, false, UT><<<grid, kThreads, 0, stream>>>(a);
  } else if (vec && chunks == 2) {
    k_edge_mlp<2, true, false, UT><<<grid, kThreads, 0, stream>>>(a);
  } else if (vec && chunks == 4) {
    k_edge_mlp<4, true, false, UT><<<grid, kThreads, 0, stream>>>(a);
  } else if (vec && chunks == 8) {
    k_edge_mlp<8, true, false, UT><<<grid, kThreads, 0, stream>>>(a);
  } else if (chunks == 1) {
    k_edge_mlp<1, false, false, UT><<<grid, kThreads, 0, stream>>>(a);
  } else if (chunks == 2) {
    k_edge_mlp<2, false, false, UT><<<grid, kThreads, 0, stream>>>(a);
  } else if (chunks <= 4) {
    k_edge_mlp<4, false, false, UT><<<grid, kThreads, 0, stream>>>(a);
  } else {
    k_edge_mlp<8, false, false, UT><<<grid, kThreads, 0, stream>>>(a);
  }
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

int launch_edge_mlp(EdgeArgs& a, int ut, const gunet_quadgraph* graph, const float* const edges[4], int n_eblk,
                    int32_t batch, hipStream_t stream) {
  bool any_feat = false, all_feat = true, vec_ok = a.f_e % 8 == 0;
  for (int q = 0; q < 4; ++q) {
    a.edge_index[q] = graph->edge_index[q];
    a.efeat[q] = edges ? edges[q] : nullptr;
    a.n_edges[q] = graph->n_edges[q];
    if (graph->n_edges[q] > 0) {
      if (a.efeat[q] != nullptr) {
        any_feat = true;
        if (reinterpret_cast<uintptr_t>(a.efeat[q]) % 16 != 0) vec_ok = false;
      } else {
        all_feat = false;
      }
    }
  }
  // a direction whose features are absent reads nothing; the vector loader needs all of them
  dim3 grid(static_cast<unsigned>(n_eblk), 4, static_cast<unsigned>(batch));
  const int chunks = (!any_feat || a.f_e == 0) ? 0 : (a.f_e + 7) / 8;
  double edges_total = 0.0;
  for (int q = 0; q < 4; ++q) edges_total += graph->n_edges[q];
  const double u = 32.0 * ut;
  ProfScope prof(kProfEdgeMlp, stream,
                 2.0 * batch * edges_total * ((chunks ? a.f_e : 0) + (a.sender_only ? 1 : 2) * u) * u);
  if (a.sender_only && chunks != 0) {
    set_error("edge_mlp: the sender-only form takes no edge features");
    return GUNET_ERR_INVALID;
  }
  return ut == 1 ? launch_edge_mlp_ut<1>(a, chunks, vec_ok && all_feat, grid, stream)
                 : launch_edge_mlp_ut<2>(a, chunks, vec_ok && all_feat, grid, stream);
}

// GEMM with U output columns as U/128 launches of 128 columns.  `w_of(s, col)` gives the weight
// pointer of segment s for the 32-column tile starting at output column `col`.
template <typename WeightOf>
GemmArgs columns_block(const GemmArgs& g, int units, int col0, int nt, const float* bias, int bias_stride, float* out,
                       float* colsum, WeightOf w_of) {
  GemmArgs gc = g;
  for (int s = 0; s < g.n_seg; ++s)
    for (int t = 0; t < nt; ++t) gc.seg[s].w[t] = w_of(s, col0 + 32 * t);
  gc.bias = bias ? bias + col0 : nullptr;
  gc.bias_stride = bias_stride;
  gc.out = out;
  gc.ldo = units;
  gc.out_col0 = col0;
  gc.colsum = colsum;
  gc.colsum_ld = units;
  return gc;
}

// Column tiles per launch: 128 columns, except 64 for the gathering GEMM with three operand
// segments (its weights fill LDS at that width).
inline int columns_nt(const GemmArgs& g) {
  return (g.gather_table != nullptr && gemm_ktiles(g) == 12 && !bf16_mode()) ? 2 : 4;
}

// Column-sum partial rows per sample such a GEMM writes (the same for each of its column blocks).
inline int columns_colsum_rows(const GemmArgs& g, int units, int64_t batch, const void* wpack) {
  GemmArgs gc = g;
  gc.ldo = units;
  return gemm_colsum_rows(gc, columns_nt(g), batch, wpack);
}

template <typename WeightOf>
int launch_gemm_columns(const GemmArgs& g, int units, int64_t batch, const float* bias, int bias_stride, float* out,
                        float* colsum, WeightOf w_of, void* wpack, hipStream_t stream) {
  const int nt = columns_nt(g);
  for (int col0 = 0; col0 < units; col0 += 32 * nt)
    GUNET_TRY(launch_gemm(columns_block(g, units, col0, nt, bias, bias_stride, out, colsum, w_of), nt, batch, wpack, stream));
  return GUNET_OK;
}

}  // namespace

int pool_max_multi(int n_jobs, const float* const* src, const int64_t* n_src, const int* width,
                   const int32_t* const* rowptr, const int32_t* const* perm, const int64_t* n_seg, float* const* out,
                   int32_t batch, hipStream_t stream) {
  GUNET_REQUIRE(n_jobs >= 0 && n_jobs <= kPoolJobs, "pool_max_multi: %d jobs", n_jobs);
  bool vec = true;
  for (int i = 0; i < n_jobs; ++i)
    vec = vec && width[i] % 4 == 0 && reinterpret_cast<uintptr_t>(src[i]) % 16 == 0 &&
          reinterpret_cast<uintptr_t>(out[i]) % 16 == 0;
  if (!vec) {
    for (int i = 0; i < n_jobs; ++i)
      GUNET_TRY(launch_segment_reduce<true>(src[i], n_src[i], width[i], rowptr[i], perm[i], n_seg[i], batch, 1.0f,
                                            out[i], stream));
    return GUNET_OK;
  }
  PoolJobs j = {};
  unsigned blocks = 0;
  double bytes = 0.0;
  for (int i = 0; i < kPoolJobs; ++i) {
    if (i < n_jobs && batch > 0) {
      j.src[i] = src[i];
      j.out[i] = out[i];
      j.rowptr[i] = rowptr[i];
      j.perm[i] = perm[i];
      j.n_src[i] = n_src[i];
      j.n_seg[i] = n_seg[i];
      j.width[i] = width[i];
      blocks += static_cast<unsigned>(ceil_div(n_seg[i] * (width[i] / 4), kThreads));
      bytes += static_cast<double>(batch) * (4.0 * width[i] * (n_src[i] + n_seg[i]) + 4.0 * (n_src[i] + n_seg[i]));
    } else {
      j.width[i] = 4;
    }
    j.block_end[i] = blocks;
  }
  if (blocks == 0) return GUNET_OK;
  ProfScope prof(kProfSegment, stream, bytes);
  k_pool_max<<<dim3(blocks, static_cast<unsigned>(batch)), kThreads, 0, stream>>>(j);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

}  // namespace gunet

using namespace gunet;

extern "C" {

int gunet_get_sender_receiver_f32(const float* x, int64_t n_nodes, int32_t width,
                                  const int32_t* edge_index, int64_t n_edges, float* senders,
                                  float* receivers, gunet_stream_t stream) {
  GUNET_REQUIRE(n_nodes >= 0 && n_edges >= 0 && width >= 0, "get_sender_receiver: negative size");
  if (n_edges == 0 || width == 0) return GUNET_OK;
  GUNET_REQUIRE(x && edge_index && senders && receivers, "get_sender_receiver: null pointer");
  const bool vec = width % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(senders) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(receivers) % 16 == 0;
  const int64_t work = n_edges * (vec ? width / 4 : width);
  const unsigned grid = static_cast<unsigned>(ceil_div(work, kThreads));
  ProfScope prof(kProfGather, as_stream(stream));
  if (vec)
    k_gather_rows<4><<<grid, kThreads, 0, as_stream(stream)>>>(x, width, edge_index, n_edges,
                                                               senders, receivers);
  else
    k_gather_rows<1><<<grid, kThreads, 0, as_stream(stream)>>>(x, width, edge_index, n_edges,
                                                               senders, receivers);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

int gunet_segment_max_f32(const float* src, int64_t n_src, int32_t width, const int32_t* rowptr,
                          const int32_t* perm, int64_t n_segments, int32_t batch, float* out,
                          gunet_stream_t stream) {
  GUNET_REQUIRE(n_src >= 0 && width >= 0 && n_segments >= 0 && batch >= 0, "segment_max: negative size");
  GUNET_REQUIRE(n_segments == 0 || (rowptr && out), "segment_max: null pointer");
  return launch_segment_reduce<true>(src, n_src, width, rowptr, perm, n_segments, batch, 1.0f, out,
                                     as_stream(stream));
}

int gunet_segment_sum_f32(const float* src, int64_t n_src, int32_t width, const int32_t* rowptr,
                          const int32_t* perm, int64_t n_segments, int32_t batch, float scale,
                          float* out, gunet_stream_t stream) {
  GUNET_REQUIRE(n_src >= 0 && width >= 0 && n_segments >= 0 && batch >= 0, "segment_sum: negative size");
  GUNET_REQUIRE(n_segments == 0 || (rowptr && out), "segment_sum: null pointer");
  return launch_segment_reduce<false>(src, n_src, width, rowptr, perm, n_segments, batch, scale,
                                      out, as_stream(stream));
}

size_t gunet_dense_workspace_bytes(int32_t k, int32_t n_out) {
  if (k <= 0 || n_out <= 0) return 0;
  return wpack_bytes(static_cast<int>(ceil_div(k, kBK)), 4) + 256;
}

int gunet_dense_forward_f32(const float* x, int64_t rows, int32_t k, const float* kernel,
                            const float* bias, int32_t n_out, int32_t activation, float* out,
                            void* workspace, size_t workspace_bytes, gunet_stream_t stream_) {
  hipStream_t stream = as_stream(stream_);
  GUNET_REQUIRE(rows >= 0 && k > 0 && n_out > 0, "dense_forward: bad shape");
  GUNET_REQUIRE(activation == GUNET_ACT_LINEAR || activation == GUNET_ACT_RELU,
                "dense_forward: unknown activation %d", activation);
  if (rows == 0) return GUNET_OK;
  GUNET_REQUIRE(x && kernel && out, "dense_forward: null pointer");
  GUNET_REQUIRE(rows < INT32_MAX, "dense_forward: too many rows");
  const bool tiled = k % 4 == 0 && n_out % 32 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(kernel) % 16 == 0;
  if (!tiled) {
    const unsigned grid = static_cast<unsigned>(ceil_div(rows * n_out, kThreads));
    ProfScope prof(kProfDenseNaive, stream);
    k_dense_naive<<<grid, kThreads, 0, stream>>>(x, rows, k, kernel, bias, n_out,
                                                 activation == GUNET_ACT_RELU, out);
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
    GUNET_TRY(launch_gemm(g, nt, 1, workspace, stream));
  }
  return GUNET_OK;
}

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

size_t gunet_gn_layer_workspace_bytes(const gunet_quadgraph* graph, int32_t batch, int32_t units) {
  if (graph == nullptr || batch <= 0 || (units != 128 && units != 256)) return 0;
  return carve_workspace(graph, batch, units, nullptr).bytes;
}

int gunet_gn_layer_forward_f32(const gunet_quadgraph* graph, const gunet_gn_weights* w, int32_t batch,
                               const float* x, const float* x_skip, int32_t c_skip,
                               const float* const edges[4], const float* global_in, int32_t activation,
                               float* x_out, float* const edges_out[4], float* global_out,
                               void* workspace, size_t workspace_bytes, gunet_stream_t stream) {
  return gn_layer_run(graph, w, batch, x, x_skip, c_skip, edges, global_in, activation, x_out, edges_out, global_out,
                      workspace, workspace_bytes, 0, nullptr, as_stream(stream));
}

int gunet_gn_upsample_forward_f32(const gunet_quadgraph* graph, int32_t n_coarse, const gunet_gn_weights* w,
                                  int32_t batch, const float* x, const float* global_in, int32_t activation,
                                  float* x_out, float* global_out, void* workspace, size_t workspace_bytes,
                                  gunet_stream_t stream) {
  return gn_upsample_run(graph, n_coarse, w, batch, x, global_in, activation, x_out, global_out, workspace,
                         workspace_bytes, 0, nullptr, as_stream(stream));
}

}  // extern "C"

namespace gunet {

int gn_layer_run(const gunet_quadgraph* graph, const gunet_gn_weights* w, int32_t batch, const float* x,
                 const float* x_skip, int32_t c_skip, const float* const edges[4], const float* global_in,
                 int32_t activation, float* x_out, float* const edges_out[4], float* global_out, void* workspace,
                 size_t workspace_bytes, int flags, int* n_partials_out, hipStream_t stream) {
  LayerWorkspace ws;
  GUNET_TRY(check_layer_args("gn_layer", graph, w, batch, activation, workspace, workspace_bytes, &ws));
  GUNET_REQUIRE(x && x_out && edges && edges_out, "gn_layer: null tensor");
  GUNET_REQUIRE(global_in || (flags & kGnSkipPrologue && flags & kGnSkipGlobalUpdate), "gn_layer: null global_in");
  GUNET_REQUIRE(global_out || (flags & kGnSkipGlobalUpdate), "gn_layer: null global_out");
  GUNET_REQUIRE(c_skip >= 0 && c_skip % 4 == 0 && c_skip < w->c_in && (c_skip == 0) == (x_skip == nullptr),
                "gn_layer: bad skip segment (c_skip = %d of c_in = %d)", c_skip, w->c_in);
  GUNET_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(x_out) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(x_skip) % 16 == 0,
                "gn_layer: x, x_skip and x_out must be 16-byte aligned");
  for (int q = 0; q < 4; ++q)
    GUNET_REQUIRE(graph->n_edges[q] == 0 || edges_out[q], "gn_layer: null edges_out[%d]", q);
  const int relu = activation == GUNET_ACT_RELU;
  const int n = graph->n_nodes;
  const int U = w->units, u = U / 4, ut = U / 128;
  const int c_a = w->c_in - c_skip;          // width of x; x_skip supplies the remaining columns
  const LayerPtrs p = layer_ptrs(w);

  if (!(flags & kGnSkipPrologue)) GUNET_TRY(launch_prologue(p, ut, global_in, relu, 0, batch, ws, stream));

  {  // node embeddings of the four directions in one GEMM per 128 columns (hybrid_gnn.py:91-92)
    GemmArgs g = {};
    g.n_seg = c_skip ? 2 : 1;
    g.seg[0].a = x;
    g.seg[0].lda = c_a;
    g.seg[0].k = c_a;
    g.seg[0].ldw = u;
    g.seg[1].a = x_skip;
    g.seg[1].lda = c_skip;
    g.seg[1].k = c_skip;
    g.seg[1].ldw = u;
    g.n_rows = n;
    g.relu = relu;
    auto w_of = [&](int s, int col) {
      return w->node_emb_w[col / u] + (s == 0 ? 0 : static_cast<int64_t>(c_a) * u) + col % u;
    };
    GUNET_TRY(launch_gemm_columns(g, U, batch, ws.nbias, 0, ws.xq, nullptr, w_of, ws.wpack, stream));
  }

  {  // edge MLP + message column sums (hybrid_gnn.py:93-108)
    EdgeArgs a = {};
    for (int q = 0; q < 4; ++q) {
      a.edge_w[q] = w->edge_w[q];
      a.out[q] = edges_out[q];
    }
    a.n_nodes = n;
    a.f_e = w->f_e;
    a.xq = ws.xq;
    a.ebias = ws.ebias;
    a.partial = ws.epart;
    a.relu = relu;
    a.out_scale = 1.0f;
    GUNET_TRY(launch_edge_mlp(a, ut, graph, edges, ws.n_eblk, batch, stream));
  }

  // node MLP over [x | x_skip | aggregated messages]; global columns folded into the bias (:118-126)
  GemmArgs g = {};
  int64_t row_of[3];
  {
    int s = 0;
    g.seg[s].a = x;
    g.seg[s].lda = c_a;
    g.seg[s].k = c_a;
    row_of[s++] = 0;
    if (c_skip) {
      g.seg[s].a = x_skip;
      g.seg[s].lda = c_skip;
      g.seg[s].k = c_skip;
      row_of[s++] = c_a;
    }
    g.seg[s].a = ws.agg;
    g.seg[s].lda = U;
    g.seg[s].k = U;
    row_of[s++] = w->c_in;
    g.n_seg = s;
    for (int i = 0; i < s; ++i) g.seg[i].ldw = U;
    g.n_rows = n;
    g.relu = relu;
  }
  // With a per-node table of incoming edges the GEMM sums the messages while it loads them
  // (same order and scale as k_aggregate), and the aggregated tensor never exists.
  bool fused = false;
  if (graph->in_table != nullptr && U == 128 && n > 0) {
    GemmArgs gg = g;
    gg.seg[g.n_seg - 1].a = nullptr;
    gg.seg[g.n_seg - 1].lda = 0;
    gg.gather_table = graph->in_table;
    for (int q = 0; q < 4; ++q) {
      // an empty quadrant has only -1 table entries, whose (masked) loads still need a valid address
      gg.gather_src[q] = graph->n_edges[q] > 0 ? edges_out[q] : x;
      gg.gather_rows[q] = graph->n_edges[q];
    }
    gg.gather_scale = 1.0f / (3 + 1);
    gg.ldo = U;
    if (takes_resident(gg, columns_nt(gg), batch, ws.wpack)) {
      g = gg;
      fused = true;
    }
  }

  if (n > 0 && !fused) {  // neighbour aggregation (hybrid_gnn.py:110-113)
    AggArgs a;
    for (int q = 0; q < 4; ++q) {
      a.rowptr[q] = graph->rowptr[q];
      a.perm[q] = graph->perm[q];
      a.msg[q] = edges_out[q];
      a.n_edges[q] = graph->n_edges[q];
    }
    a.n_nodes = n;
    a.scale = 1.0f / (3 + 1);
    a.agg = ws.agg;
    dim3 grid(static_cast<unsigned>(ceil_div(n, kThreads / (32 * ut))), static_cast<unsigned>(batch));
    double edges_total = 0.0;
    for (int q = 0; q < 4; ++q) edges_total += graph->n_edges[q];
    ProfScope prof(kProfAggregate, stream, 4.0 * batch * (edges_total * u + edges_total + static_cast<double>(n) * U));
    if (ut == 1)
      k_aggregate<1><<<grid, kThreads, 0, stream>>>(a);
    else
      k_aggregate<2><<<grid, kThreads, 0, stream>>>(a);
    GUNET_LAUNCH_CHECK();
  }

  auto w_of = [&](int seg, int col) { return w->node_w + row_of[seg] * U + col; };
  const int n_partials = columns_colsum_rows(g, U, batch, ws.wpack);
  g.colsum_tiles = n_partials;
  GUNET_TRY(launch_gemm_columns(g, U, batch, ws.gbias, U, x_out, ws.npart, w_of, ws.wpack, stream));

  if (n_partials_out) *n_partials_out = n_partials;
  if (flags & kGnSkipGlobalUpdate) return GUNET_OK;
  return launch_global_update(p, ut, global_in, n_partials, relu, batch, ws, global_out, stream);
}

int gn_upsample_run(const gunet_quadgraph* graph, int32_t n_coarse, const gunet_gn_weights* w, int32_t batch,
                    const float* x, const float* global_in, int32_t activation, float* x_out, float* global_out,
                    void* workspace, size_t workspace_bytes, int flags, int* n_partials_out, hipStream_t stream) {
  LayerWorkspace ws;
  GUNET_TRY(check_layer_args("gn_upsample", graph, w, batch, activation, workspace, workspace_bytes, &ws));
  GUNET_REQUIRE(x && x_out, "gn_upsample: null tensor");
  GUNET_REQUIRE(global_in || (flags & kGnSkipPrologue && flags & kGnSkipGlobalUpdate), "gn_upsample: null global_in");
  GUNET_REQUIRE(global_out || (flags & kGnSkipGlobalUpdate), "gn_upsample: null global_out");
  GUNET_REQUIRE(n_coarse >= 0 && n_coarse <= graph->n_nodes, "gn_upsample: bad n_coarse");
  const int n_fine = graph->n_nodes - n_coarse;
  int64_t n_edges = 0;
  for (int q = 0; q < 4; ++q) n_edges += graph->n_edges[q];
  GUNET_REQUIRE(n_edges == n_fine,
                "gn_upsample: %lld edges for %d fine nodes (every fine node needs exactly one parent edge)",
                static_cast<long long>(n_edges), n_fine);
  GUNET_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(x_out) % 16 == 0,
                "gn_upsample: x and x_out must be 16-byte aligned");
  const int relu = activation == GUNET_ACT_RELU;
  const int U = w->units, u = U / 4, ut = U / 128;
  const LayerPtrs p = layer_ptrs(w);

  // The layer input is [x ; 0]: rows n_coarse.. are zero, so their embedding is act(bias) for every
  // fine node; its product with the receiver rows of the edge kernel is folded into the edge bias.
  if (!(flags & kGnSkipPrologue)) GUNET_TRY(launch_prologue(p, ut, global_in, relu, 1, batch, ws, stream));

  {  // embeddings of the coarse rows only
    GemmArgs g = {};
    g.n_seg = 1;
    g.seg[0].a = x;
    g.seg[0].lda = w->c_in;
    g.seg[0].k = w->c_in;
    g.seg[0].ldw = u;
    g.n_rows = n_coarse;
    g.relu = relu;
    auto w_of = [&](int, int col) { return w->node_emb_w[col / u] + col % u; };
    GUNET_TRY(launch_gemm_columns(g, U, batch, ws.nbias, 0, ws.xq, nullptr, w_of, ws.wpack, stream));
  }

  // messages: every edge goes (coarse parent -> fine node); edge features are zero.  Stored
  // already scaled by 1/4, quadrant after quadrant, in ws.agg viewed as [B, n_fine, u].
  float* msg[4];
  {
    int64_t off = 0;
    EdgeArgs a = {};
    for (int q = 0; q < 4; ++q) {
      msg[q] = ws.agg + off;
      off += static_cast<int64_t>(batch) * graph->n_edges[q] * u;
      a.edge_w[q] = w->edge_w[q];
      a.out[q] = msg[q];
    }
    a.n_nodes = n_coarse;
    a.f_e = w->f_e;
    a.xq = ws.xq;
    a.ebias = ws.ebias;
    a.partial = ws.epart;
    a.relu = relu;
    a.sender_only = 1;
    a.out_scale = 1.0f / (3 + 1);
    GUNET_TRY(launch_edge_mlp(a, ut, graph, nullptr, ws.n_eblk, batch, stream));
  }

  // node MLP.  Coarse rows: only the x columns are non-zero and only their column sums are used
  // (the rows themselves are dropped by the caller, hybrid_gnn.py:377).  Fine rows: x is zero and
  // the aggregate is the single incoming message in the columns of its quadrant, so each
  // quadrant's edges are one K=u GEMM scattered to the fine rows.
  GemmArgs gc = {};
  gc.n_seg = 1;
  gc.seg[0].a = x;
  gc.seg[0].lda = w->c_in;
  gc.seg[0].k = w->c_in;
  gc.seg[0].ldw = U;
  gc.n_rows = n_coarse;
  gc.relu = relu;
  GemmArgs gq[4] = {};
  int tiles_total = columns_colsum_rows(gc, U, batch, ws.wpack);
  for (int q = 0; q < 4; ++q) {
    GemmArgs& g = gq[q];
    g.n_seg = 1;
    g.seg[0].a = msg[q];
    g.seg[0].lda = u;
    g.seg[0].k = u;
    g.seg[0].ldw = U;
    g.n_rows = graph->n_edges[q];
    g.out_rows = n_fine;
    g.row_map = graph->edge_index[q] + 1;   // receiver column of the edge list
    g.row_map_stride = 2;
    g.row_map_sub = n_coarse;
    g.relu = relu;
    g.colsum_tile0 = tiles_total;
    tiles_total += columns_colsum_rows(g, U, batch, ws.wpack);
  }
  GUNET_REQUIRE(tiles_total <= ws.n_nblk + 5, "gn_upsample: %d column-sum rows for a buffer of %d", tiles_total,
                ws.n_nblk + 5);
  if (n_coarse > 0) {
    gc.colsum_tiles = tiles_total;
    auto w_of = [&](int, int col) { return w->node_w + col; };
    GUNET_TRY(launch_gemm_columns(gc, U, batch, ws.gbias, U, nullptr, ws.npart, w_of, ws.wpack, stream));
  }
  for (int col0 = 0; col0 < U; col0 += 128) {   // the four quadrants' problems share a launch
    GemmArgs blockq[4];
    for (int q = 0; q < 4; ++q) {
      gq[q].colsum_tiles = tiles_total;
      auto w_of = [&](int, int col) { return w->node_w + static_cast<int64_t>(w->c_in + u * q) * U + col; };
      blockq[q] = columns_block(gq[q], U, col0, 4, ws.gbias, U, x_out, ws.npart, w_of);
    }
    GUNET_TRY(launch_gemm_group(blockq, 4, 4, batch, ws.wpack, stream));
  }
  if (n_partials_out) *n_partials_out = tiles_total;
  if (flags & kGnSkipGlobalUpdate) return GUNET_OK;
  return launch_global_update(p, ut, global_in, tiles_total, relu, batch, ws, global_out, stream);
}

int gn_bridge(const gunet_quadgraph* cur_graph, const gunet_gn_weights* cur, int n_partials, const float* global_in,
              float* global_out, const float* extra, int extra_width, float* cat_out, const gunet_quadgraph* next_graph,
              const gunet_gn_weights* next, int next_fold_receiver, int32_t batch, int32_t activation, void* workspace,
              size_t workspace_bytes, hipStream_t stream) {
  LayerWorkspace wc, wn;
  GUNET_TRY(check_layer_args("gn_bridge", cur_graph, cur, batch, activation, workspace, workspace_bytes, &wc));
  GUNET_TRY(check_layer_args("gn_bridge", next_graph, next, batch, activation, workspace, workspace_bytes, &wn));
  GUNET_REQUIRE(global_in && global_out, "gn_bridge: null global");
  const int g_new = cur->globt_w ? cur->units : cur->g + cur->units + cur->units / 4;
  GUNET_REQUIRE(extra ? (extra_width > 0 && cat_out && g_new + extra_width == next->g) : g_new == next->g,
                "gn_bridge: the next layer takes a global of %d, this one gives %d (+ %d)", next->g, g_new,
                extra ? extra_width : 0);
  GUNET_REQUIRE(n_partials >= 0 && n_partials <= wc.n_nblk + 5, "gn_bridge: %d column-sum rows", n_partials);
  const LayerPtrs pc = layer_ptrs(cur), pn = layer_ptrs(next);
  const int utc = cur->units / 128, utn = next->units / 128;
  const size_t lds = sizeof(double) * 4096 + sizeof(float) * (4096 + pc.g + 160 * utc + pn.g + 4 * 32 * utn);
  GUNET_REQUIRE(lds <= 64 * 1024, "gn_bridge: global vectors of %d and %d floats do not fit", pc.g, pn.g);
  const int relu = activation == GUNET_ACT_RELU;
  ProfScope prof(kProfBridge, stream);
  return utc == 1 ? launch_bridge_utc<1>(pc, pn, utn, lds, batch, global_in, n_partials, wc, relu, global_out, extra,
                                         extra_width, cat_out, next_fold_receiver, wn, stream)
                  : launch_bridge_utc<2>(pc, pn, utn, lds, batch, global_in, n_partials, wc, relu, global_out, extra,
                                         extra_width, cat_out, next_fold_receiver, wn, stream);
}

}  // namespace gunet
