// Float32 feature kernels of the Graph U-Net forward path for gfx950 (MI355X):
//   * row gathers and CSR segment max / in-order segment sum (HBM-bound),
//   * a 128-row tiled GEMM on v_mfma_f32_32x32x2_f32 with LDS-staged operand tiles,
//     fused bias + ReLU epilogue and deterministic per-block column sums,
//   * the GeoQuadrantGCN layer: four-direction node-embedding GEMM, gathered edge MLP on MFMA,
//     receiver-CSR aggregation, node MLP GEMM and the small global-feature updates.
// All reductions are atomic-free and run in a fixed order, so outputs are bitwise reproducible.
#include <cfloat>

#include "common.h"

namespace gunet {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
  const int beg = rowptr[seg], end = rowptr[seg + 1];
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = IS_MAX ? -FLT_MAX : 0.0f;
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
  ProfScope prof(kProfSegment, stream);
  if (vec)
    k_segment_reduce<4, IS_MAX><<<grid, kThreads, 0, stream>>>(src, n_src, width, rowptr, perm,
                                                               n_segments, scale, out);
  else
    k_segment_reduce<1, IS_MAX><<<grid, kThreads, 0, stream>>>(src, n_src, width, rowptr, perm,
                                                               n_segments, scale, out);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

// ---------------------------------------------------------------- GEMM on fp32 MFMA

// out[b, r, :] = act( sum_s A_s[b, r, :] @ W_s + bias[b] ),  r < n_rows, for up to three operand
// segments that are concatenated along K in the reference (x | aggregated messages, or
// upsampled x | skip x | aggregated messages).  One workgroup = 128 rows x (32*NT) columns,
// four waves of 32 rows each; K is consumed in tiles of 32 staged through LDS.
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
  int n_rows;                 // rows per sample
  const float* bias;          // [.., 32*NT] or null
  int bias_stride;            // floats between samples (0 = shared)
  float* out;                 // sample 0
  int ldo;
  int out_col0;               // first output column of this launch
  float* colsum;              // [B][gridDim.x][32*NT] per-block column sums of the output, or null
  int relu;
};

constexpr int kBM = 128;
constexpr int kBK = 32;
constexpr int kLda = kBK + 4;   // padded LDS row of the A tile (conflict-free ds_read_b128)

template <int NT>
__global__ __launch_bounds__(kThreads, 2) void k_gemm_f32(GemmArgs g) {
  constexpr int BN = 32 * NT;
  __shared__ __attribute__((aligned(16))) float As[kBM * kLda];
  __shared__ __attribute__((aligned(16))) float Bs[kBK * BN];
  __shared__ float red[4 * BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t b = blockIdx.y;
  const int row0 = blockIdx.x * kBM;

  // staging assignment: A tile = 128 rows x 8 float4, B tile = 32 rows x 8*NT float4
  const int a_c4 = tid & 7, a_r = tid >> 3;
  float4 a_reg[4];
  float4 b_reg[NT];

  int n_tiles = 0;
  for (int s = 0; s < g.n_seg; ++s) n_tiles += (g.seg[s].k + kBK - 1) / kBK;

  auto load_tile = [&](int t) {
    int s = 0;
    while (true) {
      const int nt = (g.seg[s].k + kBK - 1) / kBK;
      if (t < nt) break;
      t -= nt;
      ++s;
    }
    const GemmSeg& sg = g.seg[s];
    const int k0 = t * kBK;
    const float* abase = sg.a + (b * g.n_rows + row0) * static_cast<int64_t>(sg.lda);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = a_r + 32 * j;
      const int k = k0 + 4 * a_c4;
      if (row0 + r < g.n_rows && k < sg.k)
        a_reg[j] = *reinterpret_cast<const float4*>(abase + static_cast<int64_t>(r) * sg.lda + k);
      else
        a_reg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int idx = tid + kThreads * j;
      const int kr = idx / (8 * NT), c4 = idx % (8 * NT);
      const int k = k0 + kr;
      if (k < sg.k)
        b_reg[j] = *reinterpret_cast<const float4*>(sg.w[c4 >> 3] +
                                                    static_cast<int64_t>(k) * sg.ldw + 4 * (c4 & 7));
      else
        b_reg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(&As[(a_r + 32 * j) * kLda + 4 * a_c4]) = a_reg[j];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int idx = tid + kThreads * j;
      const int kr = idx / (8 * NT), c4 = idx % (8 * NT);
      *reinterpret_cast<float4*>(&Bs[kr * BN + 4 * c4]) = b_reg[j];
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

  load_tile(0);
  store_tile();
  __syncthreads();
  for (int t = 0; t < n_tiles; ++t) {
    if (t + 1 < n_tiles) load_tile(t + 1);
    // each 8-wide K chunk: half-wave h feeds k = 8c+4h .. 8c+4h+3, one MFMA per component
#pragma unroll
    for (int c = 0; c < kBK / 8; ++c) {
      const float4 av =
          *reinterpret_cast<const float4*>(&As[(32 * wave + li) * kLda + 8 * c + 4 * lh]);
      const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int n = 0; n < NT; ++n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float bv = Bs[(8 * c + 4 * lh + j) * BN + 32 * n + li];
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], bv, acc[n], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (t + 1 < n_tiles) {
      store_tile();
      __syncthreads();
    }
  }

  // epilogue: accumulator register r of lane (li, lh) is row (r&3) + 8*(r>>2) + 4*lh, column li
  float* obase = g.out + (b * g.n_rows + row0) * static_cast<int64_t>(g.ldo) + g.out_col0;
  const float* bias = g.bias ? g.bias + b * g.bias_stride : nullptr;
  float cs[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const float bv = bias ? bias[32 * n + li] : 0.0f;
    cs[n] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float v = activate(acc[n][r] + bv, g.relu);
      if (row0 + row < g.n_rows) {
        obase[static_cast<int64_t>(row) * g.ldo + 32 * n + li] = v;
        cs[n] += v;
      }
    }
  }
  if (g.colsum != nullptr) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      cs[n] += __shfl_xor(cs[n], 32);
      if (lh == 0) red[wave * BN + 32 * n + li] = cs[n];
    }
    __syncthreads();
    if (tid < BN) {
      const float s = ((red[tid] + red[BN + tid]) + red[2 * BN + tid]) + red[3 * BN + tid];
      g.colsum[(b * gridDim.x + blockIdx.x) * BN + tid] = s;
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

int launch_gemm(const GemmArgs& g, int nt, int64_t batch, hipStream_t stream) {
  if (g.n_rows == 0 || batch == 0) return GUNET_OK;
  dim3 grid(static_cast<unsigned>(ceil_div(g.n_rows, kBM)), static_cast<unsigned>(batch));
  ProfScope prof(kProfGemm, stream);
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

// ---------------------------------------------------------------- GeoQuadrantGCN pieces

constexpr int kU = 32;           // per-direction width (units / 4)
constexpr int kUnits = 128;
constexpr int kEdgeGroups = 4;   // 32-edge groups per wave
constexpr int kEdgesPerBlock = 4 * 32 * kEdgeGroups;

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

constexpr int kSmallThreads = 1024;   // the two per-sample kernels: 8 k-groups x 128 columns
constexpr int kGroups = kSmallThreads / kUnits;

// Sum of `n` per-group partials for column `col`, in group order.
__device__ inline float combine_groups(const float* part, int col) {
  float s = part[col];
#pragma unroll
  for (int g = 1; g < kGroups; ++g) s += part[g * kUnits + col];
  return s;
}

// Per sample: the chained global transforms (hybrid_gnn.py:98-99), folded into the bias each
// direction's edge MLP sees; the node MLP's global columns folded into its bias (:122-125);
// and the four node-embedding biases laid side by side.  Every dot product is split over 8
// thread groups by k (k = group, group + 8, ...) and the 8 partials are added in group order.
__global__ __launch_bounds__(kSmallThreads) void k_layer_prologue(LayerPtrs p,
                                                                  const float* __restrict__ global_in,
                                                                  int relu, float* __restrict__ nbias,
                                                                  float* __restrict__ ebias,
                                                                  float* __restrict__ gbias) {
  extern __shared__ float sh[];
  float* gl = sh;                      // [g]
  float* gq = gl + p.g;                // [4][32]
  float* part = gq + 4 * kU;           // [8][128]
  const int tid = threadIdx.x;
  const int grp = tid >> 7, col = tid & 127;
  const int64_t b = blockIdx.x;
  for (int k = tid; k < p.g; k += blockDim.x) gl[k] = global_in[b * p.g + k];
  __syncthreads();
  for (int q = 0; q < 4; ++q) {
    const float* src = q == 0 ? gl : gq + (q - 1) * kU;
    const int kk = q == 0 ? p.g : kU;
    if (col < kU) {
      float s = 0.0f;
      for (int k = grp; k < kk; k += kGroups) s = fmaf(src[k], p.glob_w[q][k * kU + col], s);
      part[grp * kUnits + col] = s;
    }
    __syncthreads();
    if (tid < kU) {
      float s = combine_groups(part, tid);
      if (p.glob_b[q]) s += p.glob_b[q][tid];
      gq[q * kU + tid] = activate(s, relu);
    }
    __syncthreads();
  }
  {
    const float* w = p.node_w + static_cast<int64_t>(p.c_in + kUnits) * kUnits;
    float s = 0.0f;
    for (int k = grp; k < p.g; k += kGroups) s = fmaf(gl[k], w[static_cast<int64_t>(k) * kUnits + col], s);
    part[grp * kUnits + col] = s;
  }
  __syncthreads();
  if (tid < kUnits) {
    float s = combine_groups(part, tid);
    if (p.node_b) s += p.node_b[tid];
    gbias[b * kUnits + tid] = s;
    const int q = tid >> 5, j = tid & 31;
    const float* w = p.edge_w[q] + static_cast<int64_t>(p.f_e + 2 * kU) * kU;
    float e = 0.0f;
    for (int k = 0; k < kU; ++k) e = fmaf(gq[q * kU + k], w[k * kU + j], e);
    if (p.edge_b[q]) e += p.edge_b[q][j];
    ebias[b * kUnits + tid] = e;
    if (b == 0) nbias[tid] = p.node_emb_b[q] ? p.node_emb_b[q][j] : 0.0f;
  }
}

struct EdgeArgs {
  const int32_t* edge_index[4];
  const float* efeat[4];    // [B, E_q, f_e] or null
  const float* edge_w[4];   // [f_e + 96, 32]
  float* out[4];            // [B, E_q, 32]
  int n_edges[4];
  int n_nodes;
  int f_e;
  const float* xq;          // [B, N, 128] node embeddings, direction q in columns 32q..32q+31
  const float* ebias;       // [B, 4, 32]
  float* partial;           // [B, 4, gridDim.x, 32] column sums of the messages
  int relu;
};

// Edge MLP for 32 edges per wave step: the A operand row of edge e is
// [edge features | embedding of the sender | embedding of the receiver], gathered straight into
// registers as float4 pieces (half-wave h takes floats 8c+4h..8c+4h+3 of every 8-float chunk);
// the matching weight rows stay in registers for the whole workgroup.
template <int FE_CHUNKS, bool FE_VEC>
__global__ __launch_bounds__(kThreads) void k_edge_mlp(EdgeArgs a) {
  constexpr int CH = FE_CHUNKS + 8;
  __shared__ float red[4 * kU];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int q = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int n_edges = a.n_edges[q];
  const int64_t block_base = static_cast<int64_t>(blockIdx.x) * kEdgesPerBlock;
  float colsum = 0.0f;

  if (block_base < n_edges) {
    const float* w = a.edge_w[q];
    float wreg[CH * 4];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int row;
        if (c < FE_CHUNKS) {
          row = 8 * c + 4 * lh + j;
          wreg[4 * c + j] = row < a.f_e ? w[row * kU + li] : 0.0f;
        } else {
          row = a.f_e + 8 * (c - FE_CHUNKS) + 4 * lh + j;
          wreg[4 * c + j] = w[row * kU + li];
        }
      }
    }
    const float bias = a.ebias[(b * 4 + q) * kU + li];
    const int2* ei = reinterpret_cast<const int2*>(a.edge_index[q]);
    const float* xq = a.xq + b * a.n_nodes * static_cast<int64_t>(kUnits) + kU * q;
    const float* ef = a.efeat[q] ? a.efeat[q] + b * n_edges * static_cast<int64_t>(a.f_e) : nullptr;
    float* out = a.out[q] + b * n_edges * static_cast<int64_t>(kU);

    for (int grp = 0; grp < kEdgeGroups; ++grp) {
      const int64_t e0 = block_base + (wave * kEdgeGroups + grp) * 32;
      if (e0 >= n_edges) break;
      const int64_t e = e0 + li < n_edges ? e0 + li : n_edges - 1;
      const int2 sr = ei[e];
      float4 av[CH];
      if (FE_CHUNKS > 0) {
        if (FE_VEC) {
          const float4* pf = reinterpret_cast<const float4*>(ef + e * a.f_e) + lh;
#pragma unroll
          for (int c = 0; c < FE_CHUNKS; ++c) av[c] = pf[2 * c];
        } else {
#pragma unroll
          for (int c = 0; c < FE_CHUNKS; ++c) {
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int k = 8 * c + 4 * lh + j;
              t[j] = (ef != nullptr && k < a.f_e) ? ef[e * a.f_e + k] : 0.0f;
            }
            av[c] = make_float4(t[0], t[1], t[2], t[3]);
          }
        }
      }
      const float4* ps = reinterpret_cast<const float4*>(xq + static_cast<int64_t>(sr.x) * kUnits) + lh;
      const float4* pr = reinterpret_cast<const float4*>(xq + static_cast<int64_t>(sr.y) * kUnits) + lh;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        av[FE_CHUNKS + c] = ps[2 * c];
        av[FE_CHUNKS + 4 + c] = pr[2 * c];
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].x, wreg[4 * c + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].y, wreg[4 * c + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].z, wreg[4 * c + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].w, wreg[4 * c + 3], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = e0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = activate(acc[r] + bias, a.relu);
        if (row < n_edges) {
          out[row * kU + li] = v;
          colsum += v;
        }
      }
    }
  }
  colsum += __shfl_xor(colsum, 32);
  if (lh == 0) red[wave * kU + li] = colsum;
  __syncthreads();
  if (tid < kU) {
    const float s = ((red[tid] + red[kU + tid]) + red[2 * kU + tid]) + red[3 * kU + tid];
    a.partial[((b * 4 + q) * gridDim.x + blockIdx.x) * kU + tid] = s;
  }
}

struct AggArgs {
  const int32_t* rowptr[4];
  const int32_t* perm[4];
  const float* msg[4];   // [B, E_q, 32]
  int n_edges[4];
  int n_nodes;
  float scale;
  float* agg;            // [B, N, 128]
};

// Receiver-CSR aggregation (unsorted_segment_sum, hybrid_gnn.py:110-113): 32 lanes own one
// receiver's 128 output floats, 8 lanes x float4 per direction; messages are added in the
// reference's edge order, starting from zero, then scaled.
__global__ __launch_bounds__(kThreads) void k_aggregate(AggArgs a) {
  const int tid = threadIdx.x;
  const int64_t v = static_cast<int64_t>(blockIdx.x) * (kThreads / 32) + (tid >> 5);
  if (v >= a.n_nodes) return;
  const int64_t b = blockIdx.y;
  const int sub = tid & 31, q = sub >> 3, c4 = sub & 7;
  const int32_t* rowptr = q == 0 ? a.rowptr[0] : q == 1 ? a.rowptr[1] : q == 2 ? a.rowptr[2] : a.rowptr[3];
  const int32_t* perm = q == 0 ? a.perm[0] : q == 1 ? a.perm[1] : q == 2 ? a.perm[2] : a.perm[3];
  const float* msg = q == 0 ? a.msg[0] : q == 1 ? a.msg[1] : q == 2 ? a.msg[2] : a.msg[3];
  const int ne = q == 0 ? a.n_edges[0] : q == 1 ? a.n_edges[1] : q == 2 ? a.n_edges[2] : a.n_edges[3];
  msg += b * ne * static_cast<int64_t>(kU) + 4 * c4;
  const int beg = rowptr[v], end = rowptr[v + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = beg; s < end; ++s) {
    const float4 m = *reinterpret_cast<const float4*>(msg + static_cast<int64_t>(perm[s]) * kU);
    acc.x += m.x;
    acc.y += m.y;
    acc.z += m.z;
    acc.w += m.w;
  }
  acc.x *= a.scale;
  acc.y *= a.scale;
  acc.z *= a.scale;
  acc.w *= a.scale;
  *reinterpret_cast<float4*>(a.agg + (b * a.n_nodes + v) * kUnits + kU * q + 4 * c4) = acc;
}

// Per sample: finish the node / edge column sums (fp64 accumulation over the per-block
// partials, block i handled by thread group i mod 8, groups combined in order), scale them and
// apply the global transform (hybrid_gnn.py:130-135).
__global__ __launch_bounds__(kSmallThreads) void k_global_update(LayerPtrs p,
                                                                 const float* __restrict__ global_in,
                                                                 const float* __restrict__ npart, int n_nblk,
                                                                 const float* __restrict__ epart, int n_eblk,
                                                                 int relu, float* __restrict__ global_out) {
  extern __shared__ __attribute__((aligned(16))) float gsh[];
  double* dpart = reinterpret_cast<double*>(gsh);                    // [8][128]
  float* part = reinterpret_cast<float*>(dpart + kGroups * kUnits);  // [8][128]
  float* gin = part + kGroups * kUnits;                              // [g + 128 + 32]
  const int tid = threadIdx.x;
  const int grp = tid >> 7, col = tid & 127;
  const int64_t b = blockIdx.x;
  for (int k = tid; k < p.g; k += blockDim.x) gin[k] = global_in[b * p.g + k];
  {
    double s = 0.0;
    const float* src = npart + b * n_nblk * static_cast<int64_t>(kUnits) + col;
    for (int i = grp; i < n_nblk; i += kGroups) s += static_cast<double>(src[static_cast<int64_t>(i) * kUnits]);
    dpart[grp * kUnits + col] = s;
  }
  __syncthreads();
  if (tid < kUnits) {
    double s = dpart[tid];
    for (int g = 1; g < kGroups; ++g) s += dpart[g * kUnits + tid];
    const float denom = p.gsn * 300.0f * 300.0f;
    gin[p.g + tid] = (static_cast<float>(s) * 1.0f) / denom;
  }
  __syncthreads();
  {
    const int q = col >> 5, c = col & 31;
    double s = 0.0;
    const float* src = epart + (b * 4 + q) * n_eblk * static_cast<int64_t>(kU) + c;
    for (int i = grp; i < n_eblk; i += kGroups) s += static_cast<double>(src[static_cast<int64_t>(i) * kU]);
    dpart[grp * kUnits + col] = s;
  }
  __syncthreads();
  if (tid < kU) {
    float total = 0.0f;
    for (int q = 0; q < 4; ++q) {
      double s = dpart[q * kU + tid];
      for (int g = 1; g < kGroups; ++g) s += dpart[g * kUnits + q * kU + tid];
      total += static_cast<float>(s);
    }
    const float denom = p.gse * 300.0f * 300.0f * 8.0f;
    gin[p.g + kUnits + tid] = (total * 1.0f) / denom;
  }
  __syncthreads();
  const int kin = p.g + kUnits + kU;
  if (p.globt_w != nullptr) {
    float s = 0.0f;
    for (int k = grp; k < kin; k += kGroups) s = fmaf(gin[k], p.globt_w[static_cast<int64_t>(k) * kUnits + col], s);
    part[grp * kUnits + col] = s;
    __syncthreads();
    if (tid < kUnits) {
      float t = combine_groups(part, tid);
      if (p.globt_b) t += p.globt_b[tid];
      global_out[b * kUnits + tid] = activate(t, relu);
    }
  } else {
    for (int k = tid; k < kin; k += blockDim.x) global_out[b * kin + k] = gin[k];
  }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct LayerWorkspace {
  float* xq;      // [B, N, 128]
  float* agg;     // [B, N, 128]
  float* nbias;   // [128]
  float* ebias;   // [B, 128]
  float* gbias;   // [B, 128]
  float* npart;   // [B, n_nblk, 128]
  float* epart;   // [B, 4, n_eblk, 32]
  int n_nblk, n_eblk;
  size_t bytes;
};

LayerWorkspace carve_workspace(const gunet_quadgraph* g, int32_t batch, void* base) {
  LayerWorkspace ws;
  const size_t n = static_cast<size_t>(g->n_nodes), bsz = static_cast<size_t>(batch);
  int max_e = 0;
  for (int q = 0; q < 4; ++q) max_e = g->n_edges[q] > max_e ? g->n_edges[q] : max_e;
  ws.n_nblk = static_cast<int>(ceil_div(g->n_nodes, kBM));
  ws.n_eblk = static_cast<int>(ceil_div(max_e > 0 ? max_e : 1, kEdgesPerBlock));
  size_t off = 0;
  auto take = [&](size_t floats) {
    float* p = base ? reinterpret_cast<float*>(static_cast<char*>(base) + off) : nullptr;
    off = align_up(off + floats * sizeof(float), 256);
    return p;
  };
  ws.xq = take(bsz * n * kUnits);
  ws.agg = take(bsz * n * kUnits);
  ws.nbias = take(kUnits);
  ws.ebias = take(bsz * kUnits);
  ws.gbias = take(bsz * kUnits);
  ws.npart = take(bsz * ws.n_nblk * kUnits);
  ws.epart = take(bsz * 4 * ws.n_eblk * kU);
  ws.bytes = off;
  return ws;
}

}  // namespace
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

int gunet_dense_forward_f32(const float* x, int64_t rows, int32_t k, const float* kernel,
                            const float* bias, int32_t n_out, int32_t activation, float* out,
                            gunet_stream_t stream_) {
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
    g.bias_stride = 0;
    g.out = out;
    g.ldo = n_out;
    g.out_col0 = col0;
    g.colsum = nullptr;
    g.relu = activation == GUNET_ACT_RELU;
    GUNET_TRY(launch_gemm(g, nt, 1, stream));
  }
  return GUNET_OK;
}

size_t gunet_gn_layer_workspace_bytes(const gunet_quadgraph* graph, int32_t batch, int32_t units) {
  if (graph == nullptr || batch <= 0 || units != kUnits) return 0;
  return carve_workspace(graph, batch, nullptr).bytes;
}

int gunet_gn_layer_forward_f32(const gunet_quadgraph* graph, const gunet_gn_weights* w,
                               int32_t batch, const float* x, const float* const edges[4],
                               const float* global_in, int32_t activation, float* x_out,
                               float* const edges_out[4], float* global_out, void* workspace,
                               size_t workspace_bytes, gunet_stream_t stream_) {
  hipStream_t stream = as_stream(stream_);
  GUNET_REQUIRE(graph && w, "gn_layer: null graph or weights");
  GUNET_REQUIRE(batch > 0, "gn_layer: batch must be positive");
  GUNET_REQUIRE(activation == GUNET_ACT_LINEAR || activation == GUNET_ACT_RELU,
                "gn_layer: unknown activation %d", activation);
  if (w->units != kUnits) {
    set_error("gn_layer: units = %d is not implemented (only %d)", w->units, kUnits);
    return GUNET_ERR_UNSUPPORTED;
  }
  GUNET_REQUIRE(w->c_in > 0 && w->c_in % 4 == 0, "gn_layer: c_in = %d must be a positive multiple of 4", w->c_in);
  GUNET_REQUIRE(w->g > 0 && w->g <= 4096, "gn_layer: global width %d outside 1..4096", w->g);
  if (w->f_e < 0 || w->f_e > 64) {
    set_error("gn_layer: edge feature width %d outside 0..64 is not implemented", w->f_e);
    return GUNET_ERR_UNSUPPORTED;
  }
  GUNET_REQUIRE(graph->n_nodes >= 0, "gn_layer: negative node count");
  GUNET_REQUIRE(x && global_in && x_out && global_out && edges && edges_out, "gn_layer: null tensor");
  GUNET_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(x_out) % 16 == 0,
                "gn_layer: x and x_out must be 16-byte aligned");
  GUNET_REQUIRE(w->node_w, "gn_layer: node_transform kernel is null");
  for (int q = 0; q < 4; ++q) {
    GUNET_REQUIRE(graph->n_edges[q] >= 0, "gn_layer: negative edge count");
    GUNET_REQUIRE(w->node_emb_w[q] && w->edge_w[q] && w->glob_w[q], "gn_layer: null kernel for direction %d", q);
    if (graph->n_edges[q] > 0) {
      GUNET_REQUIRE(graph->edge_index[q] && graph->perm[q] && edges_out[q],
                    "gn_layer: null edge tensor for direction %d", q);
    }
    GUNET_REQUIRE(graph->n_nodes == 0 || graph->rowptr[q], "gn_layer: null rowptr for direction %d", q);
  }
  const LayerWorkspace ws = carve_workspace(graph, batch, workspace);
  if (workspace == nullptr || workspace_bytes < ws.bytes) {
    set_error("gn_layer: workspace of %zu bytes is smaller than the %zu required", workspace_bytes, ws.bytes);
    return GUNET_ERR_WORKSPACE;
  }
  GUNET_REQUIRE(reinterpret_cast<uintptr_t>(workspace) % 16 == 0, "gn_layer: workspace must be 16-byte aligned");
  const int relu = activation == GUNET_ACT_RELU;
  const int n = graph->n_nodes;

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

  {
    ProfScope prof(kProfPrologue, stream);
    k_layer_prologue<<<batch, kSmallThreads, sizeof(float) * (w->g + 4 * kU + kGroups * kUnits), stream>>>(p, global_in, relu, ws.nbias,
                                                                              ws.ebias, ws.gbias);
    GUNET_LAUNCH_CHECK();
  }

  {  // node embeddings of the four directions in one GEMM (hybrid_gnn.py:91-92)
    GemmArgs g = {};
    g.n_seg = 1;
    g.seg[0].a = x;
    g.seg[0].lda = w->c_in;
    g.seg[0].k = w->c_in;
    g.seg[0].ldw = kU;
    for (int q = 0; q < 4; ++q) g.seg[0].w[q] = w->node_emb_w[q];
    g.n_rows = n;
    g.bias = ws.nbias;
    g.bias_stride = 0;
    g.out = ws.xq;
    g.ldo = kUnits;
    g.relu = relu;
    GUNET_TRY(launch_gemm(g, 4, batch, stream));
  }

  {  // edge MLP + message column sums (hybrid_gnn.py:93-108)
    EdgeArgs a;
    bool any_feat = false, vec_ok = w->f_e % 8 == 0;
    for (int q = 0; q < 4; ++q) {
      a.edge_index[q] = graph->edge_index[q];
      a.efeat[q] = edges[q];
      a.edge_w[q] = w->edge_w[q];
      a.out[q] = edges_out[q];
      a.n_edges[q] = graph->n_edges[q];
      if (graph->n_edges[q] > 0 && edges[q] != nullptr) {
        any_feat = true;
        if (reinterpret_cast<uintptr_t>(edges[q]) % 16 != 0) vec_ok = false;
      }
    }
    a.n_nodes = n;
    a.f_e = w->f_e;
    a.xq = ws.xq;
    a.ebias = ws.ebias;
    a.partial = ws.epart;
    a.relu = relu;
    // a direction whose features are absent reads nothing; with vector loads every direction
    // must have them, otherwise fall back to the guarded scalar loader
    bool all_feat = true;
    for (int q = 0; q < 4; ++q)
      if (graph->n_edges[q] > 0 && edges[q] == nullptr) all_feat = false;
    dim3 grid(static_cast<unsigned>(ws.n_eblk), 4, static_cast<unsigned>(batch));
    ProfScope prof(kProfEdgeMlp, stream);
    const int chunks = (!any_feat || w->f_e == 0) ? 0 : (w->f_e + 7) / 8;
    if (chunks == 0) {
      k_edge_mlp<0, false><<<grid, kThreads, 0, stream>>>(a);
    } else if (vec_ok && all_feat && chunks == 2) {
      k_edge_mlp<2, true><<<grid, kThreads, 0, stream>>>(a);
    } else if (vec_ok && all_feat && chunks == 4) {
      k_edge_mlp<4, true><<<grid, kThreads, 0, stream>>>(a);
    } else if (chunks == 1) {
      k_edge_mlp<1, false><<<grid, kThreads, 0, stream>>>(a);
    } else if (chunks == 2) {
      k_edge_mlp<2, false><<<grid, kThreads, 0, stream>>>(a);
    } else if (chunks <= 4) {
      k_edge_mlp<4, false><<<grid, kThreads, 0, stream>>>(a);
    } else {
      k_edge_mlp<8, false><<<grid, kThreads, 0, stream>>>(a);
    }
    GUNET_LAUNCH_CHECK();
  }

  if (n > 0) {  // neighbour aggregation (hybrid_gnn.py:110-113)
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
    dim3 grid(static_cast<unsigned>(ceil_div(n, kThreads / 32)), static_cast<unsigned>(batch));
    ProfScope prof(kProfAggregate, stream);
    k_aggregate<<<grid, kThreads, 0, stream>>>(a);
    GUNET_LAUNCH_CHECK();
  }

  {  // node MLP over [x | aggregated messages], global columns folded into the bias (:118-126)
    GemmArgs g = {};
    g.n_seg = 2;
    g.seg[0].a = x;
    g.seg[0].lda = w->c_in;
    g.seg[0].k = w->c_in;
    g.seg[0].ldw = kUnits;
    g.seg[1].a = ws.agg;
    g.seg[1].lda = kUnits;
    g.seg[1].k = kUnits;
    g.seg[1].ldw = kUnits;
    for (int t = 0; t < 4; ++t) {
      g.seg[0].w[t] = w->node_w + 32 * t;
      g.seg[1].w[t] = w->node_w + static_cast<int64_t>(w->c_in) * kUnits + 32 * t;
    }
    g.n_rows = n;
    g.bias = ws.gbias;
    g.bias_stride = kUnits;
    g.out = x_out;
    g.ldo = kUnits;
    g.colsum = ws.npart;
    g.relu = relu;
    GUNET_TRY(launch_gemm(g, 4, batch, stream));
  }

  {
    ProfScope prof(kProfGlobalUpdate, stream);
    k_global_update<<<batch, kSmallThreads,
                      sizeof(double) * kGroups * kUnits + sizeof(float) * (kGroups * kUnits + w->g + kUnits + kU), stream>>>(
        p, global_in, ws.npart, n > 0 ? ws.n_nblk : 0, ws.epart, ws.n_eblk, relu, global_out);
    GUNET_LAUNCH_CHECK();
  }
  return GUNET_OK;
}

}  // extern "C"
