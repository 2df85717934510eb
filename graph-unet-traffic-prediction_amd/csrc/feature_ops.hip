// Float32 feature kernels of the Graph U-Net forward path for gfx950 (MI355X):
//   * row gathers and CSR segment max / in-order segment sum (HBM-bound),
//   * the GeoQuadrantGCN layer: four-direction node-embedding GEMM, gathered edge MLP on MFMA,
//     receiver-CSR aggregation, node MLP GEMM (the GEMMs themselves live in gemm_ops.hip) and the
//     small per-sample global-feature kernels.
// All reductions are atomic-free and run in a fixed order, so outputs are bitwise reproducible.
#include <algorithm>
#include <cfloat>
#include <type_traits>

#include "gemm.h"

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

// ---------------------------------------------------------------- GeoQuadrantGCN pieces
//
// Kernels of one GN layer, templated on UT = units / 128: U = 128*UT output columns per node,
// u = 32*UT per direction.  UT = 1 is the shipped checkpoint's width, UT = 2 the 256-wide variant.

// The edge kernel's workgroup is four waves of EG consecutive 32-edge groups each: EG = 8 for large
// edge lists, 2 where that would leave most of the chip idle (every group is ~50 dependent MFMAs).
constexpr int kEdgeGroupsLarge = 8, kEdgeGroupsSmall = 2;
constexpr int edges_per_block(int eg) { return 4 * 32 * eg; }

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
// direction's edge MLP sees, and the node MLP's global columns folded into its bias (:122-125).
// Everything that does not depend on the
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
                                                  ProloguePre<UT>& pre) {
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
                                                                  float* __restrict__ ebias,
                                                                  float* __restrict__ gbias) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  float* part = sh;                      // 4096 floats, reused by every reduction
  float* gl = part + 4096;               // [g]
  float* gq = gl + p.g;                  // [4][u]
  const int64_t b = blockIdx.x;
  for (int k = threadIdx.x; k < p.g; k += blockDim.x) gl[k] = global_in[b * p.g + k];
  ProloguePre<UT> pre;
  prologue_prefetch<UT, false>(p, relu, fold_receiver, b, pre);
  prologue_compute<UT, false>(p, gl, relu, b, part, gq, ebias, gbias, pre);
}

struct EdgeArgs {
  const int32_t* edge_index[4];
  const float* efeat[4];    // [B, E_q, f_e] ([E_q, f_e] with ef_shared) or null
  int ef_shared;
  const float* edge_w[4];   // [f_e + 3u, u]
  float* out[4];            // [B, E_q, u]
  int n_edges[4];
  int n_nodes;
  int f_e;
  const float* xq;          // [B, N, U] node embeddings, direction q in columns u*q .. u*q+u-1
  const float* ebias;       // [B, 4, u]
  double* partial;          // [B, 4, gridDim.x, u] column sums of the messages
  int relu;
  int act;                  // activation code (the tuned kernels take linear / relu; others go the generic way)
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

template <int FE_CHUNKS, bool FE_VEC, bool SENDER_ONLY, int UT, int EG>
__global__ __launch_bounds__(kThreads) void k_edge_mlp(EdgeArgs a) {
  using Rows = EdgeRows<FE_CHUNKS, SENDER_ONLY, UT>;
  constexpr int CH = Rows::CH;
  constexpr int U = 128 * UT, u = 32 * UT;
  constexpr bool PIPE = UT == 1;   // two row buffers fit the register file only at u = 32
  // Weight image at u = 32: for 8-row chunk c one 1 KiB piece in lane order -- lane (lh, li) finds
  // rows 8c + 4lh + 0..3 of column li as one float4, so one ds_read_b128 feeds four MFMAs.  At
  // u = 64 the plain row-major image with one ds_read_b32 per MFMA is the faster of the two
  // (measured A/B on the dense 256-wide workload: 18.3 against 19.9 ms per step).
  constexpr bool LANE_IMAGE = UT == 1;
  __shared__ __attribute__((aligned(16))) float Ws[CH * 8 * u];
  __shared__ double red[4 * u];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int q = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int n_edges = a.n_edges[q];
  const int64_t block_base = static_cast<int64_t>(blockIdx.x) * edges_per_block(EG);
  // column sums: float32 over each group's 32 edges in a fixed order, float64 over the groups
  double colsum[UT];
#pragma unroll
  for (int t = 0; t < UT; ++t) colsum[t] = 0.0;

  if (block_base < n_edges) {   // uniform over the workgroup
    const float* w = a.edge_w[q];
    for (int i = tid; i < CH * 8 * u; i += kThreads) {
      const int row = i / u, col = i % u;
      float v;
      if (row < 8 * FE_CHUNKS)
        v = row < a.f_e ? w[row * u + col] : 0.0f;
      else
        v = w[(a.f_e + row - 8 * FE_CHUNKS) * u + col];
      if constexpr (LANE_IMAGE)
        Ws[((((row >> 3) * UT + (col >> 5)) * 64) + ((row >> 2) & 1) * 32 + (col & 31)) * 4 + (row & 3)] = v;
      else
        Ws[i] = v;
    }
    __syncthreads();
    const float4* wl = reinterpret_cast<const float4*>(Ws) + lane;
    float bias[UT];
#pragma unroll
    for (int t = 0; t < UT; ++t) bias[t] = a.ebias[(b * 4 + q) * u + 32 * t + li];
    const int2* ei = reinterpret_cast<const int2*>(a.edge_index[q]);
    const float* xq = a.xq + b * a.n_nodes * static_cast<int64_t>(U) + u * q;
    const float* ef = a.efeat[q] ? a.efeat[q] + (a.ef_shared ? 0 : b * n_edges * static_cast<int64_t>(a.f_e)) : nullptr;
    float* out = a.out[q] + b * n_edges * static_cast<int64_t>(u);
    const int64_t wave_base = block_base + static_cast<int64_t>(wave) * EG * 32;
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
      if constexpr (!LANE_IMAGE) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const float a4[4] = {cur.v[c].x, cur.v[c].y, cur.v[c].z, cur.v[c].w};
#pragma unroll
          for (int t = 0; t < UT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], Ws[(8 * c + 4 * lh + j) * u + 32 * t + li], acc[t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int c = 0; LANE_IMAGE && c < CH; ++c) {
        float4 wv[UT];
#pragma unroll
        for (int t = 0; t < UT; ++t) wv[t] = wl[(c * UT + t) * 64];
#pragma unroll
        for (int t = 0; t < UT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.v[c].x, wv[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < UT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.v[c].y, wv[t].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < UT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.v[c].z, wv[t].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < UT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.v[c].w, wv[t].w, acc[t], 0, 0, 0);
      }
      float gs[UT];
#pragma unroll
      for (int t = 0; t < UT; ++t) gs[t] = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = e0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (FULL || row < n_edges) {
#pragma unroll
          for (int t = 0; t < UT; ++t) {
            const float v = activate(acc[t][r] + bias[t], a.relu);
            out[row * u + 32 * t + li] = v * a.out_scale;
            gs[t] += v;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < UT; ++t) {
        gs[t] += __shfl_xor(gs[t], 32);
        colsum[t] += static_cast<double>(gs[t]);
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
        if (grp + 1 < EG && (FULL || e0 + 32 < n_edges)) {
          if (grp + 2 < EG) sr_nxt2 = ei[FULL ? e0 + 64 + li : edge_of(grp + 2)];
          gather_edge_rows<FE_CHUNKS, FE_VEC, SENDER_ONLY, UT>(xq, ef, a.f_e, FULL ? e0 + 32 + li : edge_of(grp + 1),
                                                               sr_nxt, lh, nxt);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the requests above the matrix work
        multiply(full_tag, e0, cur);
        __builtin_amdgcn_sched_barrier(0);
      };
      Rows r0, r1;
      int2 sr_a, sr_b;
      if (wave_base + EG * 32 <= n_edges) {
        sr_a = ei[wave_base + 32 + li];
        gather_edge_rows<FE_CHUNKS, FE_VEC, SENDER_ONLY, UT>(xq, ef, a.f_e, wave_base + li, ei[wave_base + li], lh, r0);
#pragma unroll
        for (int grp = 0; grp < EG; grp += 2) {
          step(std::true_type{}, grp, r0, r1, sr_a, sr_b);
          step(std::true_type{}, grp + 1, r1, r0, sr_b, sr_a);
        }
      } else if (wave_base < n_edges) {
        sr_a = ei[edge_of(1)];
        gather_edge_rows<FE_CHUNKS, FE_VEC, SENDER_ONLY, UT>(xq, ef, a.f_e, edge_of(0), ei[edge_of(0)], lh, r0);
        for (int grp = 0; grp < EG && wave_base + grp * 32 < n_edges; grp += 2) {
          step(std::false_type{}, grp, r0, r1, sr_a, sr_b);
          if (wave_base + (grp + 1) * 32 < n_edges) step(std::false_type{}, grp + 1, r1, r0, sr_b, sr_a);
        }
      }
    } else {
      for (int grp = 0; grp < EG && wave_base + grp * 32 < n_edges; ++grp) {
        Rows cur;
        gather_edge_rows<FE_CHUNKS, FE_VEC, SENDER_ONLY, UT>(xq, ef, a.f_e, edge_of(grp), ei[edge_of(grp)], lh, cur);
        multiply(std::false_type{}, wave_base + grp * 32, cur);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < UT; ++t)
    if (lh == 0) red[wave * u + 32 * t + li] = colsum[t];
  __syncthreads();
  if (tid < u) {
    const double s = ((red[tid] + red[u + tid]) + red[2 * u + tid]) + red[3 * u + tid];
    a.partial[((b * 4 + q) * gridDim.x + blockIdx.x) * u + tid] = s;
  }
}

// ---------------------------------------------------------------- bf16-storage forms
//
// The whole-model forward in GUNET_MATMUL_BF16 mode keeps its intermediates -- node features, node
// embeddings, messages -- in bf16 (the bf16 matrix instruction rounds its operands to bf16 anyway; a
// tensor stored in bf16 carries the same rounding and half the bytes).  Accumulation, biases,
// activations and column sums stay float32 / float64.  These kernels run at the speed of their memory
// traffic: the matrix work is 16 times cheaper than in float32.

typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4v pack_bf16x8(float4 lo, float4 hi) {
  bf16x8 pk;
  pk[0] = static_cast<__bf16>(lo.x); pk[1] = static_cast<__bf16>(lo.y);
  pk[2] = static_cast<__bf16>(lo.z); pk[3] = static_cast<__bf16>(lo.w);
  pk[4] = static_cast<__bf16>(hi.x); pk[5] = static_cast<__bf16>(hi.y);
  pk[6] = static_cast<__bf16>(hi.z); pk[7] = static_cast<__bf16>(hi.w);
  return __builtin_bit_cast(u32x4v, pk);
}

struct Edge16Args {
  const int32_t* edge_index[4];
  const void* efeat[4];     // FE = 32: bf16 [B, E_q, 32]; FE = 16: float32 [B, E_q, 16] ([E_q, 16] with ef_shared)
  int ef_shared;
  const float* edge_w[4];   // [f_e + 3u, u] float32 (rounded to bf16 on their way into registers)
  __bf16* out[4];           // [B, E_q, 32]
  float* out32[4];          // float32 copy of the same messages, [B, E_q, 32], or null
  int n_edges[4];
  int n_nodes;
  int f_e;                  // rows of the weight matrix in front of the sender block
  const __bf16* xq;         // [B, N, 128] node embeddings, direction q in columns 32q .. 32q+31
  const float* ebias;       // [B, 4, 32]
  double* partial;          // [B, 4, gridDim.x, 32] column sums of the (unscaled) messages
  int relu;
  float out_scale;
};

constexpr int kStageLd = 36;   // floats per staged row: 16-byte aligned rows, half-waves on different banks

// Edge MLP on v_mfma_f32_32x32x16_bf16, 32 edges per wave step: the operand row of edge e is
// [edge features | embedding of the sender | embedding of the receiver] in bf16, each lane's 8
// consecutive k of a 16-wide k step one 16-byte load that is the MFMA operand as it stands.  The
// weights live in registers (one bf16x8 fragment per k step).  FE: 32 = bf16 features (the messages
// of the layer before), 16 = float32 features (the model's input edge features, rounded on the fly),
// 0 = none.  Rows of group g + 1 and indices of group g + 2 are in flight behind the work of group g.
template <int FE, bool SENDER_ONLY>
struct Edge16Rows {
  static constexpr int KS_F = FE == 32 ? 2 : (FE == 16 ? 1 : 0);
  static constexpr int KS = KS_F + (SENDER_ONLY ? 2 : 4);
  u32x4v v[KS];
  float4 f[FE == 16 ? 2 : 1];   // float32 features wait here and are rounded when they are used
};

template <int FE, bool SENDER_ONLY, int EG>
__global__ __launch_bounds__(kThreads) void k_edge_mlp16(Edge16Args a) {
  using Rows = Edge16Rows<FE, SENDER_ONLY>;
  constexpr int KS_F = Rows::KS_F, KS = Rows::KS;
  __shared__ __attribute__((aligned(16))) float Stage[4][32 * kStageLd];
  __shared__ double red[4 * 32];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int q = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int n_edges = a.n_edges[q];
  const int64_t block_base = static_cast<int64_t>(blockIdx.x) * edges_per_block(EG);
  double colsum = 0.0;

  if (block_base < n_edges) {   // uniform over the workgroup
    bf16x8 wf[KS];
    {
      const float* w = a.edge_w[q];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int r0 = (s < KS_F ? 16 * s : a.f_e + 16 * (s - KS_F)) + 8 * lh;
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[s][j] = static_cast<__bf16>(w[(r0 + j) * 32 + li]);
      }
    }
    const float bias = a.ebias[(b * 4 + q) * 32 + li];
    const int2* ei = reinterpret_cast<const int2*>(a.edge_index[q]);
    const __bf16* xq = a.xq + b * a.n_nodes * static_cast<int64_t>(128) + 32 * q;
    const char* ef = static_cast<const char*>(a.efeat[q]);
    if (FE != 0 && !a.ef_shared) ef += b * n_edges * static_cast<int64_t>(64);   // 32 bf16 or 16 floats per edge
    __bf16* out = a.out[q] + b * n_edges * static_cast<int64_t>(32);
    float* out32 = a.out32[q] ? a.out32[q] + b * n_edges * static_cast<int64_t>(32) : nullptr;
    float* st = Stage[wave];
    const int64_t wave_base = block_base + static_cast<int64_t>(wave) * EG * 32;
    auto edge_of = [&](int grp) {
      const int64_t e = wave_base + grp * 32 + li;
      return e < n_edges ? e : static_cast<int64_t>(n_edges) - 1;   // clamped lanes are masked at the store
    };
    auto gather = [&](int64_t e, int2 sr, Rows& rows) {
      if constexpr (FE == 32) {
        const u32x4v* pf = reinterpret_cast<const u32x4v*>(ef + e * 64) + lh;
        rows.v[0] = pf[0];
        rows.v[1] = pf[2];
      }
      if constexpr (FE == 16) {
        const float4* pf = reinterpret_cast<const float4*>(ef + e * 64) + 2 * lh;
        rows.f[0] = pf[0];
        rows.f[1] = pf[1];
      }
      const u32x4v* ps = reinterpret_cast<const u32x4v*>(xq + static_cast<int64_t>(sr.x) * 128) + lh;
      rows.v[KS_F] = ps[0];
      rows.v[KS_F + 1] = ps[2];
      if constexpr (!SENDER_ONLY) {
        const u32x4v* pr = reinterpret_cast<const u32x4v*>(xq + static_cast<int64_t>(sr.y) * 128) + lh;
        rows.v[KS_F + 2] = pr[0];
        rows.v[KS_F + 3] = pr[2];
      }
    };
    auto multiply = [&](auto full_tag, int64_t e0, Rows& cur) {
      constexpr bool FULL = decltype(full_tag)::value;
      if constexpr (FE == 16) cur.v[0] = pack_bf16x8(cur.f[0], cur.f[1]);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.v[s]), wf[s], acc, 0, 0, 0);
      float gs = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = activate(acc[r] + bias, a.relu);
        if (FULL || e0 + row < n_edges) gs += v;
        st[row * kStageLd + li] = v * a.out_scale;
      }
      gs += __shfl_xor(gs, 32);
      colsum += static_cast<double>(gs);
      // the 32 x 32 tile leaves row-wise: the group's bf16 rows are 2 KiB of consecutive memory
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 16 * i + (lane >> 2), c8 = 8 * (lane & 3);
        const float4 v0 = lds_load_f4(&st[row * kStageLd + c8]);
        const float4 v1 = lds_load_f4(&st[row * kStageLd + c8 + 4]);
        if (FULL || e0 + row < n_edges)
          *reinterpret_cast<u32x4v*>(out + (e0 + row) * 32 + c8) = pack_bf16x8(v0, v1);
      }
      if (out32 != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 8 * i + (lane >> 3), c4 = 4 * (lane & 7);
          const float4 v = lds_load_f4(&st[row * kStageLd + c4]);
          if (FULL || e0 + row < n_edges) *reinterpret_cast<float4*>(out32 + (e0 + row) * 32 + c4) = v;
        }
      }
    };
    auto step = [&](auto full_tag, int grp, Rows& cur, Rows& nxt, int2& sr_nxt, int2& sr_nxt2) {
      constexpr bool FULL = decltype(full_tag)::value;
      const int64_t e0 = wave_base + grp * 32;
      if (grp + 1 < EG && (FULL || e0 + 32 < n_edges)) {
        if (grp + 2 < EG) sr_nxt2 = ei[FULL ? e0 + 64 + li : edge_of(grp + 2)];
        gather(FULL ? e0 + 32 + li : edge_of(grp + 1), sr_nxt, nxt);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the requests above the work of this group
      multiply(full_tag, e0, cur);
      __builtin_amdgcn_sched_barrier(0);
    };
    Rows r0, r1;
    int2 sr_a, sr_b;
    if (wave_base + EG * 32 <= n_edges) {
      sr_a = ei[wave_base + 32 + li];
      gather(wave_base + li, ei[wave_base + li], r0);
#pragma unroll
      for (int grp = 0; grp < EG; grp += 2) {
        step(std::true_type{}, grp, r0, r1, sr_a, sr_b);
        step(std::true_type{}, grp + 1, r1, r0, sr_b, sr_a);
      }
    } else if (wave_base < n_edges) {
      sr_a = ei[edge_of(1)];
      gather(edge_of(0), ei[edge_of(0)], r0);
      for (int grp = 0; grp < EG && wave_base + grp * 32 < n_edges; grp += 2) {
        step(std::false_type{}, grp, r0, r1, sr_a, sr_b);
        if (wave_base + (grp + 1) * 32 < n_edges) step(std::false_type{}, grp + 1, r1, r0, sr_b, sr_a);
      }
    }
  }
  if (lh == 0) red[wave * 32 + li] = colsum;
  __syncthreads();
  if (tid < 32) {
    const double s = ((red[tid] + red[32 + tid]) + red[64 + tid]) + red[96 + tid];
    a.partial[((b * 4 + q) * gridDim.x + blockIdx.x) * 32 + tid] = s;
  }
}

// The five segment maxima of a 2x2 pooling step on bf16 tensors (k_pool_max on 8 columns per
// thread): a maximum of bf16 values is one of them, so pooling stays exact.
__global__ void k_pool_max16(PoolJobs j) {
  int job = 0;
  while (job < kPoolJobs - 1 && blockIdx.x >= j.block_end[job]) ++job;
  const unsigned block = blockIdx.x - (job > 0 ? j.block_end[job - 1] : 0u);
  const int width = j.width[job], wv = width / 8;
  const int64_t n_seg = j.n_seg[job];
  const int64_t idx = static_cast<int64_t>(block) * blockDim.x + threadIdx.x;
  if (idx >= n_seg * wv) return;
  const int64_t seg = idx / wv;
  const int c = static_cast<int>(idx % wv) * 8;
  const int64_t b = blockIdx.y;
  const __bf16* s = reinterpret_cast<const __bf16*>(j.src[job]) + b * j.n_src[job] * width + c;
  const int32_t* rowptr = j.rowptr[job];
  const int32_t* perm = j.perm[job];
  const int beg = rowptr[seg], end = rowptr[seg + 1];
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = -FLT_MAX;
  for (int p = beg; p < end; p += 4) {   // a 2x2 cell has at most four members: all rows in flight at once
    u32x4v v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pi = p + i < end ? p + i : beg;   // a repeated member does not change a maximum
      v[i] = *reinterpret_cast<const u32x4v*>(s + static_cast<int64_t>(perm[pi]) * width);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[2 * k] = fmaxf(acc[2 * k], __builtin_bit_cast(float, v[i][k] << 16));
        acc[2 * k + 1] = fmaxf(acc[2 * k + 1], __builtin_bit_cast(float, v[i][k] & 0xFFFF0000u));
      }
  }
  // an empty segment's -FLT_MAX rounds to -infinity in bf16; every segment of a pooling plan has members
  *reinterpret_cast<u32x4v*>(reinterpret_cast<__bf16*>(j.out[job]) + (b * n_seg + seg) * width + c) =
      pack_bf16x8(make_float4(acc[0], acc[1], acc[2], acc[3]), make_float4(acc[4], acc[5], acc[6], acc[7]));
}

// Node MLP of the fine rows of an upsampling layer in bf16 storage: for each quadrant's edges
// (parent -> fine node) one K = 32 product message[e] . W_q (32 x 128) + bias, activation, written
// to the fine node's row (every fine node has exactly one incoming edge), and the column sums.
struct Scatter16Args {
  const __bf16* msg[4];        // [B, n_rows[q], 32]
  const int32_t* recv[4];      // receiver of edge r at recv[q][2 * r]
  int n_rows[4];
  const float* w[4];           // [32, 128] float32, row stride 128
  const float* bias;           // [B, 128]
  __bf16* out;                 // [B, out_rows, 128]
  int out_rows, row_sub;
  double* colsum;              // [B, colsum_tiles, 128]
  int colsum_tiles;
  int colsum_tile0[4];
  unsigned block_end[4];       // running total of the quadrants' workgroups
  int gpw;                     // 32-row groups per wave
  int relu;
};

__global__ __launch_bounds__(kThreads) void k_scatter_gemm16(Scatter16Args a) {
  __shared__ __attribute__((aligned(16))) float Stage[4][32 * kStageLd];
  __shared__ double red[4 * 128];
  int q = 0;
  while (q < 3 && blockIdx.x >= a.block_end[q]) ++q;
  const int blk = static_cast<int>(blockIdx.x - (q > 0 ? a.block_end[q - 1] : 0u));
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int64_t b = blockIdx.y;
  const int n_rows = a.n_rows[q];
  bf16x8 wf[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) wf[s][n][jj] = static_cast<__bf16>(a.w[q][(16 * s + 8 * lh + jj) * 128 + 32 * n + li]);
  float bias[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) bias[n] = a.bias[b * 128 + 32 * n + li];
  double cs[4] = {0.0, 0.0, 0.0, 0.0};
  const __bf16* msg = a.msg[q] + b * n_rows * static_cast<int64_t>(32);
  __bf16* out = a.out + b * a.out_rows * static_cast<int64_t>(128);
  float* st = Stage[wave];
  const int groups = (n_rows + 31) / 32;
  const int g_begin = (blk * 4 + wave) * a.gpw;
  const int g_end = min(g_begin + a.gpw, groups);
  auto row_of = [&](int grp) {
    const int r = grp * 32 + li;
    return r < n_rows ? r : n_rows - 1;
  };
  u32x4v cur[2], nxt[2];
  int orow_cur = 0, orow_nxt = 0;
  auto fetch = [&](int grp, u32x4v (&rows)[2], int& orow) {
    const int r = row_of(grp);
    const u32x4v* p = reinterpret_cast<const u32x4v*>(msg + static_cast<int64_t>(r) * 32) + lh;
    rows[0] = p[0];
    rows[1] = p[2];
    orow = a.recv[q][2 * static_cast<int64_t>(r)] - a.row_sub;
  };
  if (g_begin < g_end) fetch(g_begin, cur, orow_cur);
  for (int grp = g_begin; grp < g_end; ++grp) {
    if (grp + 1 < g_end) fetch(grp + 1, nxt, orow_nxt);
    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[s]), wf[s][n], acc[n], 0, 0, 0);
    }
    const int row0 = grp * 32;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float gs = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = activate(acc[n][r] + bias[n], a.relu);
        if (row0 + row < n_rows) gs += v;
        st[row * kStageLd + li] = v;
      }
      gs += __shfl_xor(gs, 32);
      cs[n] += static_cast<double>(gs);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 16 * i + (lane >> 2), c8 = 8 * (lane & 3);
        const float4 v0 = lds_load_f4(&st[row * kStageLd + c8]);
        const float4 v1 = lds_load_f4(&st[row * kStageLd + c8 + 4]);
        const int orow = __shfl(orow_cur, row);
        if (row0 + row < n_rows)
          *reinterpret_cast<u32x4v*>(out + static_cast<int64_t>(orow) * 128 + 32 * n + c8) = pack_bf16x8(v0, v1);
      }
    }
    cur[0] = nxt[0];
    cur[1] = nxt[1];
    orow_cur = orow_nxt;
  }
#pragma unroll
  for (int n = 0; n < 4; ++n)
    if (lh == 0) red[wave * 128 + 32 * n + li] = cs[n];
  __syncthreads();
  if (tid < 128) {
    const double s = ((red[tid] + red[128 + tid]) + red[256 + tid]) + red[384 + tid];
    a.colsum[(b * a.colsum_tiles + a.colsum_tile0[q] + blk) * 128 + tid] = s;
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
__device__ __forceinline__ void global_update_body(const LayerPtrs& p, const double* __restrict__ npart, int n_nblk,
                                                   const double* __restrict__ epart, int n_eblk, int relu, int64_t b,
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
    const double2* src = reinterpret_cast<const double2*>(npart + b * n_nblk * static_cast<int64_t>(U)) + 2 * cq;
    constexpr int QD = U / 2;   // double2 per partial row
    int i = sub;
    for (; i + 3 * S < n_nblk; i += 4 * S) {   // four partial rows in flight, added in order
      double2 v[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j][0] = src[static_cast<int64_t>(i + j * S) * QD];
        v[j][1] = src[static_cast<int64_t>(i + j * S) * QD + 1];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s0 += v[j][0].x;
        s1 += v[j][0].y;
        s2 += v[j][1].x;
        s3 += v[j][1].y;
      }
    }
    for (; i < n_nblk; i += S) {
      const double2 a0 = src[static_cast<int64_t>(i) * QD], a1 = src[static_cast<int64_t>(i) * QD + 1];
      s0 += a0.x;
      s1 += a0.y;
      s2 += a1.x;
      s3 += a1.y;
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
    const double* src = epart + (b * 4 + q) * n_eblk * static_cast<int64_t>(u) + c;
    int i = grp;
    for (; i + 7 * EG < n_eblk; i += 8 * EG) {   // eight partial rows in flight, added in order
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[static_cast<int64_t>(i + j * EG) * u];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; i < n_eblk; i += EG) s += src[static_cast<int64_t>(i) * u];
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
                                                                 const double* __restrict__ npart, int n_nblk,
                                                                 const double* __restrict__ epart, int n_eblk,
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
                                                                 const double* __restrict__ npart, int n_nblk,
                                                                 const double* __restrict__ epart, int n_eblk, int relu,
                                                                 float* __restrict__ global_out,
                                                                 const float* __restrict__ extra, int extra_width,
                                                                 float* __restrict__ cat_out, LayerPtrs pn,
                                                                 int fold_receiver,
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
  prologue_prefetch<UTN, true>(pn, relu, fold_receiver, b, pre);
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

// ---------------------------------------------------------------- layers wider than 256 units
//
// GraphUNet(units >= 128) builds 512- and 1024-wide blocks (models/graph_unet.py:39-55).  They take
// plain forms of the per-sample kernels (one thread per output column, k walked in order) and the
// generic GEMM for the edge MLP; the tuned kernels above serve the shipped 128 / 256 widths.

// Per sample: gbias = global . W_node[c_in + U .., :] + b; the chained global transforms; ebias of
// direction q = chain_q . W_edge_q[f_e + 2u .., :] + b (see k_layer_prologue).
__global__ __launch_bounds__(kSmallThreads) void k_layer_prologue_wide(LayerPtrs p, int U,
                                                                       const float* __restrict__ global_in, int act,
                                                                       float* __restrict__ ebias,
                                                                       float* __restrict__ gbias) {
  extern __shared__ __attribute__((aligned(16))) float wsh[];
  const int u = U / 4;
  float* gl = wsh;          // [g]
  float* gq = gl + p.g;     // [4][u]
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  for (int k = tid; k < p.g; k += blockDim.x) gl[k] = global_in[b * p.g + k];
  __syncthreads();
  for (int c = tid; c < U; c += blockDim.x) {
    const float* w = p.node_w + static_cast<int64_t>(p.c_in + U) * U + c;
    float s = 0.0f;
    for (int k = 0; k < p.g; ++k) s = fmaf(gl[k], w[static_cast<int64_t>(k) * U], s);
    if (p.node_b) s += p.node_b[c];
    gbias[b * U + c] = s;
  }
  for (int q = 0; q < 4; ++q) {
    const float* src = q == 0 ? gl : gq + (q - 1) * u;
    const int kin = q == 0 ? p.g : u;
    for (int j = tid; j < u; j += blockDim.x) {
      float s = 0.0f;
      for (int k = 0; k < kin; ++k) s = fmaf(src[k], p.glob_w[q][k * u + j], s);
      if (p.glob_b[q]) s += p.glob_b[q][j];
      gq[q * u + j] = activate_any(s, act);
    }
    __syncthreads();
  }
  for (int c = tid; c < U; c += blockDim.x) {
    const int q = c / u, j = c % u;
    const float* w = p.edge_w[q] + static_cast<int64_t>(p.f_e + 2 * u) * u + j;
    float s = 0.0f;
    for (int k = 0; k < u; ++k) s = fmaf(gq[q * u + k], w[k * u], s);
    if (p.edge_b[q]) s += p.edge_b[q][j];
    ebias[b * U + c] = s;
  }
}

// Per sample: node / edge column sums from the float64 partials, scales, glob_transform or identity
// (see global_update_body).
__global__ __launch_bounds__(kSmallThreads) void k_global_update_wide(LayerPtrs p, int U,
                                                                      const float* __restrict__ global_in,
                                                                      const double* __restrict__ npart, int n_nblk,
                                                                      const double* __restrict__ epart, int n_eblk,
                                                                      int act, float* __restrict__ global_out) {
  extern __shared__ __attribute__((aligned(16))) float wsh[];
  const int u = U / 4;
  float* gin = wsh;   // [g + U + u]
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  for (int k = tid; k < p.g; k += blockDim.x) gin[k] = global_in[b * p.g + k];
  for (int c = tid; c < U; c += blockDim.x) {
    const double* src = npart + b * n_nblk * static_cast<int64_t>(U) + c;
    double s = 0.0;
    for (int i = 0; i < n_nblk; ++i) s += src[static_cast<int64_t>(i) * U];
    gin[p.g + c] = (static_cast<float>(s) * 1.0f) / (p.gsn * 300.0f * 300.0f);
  }
  for (int j = tid; j < u; j += blockDim.x) {
    float total = 0.0f;
    for (int q = 0; q < 4; ++q) {
      const double* src = epart + (b * 4 + q) * n_eblk * static_cast<int64_t>(u) + j;
      double s = 0.0;
      for (int i = 0; i < n_eblk; ++i) s += src[static_cast<int64_t>(i) * u];
      total += static_cast<float>(s);
    }
    gin[p.g + U + j] = (total * 1.0f) / (p.gse * 300.0f * 300.0f * 8.0f);
  }
  __syncthreads();
  const int kin = p.g + U + u;
  if (p.globt_w != nullptr) {
    for (int c = tid; c < U; c += blockDim.x) {
      float s = 0.0f;
      for (int k = 0; k < kin; ++k) s = fmaf(gin[k], p.globt_w[static_cast<int64_t>(k) * U + c], s);
      if (p.globt_b) s += p.globt_b[c];
      global_out[b * U + c] = activate_any(s, act);
    }
  } else {
    for (int k = tid; k < kin; k += blockDim.x) global_out[b * kin + k] = gin[k];
  }
}

// out[r, 0..wp) = [in[r, 0..w) | zeros]: rows of a width the float4 loaders cannot take, padded to one they can
__global__ void k_pad_rows(const float* __restrict__ in, int w, int64_t rows, int wp, float* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= rows * wp) return;
  const int64_t r = i / wp;
  const int c = static_cast<int>(i % wp);
  out[i] = c < w ? in[r * w + c] : 0.0f;
}

// out[b, :] = [first[b, :w0] | second[b, :w1]]
__global__ void k_concat_rows(const float* __restrict__ first, int w0, const float* __restrict__ second, int w1,
                              float* __restrict__ out) {
  const int64_t b = blockIdx.x;
  for (int k = threadIdx.x; k < w0 + w1; k += blockDim.x)
    out[b * (w0 + w1) + k] = k < w0 ? first[b * w0 + k] : second[b * w1 + (k - w0)];
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Which form of the edge kernel a layer on `g` with `batch` samples takes, and how many partial
// rows per direction it therefore writes: 8 groups per wave unless that leaves more than half of
// the compute units without a workgroup (below that the shorter dependent chain of the 2-group form
// outweighs the four times as many copies of the weights into LDS; measured crossover).
struct EdgeSplit {
  int eg, n_eblk;   // eg = 0: the generic GEMM form (wide layers)
};

// Layers the tuned edge kernel does not serve: more than 64 columns per direction (its weights would
// not fit LDS), more than 64 edge features, or an activation other than linear / relu.
inline bool edge_is_generic(const gunet_gn_weights* w, int act) {
  return w->units > 256 || w->f_e > 64 || !act_tuned(act);
}

// Column-sum partial rows per sample the tiled GEMM writes for a problem of n_rows rows.
inline int tiled_colsum_rows(int n_rows, int64_t batch) {
  GemmArgs probe = {};
  probe.n_rows = n_rows;
  probe.n_seg = 1;
  probe.seg[0].k = 4;   // never a whole K tile: the weights-resident kernels do not apply
  double dummy = 0.0;
  probe.colsum = &dummy;
  return gemm_colsum_rows(probe, 1, batch, nullptr);
}

EdgeSplit edge_split(const gunet_quadgraph* g, const gunet_gn_weights* w, int64_t batch, int act) {
  int max_e = 1;
  for (int q = 0; q < 4; ++q) max_e = g->n_edges[q] > max_e ? g->n_edges[q] : max_e;
  if (edge_is_generic(w, act)) {   // one GEMM per direction; the reader adds the same number of rows of each
    int rows = 1;
    for (int q = 0; q < 4; ++q) rows = std::max(rows, g->n_edges[q] > 0 ? tiled_colsum_rows(g->n_edges[q], batch) : 0);
    return {0, rows};
  }
  const int64_t large = ceil_div(max_e, edges_per_block(kEdgeGroupsLarge));
  EdgeSplit sp;
  sp.eg = 2 * large * 4 * batch < static_cast<int64_t>(device_cu_count()) ? kEdgeGroupsSmall : kEdgeGroupsLarge;
  sp.n_eblk = static_cast<int>(ceil_div(max_e, edges_per_block(sp.eg)));
  return sp;
}

struct LayerWorkspace {
  float* xq;      // [B, N, U]            node embeddings (generic) / coarse rows only (upsample)
  float* agg;     // [B, N, U]            aggregated messages (generic) / messages [B, N_fine, u]
  float* ebias;   // [B, U]
  float* gbias;   // [B, U]
  double* npart;  // [B, n_nblk + 5, U]   column-sum partials of the node MLP (one row per workgroup or wave)
  double* epart;  // [B, 4, n_eblk_cap, u] column-sum partials of the edge MLP (one row per workgroup)
  void* wpack;    // weight image of the GEMM in flight (split-bf16 matmul modes)
  int n_nblk;
  int n_eblk_cap;   // rows per direction the buffer holds (the small-group form of the edge kernel)
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
  ws.n_eblk_cap = static_cast<int>(ceil_div(max_e > 0 ? max_e : 1, kBM)) + 1;   // the generic form: one row per 128 edges at most
  size_t off = 0;
  auto take = [&](size_t floats) {
    float* p = base ? reinterpret_cast<float*>(static_cast<char*>(base) + off) : nullptr;
    off = align_up(off + floats * sizeof(float), 256);
    return p;
  };
  // the folded biases come first, at offsets that depend on the batch only: the kernel that finishes
  // one layer's global and prepares the next layer's biases works on two layouts at once
  ws.ebias = take(bsz * 128 * kMaxUT);
  ws.gbias = take(bsz * 128 * kMaxUT);
  ws.xq = take(bsz * n * U);
  ws.agg = take(bsz * n * U);
  ws.npart = reinterpret_cast<double*>(take(2 * bsz * (ws.n_nblk + 5) * U));
  ws.epart = reinterpret_cast<double*>(take(2 * bsz * 4 * ws.n_eblk_cap * u));
  ws.wpack = take(wpack_bytes((kMaxCin + 128 * kMaxUT) / kBK + 2, 4) / sizeof(float));
  ws.bytes = off;
  return ws;
}

int check_layer_args(const char* fn, const gunet_quadgraph* graph, const gunet_gn_weights* w, int32_t batch,
                     int32_t activation, const void* workspace, size_t workspace_bytes, LayerWorkspace* ws) {
  GUNET_REQUIRE(graph && w, "%s: null graph or weights", fn);
  GUNET_REQUIRE(batch > 0, "%s: batch must be positive", fn);
  GUNET_REQUIRE(act_valid(activation), "%s: unknown activation %d", fn, activation);
  if (w->units != 128 && w->units != 256 && w->units != 512 && w->units != 1024) {
    set_error("%s: units = %d is not implemented (128, 256, 512 and 1024 are)", fn, w->units);
    return GUNET_ERR_UNSUPPORTED;
  }
  GUNET_REQUIRE(w->c_in > 0 && w->c_in % 4 == 0 && w->c_in <= kMaxCin,
                "%s: c_in = %d must be a positive multiple of 4, at most %d", fn, w->c_in, kMaxCin);
  GUNET_REQUIRE(w->g > 0 && w->g <= 4096, "%s: global width %d outside 1..4096", fn, w->g);
  if (w->f_e < 0 || w->f_e > 256 || (w->f_e > 64 && w->f_e % 4 != 0)) {
    set_error("%s: edge feature width %d is not implemented (0..64, or a multiple of 4 up to 256)", fn, w->f_e);
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

// `relu`: the layer's activation code; linear (0) and relu (1) on layers up to 256 wide take the tuned kernels
int launch_prologue(const LayerPtrs& p, int ut, const float* global_in, int relu, int fold_receiver,
                    int32_t batch, const LayerWorkspace& ws, hipStream_t stream) {
  ProfScope prof(kProfPrologue, stream);
  if (ut > 2 || !act_tuned(relu)) {
    GUNET_REQUIRE(!fold_receiver, "layer_prologue: the zero-row-free upsampling form is not implemented above 256 units");
    k_layer_prologue_wide<<<batch, kSmallThreads, sizeof(float) * (p.g + 128 * ut), stream>>>(p, 128 * ut, global_in, relu,
                                                                                            ws.ebias, ws.gbias);
    GUNET_LAUNCH_CHECK();
    return GUNET_OK;
  }
  const size_t lds = sizeof(float) * (4096 + p.g + 4 * 32 * ut);
  if (ut == 1)
    k_layer_prologue<1><<<batch, kSmallThreads, lds, stream>>>(p, global_in, relu, fold_receiver,
                                                                 ws.ebias, ws.gbias);
  else
    k_layer_prologue<2><<<batch, kSmallThreads, lds, stream>>>(p, global_in, relu, fold_receiver,
                                                                 ws.ebias, ws.gbias);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

int launch_global_update(const LayerPtrs& p, int ut, const float* global_in, int n_nblk, int n_eblk, int relu,
                         int32_t batch, const LayerWorkspace& ws, float* global_out, hipStream_t stream) {
  ProfScope prof(kProfGlobalUpdate, stream);
  if (ut > 2 || !act_tuned(relu)) {   // (`relu` is the activation code, as in launch_prologue)
    k_global_update_wide<<<batch, kSmallThreads, sizeof(float) * (p.g + 160 * ut), stream>>>(
        p, 128 * ut, global_in, ws.npart, n_nblk, ws.epart, n_eblk, relu, global_out);
    GUNET_LAUNCH_CHECK();
    return GUNET_OK;
  }
  const size_t lds = sizeof(double) * 4096 + sizeof(float) * (4096 + p.g + 160 * ut);
  if (ut == 1)
    k_global_update<1><<<batch, kSmallThreads, lds, stream>>>(p, global_in, ws.npart, n_nblk, ws.epart,
                                                                n_eblk, relu, global_out);
  else
    k_global_update<2><<<batch, kSmallThreads, lds, stream>>>(p, global_in, ws.npart, n_nblk, ws.epart,
                                                                n_eblk, relu, global_out);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

template <int UTC>
int launch_bridge_utc(const LayerPtrs& pc, const LayerPtrs& pn, int utn, size_t lds, int32_t batch, const float* global_in,
                      int n_nblk, int n_eblk, const LayerWorkspace& wc, int relu, float* global_out, const float* extra,
                      int extra_width, float* cat_out, int fold_receiver, const LayerWorkspace& wn, hipStream_t stream) {
  if (utn == 1)
    k_global_bridge<UTC, 1><<<batch, kSmallThreads, lds, stream>>>(pc, global_in, wc.npart, n_nblk, wc.epart, n_eblk,
                                                                    relu, global_out, extra, extra_width, cat_out, pn,
                                                                    fold_receiver, wn.ebias, wn.gbias);
  else
    k_global_bridge<UTC, 2><<<batch, kSmallThreads, lds, stream>>>(pc, global_in, wc.npart, n_nblk, wc.epart, n_eblk,
                                                                    relu, global_out, extra, extra_width, cat_out, pn,
                                                                    fold_receiver, wn.ebias, wn.gbias);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

template <int UT, int EG>
int launch_edge_mlp_inst(EdgeArgs& a, int chunks, bool vec, dim3 grid, hipStream_t stream) {
  if (a.sender_only) {
    k_edge_mlp<0, false, true, UT, EG><<<grid, kThreads, 0, stream>>>(a);
  } else if (chunks == 0) {
    k_edge_mlp<0, false, false, UT, EG><<<grid, kThreads, 0, stream>>>(a);
  } else if (vec && chunks == 2) {
    k_edge_mlp<2, true, false, UT, EG><<<grid, kThreads, 0, stream>>>(a);
  } else if (vec && chunks == 4) {
    k_edge_mlp<4, true, false, UT, EG><<<grid, kThreads, 0, stream>>>(a);
  } else if (vec && chunks == 8) {
    k_edge_mlp<8, true, false, UT, EG><<<grid, kThreads, 0, stream>>>(a);
  } else if (chunks == 1) {
    k_edge_mlp<1, false, false, UT, EG><<<grid, kThreads, 0, stream>>>(a);
  } else if (chunks == 2) {
    k_edge_mlp<2, false, false, UT, EG><<<grid, kThreads, 0, stream>>>(a);
  } else if (chunks <= 4) {
    k_edge_mlp<4, false, false, UT, EG><<<grid, kThreads, 0, stream>>>(a);
  } else {
    k_edge_mlp<8, false, false, UT, EG><<<grid, kThreads, 0, stream>>>(a);
  }
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

template <int UT>
int launch_edge_mlp_ut(EdgeArgs& a, int chunks, bool vec, int eg, dim3 grid, hipStream_t stream) {
  return eg == kEdgeGroupsLarge ? launch_edge_mlp_inst<UT, kEdgeGroupsLarge>(a, chunks, vec, grid, stream)
                                : launch_edge_mlp_inst<UT, kEdgeGroupsSmall>(a, chunks, vec, grid, stream);
}

// The edge MLP of a wide layer (hybrid_gnn.py:93-108) as four generic GEMMs, one per direction, over
// the operand segments [edge features | embedding of the sender | embedding of the receiver], the
// last two gathered through the edge list; bias = the folded global term; column sums into epart.
int launch_edge_generic(const EdgeArgs& a, const gunet_gn_weights* w, const gunet_quadgraph* graph,
                        const float* const edges[4], int n_eblk, int32_t batch, void* wpack, float* pad_scratch,
                        hipStream_t stream) {
  const int U = w->units, u = U / 4;
  GUNET_REQUIRE(!a.sender_only, "edge_mlp: the sender-only form is not implemented above 256 units");
  // quadrants write different numbers of partial rows; the reader adds n_eblk of each
  GUNET_HIP_TRY(hipMemsetAsync(a.partial, 0, sizeof(double) * batch * 4 * n_eblk * u, stream));
  for (int q = 0; q < 4; ++q) {
    if (graph->n_edges[q] == 0) continue;
    GemmArgs g = {};
    int row_of[3], s = 0;
    if (w->f_e > 0 && edges != nullptr && edges[q] != nullptr) {
      g.seg[s].a = edges[q];
      g.seg[s].lda = w->f_e;
      g.seg[s].shared = a.ef_shared;
      if (w->f_e % 4 != 0 || reinterpret_cast<uintptr_t>(edges[q]) % 16 != 0) {
        // a width the float4 loads cannot take (the 2 zero columns of the upsampling layer): a
        // zero-padded copy; the weight rows past f_e are zero in LDS, k stays f_e
        const int wp = (w->f_e + 3) / 4 * 4;
        const int64_t rows = static_cast<int64_t>(a.ef_shared ? 1 : batch) * graph->n_edges[q];
        GUNET_REQUIRE(rows * wp <= static_cast<int64_t>(batch) * a.n_nodes * U, "edge_mlp: no room to pad %lld edge rows",
                      static_cast<long long>(rows));
        k_pad_rows<<<static_cast<unsigned>(ceil_div(rows * wp, kThreads)), kThreads, 0, stream>>>(edges[q], w->f_e, rows, wp,
                                                                                                pad_scratch);
        GUNET_LAUNCH_CHECK();
        g.seg[s].a = pad_scratch;
        g.seg[s].lda = wp;
      }
      g.seg[s].k = w->f_e;
      row_of[s++] = 0;
    }
    for (int side = 0; side < 2; ++side) {
      g.seg[s].a = a.xq + u * q;
      g.seg[s].lda = U;
      g.seg[s].k = u;
      g.seg[s].row_idx = graph->edge_index[q] + side;
      g.seg[s].idx_stride = 2;
      g.seg[s].a_rows = a.n_nodes;
      row_of[s++] = w->f_e + side * u;
    }
    g.n_seg = s;
    for (int i = 0; i < s; ++i) g.seg[i].ldw = u;
    g.n_rows = graph->n_edges[q];
    g.relu = a.relu;
    g.act = a.act;
    g.colsum_tiles = 4 * n_eblk;
    g.colsum_tile0 = q * n_eblk;
    GUNET_REQUIRE(gemm_colsum_rows([&] { GemmArgs t = g; double d; t.colsum = &d; return t; }(), 1, batch, nullptr) <= n_eblk,
                  "edge_mlp: more column-sum rows than the layer planned");
    auto w_of = [&](int seg, int col) { return w->edge_w[q] + static_cast<int64_t>(row_of[seg]) * u + col; };
    const int nt = u >= 128 ? 4 : u / 32;
    for (int col0 = 0; col0 < u; col0 += 32 * nt)
      GUNET_TRY(launch_gemm(columns_block(g, u, col0, nt, a.ebias + q * u, U, a.out[q], a.partial, w_of), nt, batch, wpack,
                            stream));
  }
  return GUNET_OK;
}

// pad_scratch: room for batch * max_q E_q * 4 floats (the layer's aggregate buffer, idle at this point)
int launch_edge_mlp(EdgeArgs& a, int ut, const gunet_quadgraph* graph, const gunet_gn_weights* w,
                    const float* const edges[4], int32_t batch, void* wpack, float* pad_scratch, hipStream_t stream) {
  const EdgeSplit split = edge_split(graph, w, batch, a.act);
  const int n_eblk = split.n_eblk;
  if (split.eg == 0) {
    double edges_total = 0.0;
    for (int q = 0; q < 4; ++q) edges_total += graph->n_edges[q];
    ProfScope prof(kProfEdgeMlp, stream, 2.0 * batch * edges_total * (w->f_e + 2.0 * (w->units / 4)) * (w->units / 4));
    return launch_edge_generic(a, w, graph, edges, n_eblk, batch, wpack, pad_scratch, stream);
  }
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
  return ut == 1 ? launch_edge_mlp_ut<1>(a, chunks, vec_ok && all_feat, split.eg, grid, stream)
                 : launch_edge_mlp_ut<2>(a, chunks, vec_ok && all_feat, split.eg, grid, stream);
}

template <int FE, bool SO>
int launch_edge_mlp16_inst(const Edge16Args& a, int eg, dim3 grid, hipStream_t stream) {
  if (eg == kEdgeGroupsLarge)
    k_edge_mlp16<FE, SO, kEdgeGroupsLarge><<<grid, kThreads, 0, stream>>>(a);
  else
    k_edge_mlp16<FE, SO, kEdgeGroupsSmall><<<grid, kThreads, 0, stream>>>(a);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

// Edge MLP of a 128-wide layer in bf16 storage.  edges[q]: bf16 [B, E_q, 32] (f_e = 32), or float32
// [B, E_q, 16] / [E_q, 16] (f_e = 16, edges_f32), or null for the sender-only form.
int launch_edge_mlp16(Edge16Args& a, const gunet_quadgraph* graph, const gunet_gn_weights* w, const void* const edges[4],
                      bool edges_f32, bool sender_only, int32_t batch, hipStream_t stream) {
  const EdgeSplit split = edge_split(graph, w, batch, a.relu ? GUNET_ACT_RELU : GUNET_ACT_LINEAR);
  GUNET_REQUIRE(split.eg != 0 && w->units == 128, "edge_mlp16: a 128-wide layer is needed");
  double edges_total = 0.0;
  for (int q = 0; q < 4; ++q) {
    a.edge_index[q] = graph->edge_index[q];
    a.efeat[q] = edges ? edges[q] : nullptr;
    a.n_edges[q] = graph->n_edges[q];
    edges_total += graph->n_edges[q];
    GUNET_REQUIRE(sender_only || graph->n_edges[q] == 0 || a.efeat[q] != nullptr, "edge_mlp16: null edge features %d", q);
    GUNET_REQUIRE(reinterpret_cast<uintptr_t>(a.efeat[q]) % 16 == 0 && reinterpret_cast<uintptr_t>(a.out[q]) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(a.out32[q]) % 16 == 0,
                  "edge_mlp16: edge tensors must be 16-byte aligned");
  }
  a.f_e = w->f_e;
  dim3 grid(static_cast<unsigned>(split.n_eblk), 4, static_cast<unsigned>(batch));
  const int fe_used = sender_only ? 0 : w->f_e;
  ProfScope prof(kProfEdgeMlp, stream, 2.0 * batch * edges_total * (fe_used + (sender_only ? 1 : 2) * 32.0) * 32.0);
  if (sender_only) return launch_edge_mlp16_inst<0, true>(a, split.eg, grid, stream);
  if (edges_f32) {
    GUNET_REQUIRE(w->f_e == 16, "edge_mlp16: float32 edge features must be 16 wide, not %d", w->f_e);
    return launch_edge_mlp16_inst<16, false>(a, split.eg, grid, stream);
  }
  GUNET_REQUIRE(w->f_e == 32, "edge_mlp16: bf16 edge features must be 32 wide, not %d", w->f_e);
  return launch_edge_mlp16_inst<32, false>(a, split.eg, grid, stream);
}

// Workgroups per quadrant the scatter GEMM of an upsampling layer uses (= column-sum rows it writes
// per quadrant) and the groups each of their waves walks.
struct Scatter16Split {
  int gpw, blocks[4], total;
};

Scatter16Split scatter16_split(const gunet_quadgraph* g, int64_t batch) {
  Scatter16Split sp = {};
  int max_groups = 1;
  for (int q = 0; q < 4; ++q) max_groups = std::max(max_groups, (g->n_edges[q] + 31) / 32);
  const int64_t target = std::min<int64_t>(128, std::max<int64_t>(16, 2 * device_cu_count() / (4 * std::max<int64_t>(1, batch))));
  sp.gpw = static_cast<int>(std::max<int64_t>(1, ceil_div(max_groups, 4 * target)));
  for (int q = 0; q < 4; ++q) {
    sp.blocks[q] = static_cast<int>(ceil_div((g->n_edges[q] + 31) / 32, 4 * sp.gpw));
    sp.total += sp.blocks[q];
  }
  return sp;
}

}  // namespace

int pool_max_multi(int n_jobs, const float* const* src, const int64_t* n_src, const int* width,
                   const int32_t* const* rowptr, const int32_t* const* perm, const int64_t* n_seg, float* const* out,
                   int32_t batch, hipStream_t stream, bool bf16) {
  GUNET_REQUIRE(n_jobs >= 0 && n_jobs <= kPoolJobs, "pool_max_multi: %d jobs", n_jobs);
  bool vec = true;
  const int per_thread = bf16 ? 8 : 4;   // 16 bytes either way
  for (int i = 0; i < n_jobs; ++i)
    vec = vec && width[i] % per_thread == 0 && reinterpret_cast<uintptr_t>(src[i]) % 16 == 0 &&
          reinterpret_cast<uintptr_t>(out[i]) % 16 == 0;
  GUNET_REQUIRE(vec || !bf16, "pool_max_multi: bf16 tensors need widths of 8 columns and 16-byte alignment");
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
      blocks += static_cast<unsigned>(ceil_div(n_seg[i] * (width[i] / per_thread), kThreads));
      bytes += static_cast<double>(batch) * ((bf16 ? 2.0 : 4.0) * width[i] * (n_src[i] + n_seg[i]) + 4.0 * (n_src[i] + n_seg[i]));
    } else {
      j.width[i] = per_thread;
    }
    j.block_end[i] = blocks;
  }
  if (blocks == 0) return GUNET_OK;
  ProfScope prof(kProfSegment, stream, bytes);
  if (bf16)
    k_pool_max16<<<dim3(blocks, static_cast<unsigned>(batch)), kThreads, 0, stream>>>(j);
  else
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
  StreamDeviceGuard device_guard(as_stream(stream));
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
  StreamDeviceGuard device_guard(as_stream(stream));
  GUNET_REQUIRE(n_src >= 0 && width >= 0 && n_segments >= 0 && batch >= 0, "segment_max: negative size");
  GUNET_REQUIRE(n_segments == 0 || (rowptr && out), "segment_max: null pointer");
  return launch_segment_reduce<true>(src, n_src, width, rowptr, perm, n_segments, batch, 1.0f, out,
                                     as_stream(stream));
}

int gunet_segment_sum_f32(const float* src, int64_t n_src, int32_t width, const int32_t* rowptr,
                          const int32_t* perm, int64_t n_segments, int32_t batch, float scale,
                          float* out, gunet_stream_t stream) {
  StreamDeviceGuard device_guard(as_stream(stream));
  GUNET_REQUIRE(n_src >= 0 && width >= 0 && n_segments >= 0 && batch >= 0, "segment_sum: negative size");
  GUNET_REQUIRE(n_segments == 0 || (rowptr && out), "segment_sum: null pointer");
  return launch_segment_reduce<false>(src, n_src, width, rowptr, perm, n_segments, batch, scale,
                                      out, as_stream(stream));
}

size_t gunet_gn_layer_workspace_bytes(const gunet_quadgraph* graph, int32_t batch, int32_t units) {
  if (graph == nullptr || batch <= 0 || (units != 128 && units != 256 && units != 512 && units != 1024)) return 0;
  return carve_workspace(graph, batch, units, nullptr).bytes;
}

int gunet_gn_layer_forward_f32(const gunet_quadgraph* graph, const gunet_gn_weights* w, int32_t batch,
                               const float* x, const float* x_skip, int32_t c_skip,
                               const float* const edges[4], int32_t edges_shared, const float* global_in,
                               int32_t activation, float* x_out, float* const edges_out[4], float* global_out,
                               void* workspace, size_t workspace_bytes, gunet_stream_t stream) {
  StreamDeviceGuard device_guard(as_stream(stream));
  return gn_layer_run(graph, w, batch, x, x_skip, c_skip, edges, global_in, activation, x_out, edges_out, global_out,
                      workspace, workspace_bytes, edges_shared ? kGnEdgesShared : 0, nullptr, as_stream(stream));
}

int gunet_gn_upsample_forward_f32(const gunet_quadgraph* graph, int32_t n_coarse, const gunet_gn_weights* w,
                                  int32_t batch, const float* x, const float* global_in, int32_t activation,
                                  float* x_out, float* global_out, void* workspace, size_t workspace_bytes,
                                  gunet_stream_t stream) {
  StreamDeviceGuard device_guard(as_stream(stream));
  return gn_upsample_run(graph, n_coarse, w, batch, x, global_in, activation, x_out, global_out, workspace,
                         workspace_bytes, 0, nullptr, as_stream(stream));
}

}  // extern "C"

namespace gunet {

int gn_layer_run(const gunet_quadgraph* graph, const gunet_gn_weights* w, int32_t batch, const float* x,
                 const float* x_skip, int32_t c_skip, const float* const edges[4], const float* global_in,
                 int32_t activation, float* x_out, float* const edges_out[4], float* global_out, void* workspace,
                 size_t workspace_bytes, int flags, int* n_partials_out, hipStream_t stream,
                 float* const edges_out32[4]) {
  LayerWorkspace ws;
  GUNET_TRY(check_layer_args("gn_layer", graph, w, batch, activation, workspace, workspace_bytes, &ws));
  const bool s16 = (flags & kGnStore16) != 0;
  if (s16)
    GUNET_REQUIRE(w->units == 128 && graph->in_table != nullptr && (w->c_in == 128 || w->c_in == 256) &&
                      (c_skip == 0 || c_skip == 128),
                  "gn_layer: bf16 storage needs a 128-wide layer on 128 or 256 input columns and a graph with an in-edge table");
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
  // `relu` feeds the tuned kernels (linear / relu); any other activation takes the generic kernels
  // throughout: wide per-sample kernels, tiled GEMMs, per-direction edge GEMMs, a separate aggregation
  const int relu = activation == GUNET_ACT_RELU, act = activation;
  GUNET_REQUIRE(!s16 || act_tuned(act), "gn_layer: bf16 storage serves linear and relu only");
  const int n = graph->n_nodes;
  const int U = w->units, u = U / 4, ut = U / 128;
  const int c_a = w->c_in - c_skip;          // width of x; x_skip supplies the remaining columns
  const LayerPtrs p = layer_ptrs(w);

  if (!(flags & kGnSkipPrologue)) GUNET_TRY(launch_prologue(p, ut, global_in, act, 0, batch, ws, stream));

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
    g.act = act;
    g.a16 = g.o16 = s16;
    auto w_of = [&](int s, int col) {
      return w->node_emb_w[col / u] + (s == 0 ? 0 : static_cast<int64_t>(c_a) * u) + col % u;
    };
    auto b_of = [&](int col) { return w->node_emb_b[col / u] ? w->node_emb_b[col / u] + col % u : nullptr; };
    GUNET_TRY(launch_gemm_columns_tile_bias(g, U, batch, b_of, ws.xq, w_of, ws.wpack, stream));
  }

  if (s16) {  // edge MLP + message column sums on bf16 tensors
    Edge16Args a = {};
    for (int q = 0; q < 4; ++q) {
      a.edge_w[q] = w->edge_w[q];
      a.out[q] = reinterpret_cast<__bf16*>(edges_out[q]);
      a.out32[q] = edges_out32 ? edges_out32[q] : nullptr;
    }
    a.n_nodes = n;
    a.xq = reinterpret_cast<const __bf16*>(ws.xq);
    a.ebias = ws.ebias;
    a.partial = ws.epart;
    a.relu = relu;
    a.out_scale = 1.0f;
    a.ef_shared = (flags & kGnEdgesShared) ? 1 : 0;
    GUNET_TRY(launch_edge_mlp16(a, graph, w, reinterpret_cast<const void* const*>(edges), (flags & kGnEdgesIn32) != 0, false,
                                batch, stream));
  } else {  // edge MLP + message column sums (hybrid_gnn.py:93-108)
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
    a.act = act;
    a.out_scale = 1.0f;
    a.ef_shared = (flags & kGnEdgesShared) ? 1 : 0;
    GUNET_TRY(launch_edge_mlp(a, ut, graph, w, edges, batch, ws.wpack, ws.agg, stream));
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
    g.act = act;
    g.a16 = g.o16 = s16;
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

  GUNET_REQUIRE(!s16 || fused || n == 0, "gn_layer: bf16 storage needs the message-summing node MLP");
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
    switch (ut) {
      case 1: k_aggregate<1><<<grid, kThreads, 0, stream>>>(a); break;
      case 2: k_aggregate<2><<<grid, kThreads, 0, stream>>>(a); break;
      case 4: k_aggregate<4><<<grid, kThreads, 0, stream>>>(a); break;
      default: k_aggregate<8><<<grid, kThreads, 0, stream>>>(a); break;
    }
    GUNET_LAUNCH_CHECK();
  }

  auto w_of = [&](int seg, int col) { return w->node_w + row_of[seg] * U + col; };
  const int n_partials = columns_colsum_rows(g, U, batch, ws.wpack);
  g.colsum_tiles = n_partials;
  GUNET_TRY(launch_gemm_columns(g, U, batch, ws.gbias, U, x_out, ws.npart, w_of, ws.wpack, stream));

  if (n_partials_out) *n_partials_out = n_partials;
  if (flags & kGnSkipGlobalUpdate) return GUNET_OK;
  return launch_global_update(p, ut, global_in, n_partials, edge_split(graph, w, batch, act).n_eblk, act, batch, ws,
                              global_out, stream);
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
  if (w->units > 256 || !act_tuned(activation)) {
    set_error("gn_upsample: the zero-row-free form is implemented for 128 and 256 units with a linear or relu "
              "activation; run the layer on [x ; 0] (gunet_gn_layer_forward_f32 on the upsampling graph) for %d units, "
              "activation %d", w->units, activation);
    return GUNET_ERR_UNSUPPORTED;
  }
  const int relu = activation == GUNET_ACT_RELU;
  const int U = w->units, u = U / 4, ut = U / 128;
  const LayerPtrs p = layer_ptrs(w);
  const bool s16 = (flags & kGnStore16) != 0;
  GUNET_REQUIRE(!s16 || (U == 128 && w->c_in == 128), "gn_upsample: bf16 storage needs a 128-wide layer on 128 input columns");

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
    g.a16 = g.o16 = s16;
    auto w_of = [&](int, int col) { return w->node_emb_w[col / u] + col % u; };
    auto b_of = [&](int col) { return w->node_emb_b[col / u] ? w->node_emb_b[col / u] + col % u : nullptr; };
    GUNET_TRY(launch_gemm_columns_tile_bias(g, U, batch, b_of, ws.xq, w_of, ws.wpack, stream));
  }

  if (s16) {
    // the same three steps on bf16 tensors: messages (scaled by 1/4) in ws.agg viewed as bf16
    // [B, n_fine, 32]; column sums of the coarse rows; one scatter GEMM for the four quadrants
    __bf16* msg16[4];
    {
      int64_t off = 0;
      Edge16Args a = {};
      for (int q = 0; q < 4; ++q) {
        msg16[q] = reinterpret_cast<__bf16*>(ws.agg) + off;
        off += static_cast<int64_t>(batch) * graph->n_edges[q] * u;
        a.edge_w[q] = w->edge_w[q];
        a.out[q] = msg16[q];
      }
      a.n_nodes = n_coarse;
      a.xq = reinterpret_cast<const __bf16*>(ws.xq);
      a.ebias = ws.ebias;
      a.partial = ws.epart;
      a.relu = relu;
      a.out_scale = 1.0f / (3 + 1);
      GUNET_TRY(launch_edge_mlp16(a, graph, w, nullptr, false, true, batch, stream));
    }
    GemmArgs gc = {};
    gc.n_seg = 1;
    gc.seg[0].a = x;
    gc.seg[0].lda = w->c_in;
    gc.seg[0].k = w->c_in;
    gc.seg[0].ldw = U;
    gc.n_rows = n_coarse;
    gc.relu = relu;
    gc.a16 = 1;
    const int coarse_tiles = columns_colsum_rows(gc, U, batch, ws.wpack);
    const Scatter16Split sp = scatter16_split(graph, batch);
    const int tiles_total = coarse_tiles + sp.total;
    GUNET_REQUIRE(tiles_total <= ws.n_nblk + 5, "gn_upsample: %d column-sum rows for a buffer of %d", tiles_total, ws.n_nblk + 5);
    if (n_coarse > 0) {
      gc.colsum_tiles = tiles_total;
      auto w_of = [&](int, int col) { return w->node_w + col; };
      GUNET_TRY(launch_gemm_columns(gc, U, batch, ws.gbias, U, nullptr, ws.npart, w_of, ws.wpack, stream));
    }
    if (sp.total > 0) {
      Scatter16Args a = {};
      unsigned blocks = 0;
      int tile0 = coarse_tiles;
      for (int q = 0; q < 4; ++q) {
        a.msg[q] = msg16[q];
        a.recv[q] = graph->n_edges[q] > 0 ? graph->edge_index[q] + 1 : nullptr;
        a.n_rows[q] = graph->n_edges[q];
        a.w[q] = w->node_w + static_cast<int64_t>(w->c_in + u * q) * U;
        a.colsum_tile0[q] = tile0;
        tile0 += sp.blocks[q];
        blocks += static_cast<unsigned>(sp.blocks[q]);
        a.block_end[q] = blocks;
      }
      a.bias = ws.gbias;
      a.out = reinterpret_cast<__bf16*>(x_out);
      a.out_rows = n_fine;
      a.row_sub = n_coarse;
      a.colsum = ws.npart;
      a.colsum_tiles = tiles_total;
      a.gpw = sp.gpw;
      a.relu = relu;
      ProfScope prof(kProfGemmTiled, stream, 2.0 * batch * n_fine * static_cast<double>(u) * U);
      k_scatter_gemm16<<<dim3(blocks, static_cast<unsigned>(batch)), kThreads, 0, stream>>>(a);
      GUNET_LAUNCH_CHECK();
    }
    if (n_partials_out) *n_partials_out = tiles_total;
    if (flags & kGnSkipGlobalUpdate) return GUNET_OK;
    return launch_global_update(p, ut, global_in, tiles_total, edge_split(graph, w, batch, activation).n_eblk, relu, batch, ws,
                                global_out, stream);
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
    GUNET_TRY(launch_edge_mlp(a, ut, graph, w, nullptr, batch, ws.wpack, ws.agg, stream));
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
  return launch_global_update(p, ut, global_in, tiles_total, edge_split(graph, w, batch, activation).n_eblk, relu, batch, ws,
                              global_out, stream);
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
  if (utc > 2 || utn > 2 || !act_tuned(activation)) {
    // a wide layer on either side, or an activation beyond relu: the layers' own kernels, one after the other
    const int relu_ = activation;
    GUNET_TRY(launch_global_update(pc, utc, global_in, n_partials, edge_split(cur_graph, cur, batch, activation).n_eblk, relu_,
                                   batch, wc, global_out, stream));
    const float* next_in = global_out;
    if (extra != nullptr) {
      k_concat_rows<<<batch, 256, 0, stream>>>(global_out, g_new, extra, extra_width, cat_out);
      GUNET_LAUNCH_CHECK();
      next_in = cat_out;
    }
    return launch_prologue(pn, utn, next_in, relu_, next_fold_receiver, batch, wn, stream);
  }
  const size_t lds = sizeof(double) * 4096 + sizeof(float) * (4096 + pc.g + 160 * utc + pn.g + 4 * 32 * utn);
  GUNET_REQUIRE(lds <= 64 * 1024, "gn_bridge: global vectors of %d and %d floats do not fit", pc.g, pn.g);
  const int relu = activation == GUNET_ACT_RELU;
  ProfScope prof(kProfBridge, stream);
  const int n_eblk = edge_split(cur_graph, cur, batch, activation).n_eblk;
  return utc == 1 ? launch_bridge_utc<1>(pc, pn, utn, lds, batch, global_in, n_partials, n_eblk, wc, relu, global_out,
                                         extra, extra_width, cat_out, next_fold_receiver, wn, stream)
                  : launch_bridge_utc<2>(pc, pn, utn, lds, batch, global_in, n_partials, n_eblk, wc, relu, global_out,
                                         extra, extra_width, cat_out, next_fold_receiver, wn, stream);
}

}  // namespace gunet
