// Internal interface of the GEMM kernels (gemm_ops.hip) for the layer code (feature_ops.hip).
#pragma once

#include <algorithm>

#include "common.h"

namespace gunet {

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
  // Gathered segment (tiled float32 kernel only): row r of the operand is row
  // row_idx[r * idx_stride] of `a`, a [B][a_rows][lda] tensor (null = row r itself)
  const int32_t* row_idx;
  int idx_stride;
  int a_rows;
  int shared;          // tiled float32 kernel only: one [n_rows][lda] operand for all samples
};

struct GemmArgs {
  GemmSeg seg[3];
  int n_seg;
  int n_rows;                 // A rows per sample
  const float* bias;          // [.., 32*NT] or null
  int bias_stride;            // floats between samples (0 = shared)
  int paired;                 // weights-resident kernel, 64-column form: one launch does both 64-column halves
                              //   of a 128-column block (w[0..3] set); the two workgroups that share a row
                              //   range sit on the same XCD, so the second read of the operands hits its L2
  const float* bias_tile[4];  // with bias_by_tile: 32 floats of bias per column tile (null = none), shared
  int bias_by_tile;           //   by all samples -- the four directions' embedding biases where they lie
  float* out;                 // sample 0; null = compute the column sums only
  int ldo;
  int out_col0;               // first output column of this launch
  int out_rows;               // rows per sample of the output tensor
  const int32_t* row_map;     // A row r -> output row row_map[r * row_map_stride] - row_map_sub
  int row_map_stride;         //   (null = identity)
  int row_map_sub;
  double* colsum;             // partial column sums of the activated output, or null:
  int colsum_tiles;           //   [B][colsum_tiles][colsum_ld], this launch fills rows colsum_tile0 ..
  int colsum_tile0;           //   colsum_tile0 + gemm_colsum_rows() - 1 (one per workgroup or per wave)
  int colsum_ld;              //   doubles (0 = 32*NT), this launch's columns start at out_col0
                              //   Every kernel forms the float32 sum of each aligned 32-row group in one
                              //   fixed order and adds the group sums in float64, so the total does not
                              //   depend (beyond float64 rounding) on which kernel ran or how it split the rows.
  int relu;
  int act;                    // activation code when it is neither linear nor relu (GUNET_ACT_TANH ..; `relu` is 0
                              // then): only the tiled float32 kernel serves it
  // Gathered last segment (weights-resident kernel only): when gather_table is set, the last
  // segment (k = 128) is not read from seg[].a; row v's column block q is formed while loading as
  // gather_scale * (sum over j of row gather_table[(v * 4 + q) * 3 + j] (>= 0, in that order) of
  // gather_src[q], a [B][gather_rows[q]][32] tensor).
  const int32_t* gather_table;
  const float* gather_src[4];
  int gather_rows[4];
  float gather_scale;
  int gather_fold;            // set by the launcher: gather_scale is a power of two and goes into the weight image
  // bf16 storage (weights-resident kernels only): with a16 every operand segment and every gathered
  // source tensor holds bf16 (the `a` / gather_src pointers are really const __bf16*, lda counts
  // bf16 elements); with o16 `out` is written as bf16 (really a __bf16*, ldo counts bf16 elements).
  int a16;
  int o16;
};

constexpr int kBM = 128;
constexpr int kBK = 32;
constexpr int kRowGroup = 32;   // rows of one wave = rows of one column-sum partial

constexpr int kMaxCin = 2048;    // widest layer input the workspaces are sized for
constexpr int kMaxUT = 8;        // widest layer: 128 * kMaxUT columns

inline int gemm_ktiles(const GemmArgs& g) {
  int n = 0;
  for (int s = 0; s < g.n_seg; ++s) n += (g.seg[s].k + kBK - 1) / kBK;
  return n;
}

// Bytes of weight image a launch of `ktiles` K tiles and `nt` column tiles needs in the split modes.
inline size_t wpack_bytes(int ktiles, int nt) {
  return static_cast<size_t>(ktiles) * 3 * 4 * 32 * nt * sizeof(uint4);
}

bool bf16_mode();        // gunet_set_matmul_mode(GUNET_MATMUL_BF16) is in force
int device_cu_count();   // compute units of the current device

// Which kernel family a launch gets, and how many column-sum partial rows per sample it writes.
bool takes_resident(const GemmArgs& g, int nt, int64_t batch, const void* wpack);
int gemm_colsum_rows(const GemmArgs& g, int nt, int64_t batch, const void* wpack);

// One GEMM of 32 * nt columns (wpack: scratch of wpack_bytes(gemm_ktiles(g), nt) for the split
// modes, may be null in GUNET_MATMUL_F32 mode); several problems of equal nt as one launch.
int launch_gemm(GemmArgs g, int nt, int64_t batch, void* wpack, hipStream_t stream);
int launch_gemm_group(const GemmArgs* gs, int n, int nt, int64_t batch, void* wpack, hipStream_t stream);

// gunet_dense_forward_f32 with optional bf16 storage: a16 = x holds bf16 (k a multiple of 32), o16 =
// out is written as bf16 (weights-resident kernels; see GemmArgs).
int dense_run(const float* x, int64_t rows, int32_t k, const float* kernel, const float* bias, int32_t n_out,
              int32_t activation, float* out, void* workspace, size_t workspace_bytes, int a16, int o16,
              hipStream_t stream);

// GEMM with U output columns as U/128 launches of 128 columns.  `w_of(s, col)` gives the weight
// pointer of segment s for the 32-column tile starting at output column `col`.
template <typename WeightOf>
GemmArgs columns_block(const GemmArgs& g, int units, int col0, int nt, const float* bias, int bias_stride, float* out,
                       double* colsum, WeightOf w_of) {
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
  gc.paired = columns_nt(g) == 2 && units % 128 == 0;   // as launch_gemm_columns will launch it
  return gemm_colsum_rows(gc, columns_nt(g), batch, wpack);
}

template <typename WeightOf>
int launch_gemm_columns(const GemmArgs& g, int units, int64_t batch, const float* bias, int bias_stride, float* out,
                        double* colsum, WeightOf w_of, void* wpack, hipStream_t stream) {
  const int nt = columns_nt(g);
  if (nt == 2 && units % 128 == 0) {   // the 64-column gathering form: both halves of a 128-column block per launch
    for (int col0 = 0; col0 < units; col0 += 128) {
      GemmArgs gc = columns_block(g, units, col0, 4, bias, bias_stride, out, colsum, w_of);
      gc.paired = 1;
      GUNET_TRY(launch_gemm(gc, 2, batch, wpack, stream));
    }
    return GUNET_OK;
  }
  for (int col0 = 0; col0 < units; col0 += 32 * nt)
    GUNET_TRY(launch_gemm(columns_block(g, units, col0, nt, bias, bias_stride, out, colsum, w_of), nt, batch, wpack, stream));
  return GUNET_OK;
}

// The same with the bias given per 32-column tile: bias_of(col) points at the 32 bias values of the
// tile that starts at output column `col` (or is null).
template <typename WeightOf, typename BiasOf>
int launch_gemm_columns_tile_bias(const GemmArgs& g, int units, int64_t batch, BiasOf bias_of, float* out,
                                  WeightOf w_of, void* wpack, hipStream_t stream) {
  const int nt = columns_nt(g);
  for (int col0 = 0; col0 < units; col0 += 32 * nt) {
    GemmArgs gc = columns_block(g, units, col0, nt, nullptr, 0, out, nullptr, w_of);
    gc.bias_by_tile = 1;
    for (int t = 0; t < nt; ++t) gc.bias_tile[t] = bias_of(col0 + 32 * t);
    GUNET_TRY(launch_gemm(gc, nt, batch, wpack, stream));
  }
  return GUNET_OK;
}

}  // namespace gunet
