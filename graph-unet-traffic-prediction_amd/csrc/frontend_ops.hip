// What sits either side of the Graph U-Net forward in the reference's pipeline:
//   before: graph construction from the directional maps (data/data_utils.py:84-150), the two-layer
//           street-map CNN evaluated at the node pixels, the edge features gathered from it and
//           the per-sample model inputs (models/base_model.py:274-312);
//   after:  the prediction as a uint8 image (submission.py:69-94) and the error sums behind the
//           metrics (evaluation.py:173-244).
// Small HBM-bound kernels; graph construction, CNN and edge features depend on the city only.
#include "common.h"

namespace gunet {
namespace {

constexpr int kThreads = 256;
constexpr int kC1 = 32;   // channels of the first convolution
constexpr int kC2 = 8;    // channels of the second convolution

// street encoding of one node per half-wave; lane = channel of the first convolution.
//   h[d][c]  = relu(sum_e img[p + d + e] / 255 * w1[e][0][c])      (zero outside the image)
//   out[o]   = relu(sum_d sum_c h[d][c] * w2[d][c][o])             (h is zero outside the image)
// with d, e over the 3x3 window (cross-correlation, 'SAME' zero padding as tf.keras Conv2D).
// The sum over c is a fixed xor-shuffle tree over the 32 lanes.
__global__ __launch_bounds__(kThreads) void k_street_encoding(const uint8_t* __restrict__ img, int height,
                                                              int width, const float* __restrict__ w1,
                                                              const float* __restrict__ w2,
                                                              const int64_t* __restrict__ node_loc,
                                                              int64_t n_nodes, float* __restrict__ out) {
  const int64_t node = static_cast<int64_t>(blockIdx.x) * (kThreads / 32) + threadIdx.x / 32;
  const int c = threadIdx.x % 32;
  const bool live = node < n_nodes;   // dead half-waves still take part in the shuffles
  const int64_t nd = live ? node : 0;
  const int py = static_cast<int>(node_loc[2 * nd]), px = static_cast<int>(node_loc[2 * nd + 1]);
  float patch[5][5];   // img / 255 around the node, zero outside
#pragma unroll
  for (int y = 0; y < 5; ++y)
#pragma unroll
    for (int x = 0; x < 5; ++x) {
      const int iy = py + y - 2, ix = px + x - 2;
      const bool in = iy >= 0 && iy < height && ix >= 0 && ix < width;
      patch[y][x] = in ? static_cast<float>(img[static_cast<int64_t>(iy) * width + ix]) / 255.0f : 0.0f;
    }
  float k1[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) k1[e] = w1[e * kC1 + c];
  float acc[kC2];
#pragma unroll
  for (int o = 0; o < kC2; ++o) acc[o] = 0.0f;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int qy = py + dy - 1, qx = px + dx - 1;
      float h = 0.0f;
#pragma unroll
      for (int ey = 0; ey < 3; ++ey)
#pragma unroll
        for (int ex = 0; ex < 3; ++ex) h = fmaf(patch[dy + ey][dx + ex], k1[ey * 3 + ex], h);
      h = fmaxf(h, 0.0f);
      if (qy < 0 || qy >= height || qx < 0 || qx >= width) h = 0.0f;   // padding of the second conv
      const float* k2 = w2 + ((dy * 3 + dx) * kC1 + c) * kC2;
#pragma unroll
      for (int o = 0; o < kC2; ++o) acc[o] = fmaf(h, k2[o], acc[o]);
    }
#pragma unroll
  for (int o = 0; o < kC2; ++o) {
    float v = acc[o];
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    acc[o] = fmaxf(v, 0.0f);
  }
  if (live && c < kC2) {
    float v = acc[0];
#pragma unroll
    for (int o = 1; o < kC2; ++o) v = c == o ? acc[o] : v;
    out[node * kC2 + c] = v;
  }
}

// out[e] = [enc[sender(e)] | enc[receiver(e)]]
__global__ void k_edge_features(const float* __restrict__ enc, int width, const int32_t* __restrict__ edge_index,
                                int64_t n_edges, float* __restrict__ out) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t e = idx / (2 * width);
  if (e >= n_edges) return;
  const int j = static_cast<int>(idx % (2 * width));
  const int32_t v = edge_index[2 * e + (j >= width ? 1 : 0)];
  out[idx] = enc[static_cast<int64_t>(v) * width + (j >= width ? j - width : j)];
}

constexpr int kRowsPerBlock = 256;

// x[b, n] = [seed[b, n] | enc[n]]; partial[b, block] = column sums of this block's rows.  Thread
// (half, c) owns column c of every second row; its sum and the other half's are added in order.
__global__ __launch_bounds__(kThreads) void k_concat_inputs(const float* __restrict__ seed, int s_width,
                                                            const float* __restrict__ enc, int e_width,
                                                            int64_t n_nodes, float* __restrict__ x,
                                                            float* __restrict__ partial) {
  __shared__ float red[kThreads];
  const int64_t b = blockIdx.y;
  const int c_out = s_width + e_width;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kRowsPerBlock;
  const int64_t row1 = row0 + kRowsPerBlock < n_nodes ? row0 + kRowsPerBlock : n_nodes;
  const int half = threadIdx.x / 128, lane_c = threadIdx.x % 128;
  for (int c0 = 0; c0 < c_out; c0 += 128) {
    const int c = c0 + lane_c;
    float s = 0.0f;
    if (c < c_out) {
      for (int64_t r = row0 + half; r < row1; r += 2) {
        const float v = c < s_width ? seed[(b * n_nodes + r) * s_width + c] : enc[r * e_width + (c - s_width)];
        x[(b * n_nodes + r) * c_out + c] = v;
        s += v;
      }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (half == 0 && c < c_out) partial[(b * gridDim.x + blockIdx.x) * c_out + c] = red[lane_c] + red[128 + lane_c];
    __syncthreads();
  }
}

// global[b] = [column sums / 90000 | time encoding | one-hot weekday]
__global__ void k_finish_global(const float* __restrict__ partial, int n_blocks, int c_out,
                                const float* __restrict__ time_enc, int t_width, const int32_t* __restrict__ weekday,
                                float* __restrict__ global_out) {
  const int64_t b = blockIdx.x;
  const int g = c_out + t_width + 7;
  for (int c = threadIdx.x; c < g; c += blockDim.x) {
    float v;
    if (c < c_out) {
      double s = 0.0;
      for (int i = 0; i < n_blocks; ++i) s += static_cast<double>(partial[(b * n_blocks + i) * c_out + c]);
      v = static_cast<float>(s) / (300.0f * 300.0f);
    } else if (c < c_out + t_width) {
      v = time_enc[b * t_width + (c - c_out)];
    } else {
      v = weekday[b] == c - c_out - t_width ? 1.0f : 0.0f;
    }
    global_out[b * g + c] = v;
  }
}


// ---- graph from directional maps ------------------------------------------------------------

__constant__ int kDirDx[9] = {0, 1, 1, 1, 0, -1, -1, -1, 0};
__constant__ int kDirDy[9] = {-1, -1, 0, 1, 1, 1, 0, -1, 0};

// bit d of bits[p] = pixel p has an edge in direction d (maps that leave the image are masked as
// mask_invalid_ids does); bit 8 = self-loop of a node pixel.
__global__ void k_pixel_bits(const uint8_t* __restrict__ maps, int height, int width, int selfloops,
                             int32_t* __restrict__ node_flag, int32_t* __restrict__ edge_count,
                             uint16_t* __restrict__ bits) {
  const int64_t n = static_cast<int64_t>(height) * width;
  const int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int y = static_cast<int>(p / width), x = static_cast<int>(p % width);
  unsigned b = 0;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const int ty = y + kDirDy[d], tx = x + kDirDx[d];
    const bool inside = ty >= 0 && ty < height && tx >= 0 && tx < width;
    if (inside && maps[d * n + p] != 0) b |= 1u << d;
  }
  const int node = b != 0;
  if (node && selfloops) b |= 1u << 8;
  node_flag[p] = node;
  edge_count[p] = __popc(b);
  bits[p] = static_cast<uint16_t>(b);
}

__global__ void k_emit_graph(const uint16_t* __restrict__ bits, const int32_t* __restrict__ node_rank,
                             const int32_t* __restrict__ edge_off, int height, int width,
                             int64_t* __restrict__ node_loc, int32_t* __restrict__ edge_index,
                             int64_t* __restrict__ direction, uint8_t* __restrict__ node_mask) {
  const int64_t n = static_cast<int64_t>(height) * width;
  const int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const unsigned b = bits[p];
  if (node_mask != nullptr) node_mask[p] = b != 0;
  if (b == 0) return;
  const int32_t id = node_rank[p];
  node_loc[2 * id] = p / width;
  node_loc[2 * id + 1] = p % width;
  int64_t e = edge_off[p];
#pragma unroll
  for (int d = 0; d < 9; ++d) {
    if (!((b >> d) & 1)) continue;
    const int64_t t = p + kDirDx[d] + static_cast<int64_t>(kDirDy[d]) * width;
    edge_index[2 * e] = id;
    edge_index[2 * e + 1] = bits[t] != 0 ? node_rank[t] : 0;   // a non-node target maps to node 0
    direction[e] = d;
    ++e;
  }
}

// ---- prediction image -----------------------------------------------------------------------

// out[b, t, y, x, c] = saturate_u8(rint(pred[b, node(y, x), channels * horizon[t] + c] * scale)),
// one thread per (sample, node, horizon, channel); pixels that are no node keep the zero fill.
__global__ void k_prediction_image(const float* __restrict__ pred, int64_t n_nodes, int pred_width,
                                   const int64_t* __restrict__ node_loc, int height, int width,
                                   const int32_t* __restrict__ horizons, int n_horizons, int channels, float scale,
                                   int64_t total, uint8_t* __restrict__ out) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = static_cast<int>(idx % channels);
  int64_t r = idx / channels;
  const int t = static_cast<int>(r % n_horizons);
  r /= n_horizons;
  const int64_t node = r % n_nodes, b = r / n_nodes;
  const int64_t y = node_loc[2 * node], x = node_loc[2 * node + 1];
  if (y < 0 || y >= height || x < 0 || x >= width) return;
  const float v = rintf(pred[(b * n_nodes + node) * pred_width + channels * horizons[t] + c] * scale);
  const float clipped = fminf(fmaxf(v, 0.0f), 255.0f);
  out[(((b * n_horizons + t) * height + y) * width + x) * channels + c] = static_cast<uint8_t>(clipped);
}

// ---- error sums -----------------------------------------------------------------------------

// partial[b, block, 0, c] = sum over this block's rows of (t - p)^2, [.., 1, c] of |t - p|
__global__ __launch_bounds__(kThreads) void k_error_partials(const float* __restrict__ y_true,
                                                             const float* __restrict__ y_pred, int64_t n_rows,
                                                             int width, float* __restrict__ partial) {
  __shared__ float red[2][kThreads];
  const int64_t b = blockIdx.y;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kRowsPerBlock;
  const int64_t row1 = row0 + kRowsPerBlock < n_rows ? row0 + kRowsPerBlock : n_rows;
  const int half = threadIdx.x / 128, lane_c = threadIdx.x % 128;
  for (int c0 = 0; c0 < width; c0 += 128) {
    const int c = c0 + lane_c;
    float sq = 0.0f, ab = 0.0f;
    if (c < width) {
      for (int64_t r = row0 + half; r < row1; r += 2) {
        const float d = y_true[(b * n_rows + r) * width + c] - y_pred[(b * n_rows + r) * width + c];
        sq = fmaf(d, d, sq);
        ab += fabsf(d);
      }
    }
    red[0][threadIdx.x] = sq;
    red[1][threadIdx.x] = ab;
    __syncthreads();
    if (half == 0 && c < width) {
      float* dst = partial + ((b * gridDim.x + blockIdx.x) * 2) * static_cast<int64_t>(width) + c;
      dst[0] = red[0][lane_c] + red[0][128 + lane_c];
      dst[width] = red[1][lane_c] + red[1][128 + lane_c];
    }
    __syncthreads();
  }
}

__global__ void k_error_finish(const float* __restrict__ partial, int n_blocks, int width, double* __restrict__ out) {
  const int64_t b = blockIdx.x;
  for (int i = threadIdx.x; i < 2 * width; i += blockDim.x) {
    double s = 0.0;
    for (int k = 0; k < n_blocks; ++k) s += static_cast<double>(partial[(b * n_blocks + k) * 2 * static_cast<int64_t>(width) + i]);
    out[b * 2 * width + i] = s;
  }
}

}  // namespace
}  // namespace gunet

using namespace gunet;

extern "C" {

int gunet_street_encoding_f32(const uint8_t* street_map, int32_t height, int32_t width, const float* cnn1_kernel,
                              const float* cnn2_kernel, const int64_t* node_loc, int64_t n_nodes, float* out,
                              gunet_stream_t stream) {
  StreamDeviceGuard device_guard(as_stream(stream));
  GUNET_REQUIRE(height > 0 && width > 0 && n_nodes >= 0, "street_encoding: bad size");
  if (n_nodes == 0) return GUNET_OK;
  GUNET_REQUIRE(street_map && cnn1_kernel && cnn2_kernel && node_loc && out, "street_encoding: null pointer");
  const unsigned grid = static_cast<unsigned>(ceil_div(n_nodes, kThreads / 32));
  k_street_encoding<<<grid, kThreads, 0, as_stream(stream)>>>(street_map, height, width, cnn1_kernel, cnn2_kernel,
                                                              node_loc, n_nodes, out);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

int gunet_edge_features_f32(const float* encoding, int64_t n_nodes, int32_t width, const int32_t* edge_index,
                            int64_t n_edges, float* out, gunet_stream_t stream) {
  StreamDeviceGuard device_guard(as_stream(stream));
  GUNET_REQUIRE(n_nodes >= 0 && n_edges >= 0 && width > 0, "edge_features: bad size");
  if (n_edges == 0) return GUNET_OK;
  GUNET_REQUIRE(encoding && edge_index && out, "edge_features: null pointer");
  const unsigned grid = static_cast<unsigned>(ceil_div(n_edges * 2 * width, kThreads));
  k_edge_features<<<grid, kThreads, 0, as_stream(stream)>>>(encoding, width, edge_index, n_edges, out);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

size_t gunet_model_inputs_workspace_bytes(int64_t n_nodes, int32_t batch, int32_t seed_width, int32_t enc_width) {
  if (n_nodes < 0 || batch <= 0 || seed_width < 0 || enc_width < 0) return 0;
  return sizeof(float) * static_cast<size_t>(batch) * static_cast<size_t>(ceil_div(n_nodes > 0 ? n_nodes : 1, kRowsPerBlock)) *
         static_cast<size_t>(seed_width + enc_width);
}

int gunet_model_inputs_f32(const float* seed_nodes, int32_t seed_width, const float* encoding, int32_t enc_width,
                           const float* time_enc, int32_t time_width, const int32_t* weekday, int64_t n_nodes,
                           int32_t batch, float* x, float* global_out, void* workspace, size_t workspace_bytes,
                           gunet_stream_t stream) {
  StreamDeviceGuard device_guard(as_stream(stream));
  GUNET_REQUIRE(n_nodes > 0 && batch > 0 && seed_width > 0 && enc_width >= 0 && time_width >= 0,
                "model_inputs: bad size");
  GUNET_REQUIRE(seed_nodes && (encoding || enc_width == 0) && (time_enc || time_width == 0) && weekday && x &&
                    global_out,
                "model_inputs: null pointer");
  const size_t need = gunet_model_inputs_workspace_bytes(n_nodes, batch, seed_width, enc_width);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("model_inputs: workspace of %zu bytes is smaller than the %zu required", workspace_bytes, need);
    return GUNET_ERR_WORKSPACE;
  }
  const int n_blocks = static_cast<int>(ceil_div(n_nodes, kRowsPerBlock));
  float* partial = static_cast<float*>(workspace);
  dim3 grid(static_cast<unsigned>(n_blocks), static_cast<unsigned>(batch));
  k_concat_inputs<<<grid, kThreads, 0, as_stream(stream)>>>(seed_nodes, seed_width, encoding, enc_width, n_nodes, x,
                                                            partial);
  GUNET_LAUNCH_CHECK();
  k_finish_global<<<static_cast<unsigned>(batch), kThreads, 0, as_stream(stream)>>>(
      partial, n_blocks, seed_width + enc_width, time_enc, time_width, weekday, global_out);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

int gunet_image_to_graph(const uint8_t* maps, int32_t height, int32_t width, int32_t add_selfloops,
                         int64_t* node_loc, int32_t* edge_index, int64_t* direction, uint8_t* node_mask,
                         int64_t counts_host[2], gunet_stream_t stream_) {
  StreamDeviceGuard device_guard(as_stream(stream_));
  hipStream_t stream = as_stream(stream_);
  GUNET_REQUIRE(counts_host != nullptr, "image_to_graph: null counts_host");
  counts_host[0] = counts_host[1] = 0;
  GUNET_REQUIRE(height > 0 && width > 0 && static_cast<int64_t>(height) * width < (int64_t{1} << 27),
                "image_to_graph: bad image size %d x %d", height, width);
  GUNET_REQUIRE(maps && node_loc && edge_index && direction, "image_to_graph: null pointer");
  const int64_t n = static_cast<int64_t>(height) * width;
  DevBuf flags, ranks, bits;
  GUNET_HIP_TRY(flags.alloc(sizeof(int32_t) * 2 * n));
  GUNET_HIP_TRY(ranks.alloc(sizeof(int32_t) * 2 * (n + 1)));
  GUNET_HIP_TRY(bits.alloc(sizeof(uint16_t) * n));
  int32_t* node_flag = flags.as<int32_t>();
  int32_t* edge_count = node_flag + n;
  int32_t* node_rank = ranks.as<int32_t>();
  int32_t* edge_off = node_rank + (n + 1);
  const unsigned grid = static_cast<unsigned>(ceil_div(n, kThreads));
  k_pixel_bits<<<grid, kThreads, 0, stream>>>(maps, height, width, add_selfloops != 0, node_flag, edge_count,
                                               bits.as<uint16_t>());
  GUNET_LAUNCH_CHECK();
  GUNET_TRY(scan_exclusive_i32(node_flag, node_rank, n, stream));
  GUNET_TRY(scan_exclusive_i32(edge_count, edge_off, n, stream));
  k_emit_graph<<<grid, kThreads, 0, stream>>>(bits.as<uint16_t>(), node_rank, edge_off, height, width, node_loc,
                                               edge_index, direction, node_mask);
  GUNET_LAUNCH_CHECK();
  int32_t totals[2];
  GUNET_HIP_TRY(hipMemcpyAsync(&totals[0], node_rank + n, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  GUNET_HIP_TRY(hipMemcpyAsync(&totals[1], edge_off + n, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  GUNET_HIP_TRY(hipStreamSynchronize(stream));
  counts_host[0] = totals[0];
  counts_host[1] = totals[1];
  return GUNET_OK;
}

int gunet_prediction_image_u8(const float* pred, int64_t n_nodes, int32_t pred_width, int32_t batch,
                              const int64_t* node_loc, int32_t height, int32_t width, const int32_t* horizons,
                              int32_t n_horizons, int32_t channels, float scale, uint8_t* out,
                              gunet_stream_t stream_) {
  StreamDeviceGuard device_guard(as_stream(stream_));
  hipStream_t stream = as_stream(stream_);
  GUNET_REQUIRE(n_nodes >= 0 && batch > 0 && height > 0 && width > 0 && n_horizons > 0 && channels > 0 &&
                    pred_width >= channels,
                "prediction_image: bad size");
  GUNET_REQUIRE(out && horizons && (n_nodes == 0 || (pred && node_loc)), "prediction_image: null pointer");
  const size_t bytes = static_cast<size_t>(batch) * n_horizons * height * width * channels;
  GUNET_HIP_TRY(hipMemsetAsync(out, 0, bytes, stream));
  const int64_t total = static_cast<int64_t>(batch) * n_nodes * n_horizons * channels;
  if (total == 0) return GUNET_OK;
  k_prediction_image<<<static_cast<unsigned>(ceil_div(total, kThreads)), kThreads, 0, stream>>>(
      pred, n_nodes, pred_width, node_loc, height, width, horizons, n_horizons, channels, scale, total, out);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

size_t gunet_error_sums_workspace_bytes(int64_t n_rows, int32_t batch, int32_t width) {
  if (n_rows < 0 || batch <= 0 || width <= 0) return 0;
  return sizeof(float) * 2 * static_cast<size_t>(batch) * static_cast<size_t>(ceil_div(n_rows > 0 ? n_rows : 1, kRowsPerBlock)) *
         static_cast<size_t>(width);
}

int gunet_error_sums_f32(const float* y_true, const float* y_pred, int64_t n_rows, int32_t width, int32_t batch,
                         double* sums, void* workspace, size_t workspace_bytes, gunet_stream_t stream_) {
  StreamDeviceGuard device_guard(as_stream(stream_));
  hipStream_t stream = as_stream(stream_);
  GUNET_REQUIRE(n_rows > 0 && width > 0 && batch > 0, "error_sums: bad size");
  GUNET_REQUIRE(y_true && y_pred && sums, "error_sums: null pointer");
  const size_t need = gunet_error_sums_workspace_bytes(n_rows, batch, width);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("error_sums: workspace of %zu bytes is smaller than the %zu required", workspace_bytes, need);
    return GUNET_ERR_WORKSPACE;
  }
  const int n_blocks = static_cast<int>(ceil_div(n_rows, kRowsPerBlock));
  dim3 grid(static_cast<unsigned>(n_blocks), static_cast<unsigned>(batch));
  k_error_partials<<<grid, kThreads, 0, stream>>>(y_true, y_pred, n_rows, width, static_cast<float*>(workspace));
  GUNET_LAUNCH_CHECK();
  k_error_finish<<<static_cast<unsigned>(batch), kThreads, 0, stream>>>(static_cast<float*>(workspace), n_blocks, width,
                                                                        sums);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

}  // extern "C"
