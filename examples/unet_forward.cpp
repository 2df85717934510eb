// The Graph U-Net forward from a compiled caller: nothing but <hip/hip_runtime.h> and gunet_hip.h.
//
//   unet_forward [height width batch [dump.bin]]
//
// Builds the 8-neighbour grid graph of a height x width image on the host (node and edge order of
// data/data_utils.py:96-150 image_to_graph without self-loops), splits it into the quadrant lists,
// creates the plan, fills the checkpoint-shaped weight table (models/graph_unet.py:21-59 at the
// shipped widths: 104 -> 128, depth 3, 96 outputs) with reproducible pseudo-random values, runs the
// forward twice and checks that both runs give the same bits.  With a dump path every tensor (graph,
// inputs, weights under their checkpoint names, prediction) is written as
//   [u32 name length][name][u32 ndim][i64 dims...][u32 dtype: 0 f32, 1 i32, 2 i64][data]
// so that a checker outside this program can recompute the prediction.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gunet_hip.h"

#define HIP_OK(expr)                                                                    \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      std::fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      std::exit(2);                                                                     \
    }                                                                                   \
  } while (0)

#define GUNET_CALL(expr)                                                                \
  do {                                                                                  \
    int rc_ = (expr);                                                                   \
    if (rc_ != GUNET_OK) {                                                              \
      std::fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #expr, rc_, gunet_last_error()); \
      std::exit(3);                                                                     \
    }                                                                                   \
  } while (0)

namespace {

struct Rng {   // splitmix64: reproducible everywhere
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  }
  float uniform(float lo, float hi) { return lo + (hi - lo) * static_cast<float>((next() >> 40) * (1.0 / 16777216.0)); }
};

FILE* g_dump = nullptr;

void dump(const std::string& name, const void* data, const std::vector<int64_t>& dims, uint32_t dtype) {
  if (!g_dump) return;
  const uint32_t len = static_cast<uint32_t>(name.size()), nd = static_cast<uint32_t>(dims.size());
  size_t count = 1;
  for (int64_t d : dims) count *= static_cast<size_t>(d);
  const size_t width = dtype == 2 ? 8 : 4;
  std::fwrite(&len, 4, 1, g_dump);
  std::fwrite(name.data(), 1, len, g_dump);
  std::fwrite(&nd, 4, 1, g_dump);
  std::fwrite(dims.data(), 8, nd, g_dump);
  std::fwrite(&dtype, 4, 1, g_dump);
  std::fwrite(data, width, count, g_dump);
}

std::vector<void*> g_device;   // everything uploaded, freed at exit

template <typename T>
T* upload(const std::vector<T>& host) {
  void* p = nullptr;
  HIP_OK(hipMalloc(&p, host.size() * sizeof(T) + 16));
  if (!host.empty()) HIP_OK(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  g_device.push_back(p);
  return static_cast<T*>(p);
}

// Dense kernel [rows, cols] (Glorot-uniform) or bias [cols] (U(-0.1, 0.1)) under its checkpoint name.
const float* param(Rng& rng, const std::string& name, int rows, int cols) {
  std::vector<float> host(static_cast<size_t>(rows > 0 ? rows : 1) * cols);
  const float lim = rows > 0 ? std::sqrt(6.0f / static_cast<float>(rows + cols)) : 0.1f;
  for (float& v : host) v = rng.uniform(-lim, lim);
  if (rows > 0)
    dump(name, host.data(), {rows, cols}, 0);
  else
    dump(name, host.data(), {cols}, 0);
  return upload(host);
}

// One GeoQuadrantGCN (layers/hybrid_gnn.py:41-70) with the variable names of the checkpoint.
gunet_gn_weights gn_layer(Rng& rng, const std::string& prefix, int c_in, int f_e, int g, int units, bool glob_transform) {
  gunet_gn_weights w = {};
  const int u = units / 4;
  w.c_in = c_in;
  w.f_e = f_e;
  w.g = g;
  w.units = units;
  for (int q = 0; q < 4; ++q) {
    const std::string d = "/" + std::to_string(q);
    w.node_emb_w[q] = param(rng, prefix + "/node_embeddings" + d + "/kernel", c_in, u);
    w.node_emb_b[q] = param(rng, prefix + "/node_embeddings" + d + "/bias", 0, u);
    w.edge_w[q] = param(rng, prefix + "/edge_transforms" + d + "/kernel", f_e + 3 * u, u);
    w.edge_b[q] = param(rng, prefix + "/edge_transforms" + d + "/bias", 0, u);
    w.glob_w[q] = param(rng, prefix + "/glob_transforms" + d + "/kernel", q == 0 ? g : u, u);
    w.glob_b[q] = param(rng, prefix + "/glob_transforms" + d + "/bias", 0, u);
  }
  w.node_w = param(rng, prefix + "/node_transform/kernel", c_in + units + g, units);
  w.node_b = param(rng, prefix + "/node_transform/bias", 0, units);
  if (glob_transform) {
    w.globt_w = param(rng, prefix + "/glob_transform/kernel", g + units + u, units);
    w.globt_b = param(rng, prefix + "/glob_transform/bias", 0, units);
  }
  w.global_scale_nodes = 1.0f;
  w.global_scale_edges = 1.0f;
  return w;
}

}  // namespace

int main(int argc, char** argv) {
  const int height = argc > 1 ? std::atoi(argv[1]) : 64;
  const int width = argc > 2 ? std::atoi(argv[2]) : 48;
  const int batch = argc > 3 ? std::atoi(argv[3]) : 2;
  if (argc > 4) g_dump = std::fopen(argv[4], "wb");
  if (height < 2 || width < 2 || batch < 1 || (argc > 4 && !g_dump)) {
    std::fprintf(stderr, "usage: unet_forward [height>=2 width>=2 batch>=1 [dump.bin]]\n");
    return 1;
  }
  const int depth = 3, units = 128, u = units / 4, c_in = 104, f_e0 = 16, g0 = 113, out_units = 96;

  // ---- the graph: every pixel a node (raster order), edges ordered by (source pixel, direction)
  static const int kDy[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, kDx[8] = {0, 1, 1, 1, 0, -1, -1, -1};
  const int64_t n_nodes = static_cast<int64_t>(height) * width;
  std::vector<int64_t> node_loc(2 * n_nodes), direction;
  std::vector<int32_t> edge_index;
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const int64_t p = static_cast<int64_t>(y) * width + x;
      node_loc[2 * p] = y;
      node_loc[2 * p + 1] = x;
      for (int d = 0; d < 8; ++d) {
        const int ty = y + kDy[d], tx = x + kDx[d];
        if (ty < 0 || ty >= height || tx < 0 || tx >= width) continue;
        edge_index.push_back(static_cast<int32_t>(p));
        edge_index.push_back(static_cast<int32_t>(static_cast<int64_t>(ty) * width + tx));
        direction.push_back(d);
      }
    }
  const int64_t n_edges = static_cast<int64_t>(direction.size());
  dump("node_loc", node_loc.data(), {n_nodes, 2}, 2);
  dump("edge_index", edge_index.data(), {n_edges, 2}, 1);
  dump("edge_direction", direction.data(), {n_edges}, 2);

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  const int64_t* d_loc = upload(node_loc);
  const int32_t* d_edges = upload(edge_index);
  const int64_t* d_dir = upload(direction);

  // data/data_utils.py:47-58: the four quadrant lists
  int32_t* quad[4];
  int64_t n_quad[4];
  for (int q = 0; q < 4; ++q) quad[q] = upload(std::vector<int32_t>(2 * n_edges));
  GUNET_CALL(gunet_quadrant_partition(d_edges, d_dir, n_edges, quad, n_quad, stream));

  gunet_plan* plan = nullptr;
  GUNET_CALL(gunet_plan_create(d_loc, n_nodes, quad, n_quad, depth, &plan, stream));
  for (int l = 0; l <= depth; ++l) {
    gunet_quadgraph g;
    GUNET_CALL(gunet_plan_level(plan, l, &g, nullptr));
    std::printf("level %d: %d nodes, %d + %d + %d + %d edges\n", l, g.n_nodes, g.n_edges[0], g.n_edges[1], g.n_edges[2],
                g.n_edges[3]);
  }

  // ---- weights (models/graph_unet.py:37-58)
  Rng rng{2021};
  gunet_unet_weights w = {};
  w.depth = depth;
  w.c_in = c_in;
  w.out_units = out_units;
  w.in_kernel = param(rng, "in_layer/kernel", c_in, units);
  w.in_bias = param(rng, "in_layer/bias", 0, units);
  for (int d = 0; d < depth; ++d)
    w.down[d] = gn_layer(rng, "downsampling_layers/" + std::to_string(d) + "/gcns/0", units, d == 0 ? f_e0 : u,
                         d == 0 ? g0 : units, units, true);
  w.bottleneck[0] = gn_layer(rng, "bottleneck_gcn_1", units, u, units, units, true);
  w.bottleneck[1] = gn_layer(rng, "bottleneck_gcn_2", units, u, units, units, true);
  for (int d = 0; d < depth; ++d) {
    const int lvl = depth - 1 - d;
    const std::string p = "upsampling_layers/" + std::to_string(d);
    w.up_sampling[d] = gn_layer(rng, p + "/upsampling/gcn", units, 2, units + (lvl == 0 ? g0 : units), units, true);
    w.up_gcn[d][0] = gn_layer(rng, p + "/gcns/0", 2 * units, lvl == 0 ? f_e0 : u, units, units, true);
    w.up_gcn[d][1] = gn_layer(rng, p + "/gcns/1", units, u, units, units, d != depth - 1);
  }
  w.out_kernel = param(rng, "out_model/kernel", units, out_units);

  // ---- inputs: x [B, N, 104], edge features [B, E_q, 16], global [B, 113]
  std::vector<float> x(static_cast<size_t>(batch) * n_nodes * c_in), glob(static_cast<size_t>(batch) * g0);
  for (float& v : x) v = rng.uniform(0.0f, 1.0f) < 0.4f ? rng.uniform(0.0f, 1.0f) : 0.0f;
  for (float& v : glob) v = rng.uniform(0.0f, 1.0f);
  dump("x", x.data(), {batch, n_nodes, c_in}, 0);
  dump("global", glob.data(), {batch, g0}, 0);
  const float* d_x = upload(x);
  const float* d_glob = upload(glob);
  const float* d_feat[4];
  for (int q = 0; q < 4; ++q) {
    std::vector<float> feat(static_cast<size_t>(batch) * n_quad[q] * f_e0);
    for (float& v : feat) v = rng.uniform(0.0f, 1.0f);
    dump("edges/" + std::to_string(q), feat.data(), {batch, n_quad[q], f_e0}, 0);
    d_feat[q] = upload(feat);
  }

  const size_t ws_bytes = gunet_unet_workspace_bytes(plan, &w, batch);
  if (ws_bytes == 0) {
    std::fprintf(stderr, "workspace query failed: %s\n", gunet_last_error());
    return 3;
  }
  void* workspace = nullptr;
  HIP_OK(hipMalloc(&workspace, ws_bytes));
  const size_t out_count = static_cast<size_t>(batch) * n_nodes * out_units;
  float* d_out = upload(std::vector<float>(out_count));
  std::printf("workspace %.1f MB\n", ws_bytes / 1e6);

  std::vector<float> first(out_count), second(out_count);
  hipEvent_t t0, t1;
  HIP_OK(hipEventCreate(&t0));
  HIP_OK(hipEventCreate(&t1));
  float ms = 0.0f;
  for (int run = 0; run < 2; ++run) {
    HIP_OK(hipMemsetAsync(d_out, 0xff, out_count * sizeof(float), stream));
    HIP_OK(hipEventRecord(t0, stream));
    GUNET_CALL(gunet_unet_forward_f32(plan, &w, batch, d_x, d_feat, /*edges_shared=*/0, d_glob, GUNET_ACT_RELU, GUNET_ACT_RELU, d_out, nullptr,
                                      nullptr, workspace, ws_bytes, stream));
    HIP_OK(hipEventRecord(t1, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipEventElapsedTime(&ms, t0, t1));
    HIP_OK(hipMemcpy((run == 0 ? first : second).data(), d_out, out_count * sizeof(float), hipMemcpyDeviceToHost));
  }
  dump("prediction", second.data(), {batch, n_nodes, out_units}, 0);
  double sum = 0.0;
  bool finite = true;
  for (float v : second) {
    finite = finite && std::isfinite(v) && v >= 0.0f;
    sum += v;
  }
  const bool same = std::memcmp(first.data(), second.data(), out_count * sizeof(float)) == 0;
  std::printf("forward of %d samples on %lld nodes: %.3f ms, checksum %.6f, runs identical: %s, finite and >= 0: %s\n", batch,
              static_cast<long long>(n_nodes), ms, sum, same ? "yes" : "NO", finite ? "yes" : "NO");

  GUNET_CALL(gunet_plan_destroy(plan));
  HIP_OK(hipFree(workspace));
  for (void* p : g_device) HIP_OK(hipFree(p));
  if (g_dump) std::fclose(g_dump);
  return same && finite ? 0 : 4;
}
