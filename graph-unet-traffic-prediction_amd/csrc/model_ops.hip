// Whole-model entry points.
//   gunet_plan_*            every index structure the U-Net needs for one city graph -- per level the
//                           quadrant graph with its CSRs and in-edge table, the 2x2 pooling maps
//                           (layers/hybrid_gnn.py:410-447) and the upsampling graphs (:267-280, :336-357)
//                           -- built once on the device and owned by an opaque handle;
//   gunet_unet_forward_f32  models/graph_unet.py:61-84 GraphUNet.call as one asynchronous,
//                           allocation-free call over caller workspace: the same kernels in the same
//                           order as the layer-by-layer entry points, so results are bit-identical.
#include <mutex>
#include <new>
#include <vector>

#include "gemm.h"

struct gunet_plan {
  struct Level {   // one quadrant graph (a U-Net level, or an upsampling graph)
    int64_t n_nodes = 0;
    int64_t* node_loc = nullptr;   // [n_nodes, 2]; null for upsampling graphs
    int32_t* edge_index[4] = {};
    int64_t n_edges[4] = {};
    int32_t* rowptr[4] = {};
    int32_t* perm[4] = {};
    int32_t* in_table = nullptr;
    int32_t max_in_degree = 0;
    gunet_quadgraph qg = {};
  };
  struct Pool {   // level l -> l + 1
    int32_t* cluster = nullptr;
    int32_t* cluster_rowptr = nullptr;
    int32_t* cluster_perm = nullptr;
    int32_t* dup_map[4] = {};
    int32_t* dup_rowptr[4] = {};
    int32_t* dup_perm[4] = {};
  };
  int depth = 0;
  int device = 0;
  std::vector<Level> levels;   // depth + 1
  std::vector<Pool> pools;     // depth
  std::vector<Level> ups;      // depth; ups[l]: level l + 1 -> level l, over n_{l+1} + n_l nodes
  std::vector<int32_t*> parents;
  std::vector<char> one_parent_edge;   // ups[l] has exactly one edge per fine node (see build_plan)
  std::vector<void*> owned;            // the chunks every array above is carved from
  char* chunk = nullptr;               // free part of the newest chunk
  size_t chunk_left = 0;
  size_t chunk_bytes = 0;              // size of a fresh chunk (set from the level-0 graph)

  ~gunet_plan() {
    for (void* p : owned) (void)hipFree(p);
  }
};

namespace gunet {
namespace {

// The plan's arrays (some 150 of them for a depth-3 plan) are carved out of a few large chunks:
// one hipMalloc per chunk instead of one per array.
template <typename T>
int plan_alloc(gunet_plan& P, T** out, int64_t count) {
  const size_t bytes = (sizeof(T) * static_cast<size_t>(count > 0 ? count : 1) + 255) / 256 * 256;
  if (bytes > P.chunk_left) {
    const size_t fresh = bytes > P.chunk_bytes ? bytes : P.chunk_bytes;
    void* p = nullptr;
    GUNET_HIP_TRY(hipMalloc(&p, fresh));
    P.owned.push_back(p);
    P.chunk = static_cast<char*>(p);
    P.chunk_left = fresh;
  }
  *out = reinterpret_cast<T*>(P.chunk);
  P.chunk += bytes;
  P.chunk_left -= bytes;
  return GUNET_OK;
}

// CSRs and in-edge table of a level whose edge lists are in place.
int finish_quadgraph(gunet_plan& P, gunet_plan::Level& L, hipStream_t stream) {
  GUNET_REQUIRE(L.n_nodes < INT32_MAX, "plan: too many nodes");
  const int32_t n = static_cast<int32_t>(L.n_nodes);
  L.qg = {};
  L.qg.n_nodes = n;
  for (int q = 0; q < 4; ++q) {
    GUNET_TRY(plan_alloc(P, &L.rowptr[q], static_cast<int64_t>(n) + 1));
    GUNET_TRY(plan_alloc(P, &L.perm[q], L.n_edges[q]));
    GUNET_TRY(gunet_csr_build(L.edge_index[q], L.n_edges[q], n, L.rowptr[q], L.perm[q], stream));
    L.qg.n_edges[q] = static_cast<int32_t>(L.n_edges[q]);
    L.qg.edge_index[q] = L.edge_index[q];
    L.qg.rowptr[q] = L.rowptr[q];
    L.qg.perm[q] = L.perm[q];
  }
  if (n > 0) {
    int32_t* table = nullptr;
    GUNET_TRY(plan_alloc(P, &table, static_cast<int64_t>(n) * 12));
    GUNET_TRY(gunet_in_edge_table(&L.qg, table, &L.max_in_degree, stream));
    if (L.max_in_degree <= 3) {
      L.in_table = table;
      L.qg.in_table = table;
    }
  }
  return GUNET_OK;
}

int build_plan(gunet_plan& P, const int64_t* node_loc, int64_t n_nodes, const int32_t* const edge_index[4],
               const int64_t n_edges[4], hipStream_t stream) {
  const int D = P.depth;
  {  // about what level 0 with its pooling and upsampling structures takes; coarser levels fit a second chunk
    int64_t e_all = 0;
    for (int q = 0; q < 4; ++q) e_all += n_edges[q];
    const int64_t guess = 240 * n_nodes + 30 * e_all;
    P.chunk_bytes = static_cast<size_t>(guess < (int64_t{256} << 10) ? (int64_t{256} << 10)
                                                                     : (guess > (int64_t{64} << 20) ? (int64_t{64} << 20) : guess));
  }
  P.levels.resize(D + 1);
  P.pools.resize(D);
  P.ups.resize(D);
  P.parents.assign(D, nullptr);
  P.one_parent_edge.assign(D, 1);
  {  // level 0: the plan keeps its own copy of the caller's index tensors
    gunet_plan::Level& L = P.levels[0];
    L.n_nodes = n_nodes;
    GUNET_TRY(plan_alloc(P, &L.node_loc, 2 * n_nodes));
    if (n_nodes > 0)
      GUNET_HIP_TRY(hipMemcpyAsync(L.node_loc, node_loc, sizeof(int64_t) * 2 * n_nodes, hipMemcpyDeviceToDevice, stream));
    for (int q = 0; q < 4; ++q) {
      L.n_edges[q] = n_edges[q];
      GUNET_TRY(plan_alloc(P, &L.edge_index[q], 2 * n_edges[q]));
      if (n_edges[q] > 0)
        GUNET_HIP_TRY(hipMemcpyAsync(L.edge_index[q], edge_index[q], sizeof(int32_t) * 2 * n_edges[q],
                                     hipMemcpyDeviceToDevice, stream));
    }
    GUNET_TRY(finish_quadgraph(P, L, stream));
  }
  for (int l = 0; l < D; ++l) {   // graph_pooling_2x2, index part (hybrid_gnn.py:415-434, 391-408)
    const gunet_plan::Level& F = P.levels[l];
    gunet_plan::Level& C = P.levels[l + 1];
    gunet_plan::Pool& pool = P.pools[l];
    const int64_t n = F.n_nodes;
    GUNET_TRY(plan_alloc(P, &pool.cluster, n));
    GUNET_TRY(plan_alloc(P, &C.node_loc, 2 * n));
    int64_t n_pooled = 0;
    GUNET_TRY(gunet_cluster_ids(F.node_loc, n, pool.cluster, C.node_loc, &n_pooled, stream));
    C.n_nodes = n_pooled;
    GUNET_TRY(plan_alloc(P, &pool.cluster_rowptr, n_pooled + 1));
    GUNET_TRY(plan_alloc(P, &pool.cluster_perm, n));
    GUNET_TRY(gunet_csr_from_keys(pool.cluster, n, static_cast<int32_t>(n_pooled), pool.cluster_rowptr,
                                  pool.cluster_perm, stream));
    for (int q = 0; q < 4; ++q) {
      const int64_t ne = F.n_edges[q];
      GUNET_TRY(plan_alloc(P, &C.edge_index[q], 2 * ne));
      GUNET_TRY(plan_alloc(P, &pool.dup_map[q], ne));
      int64_t n_new = 0;
      if (ne > 0)
        GUNET_TRY(gunet_pool_edges(F.edge_index[q], ne, pool.cluster, n > 0 ? n : 1, C.edge_index[q], pool.dup_map[q],
                                   &n_new, stream));
      C.n_edges[q] = n_new;
      GUNET_TRY(plan_alloc(P, &pool.dup_rowptr[q], n_new + 1));
      GUNET_TRY(plan_alloc(P, &pool.dup_perm[q], ne));
      GUNET_TRY(gunet_csr_from_keys(pool.dup_map[q], ne, static_cast<int32_t>(n_new), pool.dup_rowptr[q],
                                    pool.dup_perm[q], stream));
    }
    GUNET_TRY(finish_quadgraph(P, C, stream));
  }
  for (int l = 0; l < D; ++l) {   // GraphUpsampling2x2 'graph', index part (hybrid_gnn.py:270-280, 336-357)
    const gunet_plan::Level& F = P.levels[l];
    const gunet_plan::Level& C = P.levels[l + 1];
    gunet_plan::Level& U = P.ups[l];
    U.n_nodes = C.n_nodes + F.n_nodes;
    GUNET_TRY(plan_alloc(P, &P.parents[l], F.n_nodes));
    int32_t* lists[4];
    for (int q = 0; q < 4; ++q) {
      GUNET_TRY(plan_alloc(P, &U.edge_index[q], 2 * F.n_nodes));
      lists[q] = U.edge_index[q];
    }
    int64_t counts[4] = {};
    GUNET_TRY(gunet_upsample_edges(C.node_loc, C.n_nodes, F.node_loc, F.n_nodes, P.parents[l], lists, counts, stream));
    int64_t total = 0;
    for (int q = 0; q < 4; ++q) {
      U.n_edges[q] = counts[q];
      total += counts[q];
    }
    // The reference's cell id (row // 2) * (max_row + 1) + col // 2 (hybrid_gnn.py:417-423) is not
    // unique when the graph is more than twice as wide as it is high; merged cells then leave fine
    // nodes that sit in none of the four places around their parent's position and get no edge
    // (:353-357).  Such a level takes the reference's literal route in the forward: the GN layer on
    // [x ; zeros].
    P.one_parent_edge[l] = total == F.n_nodes;
    GUNET_TRY(finish_quadgraph(P, U, stream));
  }
  GUNET_HIP_TRY(hipStreamSynchronize(stream));
  return GUNET_OK;
}

// dst[b, 0..row_floats) = src[b, 0..row_floats) for rows that are src_stride / dst_stride floats apart
__global__ void k_copy_rows(const float* __restrict__ src, int64_t src_stride, float* __restrict__ dst,
                            int64_t dst_stride, int64_t row_floats) {
  const int64_t b = blockIdx.y;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < row_floats;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    dst[b * dst_stride + i] = src[b * src_stride + i];
}

int copy_rows(const float* src, int64_t src_stride, float* dst, int64_t dst_stride, int64_t row_floats, int64_t batch,
              hipStream_t stream) {
  if (row_floats == 0 || batch == 0) return GUNET_OK;
  const int64_t blocks = ceil_div(row_floats, 256);
  dim3 grid(static_cast<unsigned>(blocks < 4096 ? blocks : 4096), static_cast<unsigned>(batch));
  k_copy_rows<<<grid, 256, 0, stream>>>(src, src_stride, dst, dst_stride, row_floats);
  GUNET_LAUNCH_CHECK();
  return GUNET_OK;
}

inline int global_out_width(const gunet_gn_weights& w) {
  return w.globt_w != nullptr ? w.units : w.g + w.units + w.units / 4;
}

// Bump allocation over the caller's workspace (base == nullptr: size query).
struct Carver {
  char* base;
  size_t off = 0;
  float* take(int64_t floats) {
    off = (off + 255) & ~static_cast<size_t>(255);
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += sizeof(float) * static_cast<size_t>(floats > 0 ? floats : 0);
    return p;
  }
};

struct EdgeSet {
  float* q[4];
};

struct ModelBuffers {
  float* skip_x[GUNET_MAX_DEPTH + 1];
  EdgeSet skip_e[GUNET_MAX_DEPTH + 1];   // [0] unused: the caller's tensors
  float* skip_g[GUNET_MAX_DEPTH + 1];    // [0] unused
  float* tmp_x[2];
  EdgeSet tmp_e[2];
  float* up_in;    // [x ; zeros] and the full output of an upsampling layer on the literal route
  float* up_out;
  float* globals;        // one [B, max_g] slot per layer
  int64_t global_slot;   // floats per slot
  void* layer_ws;
  size_t layer_ws_bytes;
  size_t bytes;
};

// Checks that the weights chain (every layer's c_in, f_e, g follows from the one before, as
// models/graph_unet.py:61-84 wires them) and lays the intermediates out.
int carve_model(const char* who, const gunet_plan* P, const gunet_unet_weights* w, int64_t batch, void* workspace,
                ModelBuffers* mb, int* final_width) {
  const int D = P->depth;
  GUNET_REQUIRE(w->depth == D, "%s: weights of depth %d for a plan of depth %d", who, w->depth, D);
  GUNET_REQUIRE(D >= 1, "%s: depth %d (at least one down/up block is needed)", who, D);
  GUNET_REQUIRE(w->c_in > 0 && w->out_units > 0 && w->in_kernel && w->out_kernel, "%s: null or empty dense layer", who);
  int wx[GUNET_MAX_DEPTH + 1], we[GUNET_MAX_DEPTH + 1], wg[GUNET_MAX_DEPTH + 1];
  int max_units = 0, max_g = 0;
  size_t layer_ws = 0;
  auto check = [&](const char* name, int idx, const gunet_gn_weights& g, int c_in, int f_e, int gw,
                   const gunet_quadgraph& graph) -> int {
    GUNET_REQUIRE(g.units == 128 || g.units == 256 || g.units == 512 || g.units == 1024,
                  "%s: %s[%d] is %d wide (128, 256, 512 and 1024 are implemented)", who, name, idx, g.units);
    GUNET_REQUIRE(g.c_in == c_in && g.f_e == f_e && g.g == gw,
                  "%s: %s[%d] has c_in, f_e, g = %d, %d, %d; the layers before it give %d, %d, %d", who, name, idx,
                  g.c_in, g.f_e, g.g, c_in, f_e, gw);
    max_units = g.units > max_units ? g.units : max_units;
    const int go = global_out_width(g);
    max_g = go > max_g ? go : max_g;
    max_g = gw > max_g ? gw : max_g;
    const size_t b = gunet_gn_layer_workspace_bytes(&graph, static_cast<int32_t>(batch), g.units);
    layer_ws = b > layer_ws ? b : layer_ws;
    return GUNET_OK;
  };
  wx[0] = w->down[0].c_in;
  we[0] = w->down[0].f_e;
  wg[0] = w->down[0].g;
  GUNET_REQUIRE(wx[0] > 0 && wx[0] % 32 == 0, "%s: in_layer width %d", who, wx[0]);
  for (int l = 0; l < D; ++l) {
    GUNET_TRY(check("down", l, w->down[l], wx[l], we[l], wg[l], P->levels[l].qg));
    wx[l + 1] = w->down[l].units;
    we[l + 1] = w->down[l].units / 4;
    wg[l + 1] = global_out_width(w->down[l]);
  }
  int c = wx[D], fe = we[D], g = wg[D];
  for (int i = 0; i < 2; ++i) {
    GUNET_TRY(check("bottleneck", i, w->bottleneck[i], c, fe, g, P->levels[D].qg));
    c = w->bottleneck[i].units;
    fe = c / 4;
    g = global_out_width(w->bottleneck[i]);
  }
  for (int i = 0; i < D; ++i) {
    const int l = D - 1 - i;
    GUNET_TRY(check("up_sampling", i, w->up_sampling[i], c, 2, g + wg[l], P->ups[l].qg));
    c = w->up_sampling[i].units;
    g = global_out_width(w->up_sampling[i]);
    GUNET_TRY(check("up_gcn0", i, w->up_gcn[i][0], c + wx[l], we[l], g, P->levels[l].qg));
    c = w->up_gcn[i][0].units;
    fe = c / 4;
    g = global_out_width(w->up_gcn[i][0]);
    GUNET_TRY(check("up_gcn1", i, w->up_gcn[i][1], c, fe, g, P->levels[l].qg));
    c = w->up_gcn[i][1].units;
    fe = c / 4;
    g = global_out_width(w->up_gcn[i][1]);
  }
  *final_width = c;

  Carver cv{static_cast<char*>(workspace)};
  int64_t max_rows = 0, max_edges[4] = {};
  for (int l = 0; l <= D; ++l) {
    const gunet_plan::Level& L = P->levels[l];
    max_rows = L.n_nodes > max_rows ? L.n_nodes : max_rows;
    mb->skip_x[l] = cv.take(batch * L.n_nodes * wx[l]);
    mb->skip_g[l] = l > 0 ? cv.take(batch * wg[l]) : nullptr;
    for (int q = 0; q < 4; ++q) {
      max_edges[q] = L.n_edges[q] > max_edges[q] ? L.n_edges[q] : max_edges[q];
      mb->skip_e[l].q[q] = l > 0 ? cv.take(batch * L.n_edges[q] * we[l]) : nullptr;
    }
  }
  int64_t literal_rows = 0;   // upsampling levels that run the layer on [x ; 0] as the reference writes it
  for (int l = 0; l < D; ++l) {
    if (P->one_parent_edge[l] && w->up_sampling[D - 1 - l].units <= 256) continue;
    const gunet_plan::Level& U = P->ups[l];
    literal_rows = U.n_nodes > literal_rows ? U.n_nodes : literal_rows;
    for (int q = 0; q < 4; ++q) max_edges[q] = U.n_edges[q] > max_edges[q] ? U.n_edges[q] : max_edges[q];
  }
  mb->up_in = cv.take(batch * literal_rows * max_units);
  mb->up_out = cv.take(batch * literal_rows * max_units);
  for (int i = 0; i < 2; ++i) {
    mb->tmp_x[i] = cv.take(batch * max_rows * max_units);
    for (int q = 0; q < 4; ++q) mb->tmp_e[i].q[q] = cv.take(batch * max_edges[q] * (max_units / 4));
  }
  mb->global_slot = (batch * max_g + 63) / 64 * 64;
  mb->globals = cv.take(mb->global_slot * (2 + 4 * D));
  const size_t dense_ws = gunet_dense_workspace_bytes(w->c_in, wx[0]);
  const size_t dense_ws2 = gunet_dense_workspace_bytes(c, w->out_units);
  layer_ws = dense_ws > layer_ws ? dense_ws : layer_ws;
  layer_ws = dense_ws2 > layer_ws ? dense_ws2 : layer_ws;
  mb->layer_ws = cv.take(static_cast<int64_t>((layer_ws + 3) / 4));
  mb->layer_ws_bytes = layer_ws;
  mb->bytes = cv.off + 256;
  return GUNET_OK;
}

int unet_forward(const gunet_plan* P, const gunet_unet_weights* w, int32_t batch, const float* x,
                 const float* const edges[4], int32_t edges_shared, const float* global_in, int32_t activation,
                 int32_t out_activation, float* out, float* const edges_out[4], float* global_out, void* workspace,
                 size_t workspace_bytes, hipStream_t stream);

// Whether a forward in GUNET_MATMUL_BF16 mode keeps its intermediates in bf16: the shipped shapes
// (every block 128 wide, 16 input edge features, an in_layer of 97..128 input columns, up to 128
// output columns in whole tiles) on a graph whose levels all have an in-edge table and whose
// upsampling graphs have one parent edge per fine node.  Anything else runs the bf16 products on
// float32 tensors.
bool bf16_storage_ok(const gunet_plan* P, const gunet_unet_weights* w, int64_t batch) {
  if (!bf16_mode()) return false;
  const int D = P->depth;
  // the kernels address a whole batch of a tensor with 32-bit byte offsets
  if (batch * P->levels[0].n_nodes * 128 * 4 >= (int64_t{1} << 32)) return false;
  for (int q = 0; q < 4; ++q)
    if (batch * P->levels[0].n_edges[q] * 32 * 4 >= (int64_t{1} << 32) - 256) return false;
  auto narrow = [](const gunet_gn_weights& g) { return g.units == 128; };
  for (int l = 0; l < D; ++l)
    if (!narrow(w->down[l]) || !narrow(w->up_sampling[l]) || !narrow(w->up_gcn[l][0]) || !narrow(w->up_gcn[l][1]) ||
        !P->one_parent_edge[l])
      return false;
  if (!narrow(w->bottleneck[0]) || !narrow(w->bottleneck[1])) return false;
  for (int l = 0; l <= D; ++l)
    if (P->levels[l].n_nodes > 0 && P->levels[l].in_table == nullptr) return false;
  if (w->down[0].c_in != 128 || w->down[0].f_e != 16) return false;
  if (!(w->c_in == 128 || (w->c_in > 96 && w->c_in < 128 && w->c_in % 4 == 0))) return false;
  return w->out_units % 32 == 0 && w->out_units >= 96 && w->out_units <= 128;
}

// Intermediates of gunet_unet_forward_static_f32: one array per GPU, alive from the loading of the
// code object to the end of the process, shared by every call on that GPU.  In host code the
// array's name is only a handle for hipGetSymbolAddress; the device address it yields (one per
// device) is what kernels and memsets are given.
constexpr size_t kStaticScratchBytes = GUNET_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;

// Device address of `scratch` on device `dev`, which must be the current device.  Looked up once
// per device.
int static_scratch_of(int dev, void** out) {
  static std::mutex lock;
  static void* address[kMaxDevices] = {};
  GUNET_REQUIRE(dev >= 0 && dev < kMaxDevices, "static scratch: device index %d", dev);
  std::lock_guard<std::mutex> hold(lock);
  if (address[dev] == nullptr) {
    void* p = nullptr;
    GUNET_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    GUNET_REQUIRE(p != nullptr && reinterpret_cast<uintptr_t>(p) % 256 == 0, "static scratch: bad symbol address");
    address[dev] = p;
  }
  *out = address[dev];
  return GUNET_OK;
}

}  // namespace
}  // namespace gunet

using namespace gunet;

extern "C" {

int gunet_plan_create(const int64_t* node_loc, int64_t n_nodes, const int32_t* const edge_index[4],
                      const int64_t n_edges[4], int32_t depth, gunet_plan** plan_out, gunet_stream_t stream) {
  StreamDeviceGuard device_guard(as_stream(stream));
  GUNET_REQUIRE(plan_out != nullptr, "plan_create: plan_out is null");
  *plan_out = nullptr;
  GUNET_REQUIRE(depth >= 0 && depth <= GUNET_MAX_DEPTH, "plan_create: depth %d outside 0..%d", depth, GUNET_MAX_DEPTH);
  GUNET_REQUIRE(n_nodes >= 0 && n_nodes < INT32_MAX && edge_index && n_edges, "plan_create: bad arguments");
  GUNET_REQUIRE(n_nodes == 0 || node_loc, "plan_create: node_loc is null");
  for (int q = 0; q < 4; ++q)
    GUNET_REQUIRE(n_edges[q] >= 0 && n_edges[q] < INT32_MAX && (n_edges[q] == 0 || edge_index[q]),
                  "plan_create: bad edge list %d", q);
  gunet_plan* P = new (std::nothrow) gunet_plan;
  GUNET_REQUIRE(P != nullptr, "plan_create: out of host memory");
  P->depth = depth;
  int rc = hipGetDevice(&P->device) == hipSuccess ? GUNET_OK : GUNET_ERR_HIP;
  if (rc != GUNET_OK) set_error("plan_create: hipGetDevice failed");
  // the address of this device's static scratch buffer is looked up here, where allocating and
  // synchronising are allowed, so that gunet_unet_forward_static_f32 -- which needs a plan --
  // never makes a first-time runtime query of its own (it may run under stream capture)
  void* static_base = nullptr;
  if (rc == GUNET_OK) rc = static_scratch_of(P->device, &static_base);
  if (rc == GUNET_OK) rc = build_plan(*P, node_loc, n_nodes, edge_index, n_edges, as_stream(stream));
  if (rc != GUNET_OK) {
    delete P;
    return rc;
  }
  *plan_out = P;
  return GUNET_OK;
}

int gunet_plan_destroy(gunet_plan* plan) {
  delete plan;
  return GUNET_OK;
}

int gunet_plan_depth(const gunet_plan* plan) { return plan ? plan->depth : GUNET_ERR_INVALID; }

int gunet_release_temporaries(void) {
  temp_cache_clear();
  return GUNET_OK;
}

int gunet_plan_level(const gunet_plan* plan, int32_t level, gunet_quadgraph* graph_out, const int64_t** node_loc_out) {
  GUNET_REQUIRE(plan != nullptr, "plan_level: null plan");
  GUNET_REQUIRE(level >= 0 && level <= plan->depth, "plan_level: level %d outside 0..%d", level, plan->depth);
  if (graph_out) *graph_out = plan->levels[level].qg;
  if (node_loc_out) *node_loc_out = plan->levels[level].node_loc;
  return GUNET_OK;
}

int gunet_plan_pooling(const gunet_plan* plan, int32_t level, const int32_t** cluster_out,
                       const int32_t* dup_map_out[4]) {
  GUNET_REQUIRE(plan != nullptr, "plan_pooling: null plan");
  GUNET_REQUIRE(level >= 0 && level < plan->depth, "plan_pooling: level %d outside 0..%d", level, plan->depth - 1);
  if (cluster_out) *cluster_out = plan->pools[level].cluster;
  if (dup_map_out)
    for (int q = 0; q < 4; ++q) dup_map_out[q] = plan->pools[level].dup_map[q];
  return GUNET_OK;
}

int gunet_plan_upsampling(const gunet_plan* plan, int32_t level, gunet_quadgraph* graph_out,
                          const int32_t** parent_out) {
  GUNET_REQUIRE(plan != nullptr, "plan_upsampling: null plan");
  GUNET_REQUIRE(level >= 0 && level < plan->depth, "plan_upsampling: level %d outside 0..%d", level, plan->depth - 1);
  if (graph_out) *graph_out = plan->ups[level].qg;
  if (parent_out) *parent_out = plan->parents[level];
  return GUNET_OK;
}

int gunet_plan_pool_features_f32(const gunet_plan* plan, int32_t level, int32_t batch, const float* x, int32_t width,
                                 const float* const edges[4], int32_t edge_width, float* x_out,
                                 float* const edges_out[4], gunet_stream_t stream) {
  StreamDeviceGuard device_guard(as_stream(stream));
  GUNET_REQUIRE(plan != nullptr, "plan_pool_features: null plan");
  GUNET_REQUIRE(level >= 0 && level < plan->depth, "plan_pool_features: level %d outside 0..%d", level, plan->depth - 1);
  GUNET_REQUIRE(batch > 0 && width >= 0 && edge_width >= 0, "plan_pool_features: bad batch or width");
  int dev = 0;
  GUNET_HIP_TRY(hipGetDevice(&dev));
  GUNET_REQUIRE(dev == plan->device, "plan_pool_features: the plan lives on device %d, the current device is %d",
                plan->device, dev);
  const gunet_plan::Level& L = plan->levels[level];
  const gunet_plan::Level& C = plan->levels[level + 1];
  const gunet_plan::Pool& pool = plan->pools[level];
  const float* src[5];
  float* dst[5];
  const int32_t* rowptr[5];
  const int32_t* perm[5];
  int64_t n_src[5], n_seg[5];
  int widths[5];
  int n = 0;
  if (x != nullptr || x_out != nullptr) {
    GUNET_REQUIRE(x && x_out && width > 0, "plan_pool_features: x, x_out and width go together");
    src[n] = x;
    dst[n] = x_out;
    rowptr[n] = pool.cluster_rowptr;
    perm[n] = pool.cluster_perm;
    n_src[n] = L.n_nodes;
    n_seg[n] = C.n_nodes;
    widths[n++] = width;
  }
  if (edges != nullptr || edges_out != nullptr) {
    GUNET_REQUIRE(edges && edges_out && edge_width > 0, "plan_pool_features: edges, edges_out and edge_width go together");
    for (int q = 0; q < 4; ++q) {
      if (C.n_edges[q] == 0) continue;
      GUNET_REQUIRE(edges[q] && edges_out[q], "plan_pool_features: null edge tensor %d", q);
      src[n] = edges[q];
      dst[n] = edges_out[q];
      rowptr[n] = pool.dup_rowptr[q];
      perm[n] = pool.dup_perm[q];
      n_src[n] = L.n_edges[q];
      n_seg[n] = C.n_edges[q];
      widths[n++] = edge_width;
    }
  }
  return pool_max_multi(n, src, n_src, widths, rowptr, perm, n_seg, dst, batch, as_stream(stream));
}

size_t gunet_unet_workspace_bytes(const gunet_plan* plan, const gunet_unet_weights* weights, int32_t batch) {
  if (plan == nullptr || weights == nullptr || batch <= 0) return 0;
  ModelBuffers mb;
  int final_width = 0;
  if (carve_model("unet_workspace_bytes", plan, weights, batch, nullptr, &mb, &final_width) != GUNET_OK) return 0;
  return mb.bytes;
}

int gunet_unet_forward_f32(const gunet_plan* P, const gunet_unet_weights* w, int32_t batch, const float* x,
                           const float* const edges[4], int32_t edges_shared, const float* global_in,
                           int32_t activation, int32_t out_activation, float* out, float* const edges_out[4],
                           float* global_out, void* workspace, size_t workspace_bytes, gunet_stream_t stream) {
  StreamDeviceGuard device_guard(as_stream(stream));
  return unet_forward(P, w, batch, x, edges, edges_shared, global_in, activation, out_activation, out, edges_out, global_out,
                      workspace, workspace_bytes, as_stream(stream));
}

int gunet_unet_forward_static_f32(const gunet_plan* P, const gunet_unet_weights* w, int32_t batch, const float* x,
                                  const float* const edges[4], int32_t edges_shared, const float* global_in,
                                  int32_t activation, int32_t out_activation, float* out, float* const edges_out[4],
                                  float* global_out, gunet_stream_t stream) {
  StreamDeviceGuard device_guard(as_stream(stream));
  GUNET_REQUIRE(P && w, "unet_forward_static: null plan or weights");
  GUNET_REQUIRE(batch > 0, "unet_forward_static: batch %d", batch);
  ModelBuffers mb;
  int final_width = 0;
  GUNET_TRY(carve_model("unet_forward_static", P, w, batch, nullptr, &mb, &final_width));
  if (mb.bytes > kStaticScratchBytes) {
    set_error("unet_forward_static: this graph and batch need %zu bytes of intermediates, the library's static scratch "
              "buffer holds %zu (use gunet_unet_forward_f32 with a caller workspace)", mb.bytes, kStaticScratchBytes);
    return GUNET_ERR_WORKSPACE;
  }
  int dev = 0;
  GUNET_HIP_TRY(hipGetDevice(&dev));
  GUNET_REQUIRE(dev == P->device, "unet_forward_static: the plan lives on device %d, the current device is %d",
                P->device, dev);
  void* base = nullptr;
  GUNET_TRY(static_scratch_of(dev, &base));
  return unet_forward(P, w, batch, x, edges, edges_shared, global_in, activation, out_activation, out, edges_out,
                      global_out, base, kStaticScratchBytes, as_stream(stream));
}

}  // extern "C"

namespace gunet {
namespace {

int unet_forward(const gunet_plan* P, const gunet_unet_weights* w, int32_t batch, const float* x,
                 const float* const edges[4], int32_t edges_shared, const float* global_in, int32_t activation,
                 int32_t out_activation, float* out, float* const edges_out[4], float* global_out, void* workspace,
                 size_t workspace_bytes, hipStream_t stream) {
  const int shared0 = edges_shared ? kGnEdgesShared : 0;   // the caller's (level-0) edge features
  GUNET_REQUIRE(P && w, "unet_forward: null plan or weights");
  GUNET_REQUIRE(batch > 0, "unet_forward: batch %d", batch);
  GUNET_REQUIRE(x && edges && global_in && out, "unet_forward: null tensor");
  GUNET_REQUIRE(act_valid(activation) && act_valid(out_activation), "unet_forward: unknown activation %d / %d", activation,
                out_activation);
  if (!act_tuned(activation)) {
    set_error("unet_forward: the one-call forward serves linear and relu layers; run activation %d layer by layer "
              "(gunet_gn_layer_forward_f32 and the plan views)", activation);
    return GUNET_ERR_UNSUPPORTED;
  }
  GUNET_REQUIRE(workspace != nullptr && reinterpret_cast<uintptr_t>(workspace) % 256 == 0,
                "unet_forward: the workspace must be 256-byte aligned");
  int dev = 0;
  GUNET_HIP_TRY(hipGetDevice(&dev));
  GUNET_REQUIRE(dev == P->device, "unet_forward: the plan lives on device %d, the current device is %d", P->device, dev);
  ModelBuffers mb;
  int final_width = 0;
  GUNET_TRY(carve_model("unet_forward", P, w, batch, workspace, &mb, &final_width));
  if (workspace_bytes < mb.bytes) {
    set_error("unet_forward: workspace of %zu bytes, %zu needed", workspace_bytes, mb.bytes);
    return GUNET_ERR_WORKSPACE;
  }
  const int D = P->depth;
  const int64_t B = batch;
  int slot = 0;
  auto next_global = [&]() { return mb.globals + mb.global_slot * slot++; };
  // GUNET_MATMUL_BF16 on the shipped shapes: every intermediate tensor (node features, node
  // embeddings, messages; skip tensors included) is kept in bf16 in the first half of its buffer
  const bool s16 = bf16_storage_ok(P, w, batch) && act_tuned(out_activation);
  const int f16 = s16 ? kGnStore16 : 0;
  const int in32 = s16 ? kGnEdgesIn32 : 0;   // layers that read the caller's float32 edge features

  // in_layer (graph_unet.py:37, :66): Dense with bias, no activation
  GUNET_TRY(dense_run(x, B * P->levels[0].n_nodes, w->c_in, w->in_kernel, w->in_bias, w->down[0].c_in, GUNET_ACT_LINEAR,
                      mb.skip_x[0], mb.layer_ws, mb.layer_ws_bytes, 0, s16, stream));

  // Consecutive GN layers are joined: the global update that ends one layer and the prologue that
  // starts the next (both one small workgroup per sample) run as one kernel, gn_bridge.  `pend` is
  // the layer whose global update is still owed.
  struct {
    const gunet_quadgraph* graph;
    const gunet_gn_weights* w;
    int n_partials;
    const float* g_in;
    float* g_out;
  } pend = {};
  const int mid = kGnSkipPrologue | kGnSkipGlobalUpdate;
  auto enter = [&](const gunet_quadgraph* graph, const gunet_gn_weights* gw, int fold_receiver, const float* extra,
                   int extra_width, float* cat_out) {
    return gn_bridge(pend.graph, pend.w, pend.n_partials, pend.g_in, pend.g_out, extra, extra_width, cat_out, graph, gw,
                     fold_receiver, batch, activation, mb.layer_ws, mb.layer_ws_bytes, stream);
  };

  // down path (:67-71): GN layer, then 2x2 pooling; what enters each block stays for the way up
  const float* cur_g = global_in;
  for (int l = 0; l < D; ++l) {
    const gunet_plan::Level& L = P->levels[l];
    const gunet_plan::Level& C = P->levels[l + 1];
    const gunet_plan::Pool& pool = P->pools[l];
    const gunet_gn_weights& gw = w->down[l];
    const float* const* e_in = l == 0 ? edges : mb.skip_e[l].q;
    if (l > 0) GUNET_TRY(enter(&L.qg, &gw, 0, nullptr, 0, nullptr));
    pend = {&L.qg, &gw, 0, cur_g, mb.skip_g[l + 1]};
    GUNET_TRY(gn_layer_run(&L.qg, &gw, batch, mb.skip_x[l], nullptr, 0, e_in, cur_g, activation, mb.tmp_x[0],
                           mb.tmp_e[0].q, nullptr, mb.layer_ws, mb.layer_ws_bytes,
                           (l == 0 ? (kGnSkipGlobalUpdate | shared0 | in32) : mid) | f16, &pend.n_partials, stream));
    cur_g = mb.skip_g[l + 1];
    {  // graph_pooling_2x2 (hybrid_gnn.py:427, :401): node features and the four edge-feature sets
      const float* src[5] = {mb.tmp_x[0]};
      float* dst[5] = {mb.skip_x[l + 1]};
      const int32_t* rowptr[5] = {pool.cluster_rowptr};
      const int32_t* perm[5] = {pool.cluster_perm};
      int64_t n_src[5] = {L.n_nodes}, n_seg[5] = {C.n_nodes};
      int width[5] = {gw.units};
      for (int q = 0; q < 4; ++q) {
        src[1 + q] = mb.tmp_e[0].q[q];
        dst[1 + q] = mb.skip_e[l + 1].q[q];
        rowptr[1 + q] = pool.dup_rowptr[q];
        perm[1 + q] = pool.dup_perm[q];
        n_src[1 + q] = L.n_edges[q];
        n_seg[1 + q] = C.n_edges[q];
        width[1 + q] = gw.units / 4;
      }
      GUNET_TRY(pool_max_multi(5, src, n_src, width, rowptr, perm, n_seg, dst, batch, stream, s16));
    }
  }

  // bottleneck (:73-74)
  const float* cur_x = mb.skip_x[D];
  const float* const* cur_e = mb.skip_e[D].q;
  int side = 0;   // tmp buffer the next layer writes
  for (int i = 0; i < 2; ++i) {
    const gunet_quadgraph* graph = &P->levels[D].qg;
    float* g_out = next_global();
    GUNET_TRY(enter(graph, &w->bottleneck[i], 0, nullptr, 0, nullptr));
    pend = {graph, &w->bottleneck[i], 0, cur_g, g_out};
    GUNET_TRY(gn_layer_run(graph, &w->bottleneck[i], batch, cur_x, nullptr, 0, cur_e, cur_g, activation,
                           mb.tmp_x[side], mb.tmp_e[side].q, nullptr, mb.layer_ws, mb.layer_ws_bytes, mid | f16,
                           &pend.n_partials, stream));
    cur_x = mb.tmp_x[side];
    cur_e = mb.tmp_e[side].q;
    cur_g = g_out;
    side ^= 1;
  }

  // up path (:76-80): graph upsampling, skip concatenation, two GN layers
  for (int i = 0; i < D; ++i) {
    const int l = D - 1 - i;
    const gunet_plan::Level& L = P->levels[l];
    const gunet_plan::Level& C = P->levels[l + 1];
    const gunet_gn_weights& up = w->up_sampling[i];
    const float* target_g = l == 0 ? global_in : mb.skip_g[l];
    const int target_width = l == 0 ? w->down[0].g : global_out_width(w->down[l - 1]);
    float* g_cat = next_global();   // [global | target global], hybrid_gnn.py:375
    float* g_up = next_global();
    const bool fast = P->one_parent_edge[l] && up.units <= 256;   // else the reference's literal route
    GUNET_REQUIRE(fast || !s16, "unet_forward: bf16 storage on an upsampling level that needs the literal route");
    GUNET_TRY(enter(&P->ups[l].qg, &up, fast ? 1 : 0, target_g, target_width, g_cat));
    pend = {&P->ups[l].qg, &up, 0, g_cat, g_up};
    if (fast) {
      GUNET_TRY(gn_upsample_run(&P->ups[l].qg, static_cast<int32_t>(C.n_nodes), &up, batch, cur_x, g_cat, activation,
                                mb.tmp_x[side], nullptr, mb.layer_ws, mb.layer_ws_bytes, mid | f16, &pend.n_partials, stream));
    } else {   // hybrid_gnn.py:342-344, 373-377 as written: the layer on [x ; zeros] with zero edge features
      const int64_t n_all = C.n_nodes + L.n_nodes;
      GUNET_HIP_TRY(hipMemsetAsync(mb.up_in, 0, sizeof(float) * B * n_all * up.c_in, stream));
      GUNET_TRY(copy_rows(cur_x, C.n_nodes * up.c_in, mb.up_in, n_all * up.c_in, C.n_nodes * up.c_in, B, stream));
      const float* no_edges[4] = {};
      GUNET_TRY(gn_layer_run(&P->ups[l].qg, &up, batch, mb.up_in, nullptr, 0, no_edges, g_cat, activation, mb.up_out,
                             mb.tmp_e[side].q, nullptr, mb.layer_ws, mb.layer_ws_bytes, mid, &pend.n_partials, stream));
      GUNET_TRY(copy_rows(mb.up_out + C.n_nodes * up.units, n_all * up.units, mb.tmp_x[side], L.n_nodes * up.units,
                          L.n_nodes * up.units, B, stream));
    }
    cur_x = mb.tmp_x[side];
    cur_g = g_up;
    side ^= 1;
    const float* const* e_in = l == 0 ? edges : mb.skip_e[l].q;
    for (int j = 0; j < 2; ++j) {
      const gunet_gn_weights& gw = w->up_gcn[i][j];
      const bool last = i == D - 1 && j == 1;
      float* g_out = last && global_out ? global_out : next_global();
      float* const* e_out = mb.tmp_e[side].q;
      float* last_e[4];
      if (last && edges_out && !s16) {
        for (int q = 0; q < 4; ++q) last_e[q] = edges_out[q] ? edges_out[q] : mb.tmp_e[side].q[q];
        e_out = last_e;
      }
      // bf16 storage: the messages go to the workspace in bf16 (the node MLP sums them from there) and,
      // in the last layer, to the caller's tensors in float32 as well
      float* const* e_out32 = (last && s16) ? edges_out : nullptr;
      const int c_skip = j == 0 ? gw.c_in - up.units : 0;   // concat(x, x_skip), hybrid_gnn.py:235
      GUNET_TRY(enter(&L.qg, &gw, 0, nullptr, 0, nullptr));
      pend = {&L.qg, &gw, 0, cur_g, g_out};
      // the last layer of the model finishes its own global
      // the first layer of the last block reads the caller's edge features again (the skip graph)
      const int shared = (l == 0 && j == 0) ? shared0 : 0;
      GUNET_TRY(gn_layer_run(&L.qg, &gw, batch, cur_x, j == 0 ? mb.skip_x[l] : nullptr, c_skip, e_in, cur_g, activation,
                             mb.tmp_x[side], e_out, last ? g_out : nullptr, mb.layer_ws, mb.layer_ws_bytes,
                             (last ? kGnSkipPrologue : mid) | shared | f16 | ((l == 0 && j == 0) ? in32 : 0),
                             &pend.n_partials, stream, e_out32));
      cur_x = mb.tmp_x[side];
      e_in = e_out == last_e ? nullptr : mb.tmp_e[side].q;
      cur_g = g_out;
      side ^= 1;
    }
  }
  // out_model (:82): Dense without bias
  return dense_run(cur_x, B * P->levels[0].n_nodes, final_width, w->out_kernel, w->out_bias, w->out_units, out_activation,
                   out, mb.layer_ws, mb.layer_ws_bytes, s16, 0, stream);
}

}  // namespace
}  // namespace gunet
