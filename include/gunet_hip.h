/*
 * libgunet_hip.so -- C ABI of the MI355X (gfx950) Graph U-Net forward path.
 *
 * Every entry point is extern "C", takes plain device pointers and sizes, never
 * throws and never owns caller memory.  Return value: 0 = ok, negative = error
 * (GUNET_ERR_*); gunet_last_error() returns a thread-local message.
 *
 * The reference (LucaHermes/graph-unet-traffic-prediction) is pure Python on
 * TensorFlow and has no FFI layer; each entry point below cites the Python
 * function (file:line in the reference checkout) whose work it replaces.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - index-building entry points ("plan" functions: everything in section 1)
 *     synchronise the stream and allocate temporary device memory internally;
 *     feature entry points (section 2) are asynchronous, allocate nothing and
 *     use only the workspace the caller hands in;
 *   - feature tensors are row-major float32 with a leading batch dimension B of
 *     independent samples that share one graph: x [B,N,C], edges [B,E_q,F],
 *     global [B,G].  B = 1 reproduces the reference's single-sample call.
 */
#ifndef GUNET_HIP_H
#define GUNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GUNET_OK 0
#define GUNET_ERR_INVALID (-1)     /* bad argument (null pointer, negative size, index out of range) */
#define GUNET_ERR_HIP (-2)         /* a HIP runtime call failed */
#define GUNET_ERR_WORKSPACE (-3)   /* workspace too small */
#define GUNET_ERR_UNSUPPORTED (-4) /* configuration the library does not implement */

/* Activations (layers/hybrid_gnn.py:25, models/graph_unet.py:21 take any Keras activation name).
 * LINEAR and RELU -- the shipped configuration -- run on the tuned kernels; the others give the same
 * layer through the library's generic kernels (tiled float32 GEMMs, per-direction edge GEMMs, a separate
 * aggregation pass; exact float32 in every matmul mode).  The whole-model entry points and the
 * zero-row-free upsampling layer serve LINEAR / RELU only (GUNET_ERR_UNSUPPORTED otherwise: run the
 * layers one by one).  Keras definitions: elu with alpha = 1, selu with its two constants, softplus =
 * log(exp(x) + 1), swish = x * sigmoid(x). */
#define GUNET_ACT_LINEAR 0
#define GUNET_ACT_RELU 1
#define GUNET_ACT_TANH 2
#define GUNET_ACT_SIGMOID 3
#define GUNET_ACT_ELU 4
#define GUNET_ACT_SOFTPLUS 5
#define GUNET_ACT_SELU 6
#define GUNET_ACT_SWISH 7
#define GUNET_ACT_GELU 8          /* exact form: 0.5 x (1 + erf(x / sqrt 2)) */
#define GUNET_ACT_SOFTSIGN 9
#define GUNET_ACT_EXPONENTIAL 10
#define GUNET_ACT_HARD_SIGMOID 11 /* clip(0.2 x + 0.5, 0, 1) */

typedef void* gunet_stream_t;

const char* gunet_last_error(void);
int gunet_version(void);

/* ------------------------------------------------------------------------
 * 1. Index ("plan") functions -- integer work, bit-exact with the reference
 * ------------------------------------------------------------------------ */

/* data/data_utils.py:47-58 sample_quadrant_subgraphs (with QUADRANT_IDS :24-30).
 * Stable partition of the edge list into the NE, SE, SW, NW quadrant lists;
 * directions 0,2,4,6 go to two lists, direction 8 (self-loop) to none.
 * out_edges[q] must each hold n_edges rows of int32[2]; the number of rows
 * written to each is returned in counts_host[q]. */
int gunet_quadrant_partition(const int32_t* edge_index, const int64_t* direction, int64_t n_edges,
                             int32_t* const out_edges[4], int64_t counts_host[4],
                             gunet_stream_t stream);

/* Receiver-sorted CSR of one quadrant edge list: the index structure behind
 * tf.math.unsorted_segment_sum(edge, e_index[:,1], n_nodes)
 * (layers/hybrid_gnn.py:110-111).  rowptr[n_nodes+1]; perm[n_edges] maps a CSR
 * slot to the edge's position in the reference-ordered list, ascending inside a
 * row, so a front-to-back walk adds messages in the reference's edge order. */
int gunet_csr_build(const int32_t* edge_index, int64_t n_edges, int32_t n_nodes,
                    int32_t* rowptr, int32_t* perm, gunet_stream_t stream);

/* Generic form of the above: segments from an int32 key per element
 * (key < 0 = element belongs to no segment).  rowptr[n_segments+1];
 * perm[<= n] lists the elements of each segment in ascending order. */
int gunet_csr_from_keys(const int32_t* keys, int64_t n, int32_t n_segments,
                        int32_t* rowptr, int32_t* perm, gunet_stream_t stream);

/* layers/hybrid_gnn.py:415-429 (graph_pooling_2x2, node part) and :270-280
 * (GraphUpsampling2x2.call).  cluster[i] = tf.unique((row//2)*(max_row+1) +
 * col//2).idx, i.e. the rank of the first occurrence of node i's 2x2 cell;
 * pooled_loc = unsorted_segment_min(node_loc // 2) (may be NULL to skip).
 * cluster[n_nodes]; pooled_loc holds up to n_nodes rows of int64[2]. */
int gunet_cluster_ids(const int64_t* node_loc, int64_t n_nodes, int32_t* cluster,
                      int64_t* pooled_loc, int64_t* n_pooled_host, gunet_stream_t stream);

/* layers/hybrid_gnn.py:431-434 (transform_edges) + :391-408
 * (remove_duplicate_edges), index part: relabel through `cluster`, drop
 * duplicates keeping first occurrences (key = s*n_pre + r in int64), then drop
 * self-loops.  new_edges holds up to n_edges rows; dup_map[e] is the row of
 * edge e in new_edges, or -1 when it became a self-loop. */
int gunet_pool_edges(const int32_t* edge_index, int64_t n_edges, const int32_t* cluster,
                     int64_t n_pre, int32_t* new_edges, int32_t* dup_map,
                     int64_t* n_new_host, gunet_stream_t stream);

/* layers/hybrid_gnn.py:336-357 (GraphUpsampling2x2.call, method 'graph'):
 * parent[i] = cluster id of fine node i, edge i = (parent[i], n_coarse + i),
 * split into 4 lists by the child's place in its 2x2 cell
 * (row even/col odd -> 0, odd/odd -> 1, odd/even -> 2, even/even -> 3).
 * out_edges[q] must each hold n_fine rows. */
int gunet_upsample_edges(const int64_t* coarse_loc, int64_t n_coarse, const int64_t* fine_loc,
                         int64_t n_fine, int32_t* parent, int32_t* const out_edges[4],
                         int64_t counts_host[4], gunet_stream_t stream);

/* ------------------------------------------------------------------------
 * 2. Feature functions -- float32, asynchronous on `stream`
 * ------------------------------------------------------------------------ */

/* layers/graph_ops.py:5-17 get_sender_receiver: two row gathers.
 * x [n_nodes, width]; senders/receivers [n_edges, width]. */
int gunet_get_sender_receiver_f32(const float* x, int64_t n_nodes, int32_t width,
                                  const int32_t* edge_index, int64_t n_edges, float* senders,
                                  float* receivers, gunet_stream_t stream);

/* tf.math.unsorted_segment_max / _sum over rows (layers/hybrid_gnn.py:401,427,
 * 110).  src [B, n_src, width]; out [B, n_segments, width]; segments given as
 * CSR (rowptr, perm) from gunet_csr_from_keys.  Empty segment: -FLT_MAX (max)
 * or 0 (sum).  The sum adds rows front to back and multiplies by `scale`.  The maximum is fmaxf's,
 * started from -FLT_MAX: NaNs among the members are dropped, a segment of NaNs only gives -FLT_MAX
 * (TensorFlow propagates a NaN). */
int gunet_segment_max_f32(const float* src, int64_t n_src, int32_t width, const int32_t* rowptr,
                          const int32_t* perm, int64_t n_segments, int32_t batch, float* out,
                          gunet_stream_t stream);
int gunet_segment_sum_f32(const float* src, int64_t n_src, int32_t width, const int32_t* rowptr,
                          const int32_t* perm, int64_t n_segments, int32_t batch, float scale,
                          float* out, gunet_stream_t stream);

/* Arithmetic of every matrix product of the feature functions (Dense and the GN layer's node
 * embeddings, edge MLP and node MLP).  The accumulator is float32 in all modes, and so are all
 * tensors at the interface (inputs, outputs, weights).
 *   GUNET_MATMUL_F32     v_mfma_f32_32x32x2_f32: exact float32 products (bitwise an fmaf chain).
 *   GUNET_MATMUL_SPLIT3  each operand = sum of 3 bf16 pieces, 6 piece products on the bf16 matrix
 *                        cores: float32-class accuracy (all dropped terms <= 2^-24 relative).
 *   GUNET_MATMUL_SPLIT2  2 pieces, 3 products: about 2^-17 relative per product.
 *   GUNET_MATMUL_BF16    operands rounded to bf16, 1 product: the bf16 tolerance class (2e-2 of the
 *                        output's scale).  In this mode gunet_unet_forward_f32 / _static_f32 on a model
 *                        whose blocks are all 128 wide, over a plan with at most 3 incoming edges per
 *                        node and quadrant (every pixel-grid graph), also keeps its intermediates --
 *                        node features, node embeddings, messages, pooled features -- in bf16 inside
 *                        the workspace (each rounded once from the float32 result); edges_out and
 *                        global_out are still float32.  Other models, and the per-layer functions, store
 *                        float32.
 * The setting is process-wide and read at launch time.  No counterpart in the reference
 * (TensorFlow float32 matmuls). */
#define GUNET_MATMUL_F32 0
#define GUNET_MATMUL_SPLIT3 1
#define GUNET_MATMUL_SPLIT2 2
#define GUNET_MATMUL_BF16 3
int gunet_set_matmul_mode(int mode);
int gunet_get_matmul_mode(void);

/* Which float32 GEMM kernel the feature functions take (no counterpart in the reference; results
 * agree to rounding).  AUTO (default): the weights-resident kernels -- including the node MLP that
 * sums the messages while loading them -- where they apply and the launch is large enough to pay
 * for copying the weights into every workgroup, else the tiled kernel.  RESIDENT: wherever they
 * apply, whatever the size.  TILED: never (a separate aggregation pass feeds the node MLP). */
#define GUNET_KERNELS_AUTO 0
#define GUNET_KERNELS_RESIDENT 1
#define GUNET_KERNELS_TILED 2
int gunet_set_kernel_choice(int choice);

/* tf.keras.layers.Dense (models/graph_unet.py:37,58,66,82): out = act(x @ kernel + bias).
 * x [rows, k]; kernel [k, n_out]; bias [n_out] or NULL; out [rows, n_out].
 * workspace: gunet_dense_workspace_bytes(k, n_out) bytes, 16-byte aligned; may be NULL in
 * GUNET_MATMUL_F32 mode. */
size_t gunet_dense_workspace_bytes(int32_t k, int32_t n_out);
int gunet_dense_forward_f32(const float* x, int64_t rows, int32_t k, const float* kernel,
                            const float* bias, int32_t n_out, int32_t activation, float* out,
                            void* workspace, size_t workspace_bytes, gunet_stream_t stream);

/* One quadrant graph: 4 reference-ordered edge lists and their receiver CSRs.
 * in_table (optional, NULL = absent): for graphs whose nodes have at most 3 incoming edges per
 * quadrant (every pixel-grid quadrant graph and its 2x2 poolings), the incoming edge positions of
 * node v in quadrant q at in_table[(v * 4 + q) * 3 + 0..2], ascending, -1 padded
 * (gunet_in_edge_table).  With it the node MLP sums the messages while it loads them and the
 * separate aggregation pass is skipped. */
typedef struct gunet_quadgraph {
  int32_t n_nodes;
  int32_t n_edges[4];
  const int32_t* edge_index[4]; /* [E_q, 2] sender, receiver */
  const int32_t* rowptr[4];     /* [n_nodes + 1] */
  const int32_t* perm[4];       /* [E_q] */
  const int32_t* in_table;      /* [n_nodes, 4, 3] or NULL */
} gunet_quadgraph;

/* Per-node table of incoming edges of a quadrant graph (see gunet_quadgraph.in_table): fills
 * table int32 [n_nodes, 4, 3] from graph->rowptr / perm and returns the largest in-degree found in
 * *max_in_degree_host; the table is only valid when that is <= 3.  graph->in_table is not read.
 * A plan function (synchronises).  No counterpart in the reference (index side of
 * tf.math.unsorted_segment_sum, layers/hybrid_gnn.py:110). */
int gunet_in_edge_table(const gunet_quadgraph* graph, int32_t* table, int32_t* max_in_degree_host,
                        gunet_stream_t stream);

/* Parameters of one GeoQuadrantGCN (layers/hybrid_gnn.py:20-70), in the layout
 * of the reference's checkpoint: Dense kernels [in, out] row-major, one set per
 * direction.  u = units / 4.  bias pointers may be NULL (use_bias=False).
 * globt_w == NULL selects the identity global transform of the last layer. */
typedef struct gunet_gn_weights {
  int32_t c_in, f_e, g, units;
  const float* node_emb_w[4]; /* [c_in, u] */
  const float* node_emb_b[4]; /* [u] */
  const float* edge_w[4];     /* [f_e + 3u, u] */
  const float* edge_b[4];     /* [u] */
  const float* glob_w[4];     /* [g, u] for q = 0, [u, u] for q = 1..3 */
  const float* glob_b[4];     /* [u] */
  const float* node_w;        /* [c_in + units + g, units] */
  const float* node_b;        /* [units] */
  const float* globt_w;       /* [g + units + u, units] or NULL */
  const float* globt_b;       /* [units] or NULL */
  float global_scale_nodes;
  float global_scale_edges;
} gunet_gn_weights;

size_t gunet_gn_layer_workspace_bytes(const gunet_quadgraph* graph, int32_t batch, int32_t units);

/* layers/hybrid_gnn.py:72-137 GeoQuadrantGCN.call_concat with use_global=True.
 *   x          [B, N, c_in - c_skip]
 *   x_skip     [B, N, c_skip] or NULL (c_skip = 0): the trailing input columns, so that the caller of
 *              the up path (hybrid_gnn.py:235, x = concat(x, x_skip)) need not materialise the concat
 *   edges[q]   [B, E_q, f_e], or NULL meaning all-zero edge features; with edges_shared != 0 one
 *              [E_q, f_e] tensor serves every sample (the street-map features of
 *              models/base_model.py:294-298 depend on the city only)
 *   global_in  [B, g]
 *   x_out      [B, N, units]
 *   edges_out[q] [B, E_q, u]          (graph['edges'] after the call)
 *   global_out [B, units], or [B, g + units + u] when globt_w == NULL
 * units = 128, 256 (tuned kernels), 512 and 1024 (generic kernels) are implemented; other widths
 * return GUNET_ERR_UNSUPPORTED.
 * NaN handling: a linear layer hands a NaN on; the ReLU of a NaN is 0 here (fmaxf), NaN in
 * TensorFlow -- a diverged input shows in the linear in/out layers, not behind a ReLU. */
int gunet_gn_layer_forward_f32(const gunet_quadgraph* graph, const gunet_gn_weights* weights,
                               int32_t batch, const float* x, const float* x_skip, int32_t c_skip,
                               const float* const edges[4], int32_t edges_shared, const float* global_in,
                               int32_t activation, float* x_out, float* const edges_out[4],
                               float* global_out, void* workspace, size_t workspace_bytes,
                               gunet_stream_t stream);

/* layers/hybrid_gnn.py:336-378: the GN layer of GraphUpsampling2x2 (method 'graph').  The
 * reference runs the layer above on [x ; zeros(n_fine)] with zero edge features and keeps the
 * fine rows; this entry point returns the same fine rows and global without the zero rows:
 * `graph` is the upsampling graph over n_coarse + n_fine nodes whose edges are
 * (parent, n_coarse + i), exactly one per fine node (gunet_upsample_edges);
 *   x [B, n_coarse, c_in], global_in [B, g] (already [global | target global]),
 *   x_out [B, n_fine, units], global_out as above.  The layer's edge outputs are not returned
 * (the reference discards them, :383).  Workspace: gunet_gn_layer_workspace_bytes(graph, ...). */
int gunet_gn_upsample_forward_f32(const gunet_quadgraph* graph, int32_t n_coarse,
                                  const gunet_gn_weights* weights, int32_t batch, const float* x,
                                  const float* global_in, int32_t activation, float* x_out,
                                  float* global_out, void* workspace, size_t workspace_bytes,
                                  gunet_stream_t stream);

/* ------------------------------------------------------------------------
 * 2b. The whole model -- models/graph_unet.py:21-84
 * ------------------------------------------------------------------------ */

#define GUNET_MAX_DEPTH 8

/* Every index structure GraphUNet.call needs for one city graph, owned by the handle and immutable
 * after creation (shareable across streams and samples): per level l = 0..depth the quadrant graph
 * with its receiver CSRs and in-edge table, the 2x2 pooling maps l -> l+1
 * (layers/hybrid_gnn.py:410-447, graph_pooling_2x2 and remove_duplicate_edges, index part) and the
 * upsampling graph l+1 -> l (:270-280, :336-357).  The reference recomputes all of it with
 * tf.unique / boolean masks inside every forward.  node_loc int64 [N,2], edge_index[q] int32
 * [n_edges[q],2]: the level-0 graph (graph dict of models/base_model.py:302-312); the plan keeps its
 * own copy.  A plan function: allocates device memory (freed by gunet_plan_destroy) and
 * synchronises the stream. */
typedef struct gunet_plan gunet_plan;
int gunet_plan_create(const int64_t* node_loc, int64_t n_nodes, const int32_t* const edge_index[4],
                      const int64_t n_edges[4], int32_t depth, gunet_plan** plan_out,
                      gunet_stream_t stream);
int gunet_plan_destroy(gunet_plan* plan);
int gunet_plan_depth(const gunet_plan* plan);

/* The plan functions draw their temporary device memory from a cache inside the library (at most
 * 256 MiB of idle blocks per process, reused stream by stream) instead of allocating and freeing
 * per call; this frees every idle block of every device.  No counterpart in the reference. */
int gunet_release_temporaries(void);

/* Views into a plan (device pointers owned by the plan; any output may be NULL).
 *   level l in 0..depth:  the quadrant graph and node_loc int64 [n_nodes,2] after l poolings;
 *   pooling l in 0..depth-1: cluster int32 [n_l] (:423 new_node_ids) and, per quadrant, dup_map
 *                         int32 [E_lq] (edge -> pooled edge, -1 = became a self-loop);
 *   upsampling l in 0..depth-1: the graph over n_{l+1} + n_l nodes with edges
 *                         (parent, n_{l+1} + i) and parent int32 [n_l]. */
int gunet_plan_level(const gunet_plan* plan, int32_t level, gunet_quadgraph* graph_out,
                     const int64_t** node_loc_out);
int gunet_plan_pooling(const gunet_plan* plan, int32_t level, const int32_t** cluster_out,
                       const int32_t* dup_map_out[4]);
int gunet_plan_upsampling(const gunet_plan* plan, int32_t level, gunet_quadgraph* graph_out,
                          const int32_t** parent_out);

/* layers/hybrid_gnn.py:427 and :401 (graph_pooling_2x2, feature part) for pooling `level` -> level + 1
 * of a plan, as one launch: x_out [B, n_{l+1}, width] = unsorted_segment_max(x [B, n_l, width],
 * cluster) and, per quadrant, edges_out[q] [B, E_{l+1,q}, edge_width] = the maximum of
 * edges[q] [B, E_lq, edge_width] over the fine edges that map to each pooled edge (self-loops
 * dropped).  Either half may be left out (x = x_out = NULL, or edges = edges_out = NULL).
 * Asynchronous on `stream`, allocates nothing.  Maxima are exact: bit-identical to the reference for
 * inputs without NaNs (NaNs among the members are dropped, see gunet_segment_max_f32; TensorFlow
 * hands them on). */
int gunet_plan_pool_features_f32(const gunet_plan* plan, int32_t level, int32_t batch, const float* x,
                                 int32_t width, const float* const edges[4], int32_t edge_width,
                                 float* x_out, float* const edges_out[4], gunet_stream_t stream);

/* Parameters of GraphUNet (models/graph_unet.py:21-59) in the checkpoint's layout.  The in_layer's
 * width is down[0].c_in; every later width follows from the layers' own fields, and the forward
 * checks that they chain as the model wires them. */
typedef struct gunet_unet_weights {
  int32_t depth;
  int32_t c_in;                /* columns of the model input x (104 in the reference) */
  int32_t out_units;           /* columns of the prediction (96) */
  const float* in_kernel;      /* in_layer [c_in, down[0].c_in] */
  const float* in_bias;        /* [down[0].c_in] or NULL */
  const float* out_kernel;     /* out_model [units of the last layer, out_units] */
  const float* out_bias;       /* NULL in the reference (use_bias=False, :58) */
  gunet_gn_weights down[GUNET_MAX_DEPTH];        /* downsampling_layers/i/gcns/0 */
  gunet_gn_weights bottleneck[2];                /* bottleneck_gcn_1, bottleneck_gcn_2 */
  gunet_gn_weights up_sampling[GUNET_MAX_DEPTH]; /* upsampling_layers/i/upsampling/gcn */
  gunet_gn_weights up_gcn[GUNET_MAX_DEPTH][2];   /* upsampling_layers/i/gcns/0..1 */
} gunet_unet_weights;

/* models/graph_unet.py:61-84 GraphUNet.call (diffuse_graph=False, residual=False, method 'graph',
 * use_global=True: the configuration the reference trains), for `batch` samples on the plan's graph:
 *   x [B, N, c_in], edges[q] [B, E_q, down[0].f_e] (or, with edges_shared != 0, one [E_q, f_e]
 *   tensor for all samples), global_in [B, down[0].g]
 *   out [B, N, out_units] = out_activation(out_model(...)); GUNET_ACT_RELU fuses the max(pred, 0)
 *   of models/base_model.py:325
 *   edges_out (may be NULL, as may its entries): graph['edges'] after the call, [B, E_q, u];
 *   global_out (may be NULL): graph['global'] after the call, [B, g + units + u of the last layer].
 * Same kernels in the same order as the per-layer entry points: results are bit-identical to
 * calling those one by one.  Asynchronous on `stream`; allocates nothing; every intermediate
 * (skip tensors included) lives in `workspace` (256-byte aligned,
 * gunet_unet_workspace_bytes(plan, weights, batch) bytes; 0 = the weights do not fit the plan). */
size_t gunet_unet_workspace_bytes(const gunet_plan* plan, const gunet_unet_weights* weights,
                                  int32_t batch);
int gunet_unet_forward_f32(const gunet_plan* plan, const gunet_unet_weights* weights, int32_t batch,
                           const float* x, const float* const edges[4], int32_t edges_shared, const float* global_in,
                           int32_t activation, int32_t out_activation, float* out,
                           float* const edges_out[4], float* global_out, void* workspace,
                           size_t workspace_bytes, gunet_stream_t stream);

/* gunet_unet_forward_f32 for callers that cannot allocate device memory and have no workspace to
 * hand in: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  No
 * counterpart in the reference.  Same arguments, kernels and results as gunet_unet_forward_f32,
 * except that the intermediates live in a buffer of GUNET_STATIC_SCRATCH_BYTES that is part of the
 * library's device image: every GPU the library runs on holds one from the loading of the library
 * to the end of the process.  The call allocates and frees nothing.  A graph and batch whose
 * intermediates do not fit (gunet_unet_workspace_bytes(plan, weights, batch) >
 * GUNET_STATIC_SCRATCH_BYTES) are refused with GUNET_ERR_WORKSPACE before anything is launched.
 * All calls on one GPU share that GPU's buffer: calls that overlap in time on the same GPU (two
 * streams, two host threads) race; calls on one stream, or ordered by events, are safe. */
#define GUNET_STATIC_SCRATCH_BYTES ((size_t)64 << 20)
int gunet_unet_forward_static_f32(const gunet_plan* plan, const gunet_unet_weights* weights, int32_t batch,
                                  const float* x, const float* const edges[4], int32_t edges_shared,
                                  const float* global_in, int32_t activation, int32_t out_activation, float* out,
                                  float* const edges_out[4], float* global_out, gunet_stream_t stream);

/* ------------------------------------------------------------------------
 * 3. Either side of the model -- graph construction, input assembly (models/base_model.py:274-312),
 *    prediction image and metric sums (asynchronous and allocation-free except gunet_image_to_graph)
 * ------------------------------------------------------------------------ */

/* models/base_model.py:274-280 forward_street_encoding: street_map uint8 [height, width] / 255
 * -> Conv2D(32, 3x3, 'SAME', relu, no bias) -> Conv2D(8, 3x3, 'SAME', relu, no bias) -> values at
 * the node pixels node_loc int64 [N,2] (row, col).  cnn1_kernel float32 [3,3,1,32] and cnn2_kernel
 * [3,3,32,8] in the checkpoint's layout (ky, kx, in, out).  out float32 [N,8].  Only the node
 * pixels are evaluated (the reference computes the whole map, then gathers). */
int gunet_street_encoding_f32(const uint8_t* street_map, int32_t height, int32_t width, const float* cnn1_kernel,
                              const float* cnn2_kernel, const int64_t* node_loc, int64_t n_nodes, float* out,
                              gunet_stream_t stream);

/* models/base_model.py:282-285 compute_edge_features: out[e] = [encoding[sender] | encoding[receiver]],
 * encoding float32 [N, width], edge_index int32 [E,2], out float32 [E, 2*width]. */
int gunet_edge_features_f32(const float* encoding, int64_t n_nodes, int32_t width, const int32_t* edge_index,
                            int64_t n_edges, float* out, gunet_stream_t stream);

/* models/base_model.py:300,307-312: x[b] = [seed_nodes[b] | encoding] (float32 [B,N,seed_width+enc_width]) and
 * global[b] = [column sums of x[b] / (300*300) | time_enc[b] | one_hot(weekday[b], 7)]
 * (float32 [B, seed_width+enc_width+time_width+7]).  seed_nodes float32 [B,N,seed_width], encoding float32
 * [N,enc_width] (shared by the samples of a city), time_enc float32 [B,time_width] (the last time step's
 * encoding, x['time_enc'][:, -1]), weekday int32 [B].  The column sums are block partials added in a
 * fixed order (float64 across blocks). */
size_t gunet_model_inputs_workspace_bytes(int64_t n_nodes, int32_t batch, int32_t seed_width, int32_t enc_width);
int gunet_model_inputs_f32(const float* seed_nodes, int32_t seed_width, const float* encoding, int32_t enc_width,
                           const float* time_enc, int32_t time_width, const int32_t* weekday, int64_t n_nodes,
                           int32_t batch, float* x, float* global_out, void* workspace, size_t workspace_bytes,
                           gunet_stream_t stream);

/* data/data_utils.py:84-150 image_to_graph (with mask_invalid_ids :84-93): maps uint8 [8, height, width],
 * non-zero = the pixel has an edge in that direction (N, NE, E, SE, S, SW, W, NW; maps that would
 * leave the image are ignored).  Nodes = pixels with at least one edge, numbered in raster order;
 * edges ordered by (source pixel, direction), self-loops (direction 8) appended per node when
 * add_selfloops; a target pixel that is no node maps to node 0, as in the reference.  Outputs with
 * room for height*width nodes and 9*height*width edges: node_loc int64 [N,2] (row, col), edge_index
 * int32 [E,2], direction int64 [E], node_mask uint8 [height, width] (may be NULL); counts_host =
 * {N, E}.  A plan function (synchronises, allocates temporaries).  The reference divides the maps by
 * their maximum first; for maps that are not two-valued its node set differs (pixels below the
 * maximum are no nodes there). */
int gunet_image_to_graph(const uint8_t* maps, int32_t height, int32_t width, int32_t add_selfloops,
                         int64_t* node_loc, int32_t* edge_index, int64_t* direction, uint8_t* node_mask,
                         int64_t counts_host[2], gunet_stream_t stream);

/* submission.py:69-94 graph_to_image + predict_sample: pred float32 [B, N, pred_width] (pred_width =
 * 12 time steps x channels) -> out uint8 [B, n_horizons, height, width, channels] with
 * out[b, t, row, col, c] = clip(round_half_even(pred[b, n, channels * horizons[t] + c] * scale), 0, 255)
 * at the node pixels node_loc int64 [N,2] and 0 elsewhere.  horizons int32 [n_horizons] on the
 * device (the reference takes [0, 1, 2, 5, 8, 11]). */
int gunet_prediction_image_u8(const float* pred, int64_t n_nodes, int32_t pred_width, int32_t batch,
                              const int64_t* node_loc, int32_t height, int32_t width, const int32_t* horizons,
                              int32_t n_horizons, int32_t channels, float scale, uint8_t* out,
                              gunet_stream_t stream);

/* evaluation.py:173-244 compute_metrics, graph format: the sums every metric there is a ratio of.
 * y_true, y_pred float32 [B, N, width]; sums float64 [B, 2, width]: sums[b, 0, c] = sum_n (t - p)^2,
 * sums[b, 1, c] = sum_n |t - p|, added in a fixed order (float32 inside a block of rows, float64
 * across blocks).  These are also the metric numerators a multi-GPU run all-reduces. */
size_t gunet_error_sums_workspace_bytes(int64_t n_rows, int32_t batch, int32_t width);
int gunet_error_sums_f32(const float* y_true, const float* y_pred, int64_t n_rows, int32_t width, int32_t batch,
                         double* sums, void* workspace, size_t workspace_bytes, gunet_stream_t stream);

/* ------------------------------------------------------------------------
 * 4. Measurement -- no counterpart in the reference
 * ------------------------------------------------------------------------ */

/* Per-kernel-class device timing.  While enabled, every kernel launch of the feature functions
 * is bracketed by two hipEvents on its stream; gunet_profile_collect() waits for them and
 * returns, per class and since the previous collect, the summed milliseconds, the launch count
 * and the algorithmic work of those launches: FLOPs (2 per multiply-add of the products the
 * caller asked for, without tile padding) for the matrix classes, bytes (each operand read once,
 * each result written once) for aggregate and segment_reduce, 0 for the rest.
 * Classes: 0 gemm_tiled (k_gemm_f32), 1 edge_mlp, 2 aggregate, 3 layer_prologue,
 * 4 global_update, 5 segment_reduce, 6 gather_rows, 7 dense_naive, 8 gemm_resident_k128
 * (k_gemm_resident<*, 4>), 9 gemm_resident_k256 (k_gemm_resident<*, 8>), 10 gemm_split
 * (k_pack_weights + k_gemm_split), 11 gemm_resident_k384 (k_gemm_resident<2, 12>),
 * 12 gemm_resident_bf16 (k_gemm_resident<.., BF16 = true>), 13 global_bridge (k_global_bridge: one
 * layer's global update joined with the next layer's prologue, whole-model entry point only). */
#define GUNET_PROF_CLASSES 14
int gunet_profile_enable(int on);
int gunet_profile_collect(double ms_host[GUNET_PROF_CLASSES], int64_t launches_host[GUNET_PROF_CLASSES],
                          double work_host[GUNET_PROF_CLASSES]);
const char* gunet_profile_class_name(int cls);

#ifdef __cplusplus
}
#endif
#endif /* GUNET_HIP_H */
