// Shared helpers of libgunet_hip.so: error reporting, launch checks, RAII device scratch.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "gunet_hip.h"

namespace gunet {

void set_error(const char* fmt, ...);

#define GUNET_HIP_TRY(expr)                                                              \
  do {                                                                                   \
    hipError_t err_ = (expr);                                                            \
    if (err_ != hipSuccess) {                                                            \
      ::gunet::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                   \
                         hipGetErrorString(err_));                                       \
      return GUNET_ERR_HIP;                                                              \
    }                                                                                    \
  } while (0)

#define GUNET_LAUNCH_CHECK() GUNET_HIP_TRY(hipGetLastError())

#define GUNET_REQUIRE(cond, ...)                                                         \
  do {                                                                                   \
    if (!(cond)) {                                                                       \
      ::gunet::set_error(__VA_ARGS__);                                                   \
      return GUNET_ERR_INVALID;                                                          \
    }                                                                                    \
  } while (0)

#define GUNET_TRY(expr)                                                                  \
  do {                                                                                   \
    int rc_ = (expr);                                                                    \
    if (rc_ != GUNET_OK) return rc_;                                                     \
  } while (0)

// Temporary device memory of the plan functions comes from a small per-process cache of blocks
// instead of a hipMalloc / hipFree pair per temporary (a hipFree synchronises the whole device; a
// plan build makes some hundred temporaries).  A block is handed out again only to a request made
// for the same device and stream it was last used on -- work queued on one stream runs in order,
// so the new user cannot overtake the old one -- and the cache holds at most kTempCacheBytes.
// The stream is the one of the innermost entry point running on this thread (StreamDeviceGuard).
hipError_t temp_alloc(void** ptr, size_t bytes, size_t* granted);
void temp_release(void* ptr, size_t granted);
void temp_cache_clear();   // frees every cached block of every device
hipStream_t& entry_stream();   // thread-local

// Temporary device allocation owned by a plan function (back to the cache on scope exit).
class DevBuf {
 public:
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() {
    if (ptr_) temp_release(ptr_, granted_);
  }
  hipError_t alloc(size_t bytes) {
    if (ptr_) {
      temp_release(ptr_, granted_);
      ptr_ = nullptr;
    }
    if (bytes == 0) bytes = 16;
    return temp_alloc(&ptr_, bytes, &granted_);
  }
  template <typename T>
  T* as() const {
    return static_cast<T*>(ptr_);
  }

 private:
  void* ptr_ = nullptr;
  size_t granted_ = 0;
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Activation codes (GUNET_ACT_*): linear and relu have tuned kernels; the others run on the generic ones.
inline bool act_valid(int act) { return act >= GUNET_ACT_LINEAR && act <= GUNET_ACT_HARD_SIGMOID; }
inline bool act_tuned(int act) { return act == GUNET_ACT_LINEAR || act == GUNET_ACT_RELU; }

// Any activation by its code (the generic kernels; the tuned ones know linear and relu only).
__device__ inline float activate_any(float v, int act) {
  switch (act) {
    case GUNET_ACT_RELU: return fmaxf(v, 0.0f);
    case GUNET_ACT_TANH: return tanhf(v);
    case GUNET_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case GUNET_ACT_ELU: return v > 0.0f ? v : expm1f(v);
    case GUNET_ACT_SOFTPLUS: return fmaxf(v, 0.0f) + log1pf(expf(-fabsf(v)));
    case GUNET_ACT_SELU: return 1.0507009873554805f * (v > 0.0f ? v : 1.6732632423543772f * expm1f(v));
    case GUNET_ACT_SWISH: return v / (1.0f + expf(-v));
    case GUNET_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    case GUNET_ACT_SOFTSIGN: return v / (1.0f + fabsf(v));
    case GUNET_ACT_EXPONENTIAL: return expf(v);
    case GUNET_ACT_HARD_SIGMOID: return fminf(fmaxf(0.2f * v + 0.5f, 0.0f), 1.0f);
    default: return v;
  }
}

// Kernel classes of the optional event profiler (gunet_profile_*).
enum ProfClass {
  kProfGemmTiled = 0,
  kProfEdgeMlp,
  kProfAggregate,
  kProfPrologue,
  kProfGlobalUpdate,
  kProfSegment,
  kProfGather,
  kProfDenseNaive,
  kProfGemmRes128,
  kProfGemmRes256,
  kProfGemmSplit,
  kProfGemmRes384,
  kProfGemmResBf16,
  kProfBridge,
  kProfCount
};
static_assert(kProfCount == GUNET_PROF_CLASSES, "profiler class table out of step with the header");

// Brackets the launches issued during its lifetime with two events when profiling is enabled.
// `work` is the algorithmic work of those launches: FLOPs for the matrix kernels, bytes for the
// gather / scatter / reduce kernels.
class ProfScope {
 public:
  ProfScope(int cls, hipStream_t stream, double work = 0.0);
  ~ProfScope();
  ProfScope(const ProfScope&) = delete;
  ProfScope& operator=(const ProfScope&) = delete;

 private:
  int cls_;
  hipStream_t stream_;
  double work_;
  void* start_ = nullptr;
  void* stop_ = nullptr;
};

inline hipStream_t as_stream(gunet_stream_t s) { return static_cast<hipStream_t>(s); }

// Buffer addressing for the feature kernels: the base address lives in scalar registers (a 4-dword
// resource), a lane adds a 32-bit byte offset, the instruction a scalar offset and a 12-bit
// immediate -- no 64-bit address arithmetic in vector registers.  Accesses at or past `bytes` read
// as zero and are not written.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t BufferRsrc;
__device__ __forceinline__ BufferRsrc buffer_of(const void* p, uint32_t bytes = 0xFFFFFFFFu) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, static_cast<int>(bytes), 0x00020000);
}
__device__ __forceinline__ float4 buffer_load_f4(BufferRsrc rs, uint32_t lane_off, uint32_t uniform_off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, uniform_off, 0));
}
__device__ __forceinline__ int2 buffer_load_i2(BufferRsrc rs, uint32_t lane_off, uint32_t uniform_off) {
  return __builtin_bit_cast(int2, __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off, uniform_off, 0));
}
__device__ __forceinline__ void buffer_store_f4(BufferRsrc rs, uint32_t lane_off, uint32_t uniform_off, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, lane_off, uniform_off, 0);
}
__device__ __forceinline__ void buffer_store_f1(BufferRsrc rs, uint32_t lane_off, uint32_t uniform_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), rs, lane_off, uniform_off, 0);
}

// A 16-byte LDS load the compiler may neither merge with its neighbours nor cut into pieces (one
// ds_read_b128).  `p` must be 16-byte aligned.  Used where a wave picks up row-wise what its lanes
// stored column-wise: left to itself hipcc (ROCm 7.2) cut two adjacent float4 loads that feed a bf16
// packing into ds_read_b96 + ds_read2_b32 + ds_read_b32 pieces, and on gfx950 the first two words of
// such a ds_read_b96 then came back as zeros once in some tens of launches (always for lanes 12-15 of
// a 16-lane group, second half of a tile); waiting for the stores first did not change that, single
// ds_read_b128 loads did.
typedef float f32x4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lds_load_f4(const float* p) {
  const f32x4n v = *reinterpret_cast<const volatile f32x4n*>(p);
  return make_float4(v[0], v[1], v[2], v[3]);
}

// The ReLU of the matrix kernels' epilogues as one instruction: the maximum with 0 (a NaN comes
// out as 0, as from fmaxf; TensorFlow's ReLU hands a NaN on).  fmaxf() would first canonicalise the
// accumulator with a second instruction.
__device__ __forceinline__ float max_bare(float v, float floor) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(floor));
  return r;
}

// Every entry point that takes a stream runs on that stream's device: when it is not the current
// one (one process driving several GPUs), the guard makes it current for the length of the call, so
// launches, temporary allocations and plan ownership all land on the GPU the stream belongs to.
// The null stream belongs to whatever device is current.
class StreamDeviceGuard {
 public:
  explicit StreamDeviceGuard(hipStream_t stream) : outer_stream_(entry_stream()) {
    entry_stream() = stream;
    int current = 0, wanted = 0;
    if (stream != nullptr && hipGetDevice(&current) == hipSuccess &&
        hipStreamGetDevice(stream, &wanted) == hipSuccess && wanted != current &&
        hipSetDevice(wanted) == hipSuccess)
      previous_ = current;
  }
  ~StreamDeviceGuard() {
    entry_stream() = outer_stream_;
    if (previous_ >= 0) (void)hipSetDevice(previous_);
  }
  StreamDeviceGuard(const StreamDeviceGuard&) = delete;
  StreamDeviceGuard& operator=(const StreamDeviceGuard&) = delete;

 private:
  int previous_ = -1;
  hipStream_t outer_stream_;
};

// Exclusive prefix sum of `n` int32 values into out[0..n] (n + 1 entries; out[n] = total).
// `in` and `out` may not alias.  Allocates its own block-sum scratch.
int scan_exclusive_i32(const int32_t* in, int32_t* out, int64_t n, hipStream_t stream);

// Segments from keys: rowptr[n_segments + 1], perm = members of each segment ascending.
// keys[i * key_stride + key_offset] is the key of element i; negative keys are skipped.
int csr_from_keys(const int32_t* keys, int key_stride, int key_offset, int64_t n,
                  int32_t n_segments, int32_t* rowptr, int32_t* perm, hipStream_t stream);

// The GN layer entry points with their first kernel (the prologue: folded biases from the input
// global) and / or their last one (the global update) left out, for a caller that runs consecutive
// layers and joins each layer's global update with the next layer's prologue (gn_bridge).
// n_partials_out receives the number of column-sum rows per sample the node MLP wrote, which the
// global update (or gn_bridge) of that layer needs.  flags = 0 is the public entry point.
constexpr int kGnSkipPrologue = 1;
constexpr int kGnSkipGlobalUpdate = 2;
constexpr int kGnEdgesShared = 4;   // edges[q] is one [E_q, f_e] tensor for all samples
// bf16 storage (the whole-model forward in GUNET_MATMUL_BF16 mode; 128-wide layers on graphs with an
// in-edge table): x, x_skip, x_out, edges and edges_out hold bf16 although the parameters are typed
// float*, and so do the layer's node embeddings in the workspace.
constexpr int kGnStore16 = 8;
constexpr int kGnEdgesIn32 = 16;    // with kGnStore16: edges[q] is float32 after all (the model's 16-wide input features)
// with kGnStore16, gn_layer_run's edges_out32[q] (may be null) receive a float32 copy of the messages
int gn_layer_run(const gunet_quadgraph* graph, const gunet_gn_weights* w, int32_t batch, const float* x,
                 const float* x_skip, int32_t c_skip, const float* const edges[4], const float* global_in,
                 int32_t activation, float* x_out, float* const edges_out[4], float* global_out, void* workspace,
                 size_t workspace_bytes, int flags, int* n_partials_out, hipStream_t stream,
                 float* const edges_out32[4] = nullptr);
int gn_upsample_run(const gunet_quadgraph* graph, int32_t n_coarse, const gunet_gn_weights* w, int32_t batch,
                    const float* x, const float* global_in, int32_t activation, float* x_out, float* global_out,
                    void* workspace, size_t workspace_bytes, int flags, int* n_partials_out, hipStream_t stream);
// Global update of layer `cur` (input global global_in, result to global_out) and prologue of layer
// `next`, whose input global is that result, or [result | extra] (then also written to cat_out),
// in one launch.  Both layers use the same workspace.
int gn_bridge(const gunet_quadgraph* cur_graph, const gunet_gn_weights* cur, int n_partials, const float* global_in,
              float* global_out, const float* extra, int extra_width, float* cat_out, const gunet_quadgraph* next_graph,
              const gunet_gn_weights* next, int next_fold_receiver, int32_t batch, int32_t activation, void* workspace,
              size_t workspace_bytes, hipStream_t stream);

// Up to five row-wise segment maxima (src[i] [B, n_src[i], width[i]] -> out[i] [B, n_seg[i], width[i]]
// over the CSR rowptr[i] / perm[i]) in one launch: the node and edge features of a 2x2 pooling step.
// bf16 = true: src and out hold bf16 (the pointers are really __bf16*), widths count bf16 columns.
int pool_max_multi(int n_jobs, const float* const* src, const int64_t* n_src, const int* width,
                   const int32_t* const* rowptr, const int32_t* const* perm, const int64_t* n_seg, float* const* out,
                   int32_t batch, hipStream_t stream, bool bf16 = false);

}  // namespace gunet
