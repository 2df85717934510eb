// Integer "plan" kernels: prefix sums, stable 4-way partition, receiver-sorted CSR,
// first-occurrence unique (tf.unique semantics), 2x2 cluster ids, pooled edge lists and
// upsampling edge lists.  Everything here is bit-exact integer work; results do not depend
// on atomic arrival order (only integer counts, minima and maxima go through atomics).
#include <climits>
#include <mutex>
#include <vector>

#include "common.h"

namespace gunet {

static thread_local char g_error[768] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------- cache of temporary blocks

namespace {
struct TempBlock {
  void* ptr;
  size_t bytes;
  int device;
  hipStream_t stream;
};
constexpr size_t kTempCacheBytes = size_t{256} << 20;
std::mutex g_temp_lock;
std::vector<TempBlock> g_temp_free;   // idle blocks, oldest first
size_t g_temp_free_bytes = 0;
}  // namespace

hipStream_t& entry_stream() {
  static thread_local hipStream_t stream = nullptr;
  return stream;
}

hipError_t temp_alloc(void** ptr, size_t bytes, size_t* granted) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const hipStream_t stream = entry_stream();
  {
    std::lock_guard<std::mutex> hold(g_temp_lock);
    int best = -1;
    for (int i = 0; i < static_cast<int>(g_temp_free.size()); ++i) {
      const TempBlock& b = g_temp_free[i];
      if (b.device != dev || b.stream != stream || b.bytes < bytes || b.bytes > 4 * bytes + 4096) continue;
      if (best < 0 || b.bytes < g_temp_free[best].bytes) best = i;
    }
    if (best >= 0) {
      *ptr = g_temp_free[best].ptr;
      *granted = g_temp_free[best].bytes;
      g_temp_free_bytes -= g_temp_free[best].bytes;
      g_temp_free.erase(g_temp_free.begin() + best);
      return hipSuccess;
    }
  }
  const size_t rounded = (bytes + 4095) / 4096 * 4096;
  *granted = rounded;
  return hipMalloc(ptr, rounded);
}

void temp_release(void* ptr, size_t granted) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::vector<void*> evicted;
  {
    std::lock_guard<std::mutex> hold(g_temp_lock);
    if (granted > kTempCacheBytes / 2) {
      evicted.push_back(ptr);
    } else {
      g_temp_free.push_back({ptr, granted, dev, entry_stream()});
      g_temp_free_bytes += granted;
      while (g_temp_free_bytes > kTempCacheBytes && !g_temp_free.empty()) {
        if (g_temp_free.front().device != dev) break;   // another device's block: left to temp_cache_clear()
        evicted.push_back(g_temp_free.front().ptr);
        g_temp_free_bytes -= g_temp_free.front().bytes;
        g_temp_free.erase(g_temp_free.begin());
      }
    }
  }
  for (void* p : evicted) (void)hipFree(p);   // synchronises the device: whatever still used the block is done
}

void temp_cache_clear() {
  std::vector<TempBlock> all;
  {
    std::lock_guard<std::mutex> hold(g_temp_lock);
    all.swap(g_temp_free);
    g_temp_free_bytes = 0;
  }
  int current = 0;
  (void)hipGetDevice(&current);
  for (const TempBlock& b : all) {
    if (b.device != current) (void)hipSetDevice(b.device);
    (void)hipFree(b.ptr);
    if (b.device != current) (void)hipSetDevice(current);
  }
}

namespace {

constexpr int kThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kThreads * kScanItems;

inline unsigned grid_for(int64_t n, int threads = kThreads) {
  return static_cast<unsigned>(ceil_div(n > 0 ? n : 1, threads));
}

// ---------------------------------------------------------------- prefix sum

__device__ inline int wave_inclusive_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// Scans one tile of kScanTile values.  `n` counts a virtual trailing zero, so the caller gets
// out[n_real] = total for free.  Writes the tile total to sums[blockIdx.x] when sums != null.
__global__ __launch_bounds__(kThreads) void k_scan_tile(const int32_t* __restrict__ in,
                                                        int32_t* __restrict__ out,
                                                        int32_t* __restrict__ sums, int64_t n_in,
                                                        int64_t n_out) {
  __shared__ int wave_sum[kThreads / 64];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile + threadIdx.x * kScanItems;
  int v[kScanItems];
  int tsum = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t i = base + j;
    v[j] = (i < n_in) ? in[i] : 0;
    tsum += v[j];
  }
  const int incl = wave_inclusive_scan(tsum);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 63) wave_sum[wid] = incl;
  __syncthreads();
  int offset = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; ++w) {
    if (w < wid) offset += wave_sum[w];
    total += wave_sum[w];
  }
  int run = offset + incl - tsum;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t i = base + j;
    if (i < n_out) out[i] = run;
    run += v[j];
  }
  if (sums != nullptr && threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void k_scan_add(int32_t* __restrict__ out, const int32_t* __restrict__ offsets,
                           int64_t n_out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n_out) out[i] += offsets[i / kScanTile];
}

}  // namespace

int scan_exclusive_i32(const int32_t* in, int32_t* out, int64_t n, hipStream_t stream) {
  const int64_t n_out = n + 1;
  const int64_t tiles = ceil_div(n_out, kScanTile);
  if (tiles == 1) {
    k_scan_tile<<<1, kThreads, 0, stream>>>(in, out, nullptr, n, n_out);
    GUNET_LAUNCH_CHECK();
    return GUNET_OK;
  }
  DevBuf sums, sums_scanned;
  GUNET_HIP_TRY(sums.alloc(sizeof(int32_t) * tiles));
  GUNET_HIP_TRY(sums_scanned.alloc(sizeof(int32_t) * (tiles + 1)));
  k_scan_tile<<<static_cast<unsigned>(tiles), kThreads, 0, stream>>>(in, out, sums.as<int32_t>(), n,
                                                                     n_out);
  GUNET_LAUNCH_CHECK();
  GUNET_TRY(scan_exclusive_i32(sums.as<int32_t>(), sums_scanned.as<int32_t>(), tiles, stream));
  k_scan_add<<<grid_for(n_out), kThreads, 0, stream>>>(out, sums_scanned.as<int32_t>(), n_out);
  GUNET_LAUNCH_CHECK();
  // the scratch is freed when this frame unwinds: wait for the kernels that read it
  GUNET_HIP_TRY(hipStreamSynchronize(stream));
  return GUNET_OK;
}

namespace {

// ---------------------------------------------------------------- CSR from keys

__global__ void k_count_keys(const int32_t* __restrict__ keys, int stride, int offset, int64_t n,
                             int32_t n_segments, int32_t* __restrict__ counts,
                             int32_t* __restrict__ err) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t k = keys[i * stride + offset];
  if (k < 0) return;
  if (k >= n_segments) {
    *err = 1;
    return;
  }
  atomicAdd(&counts[k], 1);
}

__global__ void k_fill_slots(const int32_t* __restrict__ keys, int stride, int offset, int64_t n,
                             int32_t n_segments, int32_t* __restrict__ cursor,
                             int32_t* __restrict__ perm) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t k = keys[i * stride + offset];
  if (k < 0 || k >= n_segments) return;
  const int slot = atomicAdd(&cursor[k], 1);
  perm[slot] = static_cast<int32_t>(i);
}

__device__ inline void sift_down(int32_t* a, int start, int end) {
  int root = start;
  while (2 * root + 1 <= end) {
    int child = 2 * root + 1;
    if (child + 1 <= end && a[child] < a[child + 1]) ++child;
    if (a[root] >= a[child]) return;
    const int32_t t = a[root];
    a[root] = a[child];
    a[child] = t;
    root = child;
  }
}

// Slots inside a row were claimed in arrival order; sorting each row ascending restores the
// stable (input) order, so the finished CSR is independent of scheduling.
__global__ void k_sort_rows(const int32_t* __restrict__ rowptr, int32_t* __restrict__ perm,
                            int32_t n_segments) {
  const int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= n_segments) return;
  const int b = rowptr[row], len = rowptr[row + 1] - b;
  if (len < 2) return;
  int32_t* a = perm + b;
  if (len <= 16) {
    for (int i = 1; i < len; ++i) {
      const int32_t v = a[i];
      int j = i - 1;
      while (j >= 0 && a[j] > v) {
        a[j + 1] = a[j];
        --j;
      }
      a[j + 1] = v;
    }
    return;
  }
  for (int s = (len - 2) / 2; s >= 0; --s) sift_down(a, s, len - 1);
  for (int e = len - 1; e > 0; --e) {
    const int32_t t = a[0];
    a[0] = a[e];
    a[e] = t;
    sift_down(a, 0, e - 1);
  }
}

}  // namespace

int csr_from_keys(const int32_t* keys, int key_stride, int key_offset, int64_t n,
                  int32_t n_segments, int32_t* rowptr, int32_t* perm, hipStream_t stream) {
  DevBuf counts, cursor, err;
  GUNET_HIP_TRY(counts.alloc(sizeof(int32_t) * (static_cast<size_t>(n_segments) + 1)));
  GUNET_HIP_TRY(cursor.alloc(sizeof(int32_t) * (static_cast<size_t>(n_segments) + 1)));
  GUNET_HIP_TRY(err.alloc(sizeof(int32_t)));
  GUNET_HIP_TRY(hipMemsetAsync(counts.as<void>(), 0,
                               sizeof(int32_t) * (static_cast<size_t>(n_segments) + 1), stream));
  GUNET_HIP_TRY(hipMemsetAsync(err.as<void>(), 0, sizeof(int32_t), stream));
  if (n > 0) {
    k_count_keys<<<grid_for(n), kThreads, 0, stream>>>(keys, key_stride, key_offset, n, n_segments,
                                                       counts.as<int32_t>(), err.as<int32_t>());
    GUNET_LAUNCH_CHECK();
  }
  GUNET_TRY(scan_exclusive_i32(counts.as<int32_t>(), rowptr, n_segments, stream));
  int32_t err_host = 0;
  GUNET_HIP_TRY(hipMemcpyAsync(&err_host, err.as<void>(), sizeof(int32_t), hipMemcpyDeviceToHost,
                               stream));
  GUNET_HIP_TRY(hipStreamSynchronize(stream));
  GUNET_REQUIRE(err_host == 0, "csr_from_keys: a key is >= n_segments (%d)", n_segments);
  if (n > 0) {
    GUNET_HIP_TRY(hipMemcpyAsync(cursor.as<void>(), rowptr,
                                 sizeof(int32_t) * static_cast<size_t>(n_segments),
                                 hipMemcpyDeviceToDevice, stream));
    k_fill_slots<<<grid_for(n), kThreads, 0, stream>>>(keys, key_stride, key_offset, n, n_segments,
                                                       cursor.as<int32_t>(), perm);
    GUNET_LAUNCH_CHECK();
    k_sort_rows<<<grid_for(n_segments), kThreads, 0, stream>>>(rowptr, perm, n_segments);
    GUNET_LAUNCH_CHECK();
  }
  GUNET_HIP_TRY(hipStreamSynchronize(stream));
  return GUNET_OK;
}

namespace {

// ---------------------------------------------------------------- first-occurrence unique

constexpr unsigned long long kEmptyKey = ~0ull;

__device__ inline uint32_t mix64(unsigned long long x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return static_cast<uint32_t>(x);
}

// Open-addressing insert.  The table ends up holding, per distinct key, the smallest element
// index carrying it: a minimum, so the result does not depend on the order of arrival.
__global__ void k_hash_insert(const int64_t* __restrict__ keys, int64_t n,
                              unsigned long long* __restrict__ tkeys, int32_t* __restrict__ tmin,
                              uint32_t mask, int32_t* __restrict__ slot,
                              int32_t* __restrict__ err) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t k = keys[i];
  if (k < 0) {
    *err = 1;
    slot[i] = 0;
    return;
  }
  const unsigned long long key = static_cast<unsigned long long>(k);
  uint32_t h = mix64(key) & mask;
  for (;;) {
    const unsigned long long old = atomicCAS(&tkeys[h], kEmptyKey, key);
    if (old == kEmptyKey || old == key) break;
    h = (h + 1) & mask;
  }
  atomicMin(&tmin[h], static_cast<int32_t>(i));
  slot[i] = static_cast<int32_t>(h);
}

__global__ void k_mark_first(const int32_t* __restrict__ slot, const int32_t* __restrict__ tmin,
                             int64_t n, int32_t* __restrict__ flag) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (tmin[slot[i]] == static_cast<int32_t>(i)) ? 1 : 0;
}

__global__ void k_assign_labels(const int64_t* __restrict__ keys, const int32_t* __restrict__ slot,
                                const int32_t* __restrict__ tmin, const int32_t* __restrict__ rank,
                                const int32_t* __restrict__ flag, int64_t n,
                                int32_t* __restrict__ label, int64_t* __restrict__ uniq) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  label[i] = rank[tmin[slot[i]]];
  if (uniq != nullptr && flag[i]) uniq[rank[i]] = keys[i];
}

// tf.unique: label[i] = rank (in order of first appearance) of keys[i]; uniq = the distinct keys
// in that order (optional).  Keys must be non-negative.
int unique_first(const int64_t* keys, int64_t n, int32_t* label, int64_t* uniq,
                 int64_t* count_host, hipStream_t stream) {
  *count_host = 0;
  if (n == 0) return GUNET_OK;
  GUNET_REQUIRE(n < (int64_t{1} << 30), "unique_first: %lld elements exceed the supported 2^30",
                static_cast<long long>(n));
  uint64_t cap = 64;
  while (cap < static_cast<uint64_t>(2 * n)) cap <<= 1;
  DevBuf tkeys, tmin, slot, flag, rank, err;
  GUNET_HIP_TRY(tkeys.alloc(sizeof(unsigned long long) * cap));
  GUNET_HIP_TRY(tmin.alloc(sizeof(int32_t) * cap));
  GUNET_HIP_TRY(slot.alloc(sizeof(int32_t) * n));
  GUNET_HIP_TRY(flag.alloc(sizeof(int32_t) * n));
  GUNET_HIP_TRY(rank.alloc(sizeof(int32_t) * (n + 1)));
  GUNET_HIP_TRY(err.alloc(sizeof(int32_t)));
  GUNET_HIP_TRY(hipMemsetAsync(tkeys.as<void>(), 0xFF, sizeof(unsigned long long) * cap, stream));
  GUNET_HIP_TRY(hipMemsetAsync(tmin.as<void>(), 0x7F, sizeof(int32_t) * cap, stream));
  GUNET_HIP_TRY(hipMemsetAsync(err.as<void>(), 0, sizeof(int32_t), stream));
  k_hash_insert<<<grid_for(n), kThreads, 0, stream>>>(keys, n, tkeys.as<unsigned long long>(),
                                                      tmin.as<int32_t>(),
                                                      static_cast<uint32_t>(cap - 1),
                                                      slot.as<int32_t>(), err.as<int32_t>());
  GUNET_LAUNCH_CHECK();
  k_mark_first<<<grid_for(n), kThreads, 0, stream>>>(slot.as<int32_t>(), tmin.as<int32_t>(), n,
                                                     flag.as<int32_t>());
  GUNET_LAUNCH_CHECK();
  GUNET_TRY(scan_exclusive_i32(flag.as<int32_t>(), rank.as<int32_t>(), n, stream));
  k_assign_labels<<<grid_for(n), kThreads, 0, stream>>>(keys, slot.as<int32_t>(),
                                                        tmin.as<int32_t>(), rank.as<int32_t>(),
                                                        flag.as<int32_t>(), n, label, uniq);
  GUNET_LAUNCH_CHECK();
  int32_t count = 0, err_host = 0;
  GUNET_HIP_TRY(hipMemcpyAsync(&count, rank.as<int32_t>() + n, sizeof(int32_t),
                               hipMemcpyDeviceToHost, stream));
  GUNET_HIP_TRY(hipMemcpyAsync(&err_host, err.as<void>(), sizeof(int32_t), hipMemcpyDeviceToHost,
                               stream));
  GUNET_HIP_TRY(hipStreamSynchronize(stream));
  GUNET_REQUIRE(err_host == 0, "unique_first: negative key");
  *count_host = count;
  return GUNET_OK;
}

// ---------------------------------------------------------------- stable 4-way partition

__global__ void k_direction_mask(const int64_t* __restrict__ direction, int64_t n,
                                 uint8_t* __restrict__ mask, int32_t* __restrict__ err) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // bit q set = member of quadrant q; quadrants are {N,NE,E} {E,SE,S} {S,SW,W} {W,NW,N}
  constexpr uint8_t table[9] = {0x9, 0x1, 0x3, 0x2, 0x6, 0x4, 0xC, 0x8, 0x0};
  const int64_t d = direction[i];
  if (d < 0 || d > 8) {
    *err = 1;
    mask[i] = 0;
    return;
  }
  mask[i] = table[d];
}

__global__ void k_mask_flags(const uint8_t* __restrict__ mask, int64_t n, int32_t* __restrict__ f0,
                             int32_t* __restrict__ f1, int32_t* __restrict__ f2,
                             int32_t* __restrict__ f3) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t m = mask[i];
  f0[i] = m & 1;
  f1[i] = (m >> 1) & 1;
  f2[i] = (m >> 2) & 1;
  f3[i] = (m >> 3) & 1;
}

struct QuadPtrs {
  int32_t* out[4];
  const int32_t* pos[4];
};

// payload row i = rows[i] when rows != null, else (first[i], second_base + i)
__global__ void k_partition_scatter(const uint8_t* __restrict__ mask, int64_t n,
                                    const int32_t* __restrict__ rows,
                                    const int32_t* __restrict__ first, int32_t second_base,
                                    QuadPtrs p) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t m = mask[i];
  if (m == 0) return;
  int2 row;
  if (rows != nullptr) {
    row = reinterpret_cast<const int2*>(rows)[i];
  } else {
    row.x = first[i];
    row.y = second_base + static_cast<int32_t>(i);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if ((m >> q) & 1) reinterpret_cast<int2*>(p.out[q])[p.pos[q][i]] = row;
  }
}

int partition4(const uint8_t* mask, int64_t n, const int32_t* rows, const int32_t* first,
               int32_t second_base, int32_t* const out[4], int64_t counts_host[4],
               hipStream_t stream) {
  for (int q = 0; q < 4; ++q) counts_host[q] = 0;
  if (n == 0) return GUNET_OK;
  DevBuf flags, pos;
  GUNET_HIP_TRY(flags.alloc(sizeof(int32_t) * 4 * n));
  GUNET_HIP_TRY(pos.alloc(sizeof(int32_t) * 4 * (n + 1)));
  int32_t* f = flags.as<int32_t>();
  int32_t* ps = pos.as<int32_t>();
  k_mask_flags<<<grid_for(n), kThreads, 0, stream>>>(mask, n, f, f + n, f + 2 * n, f + 3 * n);
  GUNET_LAUNCH_CHECK();
  QuadPtrs p;
  for (int q = 0; q < 4; ++q) {
    GUNET_TRY(scan_exclusive_i32(f + q * n, ps + q * (n + 1), n, stream));
    p.out[q] = out[q];
    p.pos[q] = ps + q * (n + 1);
  }
  k_partition_scatter<<<grid_for(n), kThreads, 0, stream>>>(mask, n, rows, first, second_base, p);
  GUNET_LAUNCH_CHECK();
  int32_t totals[4];
  for (int q = 0; q < 4; ++q) {
    GUNET_HIP_TRY(hipMemcpyAsync(&totals[q], ps + q * (n + 1) + n, sizeof(int32_t),
                                 hipMemcpyDeviceToHost, stream));
  }
  GUNET_HIP_TRY(hipStreamSynchronize(stream));
  for (int q = 0; q < 4; ++q) counts_host[q] = totals[q];
  return GUNET_OK;
}

// ---------------------------------------------------------------- 2x2 clusters

__global__ __launch_bounds__(kThreads) void k_max_row(const int64_t* __restrict__ node_loc,
                                                      int64_t n, unsigned long long* max_row,
                                                      int32_t* __restrict__ err) {
  __shared__ unsigned long long wave_max[kThreads / 64];
  unsigned long long m = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t r = node_loc[2 * i], c = node_loc[2 * i + 1];
    if (r < 0 || c < 0) {
      *err = 1;
    } else if (static_cast<unsigned long long>(r) > m) {
      m = static_cast<unsigned long long>(r);
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const unsigned long long o = __shfl_xor(m, d);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / 64; ++w) m = wave_max[w] > m ? wave_max[w] : m;
    atomicMax(max_row, m);
  }
}

__global__ void k_pos_id(const int64_t* __restrict__ node_loc, int64_t n,
                         const unsigned long long* __restrict__ max_row,
                         int64_t* __restrict__ pos_id) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t m = static_cast<int64_t>(*max_row) + 1;
  pos_id[i] = (node_loc[2 * i] >> 1) * m + (node_loc[2 * i + 1] >> 1);
}

__global__ void k_fill_i64(int64_t* __restrict__ p, int64_t n, int64_t v) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void k_pooled_loc(const int64_t* __restrict__ node_loc,
                             const int32_t* __restrict__ cluster, int64_t n,
                             int64_t* __restrict__ pooled) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t c = cluster[i];
  __hip_atomic_fetch_min(&pooled[2 * c], node_loc[2 * i] >> 1, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_min(&pooled[2 * c + 1], node_loc[2 * i + 1] >> 1, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
}

int cluster_ids(const int64_t* node_loc, int64_t n, int32_t* cluster, int64_t* pooled_loc,
                int64_t* n_pooled_host, hipStream_t stream) {
  *n_pooled_host = 0;
  if (n == 0) return GUNET_OK;
  DevBuf max_row, pos_id, err;
  GUNET_HIP_TRY(max_row.alloc(sizeof(unsigned long long)));
  GUNET_HIP_TRY(pos_id.alloc(sizeof(int64_t) * n));
  GUNET_HIP_TRY(err.alloc(sizeof(int32_t)));
  GUNET_HIP_TRY(hipMemsetAsync(max_row.as<void>(), 0, sizeof(unsigned long long), stream));
  GUNET_HIP_TRY(hipMemsetAsync(err.as<void>(), 0, sizeof(int32_t), stream));
  const unsigned blocks = static_cast<unsigned>(ceil_div(n, kThreads) < 1024 ? ceil_div(n, kThreads) : 1024);
  k_max_row<<<blocks, kThreads, 0, stream>>>(node_loc, n, max_row.as<unsigned long long>(),
                                             err.as<int32_t>());
  GUNET_LAUNCH_CHECK();
  k_pos_id<<<grid_for(n), kThreads, 0, stream>>>(node_loc, n, max_row.as<unsigned long long>(),
                                                 pos_id.as<int64_t>());
  GUNET_LAUNCH_CHECK();
  int32_t err_host = 0;
  GUNET_HIP_TRY(hipMemcpyAsync(&err_host, err.as<void>(), sizeof(int32_t), hipMemcpyDeviceToHost,
                               stream));
  GUNET_HIP_TRY(hipStreamSynchronize(stream));
  GUNET_REQUIRE(err_host == 0, "cluster_ids: node_loc holds a negative coordinate");
  GUNET_TRY(unique_first(pos_id.as<int64_t>(), n, cluster, nullptr, n_pooled_host, stream));
  if (pooled_loc != nullptr) {
    const int64_t p = *n_pooled_host;
    k_fill_i64<<<grid_for(2 * p), kThreads, 0, stream>>>(pooled_loc, 2 * p, INT64_MAX);
    GUNET_LAUNCH_CHECK();
    k_pooled_loc<<<grid_for(n), kThreads, 0, stream>>>(node_loc, cluster, n, pooled_loc);
    GUNET_LAUNCH_CHECK();
    GUNET_HIP_TRY(hipStreamSynchronize(stream));
  }
  return GUNET_OK;
}

// ---------------------------------------------------------------- pooled edge lists

__global__ void k_edge_keys(const int32_t* __restrict__ edge_index, int64_t n_edges,
                            const int32_t* __restrict__ cluster, int64_t n_pre,
                            int64_t* __restrict__ keys, int32_t* __restrict__ err) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const int2 sr = reinterpret_cast<const int2*>(edge_index)[e];
  if (sr.x < 0 || sr.x >= n_pre || sr.y < 0 || sr.y >= n_pre) {
    *err = 1;
    keys[e] = 0;
    return;
  }
  keys[e] = static_cast<int64_t>(cluster[sr.x]) * n_pre + static_cast<int64_t>(cluster[sr.y]);
}

__global__ void k_keep_flags(const int64_t* __restrict__ uniq, int64_t n_uniq, int64_t n_pre,
                             int32_t* __restrict__ keep) {
  const int64_t u = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u < n_uniq) keep[u] = (uniq[u] / n_pre != uniq[u] % n_pre) ? 1 : 0;
}

__global__ void k_emit_edges(const int64_t* __restrict__ uniq, int64_t n_uniq, int64_t n_pre,
                             const int32_t* __restrict__ keep, const int32_t* __restrict__ newpos,
                             int32_t* __restrict__ new_edges) {
  const int64_t u = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= n_uniq || !keep[u]) return;
  int2 row;
  row.x = static_cast<int32_t>(uniq[u] / n_pre);
  row.y = static_cast<int32_t>(uniq[u] % n_pre);
  reinterpret_cast<int2*>(new_edges)[newpos[u]] = row;
}

__global__ void k_dup_map(const int32_t* __restrict__ idx, int64_t n_edges,
                          const int32_t* __restrict__ keep, const int32_t* __restrict__ newpos,
                          int32_t* __restrict__ dup_map) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const int32_t u = idx[e];
  dup_map[e] = keep[u] ? newpos[u] : -1;
}

// ---------------------------------------------------------------- upsampling edge lists

__global__ void k_child_mask(const int64_t* __restrict__ coarse_loc, int64_t n_coarse,
                             const int64_t* __restrict__ fine_loc, int64_t n_fine,
                             const int32_t* __restrict__ parent, uint8_t* __restrict__ mask,
                             int32_t* __restrict__ err) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_fine) return;
  const int64_t p = parent[i];
  if (p >= n_coarse) {
    *err = 1;
    mask[i] = 0;
    return;
  }
  // the reference compares fine - (2*coarse + 0.5) with (+-0.5, +-0.5) in float32, which for
  // image coordinates (< 2^22) is this integer test on fine - 2*coarse
  const int64_t dr = fine_loc[2 * i] - 2 * coarse_loc[2 * p];
  const int64_t dc = fine_loc[2 * i + 1] - 2 * coarse_loc[2 * p + 1];
  uint8_t m = 0;
  if (dr == 0 && dc == 1) m = 1;
  else if (dr == 1 && dc == 1) m = 2;
  else if (dr == 1 && dc == 0) m = 4;
  else if (dr == 0 && dc == 0) m = 8;
  mask[i] = m;
}

int check_err_flag(const DevBuf& err, hipStream_t stream, const char* what) {
  int32_t err_host = 0;
  GUNET_HIP_TRY(hipMemcpyAsync(&err_host, err.as<void>(), sizeof(int32_t), hipMemcpyDeviceToHost,
                               stream));
  GUNET_HIP_TRY(hipStreamSynchronize(stream));
  GUNET_REQUIRE(err_host == 0, "%s", what);
  return GUNET_OK;
}

// ---------------------------------------------------------------- in-edge table

struct CsrPtrs {
  const int32_t* rowptr[4];
  const int32_t* perm[4];
};

// table[v][q][j] = position (in quadrant q's edge list) of the j-th incoming edge of node v, or -1
__global__ void k_in_table(CsrPtrs p, int n_nodes, int32_t* __restrict__ table, int32_t* __restrict__ max_deg) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<int64_t>(n_nodes) * 4) return;
  const int v = static_cast<int>(idx >> 2), q = static_cast<int>(idx & 3);
  const int32_t* rowptr = q == 0 ? p.rowptr[0] : q == 1 ? p.rowptr[1] : q == 2 ? p.rowptr[2] : p.rowptr[3];
  const int32_t* perm = q == 0 ? p.perm[0] : q == 1 ? p.perm[1] : q == 2 ? p.perm[2] : p.perm[3];
  const int beg = rowptr[v], deg = rowptr[v + 1] - beg;
  if (deg > 0) atomicMax(max_deg, deg);
#pragma unroll
  for (int j = 0; j < 3; ++j) table[idx * 3 + j] = j < deg ? perm[beg + j] : -1;
}

}  // namespace
}  // namespace gunet

using namespace gunet;

extern "C" {

const char* gunet_last_error(void) { return g_error; }

int gunet_version(void) { return 1; }

int gunet_quadrant_partition(const int32_t* edge_index, const int64_t* direction, int64_t n_edges,
                             int32_t* const out_edges[4], int64_t counts_host[4],
                             gunet_stream_t stream_) {
  StreamDeviceGuard device_guard(as_stream(stream_));
  hipStream_t stream = as_stream(stream_);
  GUNET_REQUIRE(counts_host != nullptr && out_edges != nullptr, "quadrant_partition: null output");
  GUNET_REQUIRE(n_edges >= 0 && n_edges < INT32_MAX, "quadrant_partition: bad n_edges");
  for (int q = 0; q < 4; ++q) counts_host[q] = 0;
  if (n_edges == 0) return GUNET_OK;
  GUNET_REQUIRE(edge_index != nullptr && direction != nullptr, "quadrant_partition: null input");
  for (int q = 0; q < 4; ++q)
    GUNET_REQUIRE(out_edges[q] != nullptr, "quadrant_partition: out_edges[%d] is null", q);
  DevBuf mask, err;
  GUNET_HIP_TRY(mask.alloc(n_edges));
  GUNET_HIP_TRY(err.alloc(sizeof(int32_t)));
  GUNET_HIP_TRY(hipMemsetAsync(err.as<void>(), 0, sizeof(int32_t), stream));
  k_direction_mask<<<grid_for(n_edges), kThreads, 0, stream>>>(direction, n_edges,
                                                               mask.as<uint8_t>(),
                                                               err.as<int32_t>());
  GUNET_LAUNCH_CHECK();
  GUNET_TRY(check_err_flag(err, stream, "quadrant_partition: direction outside 0..8"));
  return partition4(mask.as<uint8_t>(), n_edges, edge_index, nullptr, 0, out_edges, counts_host,
                    stream);
}

int gunet_csr_build(const int32_t* edge_index, int64_t n_edges, int32_t n_nodes, int32_t* rowptr,
                    int32_t* perm, gunet_stream_t stream) {
  StreamDeviceGuard device_guard(as_stream(stream));
  GUNET_REQUIRE(n_edges >= 0 && n_edges < INT32_MAX && n_nodes >= 0, "csr_build: bad sizes");
  GUNET_REQUIRE(rowptr != nullptr && (n_edges == 0 || (edge_index != nullptr && perm != nullptr)),
                "csr_build: null pointer");
  return csr_from_keys(edge_index, 2, 1, n_edges, n_nodes, rowptr, perm, as_stream(stream));
}

int gunet_csr_from_keys(const int32_t* keys, int64_t n, int32_t n_segments, int32_t* rowptr,
                        int32_t* perm, gunet_stream_t stream) {
  StreamDeviceGuard device_guard(as_stream(stream));
  GUNET_REQUIRE(n >= 0 && n < INT32_MAX && n_segments >= 0, "csr_from_keys: bad sizes");
  GUNET_REQUIRE(rowptr != nullptr && (n == 0 || (keys != nullptr && perm != nullptr)),
                "csr_from_keys: null pointer");
  return csr_from_keys(keys, 1, 0, n, n_segments, rowptr, perm, as_stream(stream));
}

int gunet_cluster_ids(const int64_t* node_loc, int64_t n_nodes, int32_t* cluster,
                      int64_t* pooled_loc, int64_t* n_pooled_host, gunet_stream_t stream) {
  StreamDeviceGuard device_guard(as_stream(stream));
  GUNET_REQUIRE(n_pooled_host != nullptr, "cluster_ids: n_pooled_host is null");
  GUNET_REQUIRE(n_nodes >= 0 && n_nodes < INT32_MAX, "cluster_ids: bad n_nodes");
  GUNET_REQUIRE(n_nodes == 0 || (node_loc != nullptr && cluster != nullptr),
                "cluster_ids: null pointer");
  return cluster_ids(node_loc, n_nodes, cluster, pooled_loc, n_pooled_host, as_stream(stream));
}

int gunet_pool_edges(const int32_t* edge_index, int64_t n_edges, const int32_t* cluster,
                     int64_t n_pre, int32_t* new_edges, int32_t* dup_map, int64_t* n_new_host,
                     gunet_stream_t stream_) {
  StreamDeviceGuard device_guard(as_stream(stream_));
  hipStream_t stream = as_stream(stream_);
  GUNET_REQUIRE(n_new_host != nullptr, "pool_edges: n_new_host is null");
  *n_new_host = 0;
  GUNET_REQUIRE(n_edges >= 0 && n_edges < INT32_MAX && n_pre > 0 && n_pre < INT32_MAX,
                "pool_edges: bad sizes");
  if (n_edges == 0) return GUNET_OK;
  GUNET_REQUIRE(edge_index && cluster && new_edges && dup_map, "pool_edges: null pointer");
  DevBuf keys, idx, uniq, keep, newpos, err;
  GUNET_HIP_TRY(keys.alloc(sizeof(int64_t) * n_edges));
  GUNET_HIP_TRY(idx.alloc(sizeof(int32_t) * n_edges));
  GUNET_HIP_TRY(uniq.alloc(sizeof(int64_t) * n_edges));
  GUNET_HIP_TRY(err.alloc(sizeof(int32_t)));
  GUNET_HIP_TRY(hipMemsetAsync(err.as<void>(), 0, sizeof(int32_t), stream));
  k_edge_keys<<<grid_for(n_edges), kThreads, 0, stream>>>(edge_index, n_edges, cluster, n_pre,
                                                          keys.as<int64_t>(), err.as<int32_t>());
  GUNET_LAUNCH_CHECK();
  GUNET_TRY(check_err_flag(err, stream, "pool_edges: edge endpoint outside [0, n_pre)"));
  int64_t n_uniq = 0;
  GUNET_TRY(unique_first(keys.as<int64_t>(), n_edges, idx.as<int32_t>(), uniq.as<int64_t>(),
                         &n_uniq, stream));
  GUNET_HIP_TRY(keep.alloc(sizeof(int32_t) * n_uniq));
  GUNET_HIP_TRY(newpos.alloc(sizeof(int32_t) * (n_uniq + 1)));
  k_keep_flags<<<grid_for(n_uniq), kThreads, 0, stream>>>(uniq.as<int64_t>(), n_uniq, n_pre,
                                                          keep.as<int32_t>());
  GUNET_LAUNCH_CHECK();
  GUNET_TRY(scan_exclusive_i32(keep.as<int32_t>(), newpos.as<int32_t>(), n_uniq, stream));
  k_emit_edges<<<grid_for(n_uniq), kThreads, 0, stream>>>(uniq.as<int64_t>(), n_uniq, n_pre,
                                                          keep.as<int32_t>(),
                                                          newpos.as<int32_t>(), new_edges);
  GUNET_LAUNCH_CHECK();
  k_dup_map<<<grid_for(n_edges), kThreads, 0, stream>>>(idx.as<int32_t>(), n_edges,
                                                        keep.as<int32_t>(), newpos.as<int32_t>(),
                                                        dup_map);
  GUNET_LAUNCH_CHECK();
  int32_t n_new = 0;
  GUNET_HIP_TRY(hipMemcpyAsync(&n_new, newpos.as<int32_t>() + n_uniq, sizeof(int32_t),
                               hipMemcpyDeviceToHost, stream));
  GUNET_HIP_TRY(hipStreamSynchronize(stream));
  *n_new_host = n_new;
  return GUNET_OK;
}

int gunet_upsample_edges(const int64_t* coarse_loc, int64_t n_coarse, const int64_t* fine_loc,
                         int64_t n_fine, int32_t* parent, int32_t* const out_edges[4],
                         int64_t counts_host[4], gunet_stream_t stream_) {
  StreamDeviceGuard device_guard(as_stream(stream_));
  hipStream_t stream = as_stream(stream_);
  GUNET_REQUIRE(counts_host != nullptr && out_edges != nullptr, "upsample_edges: null output");
  for (int q = 0; q < 4; ++q) counts_host[q] = 0;
  GUNET_REQUIRE(n_coarse >= 0 && n_fine >= 0 && n_coarse + n_fine < INT32_MAX,
                "upsample_edges: bad sizes");
  if (n_fine == 0) return GUNET_OK;
  GUNET_REQUIRE(coarse_loc && fine_loc && parent, "upsample_edges: null pointer");
  for (int q = 0; q < 4; ++q)
    GUNET_REQUIRE(out_edges[q] != nullptr, "upsample_edges: out_edges[%d] is null", q);
  int64_t n_parents = 0;
  GUNET_TRY(cluster_ids(fine_loc, n_fine, parent, nullptr, &n_parents, stream));
  GUNET_REQUIRE(n_parents <= n_coarse,
                "upsample_edges: the fine graph pools to %lld nodes but the coarse graph has %lld",
                static_cast<long long>(n_parents), static_cast<long long>(n_coarse));
  DevBuf mask, err;
  GUNET_HIP_TRY(mask.alloc(n_fine));
  GUNET_HIP_TRY(err.alloc(sizeof(int32_t)));
  GUNET_HIP_TRY(hipMemsetAsync(err.as<void>(), 0, sizeof(int32_t), stream));
  k_child_mask<<<grid_for(n_fine), kThreads, 0, stream>>>(coarse_loc, n_coarse, fine_loc, n_fine,
                                                          parent, mask.as<uint8_t>(),
                                                          err.as<int32_t>());
  GUNET_LAUNCH_CHECK();
  GUNET_TRY(check_err_flag(err, stream, "upsample_edges: parent id outside the coarse graph"));
  return partition4(mask.as<uint8_t>(), n_fine, nullptr, parent, static_cast<int32_t>(n_coarse),
                    out_edges, counts_host, stream);
}

int gunet_in_edge_table(const gunet_quadgraph* graph, int32_t* table, int32_t* max_in_degree_host,
                        gunet_stream_t stream_) {
  StreamDeviceGuard device_guard(as_stream(stream_));
  hipStream_t stream = as_stream(stream_);
  GUNET_REQUIRE(graph && max_in_degree_host, "in_edge_table: null argument");
  *max_in_degree_host = 0;
  GUNET_REQUIRE(graph->n_nodes >= 0, "in_edge_table: negative node count");
  if (graph->n_nodes == 0) return GUNET_OK;
  GUNET_REQUIRE(table, "in_edge_table: null table");
  CsrPtrs p;
  for (int q = 0; q < 4; ++q) {
    GUNET_REQUIRE(graph->rowptr[q] && (graph->n_edges[q] == 0 || graph->perm[q]), "in_edge_table: null CSR %d", q);
    p.rowptr[q] = graph->rowptr[q];
    p.perm[q] = graph->perm[q];
  }
  DevBuf md;
  GUNET_HIP_TRY(md.alloc(sizeof(int32_t)));
  GUNET_HIP_TRY(hipMemsetAsync(md.as<int32_t>(), 0, sizeof(int32_t), stream));
  const int64_t work = static_cast<int64_t>(graph->n_nodes) * 4;
  k_in_table<<<grid_for(work), kThreads, 0, stream>>>(p, graph->n_nodes, table, md.as<int32_t>());
  GUNET_LAUNCH_CHECK();
  GUNET_HIP_TRY(hipMemcpyAsync(max_in_degree_host, md.as<int32_t>(), sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  GUNET_HIP_TRY(hipStreamSynchronize(stream));
  return GUNET_OK;
}

}  // extern "C"
