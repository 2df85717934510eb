// Optional per-kernel-class device timing: when enabled, every launch site brackets its kernel
// with two hipEvents on the launch stream; gunet_profile_collect() turns the pairs into
// milliseconds per class.  Off by default (the brackets cost a few microseconds each).
#include <mutex>
#include <vector>

#include "common.h"

namespace gunet {
namespace {

struct Bracket {
  int cls;
  double work;
  hipEvent_t start, stop;
};

std::mutex g_mu;
bool g_enabled = false;
std::vector<Bracket> g_open;           // recorded, not yet collected
std::vector<hipEvent_t> g_free_events; // reusable events

const char* const kNames[kProfCount] = {"gemm_tiled",    "edge_mlp",       "aggregate",   "layer_prologue",
                                        "global_update", "segment_reduce", "gather_rows", "dense_naive",
                                        "gemm_resident_k128", "gemm_resident_k256", "gemm_split",
                                        "gemm_resident_k384", "gemm_resident_bf16", "global_bridge"};

hipEvent_t take_event() {
  if (!g_free_events.empty()) {
    hipEvent_t e = g_free_events.back();
    g_free_events.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

}  // namespace

ProfScope::ProfScope(int cls, hipStream_t stream, double work) : cls_(cls), stream_(stream), work_(work) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_enabled) return;
  start_ = take_event();
  stop_ = take_event();
  if (start_ == nullptr || stop_ == nullptr) {
    start_ = stop_ = nullptr;
    return;
  }
  (void)hipEventRecord(static_cast<hipEvent_t>(start_), stream_);
}

ProfScope::~ProfScope() {
  if (start_ == nullptr) return;
  (void)hipEventRecord(static_cast<hipEvent_t>(stop_), stream_);
  std::lock_guard<std::mutex> lock(g_mu);
  g_open.push_back({cls_, work_, static_cast<hipEvent_t>(start_), static_cast<hipEvent_t>(stop_)});
}

}  // namespace gunet

using namespace gunet;

extern "C" {

int gunet_profile_enable(int on) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_enabled = on != 0;
  return GUNET_OK;
}

int gunet_profile_collect(double ms_host[GUNET_PROF_CLASSES], int64_t launches_host[GUNET_PROF_CLASSES],
                          double work_host[GUNET_PROF_CLASSES]) {
  GUNET_REQUIRE(ms_host != nullptr && launches_host != nullptr && work_host != nullptr,
                "profile_collect: null output");
  std::vector<Bracket> open;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    open.swap(g_open);
  }
  for (int c = 0; c < GUNET_PROF_CLASSES; ++c) {
    ms_host[c] = 0.0;
    launches_host[c] = 0;
    work_host[c] = 0.0;
  }
  int rc = GUNET_OK;
  for (const Bracket& b : open) {
    float ms = 0.0f;
    hipError_t e = hipEventSynchronize(b.stop);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, b.start, b.stop);
    if (e != hipSuccess) {
      set_error("profile_collect: %s", hipGetErrorString(e));
      rc = GUNET_ERR_HIP;
    } else if (b.cls >= 0 && b.cls < GUNET_PROF_CLASSES) {
      ms_host[b.cls] += ms;
      launches_host[b.cls] += 1;
      work_host[b.cls] += b.work;
    }
  }
  std::lock_guard<std::mutex> lock(g_mu);
  for (const Bracket& b : open) {
    g_free_events.push_back(b.start);
    g_free_events.push_back(b.stop);
  }
  return rc;
}

const char* gunet_profile_class_name(int cls) {
  return (cls >= 0 && cls < kProfCount) ? kNames[cls] : "";
}

}  // extern "C"
