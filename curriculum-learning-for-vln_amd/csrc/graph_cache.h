// hipGraph memoisation of fixed launch chains.
//
// The hot path is a long chain of small kernels (one per LSTM time step, ~20 per decoder step) whose host
// launch cost (~5-7 us each on this stack) exceeds their GPU time.  Each chain is a pure function of its
// argument block (device pointers, sizes), so the first call with a given argument block is stream-captured
// into a hipGraph and later calls with the SAME block replay it with one hipGraphLaunch (~10-16 us).  PyTorch's
// caching allocator hands back the same addresses every training iteration, so the steady state always hits.
// Anything that varies per call but is not an address (dropout offsets) must live in device memory, not in the
// key.  If a chain keeps missing (addresses never repeat) graphs are switched off for that chain and it
// falls back to plain launches -- results are identical either way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <list>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>

namespace vln {

extern int g_graphs_enabled;   // vln_set_graphs(); default on
extern unsigned g_prof_mask;
extern long long g_graph_stats[3];   // replays, captures, chains switched off (vln_graph_stats)

class GraphCache {
 public:
  explicit GraphCache(size_t cap = 48) : cap_(cap) {}

  // key: raw bytes of the argument block.  issue(stream) enqueues the chain and returns a VLN status.
  template <typename F>
  int run(hipStream_t st, const void* key, size_t key_bytes, F&& issue) {
    if (!g_graphs_enabled || g_prof_mask || disabled_) return issue(st);
    std::lock_guard<std::mutex> lock(mu_);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return issue(st);
    std::string k(reinterpret_cast<const char*>(key), key_bytes);
    auto it = map_.find(k);
    if (it != map_.end()) {
      lru_.splice(lru_.begin(), lru_, it->second.second);
      misses_in_a_row_ = 0;
      ++g_graph_stats[0];
      if (hipGraphLaunch(it->second.first, st) == hipSuccess) return 0;
      (void)hipGetLastError();
      return issue(st);
    }
    if (++misses_in_a_row_ > 24) {   // addresses never repeat: stop paying for captures
      disabled_ = true;
      ++g_graph_stats[2];
      return issue(st);
    }
    // Capture on a PRIVATE stream: the caller's stream is usually PyTorch's current stream = the legacy default
    // stream, which cannot be captured.  The chain is only recorded there (nothing executes); the instantiated
    // graph is then launched into the caller's stream.
    if (cs_ == nullptr && hipStreamCreateWithFlags(&cs_, hipStreamNonBlocking) != hipSuccess) {
      (void)hipGetLastError();
      cs_ = nullptr;
      return issue(st);
    }
    if (hipStreamBeginCapture(cs_, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();
      return issue(st);
    }
    ++g_graph_stats[1];
    int rc = issue(cs_);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(cs_, &g);
    if (rc != 0 || e != hipSuccess || g == nullptr) {
      if (g) (void)hipGraphDestroy(g);
      (void)hipGetLastError();
      return rc != 0 ? rc : issue(st);
    }
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess || ex == nullptr) {
      (void)hipGetLastError();
      return issue(st);
    }
    if (map_.size() >= cap_) {
      const std::string& old = lru_.back();
      auto jt = map_.find(old);
      if (jt != map_.end()) {
        (void)hipGraphExecDestroy(jt->second.first);
        map_.erase(jt);
      }
      lru_.pop_back();
    }
    lru_.push_front(k);
    map_.emplace(std::move(k), std::make_pair(ex, lru_.begin()));
    if (hipGraphLaunch(ex, st) != hipSuccess) {
      (void)hipGetLastError();
      return issue(st);
    }
    return 0;
  }

 private:
  std::mutex mu_;
  size_t cap_;
  hipStream_t cs_ = nullptr;   // capture-only stream
  bool disabled_ = false;
  int misses_in_a_row_ = 0;
  std::list<std::string> lru_;
  std::unordered_map<std::string, std::pair<hipGraphExec_t, std::list<std::string>::iterator>> map_;
};

}  // namespace vln
