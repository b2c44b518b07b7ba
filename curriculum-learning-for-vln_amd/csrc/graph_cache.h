// hipGraph memoisation of fixed launch chains.
//
// The hot path is a long chain of small kernels (one per LSTM time step, ~20 per decoder step) whose host
// launch cost (~5-7 us each on this stack) exceeds their GPU time.  Each chain is a pure function of its
// argument block (device pointers, sizes), so the first call with a given argument block is stream-captured
// into a hipGraph and later calls with the SAME block replay it with one hipGraphLaunch (~10-16 us).  PyTorch's
// caching allocator hands back the same addresses every training iteration, so the steady state always hits.
// Anything that varies per call but is not an address (dropout offsets) must live in device memory, not in the
// key.  If a chain never hits (2 x cap captures without one replay) capturing is PAUSED for that chain for a while and it
// runs as plain launches -- results are identical either way; vln_graph_stats()[2] counts the pauses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

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
  // cap: graphs kept (LRU).  A rollout keeps 2 arena generations x steps x (forward + backward) blocks alive: the reference's
  // sampled rollouts run up to MAX_EPISODE_LEN = 35 steps beside a 7-step teacher rollout (2 x 42 x 2 = 168), hence 512.
  explicit GraphCache(size_t cap = 512) : cap_(cap) {}

  // key: raw bytes of the argument block.  issue(stream) enqueues the chain and returns a VLN status.
  template <typename F>
  int run(hipStream_t st, const void* key, size_t key_bytes, F&& issue) {
    if (!g_graphs_enabled || g_prof_mask) return issue(st);
    std::lock_guard<std::mutex> lock(mu_);
    ++calls_;
    if (calls_ < paused_until_) return issue(st);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return issue(st);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) { (void)hipGetLastError(); return issue(st); }
    std::string k(reinterpret_cast<const char*>(&dev), sizeof(dev));   // an argument block is only meaningful on its device
    k.append(reinterpret_cast<const char*>(key), key_bytes);
    auto it = map_.find(k);
    if (it != map_.end()) {
      lru_.splice(lru_.begin(), lru_, it->second.second);
      captures_since_hit_ = 0;
      pause_ = kPause;
      ++g_graph_stats[0];
      if (hipGraphLaunch(it->second.first, st) == hipSuccess) return 0;
      (void)hipGetLastError();
      return issue(st);
    }
    // Giving up needs EVIDENCE that argument blocks never come back: two arena generations of any rollout length are all
    // misses by construction, so "a few misses in a row" proves nothing (the old rule switched a 13-step rollout's graphs
    // off for good before their first possible hit).  Only after 2 x cap captures without a single replay -- every entry
    // of the LRU was evicted unused, twice over -- does the chain run un-captured, and only for a while: the pause ends
    // after a number of calls (kPause, x4 per failed retry) and the cache tries again (a caller may have switched to address-stable buffers meanwhile).
    if (++captures_since_hit_ > 2 * cap_) {
      paused_until_ = calls_ + pause_;
      if (pause_ < (1ull << 22)) pause_ *= 4;        // back off: a chain that keeps failing the retry pays for it ever less often
      captures_since_hit_ = 0;
      ++g_graph_stats[2];
      if (!warned_) {
        warned_ = true;
        fprintf(stderr, "[vln] hipGraph replay paused for a launch chain: %zu captures without one replay (argument blocks do not "
                        "repeat -- per-iteration buffers need stable addresses, see ops.RolloutArena); retrying later\n", 2 * cap_);
      }
      return issue(st);
    }
    // Capture on a PRIVATE stream: the caller's stream is usually PyTorch's current stream = the legacy default
    // stream, which cannot be captured.  The chain is only recorded there (nothing executes); the instantiated
    // graph is then launched into the caller's stream.
    hipStream_t& cst = cs_[dev];
    if (cst == nullptr && hipStreamCreateWithFlags(&cst, hipStreamNonBlocking) != hipSuccess) {
      (void)hipGetLastError();
      cst = nullptr;
      return issue(st);
    }
    if (hipStreamBeginCapture(cst, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();
      return issue(st);
    }
    ++g_graph_stats[1];
    int rc = issue(cst);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(cst, &g);
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
  static constexpr int kMaxDev = 16;
  static constexpr unsigned long long kPause = 4096;
  std::mutex mu_;
  size_t cap_;
  hipStream_t cs_[kMaxDev] = {};   // capture-only streams, one per device
  unsigned long long calls_ = 0, paused_until_ = 0, pause_ = kPause;
  size_t captures_since_hit_ = 0;
  bool warned_ = false;
  std::list<std::string> lru_;
  std::unordered_map<std::string, std::pair<hipGraphExec_t, std::list<std::string>::iterator>> map_;
};

}  // namespace vln
