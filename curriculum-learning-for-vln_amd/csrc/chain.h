// Chained step launches: the dependent stages of a decoder step (skinny GEMMs, attention rows, elementwise stages) as ONE
// kernel launch.  OPT-IN (vln_set_chain(1)): measured SLOWER than one launch per stage on MI355X.  Kept as a correct,
// bit-identical A/B (tests/test_hip_modules.py::test_chained_step_kernel_equals_stage_launches) and as the record of why:
//
//   EnvDrop IL iteration, B = 64, bf16 (profiles/round2_notes.md):   one launch per stage 1.80 ms   chained 2.64 ms
//   (14 chained launches of ~124 us replace 126 launches that add up to ~65 us per step and direction).  With both fences
//   removed (wrong results, timing probe) 2.12 ms; with per-stage counters instead of per-workgroup flags 3.16 ms (~900
//   agent-scope atomic adds on a few addresses serialise at ~90 ns each).  A hand-over THROUGH MEMORY between workgroups
//   on different XCDs -- write-through + flag store, poll, invalidate, cold dependent load -- costs 6-8 us here, no less
//   than the kernel boundary it replaces (4-6 us), and the agent-scope acquires (buffer_inv sc1) of ~900 workgroups empty
//   the XCDs' L2s over and over (0.4 ms of the 2.64).  What makes the persistent recurrence's hand-off cheap (2.2 us per
//   step: data-tagged granules polled by the consuming wave itself, no flag round trip, no fence) does not carry over to
//   stages whose operands are ordinary matrices written by generic bodies.
//
// A decoder step at B = 64 is a chain of ~9 dependent launches each way, each worth 4-10 us although it moves about a
// megabyte: kernel start / end (cache invalidate + write-back, dispatch, cold first loads) is what they cost, not their
// work.  Here the stages of a step become workgroup ranges of one grid, in stage order.  A workgroup of stage s
//   * requests what does not depend on the previous stage (its weight fragments / its context block) first,
//   * waits until every workgroup of the stages it depends on has set its completion flag (agent-scope acquire),
//   * runs the stage's ordinary workgroup body (the same __device__ function the stand-alone kernel runs: results are
//     bit-identical), and
//   * releases its stores (agent scope) and sets its own flag.
// Forward progress needs no co-residency: workgroups are dispatched in index order, a workgroup only ever waits for
// stages with LOWER indices, so the lowest unfinished workgroup can always run.  Waits are bounded (a timeout raises the
// library's sticky error word, vln_persistent_check) so a broken assumption is an error, not a hang.
// Flags hold launch epochs and are never reset (chain.hip), so a captured launch replays as it is.
//
// Host side: launchers that know a chained form call chain_add(); while a ChainScope is open on the calling thread the
// stage is recorded instead of launched.  Every other launch in the library goes through VLN_LAUNCH / launch_timed, which
// first submit what has been recorded (chain_flush), so stream order is preserved whatever mix of launchers a step uses.
#pragma once
#include "vln_internal.h"

namespace vln {

enum ChainKind {
  CK_NONE = 0,
  CK_GEMM_NT,          // gemm_nt_body<TW, 2, true, 1>           grid (nb, nsplit, mb)
  CK_ATTN_FWD_0, CK_ATTN_FWD_1, CK_ATTN_FWD_2, CK_ATTN_FWD_3,     // attn_fused_body<TW, cfg, false>   grid (B)
  CK_ATTN_BWD_0, CK_ATTN_BWD_1, CK_ATTN_BWD_2, CK_ATTN_BWD_3,     // attn_fused_body<TW, cfg, true>
  CK_LSTM_PW_FWD,      // grid (blocks), gy = iterations per block
  CK_LSTM_PW_BWD,
  CK_REDUCE_EPI,
  CK_TANH_DROP_BWD,
  CK_PREP,             // envdrop_prep_body
  CK_PREP_BWD,
  CK_GATHER_STEP,      // gather_step_row<TW>: one table row per virtual block
};

constexpr int kChainMaxStages = 14;
constexpr int kChainArgBytes = 3200;
constexpr int kChainThreads = 512;
constexpr int kDepNone = -1;      // no wait
constexpr int kDepPrev = -2;      // the stage recorded just before

struct ChainStageDesc {
  int kind, first, nwg, gx, gy, gz, arg_off;
  int dep_main;      // stage whose completion the body's dependent part waits for (kDepNone: none)
  int dep_pre;       // stage the body's independent loads wait for (kDepNone: they go first thing)
  int early;         // 1: the body may request its stage-independent operand before dep_main completes
  int pad[2];
};
struct ChainArgs {
  unsigned* flags;          // one completion word per workgroup of the launch (chain.hip: epochs, never reset)
  unsigned* sticky;         // host-mapped timeout word of the device
  int nstages, pad;
  ChainStageDesc st[kChainMaxStages];
  alignas(16) unsigned char args[kChainArgBytes];
};
static_assert(sizeof(ChainArgs) <= 4096, "kernel argument block");

// ---- host side ----------------------------------------------------------------------------------------------------------
// Records the chained launches of the calling thread between construction and finish() / destruction.
class ChainScope {
 public:
  ChainScope(hipStream_t st, bool enable);
  ~ChainScope();
  int finish();            // submits what is recorded; returns the launch status
 private:
  bool owner_;
};
// false: not recording / kind not chainable here -> the caller launches its stand-alone kernel (VLN_LAUNCH flushes first).
// dtype: W_F32 / W_BF16 of the streamed operand the body is instantiated for, -1 if the body has no such operand.
bool chain_add(hipStream_t st, int kind, int gx, int gy, int gz, const void* args, int nbytes, double algo_bytes, int dtype);
// dependency overrides for the NEXT chain_add (defaults: dep_main = previous stage, dep_pre = none, early = 0)
void chain_next(int dep_main, int dep_pre, int early);
int chain_last();            // index of the stage recorded last (-1: none / not recording)
int chain_flush();           // submit the recorded stages now (no-op when nothing is recorded)
bool chain_recording();
int chain_prime();             // allocate what a chained launch needs (call outside stream capture)

extern int g_chain_mode;     // vln_set_chain: 0 stand-alone launches (default), 1 chained steps

}  // namespace vln
