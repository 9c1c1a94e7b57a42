// Launch-plan replay in C: a recorded network pass (bcp_amd/plan.py) is a constant list of C-ABI calls; bcp_replay_run walks it with
// one call from the host language instead of one foreign-function call per launch.  Also the stream fork / join the backward pass
// uses for its weight-gradient side stream, so that the whole pass -- ordering included -- lives in the list.
// No reference counterpart: the reference's host path is PyTorch's eager dispatch.
#include <mutex>
#include <cstring>
#include <vector>

#include "common.h"

namespace bcp {

union Slot {
  void* p;
  long long l;
  int i;
  float f;
  double d;
  unsigned long long u;
  size_t z;
};
static_assert(sizeof(Slot) == 8, "argument images are 8 bytes");

typedef int (*Caller)(void* fn, const Slot* a);

#define BCP_SHAPE(name, types, args) \
  static int call_##name(void* fn, const Slot* a) { return reinterpret_cast<int(*) types>(fn) args; }
#include "replay_shapes.inc"
#undef BCP_SHAPE

struct ShapeRow { const char* name; Caller call; };
static const ShapeRow kShapes[] = {
#define BCP_SHAPE(name, types, args) {#name, &call_##name},
#include "replay_shapes.inc"
#undef BCP_SHAPE
};

constexpr int kMaxArgs = 28;
struct Entry {
  Caller call;
  void* fn;
  Slot a[kMaxArgs];
};
struct Replay { std::vector<Entry> e; };

// events for bcp_stream_wait_stream: a wait takes the state of the event's LAST record at the time of the call, so a small ring
// can be re-recorded freely
constexpr int kEvRing = 64, kMaxDev = 16;
struct EvRing { hipEvent_t ev[kEvRing]; int next = 0; bool ready = false; };
static EvRing g_ring[kMaxDev];       // one per device (bcp_stream_wait_stream)
static std::mutex g_ev_mu;

}  // namespace bcp

extern "C" int bcp_replay_create(void** h) {
  BCP_REQUIRE(h, "bcp_replay_create: null");
  *h = new bcp::Replay();
  return BCP_OK;
}

extern "C" int bcp_replay_add(void* h, void* fn, const char* shape, const void* slots, int nargs) {
  BCP_REQUIRE(h && fn && shape && (slots || nargs == 0), "bcp_replay_add: null");
  BCP_REQUIRE(nargs >= 0 && nargs <= bcp::kMaxArgs && (int)strlen(shape) == nargs, "bcp_replay_add: %d arguments for shape \"%s\"", nargs, shape);
  bcp::Entry e{};
  for (const bcp::ShapeRow& r : bcp::kShapes)
    if (!strcmp(r.name, shape)) { e.call = r.call; break; }
  BCP_REQUIRE(e.call, "bcp_replay_add: no caller for the argument shape \"%s\" (regenerate csrc/replay_shapes.inc: tools/gen_replay_shapes.py)", shape);
  e.fn = fn;
  if (nargs) memcpy(e.a, slots, (size_t)nargs * sizeof(bcp::Slot));
  static_cast<bcp::Replay*>(h)->e.push_back(e);
  return BCP_OK;
}

extern "C" int bcp_replay_count(void* h) { return h ? (int)static_cast<bcp::Replay*>(h)->e.size() : 0; }

extern "C" int bcp_replay_run(void* h) {
  BCP_REQUIRE(h, "bcp_replay_run: null");
  for (const bcp::Entry& e : static_cast<bcp::Replay*>(h)->e)
    if (const int rc = e.call(e.fn, e.a)) return rc;        // the failing entry point has set the error text
  return BCP_OK;
}

// Measurement twin of bcp_replay_run (round 5, bench.py's per-op table): the same walk with HIP events recorded around chosen entries ON THE
// ENTRY'S OWN STREAM -- ev_before[i] in front of entry i, ev_after[i] behind it (NULL: none; streams[i] = the stream that entry launches on).
// The host stays as far ahead of the GPU as in the timed region, so an (event before op, event behind op) pair brackets the op's kernels as
// they run INSIDE the step -- beside the other streams' work -- which the eager profile steps of rounds 1-4 could not (their host was the
// bottleneck: every bracket held the Python between the record and the launch, 8 % on a 55 us kernel, 3 ms where the allocator stalled).
extern "C" int bcp_replay_run_timed(void* h, void* const* ev_before, void* const* ev_after, void* const* streams) {
  BCP_REQUIRE(h && ev_before && ev_after && streams, "bcp_replay_run_timed: null");
  const std::vector<bcp::Entry>& e = static_cast<bcp::Replay*>(h)->e;
  for (size_t i = 0; i < e.size(); ++i) {
    if (ev_before[i] && hipEventRecord((hipEvent_t)ev_before[i], (hipStream_t)streams[i]) != hipSuccess) { bcp::set_error("bcp_replay_run_timed: hipEventRecord failed"); return BCP_ELAUNCH; }
    if (const int rc = e[i].call(e[i].fn, e[i].a)) return rc;
    if (ev_after[i] && hipEventRecord((hipEvent_t)ev_after[i], (hipStream_t)streams[i]) != hipSuccess) { bcp::set_error("bcp_replay_run_timed: hipEventRecord failed"); return BCP_ELAUNCH; }
  }
  return BCP_OK;
}

extern "C" int bcp_replay_destroy(void* h) {
  delete static_cast<bcp::Replay*>(h);
  return BCP_OK;
}

// `waiter` does not run past this point before everything enqueued on `signaller` so far has finished.
// One ring of events PER DEVICE (an event records only on streams of the device it was created on), created lazily for the device
// that is current at the call; ring creation and the index are under a mutex -- the autograd thread replays recorded backward passes
// while the main thread enqueues the next forward.  (An event may be re-recorded while an earlier wait on it is still pending: the
// wait captured the earlier record.)
extern "C" int bcp_stream_wait_stream(void* waiter, void* signaller) {
  if (waiter == signaller) return BCP_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= bcp::kMaxDev) { bcp::set_error("bcp_stream_wait_stream: no current device"); return BCP_ELAUNCH; }
  hipEvent_t ev;
  {
    std::lock_guard<std::mutex> lock(bcp::g_ev_mu);
    bcp::EvRing& r = bcp::g_ring[dev];
    if (!r.ready) {
      for (int i = 0; i < bcp::kEvRing; ++i)
        if (hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming) != hipSuccess) { bcp::set_error("hipEventCreateWithFlags failed"); return BCP_ELAUNCH; }
      r.ready = true;
    }
    ev = r.ev[r.next];
    r.next = (r.next + 1) % bcp::kEvRing;
  }
  hipError_t e = hipEventRecord(ev, (hipStream_t)signaller);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
  if (e != hipSuccess) { bcp::set_error("bcp_stream_wait_stream: %s", hipGetErrorString(e)); return BCP_ELAUNCH; }
  return BCP_OK;
}
