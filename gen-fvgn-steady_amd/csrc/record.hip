// Native command list (include/gfv.h gfv_record_*): the launches of one recorded step, replayed from C.  See gfv_launch.h.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/gfv.h"
#include "gfv_common.h"
#include "gfv_launch.h"

struct GfvRecorder {
  struct Cmd {
    hipError_t (*run)(const void*, hipStream_t);
    hipStream_t st;
    size_t off;
  };
  std::vector<Cmd> cmds;
  std::vector<char> arena;   // argument blobs, 16-byte aligned
};

namespace {
thread_local GfvRecorder* t_active = nullptr;
std::mutex g_mu;
std::unordered_map<int64_t, GfvRecorder*> g_lists;
int64_t g_next = 1;

// events for stream-to-stream edges: a wait refers to the event's record at the time of the call, so a small ring is enough.
// One ring per DEVICE (an event belongs to the device that was current when it was created: a process that drives streams of
// several devices gets a ring on each)
constexpr int N_EVENTS = 64;
constexpr int MAX_DEVICES = 16;
struct EventRing {
  hipEvent_t ev[N_EVENTS];
  bool ready = false;
  int next = 0;
};
EventRing g_rings[MAX_DEVICES];
hipEvent_t next_event() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  EventRing& r = g_rings[dev];
  if (!r.ready) {
    for (int i = 0; i < N_EVENTS; ++i)
      if (hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    r.ready = true;
  }
  hipEvent_t e = r.ev[r.next];
  r.next = (r.next + 1) % N_EVENTS;
  return e;
}
struct WaitBlob {
  hipStream_t waited;
};
hipError_t run_wait(const void* p, hipStream_t waiter) {
  const WaitBlob& B = *static_cast<const WaitBlob*>(p);
  hipEvent_t e = next_event();
  if (!e) return hipErrorInvalidResourceHandle;   // (a dropped edge would be a data race, not an error: fail the replay)
  hipError_t rc = hipEventRecord(e, B.waited);
  if (rc != hipSuccess) return rc;
  return hipStreamWaitEvent(waiter, e, 0);
}
struct MemsetBlob {
  void* dst;
  int value;
  size_t bytes;
};
hipError_t run_memset(const void* p, hipStream_t st) {
  const MemsetBlob& B = *static_cast<const MemsetBlob*>(p);
  return hipMemsetAsync(B.dst, B.value, B.bytes, st);
}
}  // namespace

GfvRecorder* gfv_rec_active() { return t_active; }
void gfv_rec_push(GfvRecorder* r, hipError_t (*run)(const void*, hipStream_t), const void* blob, size_t bytes, hipStream_t st) {
  const size_t off = (r->arena.size() + 15) & ~(size_t)15;
  r->arena.resize(off + bytes);
  memcpy(r->arena.data() + off, blob, bytes);
  r->cmds.push_back({run, st, off});
}
hipError_t gfv_memset_rec(void* dst, int value, size_t bytes, hipStream_t st) {
  const MemsetBlob b{dst, value, bytes};
  const hipError_t rc = run_memset(&b, st);
  if (GfvRecorder* r = gfv_rec_active()) gfv_rec_push(r, run_memset, &b, sizeof(b), st);
  return rc;
}

extern "C" int gfv_record_begin(void) {
  if (t_active) return GFV_ERR_ARG;
  t_active = new GfvRecorder();
  return GFV_OK;
}
extern "C" int gfv_record_count(void) { return t_active ? (int)t_active->cmds.size() : -1; }
extern "C" int64_t gfv_record_end(void) {
  if (!t_active) return 0;
  GfvRecorder* r = t_active;
  t_active = nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  const int64_t h = g_next++;
  g_lists[h] = r;
  return h;
}
extern "C" int gfv_record_length(int64_t handle) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_lists.find(handle);
  return it == g_lists.end() ? -1 : (int)it->second->cmds.size();
}
extern "C" int gfv_record_replay(int64_t handle, int32_t first, int32_t last) {
  GfvRecorder* r = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_lists.find(handle);
    if (it == g_lists.end()) return GFV_ERR_ARG;
    r = it->second;
  }
  const int n = (int)r->cmds.size();
  if (first < 0 || last > n || first > last) return GFV_ERR_ARG;
  if (t_active) return GFV_ERR_ARG;   // (a replay inside a recording would be recorded again)
  const char* base = r->arena.data();
  for (int i = first; i < last; ++i)
    if (r->cmds[i].run(base + r->cmds[i].off, r->cmds[i].st) != hipSuccess) return GFV_ERR_LAUNCH;   // stop at the first failure
  return GFV_OK;
}
extern "C" int gfv_record_free(int64_t handle) {
  GfvRecorder* r = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_lists.find(handle);
    if (it == g_lists.end()) return GFV_ERR_ARG;
    r = it->second;
    g_lists.erase(it);
  }
  delete r;
  return GFV_OK;
}
// `waiter` waits for everything submitted to `waited` so far (event record + stream wait); recorded like a launch
extern "C" int gfv_stream_wait(void* waiter, void* waited) {
  const WaitBlob b{(hipStream_t)waited};
  hipEvent_t e = next_event();
  if (!e) return GFV_ERR_LAUNCH;
  if (hipEventRecord(e, (hipStream_t)waited) != hipSuccess || hipStreamWaitEvent((hipStream_t)waiter, e, 0) != hipSuccess)
    return GFV_ERR_LAUNCH;
  if (GfvRecorder* r = gfv_rec_active()) gfv_rec_push(r, run_wait, &b, sizeof(b), (hipStream_t)waiter);
  return GFV_OK;
}

// Issue order of a recorded step (round 6).  A step is recorded in the order the host issued it: the main stream's launches with the
// side stream's bursts (a block's weight-gradient launches and their reduction: 5 - 10 commands) in between.  On a small mesh the
// main stream's kernels last 5 - 20 us and the host needs ~4 us per command, so while it issues a burst the main queue runs dry
// (profiles/r06_timeline_cavity_unmerged.txt: holes of 30 - 110 us in front of the next block's first launch).  This pass moves every
// run of side-stream commands (its leading "side waits for main" included) behind up to `k` of the main-stream launches that follow it,
// but never across a command in which the main stream waits (a join), and only by a third of the distance to the next join (a burst
// right in front of a join - the weight images at the start of a step - stays where it is).  Moving a burst later only ADDS ordering:
// its wait then covers more of the main stream, and nothing on the main stream waits for it before the join.  Returns the number of
// runs moved.
extern "C" int gfv_record_delay_side(int64_t handle, void* main_stream, int32_t k) {
  GfvRecorder* r = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_lists.find(handle);
    if (it == g_lists.end()) return GFV_ERR_ARG;
    r = it->second;
  }
  if (k <= 0) return 0;
  const hipStream_t ms = (hipStream_t)main_stream;
  auto& c = r->cmds;
  const size_t n = c.size();
  auto is_join = [&](size_t i) { return c[i].st == ms && c[i].run == run_wait; };
  int moved_runs = 0;
  size_t i = 0;
  while (i < n) {
    if (c[i].st == ms) { ++i; continue; }
    size_t j = i;
    while (j < n && c[j].st != ms) ++j;          // the run [i, j)
    size_t d = 0, t = j;
    while (t < n && !is_join(t)) { if (c[t].st == ms) ++d; ++t; }   // main-stream launches up to the next join
    size_t want = d / 3 < (size_t)k ? d / 3 : (size_t)k;
    t = j;
    size_t got = 0;
    while (t < n && got < want && c[t].st == ms && !is_join(t)) { ++t; ++got; }
    if (got > 0) {
      std::rotate(c.begin() + i, c.begin() + j, c.begin() + t);
      ++moved_runs;
    }
    i = t > j ? t : j;
  }
  return moved_runs;
}

