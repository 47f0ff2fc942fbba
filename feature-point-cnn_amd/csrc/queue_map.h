// queue_map.h -- which hardware queue a HIP stream sits on, found once per process and device, from DEVICE timestamps.
//
// The runtime maps streams onto GPU_MAX_HW_QUEUES (4) hardware queues, a new stream onto the least-used one -- which one
// that is depends on every stream the PROCESS has made and dropped before -- and two streams on one queue run their
// kernels in a row: with the two sub-batch streams of a context on one queue the context is a one-stream context
// (round 4, 64 HD frames, bf16: the same engine gave 12 790 / 12 770 / 12 110 / 11 620 / 12 780 / 12 730 frames/s with
// 0 .. 5 unrelated streams created in the process before it).
//
// Round 4 decided "side by side" from a host wall clock around two 150 us spin kernels and a fixed threshold, probed every
// candidate against the LIVE streams of other contexts, and did so again for every context (416 probe launches = 63 ms of
// GPU time in one bench process; under a profiler's kernel serialisation it could not succeed at all).  Round 5:
//
// * ANCHORS.  The first fpc_create on a device finds one idle, library-owned stream per hardware queue (up to
//   GPU_MAX_HW_QUEUES of them) and keeps them for the life of the process.  Nobody but the probe launches on them.
// * A stream's QUEUE CLASS = the anchor it cannot run beside.  One round = one-thread kernels on the stream and on every
//   anchor; each records s_memrealtime (the constant 100 MHz clock) when it starts, waits until the HOST has issued all
//   launches of the round (a flag in pinned host memory -- host jitter between two launches, or the code object's first
//   load, therefore does not matter), then until every kernel of the round has started or a grace period (30 us) is
//   over, and records the clock again.  Kernels on different queues overlap; the one queued behind another starts after
//   that one has ended.  The verdict comes from the recorded intervals, not from a host clock.  A round is 5 launches and
//   40-80 us; a verdict is never taken against a foreign (possibly busy) stream.
// * A REGISTRY of the live contexts' streams with their classes: "beside the other contexts' main streams" is a table
//   lookup, not a probe.  And a SPARE LIST: candidates that were set aside and the streams of closed contexts stay
//   alive (idle, at most three per queue) with their class, so a later fpc_create mostly finds its streams there --
//   no probe round and no hipStreamCreate (a created-and-dropped candidate costs 0.3 ms, a probe round 0.1 ms).
// * BOUNDED.  A kernel leaves after 2 ms whatever happens (every wave reaches that exit); three inconclusive rounds in a
//   process (a profiler that serialises kernels makes every round inconclusive) switch probing off for the process:
//   streams are then taken as the runtime hands them out, which is what an unprobed build does.
// FPC_QUEUE_PROBE=0 switches all of this off; =2 logs every decision to stderr.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace fpc {
namespace qmap {

constexpr int MAX_ANCHORS = 8;
constexpr unsigned GRACE_TICKS = 3000;     // 30 us of the 100 MHz clock: how long a started kernel waits for the others
constexpr unsigned HARD_TICKS = 200000;    // 2 ms: every probe kernel has left by then

// One thread.  stamps (pinned host memory) [slot][2] = start / end on the constant clock; `started` (device) counts the
// kernels of all rounds so far; `go` (pinned host) holds the number of the last round whose launches have all been issued.
__global__ __launch_bounds__(64) void queue_stamp_kernel(unsigned long long* stamps, unsigned* started, const unsigned* go,
                                                          unsigned epoch, int slot, unsigned target, unsigned grace, unsigned hard) {
  if (threadIdx.x) return;
  unsigned long long t0, t, tg;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  stamps[2 * slot] = t0;
  __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t = t0;
  // (1) the host has issued every launch of this round (or the hard limit)
  while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != epoch && t - t0 < hard) {
    __builtin_amdgcn_s_sleep(8);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  }
  tg = t;
  // (2) every kernel of the round has started (unsigned difference: the counter runs on over the rounds), or the grace
  // period is over -- a kernel queued behind this one on the same hardware queue cannot start before this one ends
  while ((int)(__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 && t - tg < grace && t - t0 < hard) {
    __builtin_amdgcn_s_sleep(4);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  }
  stamps[2 * slot + 1] = t;
}

enum { Q_OWN_QUEUE = -1, Q_INCONCLUSIVE = -2, Q_UNKNOWN = -3 };

struct DeviceQueues {
  bool tried = false;       // anchor discovery ran
  bool futile = false;      // verdicts were inconclusive (kernel serialisation, no concurrency): no more probing in this process
  std::vector<hipStream_t> anchor;
  unsigned long long* stamps = nullptr;   // pinned host: [MAX_ANCHORS + 1][2], then the go word
  unsigned* go = nullptr;
  unsigned* started = nullptr;            // device
  unsigned epoch = 0, launched = 0;
  // idle, classified streams of the library's own that no context uses at the moment: candidates that were set aside and
  // the streams of closed contexts.  A later fpc_create takes its streams from here -- no probe round, no stream creation.
  std::vector<std::pair<hipStream_t, int>> spare;
  int rounds = 0, launches = 0, inconclusive = 0;
  double ms = 0.0;                        // host time spent in rounds (reported, never decided on)
};

struct Placed {
  const void* owner;
  int slot;             // 0 main, 1.. sub-batch streams, 100.. side streams, 200 upload
  int device;
  hipStream_t st;
  int qclass;
  bool heavy;           // main / sub-batch stream: what other contexts keep clear of
};

struct State {
  std::mutex mu;
  std::map<int, DeviceQueues> dev;
  std::vector<Placed> placed;
};
inline State& state() {
  static State* s = new State();   // (never destroyed: contexts may be closed from static destructors)
  return *s;
}

inline int probe_mode() {
  const char* e = getenv("FPC_QUEUE_PROBE");
  return e ? atoi(e) : 1;
}
inline int max_hw_queues() {
  const char* e = getenv("GPU_MAX_HW_QUEUES");
  const int n = e ? atoi(e) : 4;
  return n < 1 ? 1 : n > MAX_ANCHORS ? MAX_ANCHORS : n;
}

inline bool ensure_buffers(DeviceQueues& q) {
  if (q.stamps) return true;
  void* h = nullptr;
  if (hipHostMalloc(&h, 4096, hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipMalloc((void**)&q.started, 256) != hipSuccess) { (void)hipGetLastError(); hipHostFree(h); return false; }
  if (hipMemset(q.started, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError(); hipFree(q.started); q.started = nullptr; hipHostFree(h); return false;
  }
  memset(h, 0, 4096);
  q.stamps = static_cast<unsigned long long*>(h);
  q.go = reinterpret_cast<unsigned*>(q.stamps + 2 * (MAX_ANCHORS + 1));
  return true;
}

// The queue class of `s` (an idle stream): index of the anchor it shares a hardware queue with, Q_OWN_QUEUE if it runs
// beside all of them, Q_INCONCLUSIVE if the round's intervals name more than one anchor.
inline int classify(DeviceQueues& q, hipStream_t s) {
  const int n = (int)q.anchor.size();
  if (q.futile) return Q_INCONCLUSIVE;
  if (!ensure_buffers(q)) { q.futile = true; return Q_INCONCLUSIVE; }
  if (n == 0) return Q_OWN_QUEUE;
  const auto w0 = std::chrono::steady_clock::now();
  const unsigned epoch = ++q.epoch;
  const unsigned target = q.launched + (unsigned)(n + 1);
  q.launched = target;
  hipLaunchKernelGGL(queue_stamp_kernel, dim3(1), dim3(64), 0, s, q.stamps, q.started, q.go, epoch, n, target, GRACE_TICKS, HARD_TICKS);
  for (int j = 0; j < n; ++j)
    hipLaunchKernelGGL(queue_stamp_kernel, dim3(1), dim3(64), 0, q.anchor[j], q.stamps, q.started, q.go, epoch, j, target, GRACE_TICKS, HARD_TICKS);
  __atomic_store_n(q.go, epoch, __ATOMIC_RELEASE);
  bool ok = hipStreamSynchronize(s) == hipSuccess;
  for (int j = 0; j < n; ++j) ok = (hipStreamSynchronize(q.anchor[j]) == hipSuccess) && ok;
  ok = (hipGetLastError() == hipSuccess) && ok;
  q.rounds += 1;
  q.launches += n + 1;
  q.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
  int verdict = Q_INCONCLUSIVE;
  if (ok) {
    const unsigned long long cs = q.stamps[2 * n], ce = q.stamps[2 * n + 1];
    int apart = 0, which = -1;
    for (int j = 0; j < n; ++j) {
      const unsigned long long as = q.stamps[2 * j], ae = q.stamps[2 * j + 1];
      const bool overlap = as < ce && cs < ae;
      if (!overlap) { ++apart; which = j; }
    }
    verdict = apart == 0 ? Q_OWN_QUEUE : apart == 1 ? which : Q_INCONCLUSIVE;
  }
  if (verdict == Q_INCONCLUSIVE && ++q.inconclusive >= 3) {
    q.futile = true;
    if (probe_mode() >= 1)
      fprintf(stderr, "[fpc] stream placement: three probe rounds without a verdict (kernels are being serialised -- a profiler? -- "
                      "or the device is busy): streams are taken as the runtime hands them out from here on\n");
  }
  return verdict;
}

// One idle stream of the library's own per hardware queue.  At most 12 candidates, every round bounded.
inline void discover(DeviceQueues& q) {
  if (q.tried) return;
  q.tried = true;
  const int want = max_hw_queues();
  std::vector<hipStream_t> aside;
  for (int tries = 0; tries < 12 && (int)q.anchor.size() < want && !q.futile; ++tries) {
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
    const int k = classify(q, st);
    if (k == Q_OWN_QUEUE) q.anchor.push_back(st);
    else aside.push_back(st);   // kept alive until the search ends: a dropped one would free its queue for the next candidate
  }
  for (hipStream_t st : aside) hipStreamDestroy(st);
  if (probe_mode() == 2)
    fprintf(stderr, "[fpc] stream placement: %zu hardware queues found with %d probe rounds (%d launches, %.2f ms)%s\n", q.anchor.size(),
            q.rounds, q.launches, q.ms, q.futile ? "; verdicts inconclusive, probing off" : "");
}

// An idle stream of the library's own that is no longer needed: kept as a spare if its queue is known (at most three per
// queue), destroyed otherwise.
inline void retire(DeviceQueues& q, hipStream_t st, int qclass) {
  if (!st) return;
  int same = 0;
  for (const auto& sp : q.spare) same += sp.second == qclass;
  if (qclass >= 0 && same < 3 && !q.futile) q.spare.push_back({st, qclass});
  else hipStreamDestroy(st);
}

// A new non-blocking stream for (owner, slot), registered.  `avoid_all_own`: keep clear of every stream of the owner's
// (sub-batch and side streams), otherwise of its heavy ones only (the upload stream).  Preferred: a queue no other
// context's heavy stream sits on.
inline hipStream_t acquire(int device, const void* owner, int slot, bool heavy, bool probe, bool avoid_all_own, int* qclass_out = nullptr) {
  State& S = state();
  DeviceQueues& q = S.dev[device];
  hipStream_t st = nullptr;
  int qclass = Q_UNKNOWN;
  if (probe && !q.futile) discover(q);
  if (!probe || q.futile || q.anchor.size() < 2) {
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  } else {
    int own[MAX_ANCHORS] = {}, others[MAX_ANCHORS] = {};
    for (const Placed& p : S.placed) {
      if (p.device != device || p.qclass < 0 || p.qclass >= MAX_ANCHORS) continue;
      if (p.owner == owner) own[p.qclass] += (avoid_all_own || p.heavy) ? 1 : 0;
      else if (p.heavy) others[p.qclass] += 1;
    }
    // score of a candidate: 0 shares a queue with one of the owner's streams, 1 beside them, 2 also beside every other
    // context's heavy stream; among equals the queue with fewer of those
    // ... and what the tables say is the best a candidate can reach: the search stops there, not after six candidates
    int reachable = 0;
    for (int k = 0; k < (int)q.anchor.size(); ++k) reachable = std::max(reachable, own[k] ? 0 : others[k] ? 1 : 2);
    struct Cand { hipStream_t st; int k, score, load; };
    std::vector<Cand> cand;
    int best = -1;
    auto rate = [&](hipStream_t s, int k) {
      Cand c{s, k, 1, 0};
      if (k >= 0) {
        c.load = others[k];
        c.score = own[k] ? 0 : others[k] ? 1 : 2;
      } else if (k == Q_OWN_QUEUE) {
        c.score = 2;
      }
      return c;
    };
    // a spare stream that reaches what is reachable: taken as it is
    {
      int take = -1;
      for (size_t i = 0; i < q.spare.size(); ++i) {
        const Cand c = rate(q.spare[i].first, q.spare[i].second);
        if (c.score == reachable && (take < 0 || c.load < rate(q.spare[take].first, q.spare[take].second).load)) take = (int)i;
      }
      if (take >= 0) {
        cand.push_back(rate(q.spare[take].first, q.spare[take].second));
        best = reachable;
        q.spare.erase(q.spare.begin() + take);
      }
    }
    for (int tries = 0; tries < 6 && best < reachable && !q.futile; ++tries) {
      hipStream_t s = nullptr;
      if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
      const int k = classify(q, s);
      const Cand c = rate(s, k);
      cand.push_back(c);
      best = std::max(best, c.score);
    }
    int pick = -1;
    for (size_t i = 0; i < cand.size(); ++i)
      if (cand[i].score == best && (pick < 0 || cand[i].load < cand[pick].load)) pick = (int)i;
    for (size_t i = 0; i < cand.size(); ++i)
      if ((int)i != pick) retire(q, cand[i].st, cand[i].k);
    if (pick >= 0) {
      st = cand[pick].st;
      qclass = cand[pick].k == Q_INCONCLUSIVE ? Q_UNKNOWN : cand[pick].k;
    } else if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    if (probe_mode() == 2)
      fprintf(stderr, "[fpc] stream placement: slot %d of %p: %zu candidates, queue %d: %s\n", slot, owner, cand.size(), qclass,
              best == 2 ? "beside all" : best == 1 ? "beside this context's" : best == 0 ? "SHARES a queue" : "unprobed");
  }
  S.placed.push_back({owner, slot, device, st, qclass, heavy});
  if (qclass_out) *qclass_out = qclass;
  return st;
}

inline void release_owner(const void* owner) {
  State& S = state();
  for (size_t i = 0; i < S.placed.size();)
    if (S.placed[i].owner == owner) S.placed.erase(S.placed.begin() + i);
    else ++i;
}
inline Placed* find(const void* owner, int slot) {
  for (Placed& p : state().placed)
    if (p.owner == owner && p.slot == slot) return &p;
  return nullptr;
}

}  // namespace qmap
}  // namespace fpc
