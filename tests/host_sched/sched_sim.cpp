// CPU execution of the dataflow scheduler's dependency rules (csrc/sched_logic.h) -- the same header the device kernel
// compiles.  A randomised executor finishes announced tasks in arbitrary order (optionally from several host threads, with
// real atomics) and checks the contract the device relies on:
//   * every task of the sweep is announced exactly once, and the count equals tasks_per_matrix (the queue size the host
//     allocates, and the workgroups' exit condition);
//   * when a task is announced, everything it reads is finished:
//       D(j)   : T(j, k) for k < j, T(nt, k) for k < j
//       U(i,j) : T(i, k), T(j, k) for first_col(i) <= k < j
//       T(i,j) : D(j), and U(i, j) where it exists.
// A last argument `chaos` > 0 makes a thread sleep for a few microseconds after every chaos-th counter increment on average:
// the descheduling in the middle of a fan-out that a loaded host produces once in a hundred runs, on every run.
// usage: sched_sim nt ne lower seed threads [chaos]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <random>
#include <set>
#include <thread>
#include <tuple>
#include <vector>

#include "../../convex-combination-of-gaussian-processes_amd/csrc/sched_logic.h"

using namespace ccgp::sched;
typedef std::tuple<int, int, int> Key;   // kind, j, i

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const Shape s{std::atoi(argv[1]), std::atoi(argv[2]), std::atoi(argv[3])};
  const unsigned seed = (unsigned)std::atoi(argv[4]);
  const int nthreads = std::atoi(argv[5]);
  const int chaos = argc > 6 ? std::atoi(argv[6]) : 0;
  const int cpm = counters_per_matrix(s);
  std::vector<int> init(cpm);
  init_counters(s, init.data());
  std::vector<std::atomic<int>> c(cpm);
  for (int i = 0; i < cpm; ++i) c[i].store(init[i]);

  std::mutex mu;
  std::vector<uint64_t> ready;          // announced, not yet taken
  std::set<Key> announced, finished;
  int errors = 0;
  long running = 0;

  auto check_ready = [&](int kind, int j, int i) {   // called under mu at announcement time
    auto need = [&](int k2, int j2, int i2) {
      if (!finished.count(Key(k2, j2, i2))) {
        if (errors++ < 10) std::fprintf(stderr, "task (%d,%d,%d) announced before (%d,%d,%d) finished\n", kind, j, i, k2, j2, i2);
      }
    };
    if (kind == kD) {
      for (int k = 0; k < j; ++k) { need(kT, k, j); need(kT, k, s.nt); }
    } else if (kind == kU) {
      if (!has_U(s, i, j)) { ++errors; std::fprintf(stderr, "U(%d,%d) does not exist\n", i, j); }
      for (int k = first_col(s, i); k < j; ++k) need(kT, k, i);
      for (int k = 0; k < j; ++k) need(kT, k, j);
    } else {
      if (!has_T(s, i, j)) { ++errors; std::fprintf(stderr, "T(%d,%d) does not exist\n", i, j); }
      need(kD, j, j);
      if (has_U(s, i, j)) need(kU, j, i);
    }
  };
  auto announce_locked = [&](int kind, int j, int i) {
    if (!announced.insert(Key(kind, j, i)).second) {
      ++errors;
      std::fprintf(stderr, "task (%d,%d,%d) announced twice\n", kind, j, i);
    }
    check_ready(kind, j, i);
    ready.push_back(encode(kind, j, i, 0));
  };
  {
    std::lock_guard<std::mutex> g(mu);
    announce_locked(kD, 0, 0);
  }

  auto worker = [&](unsigned wseed) {
    std::mt19937 rng(wseed);
    for (;;) {
      uint64_t w = 0;
      {
        std::lock_guard<std::mutex> g(mu);
        if (ready.empty()) {
          if (running == 0) return;   // nothing announced, nothing running: done (or stuck -- the count check tells)
          continue;
        }
        const size_t k = rng() % ready.size();
        w = ready[k];
        ready[k] = ready.back();
        ready.pop_back();
        ++running;
      }
      const int kind = task_kind(w), j = task_j(w), i = task_i(w);
      {   // "the tile": mark finished BEFORE the arrivals, as the device releases its stores before them
        std::lock_guard<std::mutex> g(mu);
        finished.insert(Key(kind, j, i));
      }
      auto add = [&](int idx, int inc) {
        const int old = c[idx].fetch_add(inc, std::memory_order_acq_rel);
        if (chaos > 0 && rng() % (unsigned)chaos == 0) std::this_thread::sleep_for(std::chrono::microseconds(20 + rng() % 200));
        return old;
      };
      auto ann = [&](int k2, int j2, int i2) {
        std::lock_guard<std::mutex> g(mu);
        announce_locked(k2, j2, i2);
      };
      auto raise = [&](int idx, int level) {
        int old = c[idx].load(std::memory_order_acquire);
        while (hi16(old) < level && !c[idx].compare_exchange_weak(old, (level << 16) | lo16(old), std::memory_order_acq_rel)) {}
        if (chaos > 0 && rng() % (unsigned)chaos == 0) std::this_thread::sleep_for(std::chrono::microseconds(20 + rng() % 200));
        return old;
      };
      finish(s, kind, j, i, add, raise, ann);
      {
        std::lock_guard<std::mutex> g(mu);
        --running;
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t) th.emplace_back(worker, seed * 7919u + t);
  for (auto& t : th) t.join();

  const long want = tasks_per_matrix(s);
  if ((long)announced.size() != want || (long)finished.size() != want) {
    ++errors;
    std::fprintf(stderr, "announced %zu, finished %zu, tasks_per_matrix %ld\n", announced.size(), finished.size(), want);
  }
  // every task that should exist was announced
  for (int j = 0; j < s.nt; ++j) {
    if (!announced.count(Key(kD, j, j))) { ++errors; std::fprintf(stderr, "D(%d) missing\n", j); }
    for (int i = 0; i < rows(s); ++i) {
      if (has_T(s, i, j) && !announced.count(Key(kT, j, i))) { ++errors; std::fprintf(stderr, "T(%d,%d) missing\n", i, j); }
      if (has_U(s, i, j) && !announced.count(Key(kU, j, i))) { ++errors; std::fprintf(stderr, "U(%d,%d) missing\n", i, j); }
    }
  }
  std::printf("nt=%d ne=%d lower=%d threads=%d chaos=%d: %ld tasks, %d errors\n", s.nt, s.ne, s.lower, nthreads, chaos, want, errors);
  return errors ? 1 : 0;
}
