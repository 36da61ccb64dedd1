// Dependency logic of the dataflow tile scheduler of the blocked Cholesky sweep (blocked_sched.inc: chol_sched_kernel).
// Plain C++ shared by the device kernel, the host (queue sizes, initial counters) and a CPU test that executes the same
// rules with host threads (tests/host_sched/): every task must be announced exactly once, after everything it reads.
//
// Tasks of ONE matrix with nt block columns, the thin right-hand-side tile row nt and ne extra tile rows nt+1 .. nt+ne
// (prediction: cross-correlation rows; inverse / gradient: identity rows, `lower`: extra row e = nt + 1 + te is zero left of
// block column te, so its first task is T(e, te)):
//   D(j)    j = 0 .. nt-1   diagonal tile: (j >= 1: T_jj and the right-hand-side rows minus the finished panel,) factor, inverse
//   U(i,j)  j >= 1, i > j   T_ij = A_ij - sum_{k<j} L_ik L_jk'                 reads rows i and j of block columns < j
//   T(i,j)  j >= 0, i > j   L_ij = T_ij W_j'  (i = nt: the thin row)          reads U(i,j)'s tile and D(j)'s W_j
// What each task waits for (everything earlier follows transitively):
//   D(j+1)   <- T(j+1, j), T(nt, j)
//   U(i,j+1) <- T(i, j),   T(j+1, j)
//   T(i,j)   <- U(i, j) (j >= 1, i != nt),  D(j)
// Per row two words of 16 + 16 bits, per matrix one more; the ONE atomic update after which both halves admit the row's
// next task announces it:
//   cu[i] : lo = T(i, .) finished ("own"),  hi = highest pivot-row solve T(L, L-1) seen (a MAXIMUM, not a count) -> U(i, lo)
//   ct[i] : lo = 1 + U(i, .) finished,      hi = 1 + highest D(.) seen (a maximum)                               -> T(i, lo - 1)
//   cd    : lo = thin-row solves T(nt, .) finished, hi = pivot-row solves finished (both counts: strictly sequential) -> D(min)
// Why maxima: the fan-out of a diagonal block or of a pivot-row solve visits its rows one after the other, and a row that
// has received its arrival can start tasks that lead to the NEXT block column's fan-out while this one is still under way
// (own-progress arrivals announce too), so a row may see block column L + 1 before L.  "L + 1 seen" implies that block
// column L's tile is finished (L + 1 could not exist otherwise), so the maximum is the right readiness test, and a late
// arrival that changes nothing announces nothing.  (A count in that field announced tasks twice when a host thread was
// descheduled in the middle of a fan-out: tests/test_sched_logic.py runs the rules from eight threads with random sleeps.)
// Announced tasks go to a FIFO of the matrix's queue (one queue per XCD: matrix b lives on queue b % 8, so all tiles of a
// matrix share one L2); a workgroup takes the next slot index and waits until that slot is filled.  A workgroup never waits
// for a particular task, only for "one more announcement", so any number of resident workgroups makes progress.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define CCGP_HD __host__ __device__
#else
#define CCGP_HD
#endif

namespace ccgp {
namespace sched {

enum Kind : int { kD = 1, kU = 2, kT = 3 };

struct Shape {
  int nt, ne, lower;
};

// 64-bit task word, never zero: kind [62:61] | j [60:48] | i [47:32] | matrix [31:0] | bit 63
CCGP_HD inline uint64_t encode(int kind, int j, int i, int b) {
  return (1ull << 63) | ((uint64_t)kind << 61) | ((uint64_t)j << 48) | ((uint64_t)i << 32) | (uint32_t)b;
}
CCGP_HD inline int task_kind(uint64_t w) { return (int)((w >> 61) & 3); }
CCGP_HD inline int task_j(uint64_t w) { return (int)((w >> 48) & 0x1fff); }
CCGP_HD inline int task_i(uint64_t w) { return (int)((w >> 32) & 0xffff); }
CCGP_HD inline int task_b(uint64_t w) { return (int)(uint32_t)w; }

CCGP_HD inline int rows(const Shape& s) { return s.nt + 1 + s.ne; }                  // row slots per matrix
CCGP_HD inline int counters_per_matrix(const Shape& s) { return 2 * rows(s) + 1; }   // cu[rows], ct[rows], cd
CCGP_HD inline int first_col(const Shape& s, int i) {   // block column of row i's first task
  return (s.lower && i > s.nt) ? i - s.nt - 1 : 0;
}
// does row i take part in block column j's solve T(i, j) / update U(i, j)?
CCGP_HD inline bool has_T(const Shape& s, int i, int j) {
  if (i < s.nt) return i > j;
  if (i == s.nt) return true;
  return j >= first_col(s, i);
}
CCGP_HD inline bool has_U(const Shape& s, int i, int j) {
  if (j < 1 || i == s.nt) return false;
  if (i < s.nt) return i > j;
  return j > first_col(s, i) || (!s.lower);   // identity row te: the k-sum of U(e, te) is empty, its first task is T(e, te)
}

CCGP_HD inline long tasks_per_matrix(const Shape& s) {
  long t = s.nt;                                         // D
  for (int j = 0; j < s.nt; ++j) {
    t += (s.nt - 1 - j) + 1;                             // T: matrix rows below + thin row
    if (j >= 1) t += s.nt - 1 - j;                       // U
    for (int e = s.nt + 1; e <= s.nt + s.ne; ++e) t += (has_T(s, e, j) ? 1 : 0) + (has_U(s, e, j) ? 1 : 0);
  }
  return t;
}

// initial counters of one matrix (c = counters_per_matrix ints)
CCGP_HD inline void init_counters(const Shape& s, int* c) {
  const int R = rows(s);
  for (int i = 0; i < R; ++i) {
    const int f = first_col(s, i);
    c[i] = f;                                            // cu: own = f (as if T(i, 0 .. f-1) were done), pivot = 0
    // ct: lo = 1 + U done; a row whose first task is T(i, f) behaves as if U(i, 1 .. f) were done
    c[R + i] = (f + 1) | (f << 16);                      // hi = f: as if D(0 .. f-1) had been seen (the thin row's word is unused)
  }
  c[2 * R] = 0;
}

// Arrivals of a finished task.  The caller supplies three primitives on the matrix's counters (index into the
// counters_per_matrix words), each ONE atomic read-modify-write that returns the OLD word:
//   add(idx, inc)      fetch-add
//   raise(idx, level)  hi = max(hi, level), lo untouched (compare-exchange loop; returns the word it saw when hi >= level already)
// and `announce(kind, j, i)`, which queues a task of the same matrix.  The device kernel runs the two fan-outs with one lane
// per row (sched_finish in blocked_sched.inc) on the same row rules.
CCGP_HD inline int lo16(int v) { return v & 0xffff; }
CCGP_HD inline int hi16(int v) { return (v >> 16) & 0xffff; }
// did raising hi to `level` carry it over lo (the row's waiting task became ready through THIS update)?
CCGP_HD inline bool crossed(int old, int level) { return hi16(old) < lo16(old) && lo16(old) <= level; }

// D(j) finished.  Row i: returns the block column of the solve T(i, .) this made ready, or -1.
template <class Raise>
CCGP_HD inline int row_after_D(const Shape& s, int j, int i, Raise raise) {
  if (!has_T(s, i, j)) return -1;
  if (i == s.nt) return j;                               // the thin row waits for nothing else: no counter
  const int old = raise(rows(s) + i, j + 1);
  return crossed(old, j + 1) ? lo16(old) - 1 : -1;
}
// pivot-row solve T(j+1, j) finished.  Row i: returns the block column of the update U(i, .) this made ready, or -1.
template <class Raise>
CCGP_HD inline int row_after_pivot(const Shape& s, int j, int i, Raise raise) {
  if (i == s.nt || (i < s.nt && i <= j + 1)) return -1;   // finished rows, the pivot row itself, the thin row: no U(i, j+1)
  const int old = raise(i, j + 1);
  return (crossed(old, j + 1) && has_U(s, i, lo16(old))) ? lo16(old) : -1;
}

// the tasks with ONE arrival and at most one announcement: U(i, j), and T(i, j) of the thin row or of an ordinary row
template <class Add, class Announce>
CCGP_HD inline void finish_single(const Shape& s, int kind, int j, int i, Add add, Announce announce) {
  const int R = rows(s);
  if (kind == kU) {
    const int old = add(R + i, 1);                       // lo(ct[i]) -> j + 1
    if (hi16(old) >= j + 1) announce((int)kT, j, i);
  } else if (kind == kT && j + 1 < s.nt) {               // (last block column: nothing follows)
    if (i == s.nt) {                                     // thin row -> D(j+1) once the pivot-row solve is in as well
      const int old = add(2 * R, 1);
      if (hi16(old) >= j + 1) announce((int)kD, j + 1, j + 1);
    } else {                                             // an ordinary row: its own progress
      const int old = add(i, 1);                         // lo(cu[i]) -> j + 1
      if (has_U(s, i, j + 1) && hi16(old) >= j + 1) announce((int)kU, j + 1, i);
    }
  }
}
CCGP_HD inline bool is_pivot_solve(const Shape& s, int kind, int j, int i) { return kind == kT && i == j + 1 && j + 1 < s.nt; }

template <class Add, class Raise, class Announce>
CCGP_HD inline void finish(const Shape& s, int kind, int j, int i, Add add, Raise raise, Announce announce) {
  const int R = rows(s);
  if (kind == kD) {
    for (int r = j + 1; r < R; ++r) {
      const int jt = row_after_D(s, j, r, raise);
      if (jt >= 0) announce((int)kT, jt, r);
    }
  } else if (is_pivot_solve(s, kind, j, i)) {            // pivot row of the next block column
    const int old = add(2 * R, 1 << 16);
    if (lo16(old) >= j + 1) announce((int)kD, j + 1, j + 1);
    for (int r = j + 2; r < R; ++r) {
      const int ju = row_after_pivot(s, j, r, raise);
      if (ju >= 0) announce((int)kU, ju, r);
    }
  } else {
    finish_single(s, kind, j, i, add, announce);
  }
}

}  // namespace sched
}  // namespace ccgp
