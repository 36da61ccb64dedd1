#!/usr/bin/env python3
"""List-scheduling model of the blocked sweep's launches on 256 CUs (one workgroup per CU saturates its MFMA pipes):
tile (i > j) = 16.4 j + 3 us, diagonal workgroup = 11.3 j + 82 us (11/16 of a tile, then the 128-step factorisation),
trsm tile 10 us, 8 us between launches.  Reproduces the measured 64-matrix slice to a few percent (27.8 ms against
26.4 ms of update + trsm + launch gaps) and answers the round-2 review's "two half-batches skewed by nt/2 block
columns in one launch, long workgroups first": WORSE for a single chunk of 64 -- the second half's late block
columns then run alone with 32 matrices.  profiles/r03_experiments.md section 5."""
import heapq

T = 16.4


def launch_time(wgs, slots=256, gap=8.0):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for d in wgs:
        t = heapq.heappop(h) + d
        end = max(end, t)
        heapq.heappush(h, t)
    return end + gap


def col_wgs(nm, j, nt=32):
    return [11.3 * j + 82.0] * nm + [T * j + 3.0] * (nm * (nt - 1 - j))


def sweep_plain(nm):
    return sum(launch_time(col_wgs(nm, j)) + launch_time([10.0] * (nm * (32 - j))) for j in range(1, 32))


def sweep_skew(nm_half, skew):
    tot = 0.0
    for t in range(1, 32 + skew):
        w, tt = [], []
        for j in (t, t - skew):
            if 1 <= j <= 31:
                w += col_wgs(nm_half, j)
            if 0 <= j <= 30:
                tt += [10.0] * (nm_half * (32 - j))
        if w:
            tot += launch_time(sorted(w, reverse=True))
        if tt:
            tot += launch_time(tt)
    return tot


if __name__ == "__main__":
    print("plain, 64 matrices : %.2f ms   (512 matrices: %.1f ms)" % (sweep_plain(64) / 1e3, sweep_plain(512) / 1e3))
    for sk in (4, 8, 12, 16):
        print("two halves of 32, skew %2d block columns: %.2f ms" % (sk, sweep_skew(32, sk) / 1e3))
    ideal = sum(64 * ((31 - j) * (T * j + 3) + 11.3 * j + 82) for j in range(1, 32)) / 256
    print("work / 256 CUs     : %.2f ms" % (ideal / 1e3))
