"""Where a bench step's wall time goes that is NOT kernel time: from a rocprofv3 --kernel-trace database, cut the trace
into steps at the split matcher launches (exactly one per step), and for the last steps print wall time, the union of the
kernel intervals (GPU busy), the idle gaps by the kernel that precedes them, and the time in kernels shorter than 10 us.
usage: python tools/step_gaps.py <results.db> [n_steps [first_step]]   (first_step: index of the first matcher launch to use;
default = the last n_steps + 1 launches)"""
import collections
import sqlite3
import sys


def main(path, n_steps=8, first=None):
    con = sqlite3.connect(path)
    rows = list(con.execute("select name, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if "match_tiles_split_kernel" in r[0]]
    if len(marks) < n_steps + 1:
        n_steps = len(marks) - 1
    marks = marks[-(n_steps + 1):] if first is None else marks[first:first + n_steps + 1]
    n_steps = len(marks) - 1
    print(f"# {path}: {n_steps} steps (delimited by match_tiles_split_kernel launches)")
    gap_by_prev, small, fam = collections.Counter(), collections.Counter(), collections.Counter()
    gap_n = collections.Counter()
    wall = 0.0
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a:b + 1]
        wall += (seg[-1][1] - seg[0][1]) / 1e3
        cur_end = seg[0][1]
        for (n0, s0, e0), (n1, s1, e1) in zip(seg[:-1], seg[1:]):
            cur_end = max(cur_end, e0)
            if s1 > cur_end:
                gap_by_prev[n0[:110]] += (s1 - cur_end) / 1e3
                gap_n[n0[:110]] += 1
            d = (e0 - s0) / 1e3
            fam[n0[:110]] += d
            if d < 10.0:
                small[n0[:110]] += d
    gaps = sum(gap_by_prev.values())
    print(f"wall {wall / n_steps / 1e3:.3f} ms per step; idle gaps {gaps / n_steps / 1e3:.3f} ms per step; "
          f"kernels < 10 us: {sum(small.values()) / n_steps / 1e3:.3f} ms per step in {sum(1 for _ in small)} kernel names")
    print("\n# idle time by preceding kernel (us per step, count per step)")
    for k, v in gap_by_prev.most_common(15):
        print(f"{k:110s} {v / n_steps:9.1f} {gap_n[k] / n_steps:7.1f}")
    print("\n# kernel time (us per step)")
    for k, v in fam.most_common(30):
        print(f"{k:110s} {v / n_steps:9.1f}")
    print("\n# kernels shorter than 10 us (us per step)")
    for k, v in small.most_common(12):
        print(f"{k:110s} {v / n_steps:9.1f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else None)
