"""Per-kernel statistics from a rocprofv3 rocpd database (run_results.db): calls, average / total duration.
Usage: python tools/rocpd_stats.py <run_results.db> [top]"""
import re
import sqlite3
import sys


def stats(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    scols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    q = f"select s.{name_col}, count(*), avg(d.end - d.start), sum(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.{name_col}"
    return list(cur.execute(q)), cols, scols


if __name__ == "__main__":
    rows, cols, scols = stats(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    tot = sum(r[3] for r in rows)
    print(f"{'kernel':100s} {'calls':>6s} {'avg us':>9s} {'total ms':>9s} {'%':>6s}")
    for n, c, a, t in sorted(rows, key=lambda r: -r[3])[:top]:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)[:98]
        print(f"{n:100s} {c:6d} {a / 1e3:9.1f} {t / 1e6:9.2f} {100 * t / tot:6.1f}")
