"""Summarise a rocprofv3 (ROCm 7.x, rocpd sqlite) result database as text:
per-kernel launch count / total / average / min / max duration, and -- when the run collected PMC
counters -- per-kernel counter totals and per-launch averages.
usage: python tools/rocprof_summary.py <results.db> [top_n]"""
import sqlite3
import sys


def main(path, top=40):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size), max(workgroup_x), max(grid_x) "
        "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1.0
    print(f"# kernel-trace summary of {path}")
    print(f"# total kernel time {tot:.2f} ms over {sum(r[1] for r in rows)} launches")
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} "
          f"{'vgpr':>5s} {'agpr':>5s} {'lds':>7s} {'scr':>5s} {'wg':>5s}")
    for r in rows[:top]:
        print(f"{r[0][:70]:70s} {r[1]:6d} {r[2]:10.3f} {r[3]:10.1f} {r[4]:9.1f} {r[5]:9.1f} {100*r[2]/tot:6.2f} "
              f"{r[6] or 0:5d} {r[7] or 0:5d} {r[8] or 0:7d} {r[9] or 0:5d} {r[10] or 0:5d}")
    try:
        cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)")]
        if cols:
            name_col = "kernel_name" if "kernel_name" in cols else "name"
            pmc = list(cur.execute(
                f"select {name_col}, counter_name, count(*), sum(value), avg(value) from counters_collection "
                f"group by {name_col}, counter_name order by 4 desc"))
            if pmc:
                print("\n# PMC counters (summed over all dimensions per dispatch, then over dispatches)")
                print(f"{'kernel':70s} {'counter':28s} {'samples':>8s} {'sum':>18s} {'avg/sample':>16s}")
                for r in pmc[: 4 * top]:
                    print(f"{r[0][:70]:70s} {r[1][:28]:28s} {r[2]:8d} {r[3]:18.1f} {r[4]:16.1f}")
    except sqlite3.Error as e:  # schema differences between rocprofv3 versions
        print("# (no PMC table readable:", e, ")")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
