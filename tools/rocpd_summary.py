"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel table (markdown) for profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [steps] > profiles/rNN_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    m = re.match(r"void (.*)", name)
    name = m.group(1) if m else name
    name = name.replace("at::native::", "").replace("(anonymous namespace)::", "")
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    c = db.cursor()
    rows = c.execute("select name, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by name "
                     "order by sum(end-start) desc").fetchall() if _has(c, "kernels", "name") else None
    if rows is None:
        rows = c.execute("select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from rocpd_kernel_dispatch d "
                         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by sum(d.end-d.start) desc").fetchall()
    total = sum(r[2] for r in rows)
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, tot, mn, mx in rows[:45]:
        print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (short(name), n, tot / 1e6, tot / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    print("\ntotal kernel time: %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))
    if steps:
        print("(trace covers warm-up + %d timed steps + the CPU-baseline leg's zero GPU work)" % steps)


def _has(c, table, col):
    try:
        return col in [r[1] for r in c.execute("pragma table_info('%s')" % table)]
    except Exception:
        return False


if __name__ == "__main__":
    main()
