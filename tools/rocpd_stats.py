"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / % — the --stats table as text.
python tools/rocpd_stats.py results.db [out.txt]"""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["%-90s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%")]
    for n, cnt, s, a, mn, mx in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        lines.append("%-90s %8d %12.3f %10.2f %10.2f %10.2f %6.2f" % (n[:90], cnt, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    lines.append("TOTAL kernel time %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
