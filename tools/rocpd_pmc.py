"""Per-kernel PMC summary from a rocprofv3 (rocpd sqlite) counter-collection run: average counter value per dispatch.
python tools/rocpd_pmc.py results.db [out.txt]   (FETCH_SIZE / WRITE_SIZE are in KiB)"""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = c.execute("select name, counter_name, count(*), sum(counter_value), avg(counter_value), avg(duration) from pmc_events "
                     "group by name, counter_name order by 4 desc").fetchall()
    lines = ["%-78s %-12s %8s %14s %12s %10s" % ("kernel", "counter", "calls", "total_KiB", "avg_KiB", "avg_us")]
    for n, cn, cnt, tot, avg, dur in rows[:40]:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        lines.append("%-78s %-12s %8d %14.1f %12.1f %10.2f" % (n[:78], cn, cnt, tot, avg, dur / 1e3))
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
