"""Per-kernel PMC summary from a rocprofv3 (rocpd sqlite) counter-collection run: average counter value per dispatch, one column per
counter.  python tools/rocpd_pmc.py results.db [out.txt] [--digest HEX]   (FETCH_SIZE / WRITE_SIZE are in KiB)"""
import re
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    digest = None
    if "--digest" in sys.argv:
        digest = sys.argv[sys.argv.index("--digest") + 1]
        args = [a for a in args if a != digest]
    c = sqlite3.connect(args[0])
    rows = c.execute("select name, counter_name, count(*), sum(counter_value), avg(counter_value), avg(duration) from pmc_events "
                     "group by name, counter_name").fetchall()
    ctrs = sorted({r[1] for r in rows})
    per = {}
    for n, cn, cnt, tot, avg, dur in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        d = per.setdefault(n, {"calls": cnt, "dur": dur, "tot": 0.0})
        d[cn] = avg
        d["tot"] = max(d["tot"], tot)
    lines = []
    if digest:
        lines.append("# lib_digest=%s" % digest)
    lines.append("%-78s %8s %10s " % ("kernel", "calls", "avg_us") + " ".join("%22s" % c_ for c_ in ctrs))
    for n, d in sorted(per.items(), key=lambda kv: -kv[1]["dur"] * kv[1]["calls"])[:48]:
        # single-counter files keep the historical column order (... total avg_value avg_us) that bench.py parses: value is field [-2]
        if len(ctrs) == 1:
            lines.append("%-78s %-12s %8d %14.1f %12.1f %10.2f" % (n[:78], ctrs[0], d["calls"], d["tot"], d.get(ctrs[0], 0.0), d["dur"] / 1e3))
        else:
            lines.append("%-78s %8d %10.2f " % (n[:78], d["calls"], d["dur"] / 1e3) + " ".join("%22.1f" % d.get(c_, float("nan")) for c_ in ctrs))
    txt = "\n".join(lines)
    print(txt)
    if len(args) > 1:
        open(args[1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
