#!/usr/bin/env python3
"""Per-kernel duration summary from a rocprofv3 rocpd .db (this image's rocprofv3 writes sqlite by default).
usage: tools/rocpd_stats.py <results.db> [--csv out.csv]"""
import collections, sqlite3, sys

def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    acc = collections.defaultdict(list)
    for name, start, end, gx, lds in c.execute("select name, start, end, grid_x, lds_size from kernels"):
        acc[(name.split('(')[0], gx, lds)].append(end - start)
    rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
    tot = sum(sum(v) for _, v in rows)
    lines = ['"Name","GridX","LdsBytes","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
    for (n, gx, lds), v in rows:
        lines.append('"%s",%d,%d,%d,%d,%.1f,%.2f,%d,%d' % (n, gx, lds, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 3 and sys.argv[2] == "--csv":
        open(sys.argv[3], "w").write(out)
    sys.stdout.write(out)

if __name__ == "__main__":
    main()
