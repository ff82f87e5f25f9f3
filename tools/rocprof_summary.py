#!/usr/bin/env python3
"""Summarise a rocprofv3 (--kernel-trace --stats) results.db into a small text table for profiles/."""
import glob
import sqlite3
import sys


def main(path, out):
    dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
    lines = []
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        lines.append(f"# {db}")
        lines.append("## top_kernels (durations in us)")
        lines.append("name | calls | total_us | avg_us | pct")
        for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            lines.append(" | ".join(str(x) for x in r))
        lines.append("## dispatches")
        lines.append("name | duration_us | grid_x | wg_x | lds_bytes | scratch_bytes | vgpr | agpr | sgpr")
        for r in cur.execute("select name,duration/1000.0,grid_x,workgroup_x,lds_size,scratch_size,vgpr_count,"
                             "accum_vgpr_count,sgpr_count from kernels order by start"):
            lines.append(" | ".join(str(x) for x in r))
        try:
            rows = list(cur.execute("select name, counter_name, sum(value) from counters_collection "
                                    "group by name, counter_name"))
            if rows:
                lines.append("## counters (sum over dispatches)")
                lines += [" | ".join(str(x) for x in r) for r in rows]
        except sqlite3.Error:
            pass
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
