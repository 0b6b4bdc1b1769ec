#!/usr/bin/env python3
"""Per-kernel summary of a `rocprofv3 --kernel-trace --stats -d DIR` run on this image (it writes a sqlite *_results.db, no CSV):
`python tools/rocprof_stats.py DIR [top N] > profiles/rNN_<what>_kernel_stats.txt`."""
import glob
import os
import re
import sqlite3
import sys


def demangle(names):
    import shutil
    import subprocess
    tool = shutil.which("llvm-cxxfilt") or shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        out = subprocess.run([tool], input="\n".join(n[:-3] if n.endswith(".kd") else n for n in names), capture_output=True, text=True,
                             check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((?!anonymous).*$", "", name)      # drop the parameter list
    return name if len(name) <= 150 else name[:147] + "..."


def main():
    d = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dbs = sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True))
    if not dbs:
        sys.exit(f"no *_results.db under {d}")
    for db in dbs:
        con = sqlite3.connect(db)
        tables = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        kd = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
        ks = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
        rows = con.execute(f"select s.kernel_name, count(*), sum(k.end - k.start), min(k.end - k.start), max(k.end - k.start) "
                           f"from {kd} k join {ks} s on k.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
        total = sum(r[2] for r in rows)
        dm = demangle([r[0] for r in rows[:top]])
        print(f"# {os.path.relpath(db, d)}: {sum(r[1] for r in rows)} dispatches, {total / 1e6:.3f} ms of kernel time")
        print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'%':>6}  kernel")
        for name, n, tot, mn, mx in rows[:top]:
            print(f"{n:8d} {tot / 1e6:10.3f} {tot / n / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100.0 * tot / total:6.2f}  {short(dm.get(name, name))}")


if __name__ == "__main__":
    main()
