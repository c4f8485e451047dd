"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls, total / average / min / max duration.
Usage: python scripts/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/<name>.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
sym_cols = [r[1] for r in db.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in sym_cols else ("display_name" if "display_name" in sym_cols else sym_cols[-1])
rows = db.execute(f"""select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                      from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                      group by s.{name_col} order by 3 desc""").fetchall()
total = sum(r[2] for r in rows)
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---:|---:|---:|---:|---:|---:|")
for name, n, tot, mn, mx in rows:
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name) or re.match(r"_ZN4att2(\d+)", name) or re.match(r"_Z(\d+)", name)
    if m:  # Itanium mangling: <len><identifier>; template arguments are kept as a suffix
        start = m.end()
        short = name[start:start + int(m.group(1))]
        t = re.match(r"I(L[ib]\d+E)+E", name[start + int(m.group(1)):])
        if t:
            short += "<" + ",".join(re.findall(r"L[ib](\d+)E", t.group(0))) + ">"
    else:
        short = re.sub(r"\(.*", "", name)[:60]
    print(f"| {short} | {n} | {tot / 1e6:.3f} | {tot / n / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * tot / total:.1f} |")
print(f"\ntotal kernel time {total / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches")
