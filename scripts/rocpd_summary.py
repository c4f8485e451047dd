"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls, total / average / min / max duration.
Usage: python scripts/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/<name>.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
sym_cols = [r[1] for r in db.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in sym_cols else ("display_name" if "display_name" in sym_cols else sym_cols[-1])
where = ""
if "--window" in sys.argv:  # --window KERNEL FIRST LAST: only the dispatches from the FIRST-th to the LAST-th launch of KERNEL (0-based)
    i = sys.argv.index("--window")
    key, first, last = sys.argv[i + 1], int(sys.argv[i + 2]), int(sys.argv[i + 3])
    marks = [r[0] for r in db.execute(f"""select d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                                          where s.{name_col} like ? order by d.start""", (f"%{key}%",))]
    where = f"where d.start >= {marks[first]} and d.start < {marks[last]}"
    print(f"(window: between launches {first} and {last} of `{key}`: {(marks[last] - marks[first]) / 1e6:.2f} ms, "
          f"{(marks[last] - marks[first]) / 1e6 / (last - first):.3f} ms per occurrence)\n")
rows = db.execute(f"""select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                      from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id {where}
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

if "--gaps" in sys.argv:
    # idle time of the device between consecutive dispatches (start_i - latest end so far), charged to the kernel that ends
    # the gap: where the host fails to keep the queue full
    disp = db.execute(f"""select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d
                          join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
    if "--gaps-window" in sys.argv:  # KERNEL FIRST LAST: from the FIRST-th to the LAST-th dispatch of KERNEL (0-based)
        i = sys.argv.index("--gaps-window")
        key, first, last = sys.argv[i + 1], int(sys.argv[i + 2]), int(sys.argv[i + 3])
        marks = [k for k, d in enumerate(disp) if key in d[0]]
        disp = disp[marks[first]:marks[last]]
        print(f"\n(window: dispatches {marks[first]} .. {marks[last]}, between occurrences {first} and {last} of `{key}`)")
    gaps = {}
    busy_end = disp[0][2]
    idle = 0
    big = []
    prev_name = disp[0][0]
    for name, st, en in disp[1:]:
        g = st - busy_end
        if g > 2_000_000:
            big.append((g, prev_name, name))
        prev_name = name if en >= busy_end else prev_name
        if g > 0:
            idle += g
            m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name) or re.match(r"_ZN4att2(\d+)", name) or re.match(r"_Z(\d+)", name)
            short = name[m.end():m.end() + int(m.group(1))] if m else re.sub(r"\(.*", "", name)[:50]
            a = gaps.setdefault(short, [0, 0, 0])
            a[0] += 1
            a[1] += g
            a[2] = max(a[2], g)
        busy_end = max(busy_end, en)
    span = busy_end - disp[0][1]
    print(f"\ndevice idle between dispatches: {idle / 1e6:.1f} ms of a {span / 1e6:.1f} ms span ({100.0 * idle / span:.1f} %)")
    print("| gap ends at kernel | gaps | total ms | max us |")
    print("|---|---:|---:|---:|")
    for k, (n, tot, mx) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"| {k} | {n} | {tot / 1e6:.3f} | {mx / 1e3:.1f} |")
    strip = lambda n: re.sub(r"\(.*", "", n)[:90]
    for g, a, b in big:
        print("\ngap of %.1f ms between `%s` and `%s`" % (g / 1e6, strip(a), strip(b)))
