"""Per-kernel sums of PMC counters from a rocprofv3 rocpd database (development / profiling aid)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
def cols(t): return [r[1] for r in db.execute(f"pragma table_info({t})")]
pe, ip, kd, ks = cols("rocpd_pmc_event"), cols("rocpd_info_pmc"), cols("rocpd_kernel_dispatch"), cols("rocpd_info_kernel_symbol")
name_col = "kernel_name" if "kernel_name" in ks else ks[-1]
q = f"""select s.{name_col}, p.name, count(*), sum(e.value) from rocpd_pmc_event e
        join rocpd_info_pmc p on e.pmc_id = p.id
        join rocpd_kernel_dispatch d on e.event_id = d.event_id
        join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by 1, 2"""
try:
    rows = db.execute(q).fetchall()
except Exception as ex:
    print("schema:", pe, ip, kd[:12]); raise
out = {}
for kn, cn, n, v in rows:
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", kn) or re.match(r"_Z(\d+)", kn)
    short = kn[m.end():m.end() + int(m.group(1))] if m else kn[:40]
    out.setdefault(short, {})[cn] = (n, v)
want = sys.argv[2:] or None
for k, d in sorted(out.items()):
    if want and not any(w in k for w in want): continue
    print(k, {c: (f"{v:.4g}", n) for c, (n, v) in sorted(d.items())})
