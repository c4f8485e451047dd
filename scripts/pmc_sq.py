"""Where the waves of each kernel spend their cycles, from ONE rocprofv3 PMC pass with the SQ counters
    SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES
(MI355X_MICROARCH.md, PMC section: WAIT_ANY = wave parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY
= issuing; the three are disjoint and sum to WAVE_CYCLES; SQ counters are in quad-cycles, MFMA_BUSY in cycles).

Usage: python scripts/pmc_sq.py <pmc.db> > profiles/<round>_pmc_sq.md"""
import re
import sqlite3
import sys



def function_name(kn):
    """last component of an Itanium-mangled (possibly nested) kernel name, templates dropped (as in pmc_traffic.py)"""
    if kn.startswith("_Z"):
        i = 3 if kn.startswith("_ZN") else 2
        last = None
        while i < len(kn) and kn[i].isdigit():
            j = i
            while kn[j].isdigit():
                j += 1
            n = int(kn[i:j])
            last = kn[j:j + n]
            i = j + n
        return last or kn[:40]
    base = re.sub(r"<.*", "", kn.split("(")[0]).strip()
    return base.split("::")[-1].split(" ")[-1]


db = sqlite3.connect(sys.argv[1])
ks = [r[1] for r in db.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in ks else ks[-1]
rows = db.execute(f"""select s.{name_col}, p.name, count(*), sum(e.value) from rocpd_pmc_event e
                      join rocpd_info_pmc p on e.pmc_id = p.id
                      join rocpd_kernel_dispatch d on e.event_id = d.event_id
                      join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by 1, 2""").fetchall()
dur = dict(db.execute(f"""select s.{name_col}, sum(d.end - d.start) from rocpd_kernel_dispatch d
                          join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by 1""").fetchall())
tab = {}
# Round 6 ("gemm_ws 68.5 % MFMA-busy at 0.28 of peak"): template instantiations share a short name (gemm_ws_kernel =
# qkv + proj + fc1).  The counters were SUMMED over them but the time was the MAX of them, so every multi-instantiation row had its
# busy % inflated by sum / max (2.07 for gemm_ws).  Rows are now per instantiation (short name + template arguments) and the time of a
# row is the time of exactly the dispatches its counters come from.
def row_name(kn):
    short = function_name(kn)
    m = re.search(re.escape(short) + r"I(.*?)E+v", kn) if kn.startswith("_Z") else None
    return short + ("<" + m.group(1)[:28] + ">" if m else "")


seen = set()
for kn, cn, n, v in rows:
    short = row_name(kn)
    t = tab.setdefault(short, {"launches": 0, "ns": 0})
    t[cn] = t.get(cn, 0.0) + v
    if kn not in seen:
        seen.add(kn)
        t["ns"] += dur.get(kn, 0)
        t["launches"] += n
print("| kernel | launches | ms (this pass) | parked % | issue-stall % | issuing % | VALU issue % | VALU instr / wave | MFMA pipe busy % of kernel time |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
for k, t in sorted(tab.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:24]:
    wc = t.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    pct = lambda c: 100.0 * t.get(c, 0.0) / wc
    waves = max(t.get("SQ_WAVES", 0.0), 1.0)
    # MFMA_BUSY is summed over the 1024 SIMDs of the chip; kernel time in cycles at ~2.1 GHz
    busy = 100.0 * t.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * t["ns"] * 2.1) if t["ns"] else 0.0
    print(f"| {k} | {t['launches']} | {t['ns'] / 1e6:.2f} | {pct('SQ_WAIT_ANY'):.1f} | {pct('SQ_WAIT_INST_ANY'):.1f} | "
          f"{pct('SQ_ACTIVE_INST_ANY'):.1f} | {pct('SQ_ACTIVE_INST_VALU'):.1f} | {t.get('SQ_INSTS_VALU', 0.0) / waves:.0f} | {busy:.1f} |")
print("\n(percentages of SQ_WAVE_CYCLES; `MFMA pipe busy` assumes 1024 SIMDs and 2.1 GHz under load: an estimate)")
