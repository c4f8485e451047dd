"""HBM-side traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, rocpd databases).

Usage: python scripts/pmc_traffic.py <fetch.db> <write.db> > profiles/<round>_pmc_traffic.json
Corrections as prescribed by MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies
the 128-byte requests of wide reads at 64 bytes, so it is doubled; WRITE_SIZE is uncalibrated and taken as is."""
import json, re, sqlite3, sys

NAMES = {  # kernel function -> launch name used by bench.py (only kernels with one launch name)
    "attention_kernel": "vit_attention", "attention2_kernel": "vit_attention", "attention4_kernel": "vit_attention", "attention6_kernel": "vit_attention", "corr_peaks_kernel": "corr_peaks", "refine_corr_kernel": "refine_corr_generic", "refine_corr_dma_kernel": "refine_corr",
    "refine_head_kernel": "refine_head", "layernorm_kernel": "vit_layernorm", "rescore_kernel": "rescore",
    "patch_embed_split_kernel": "vit_patch_embed", "conv1_split_kernel": "dd_conv1", "gemm_wide_delta_kernel": "vit_gemm_fc2",
}

def function_name(kn):
    """last component of an Itanium-mangled (possibly nested: _ZN <len><name> ... E) kernel name, templates dropped;
    names that are already demangled lose their namespaces, template arguments and parameter list"""
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


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    ks = [r[1] for r in db.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in ks else ks[-1]
    rows = db.execute(f"""select s.{name_col}, count(*), sum(e.value) from rocpd_pmc_event e
                          join rocpd_info_pmc p on e.pmc_id = p.id
                          join rocpd_kernel_dispatch d on e.event_id = d.event_id
                          join rocpd_info_kernel_symbol s on d.kernel_id = s.id where p.name = ? group by 1""", (counter,))
    out = {}
    for kn, n, v in rows:
        short = function_name(kn)
        c = out.setdefault(short, [0, 0.0])
        c[0] += n; c[1] += v
    return out

if __name__ == "__main__":
    rd, wr = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    res = {}
    names = NAMES
    what = "`python bench.py --steps 1 --warmup 0 --no-cpu-baseline`"
    if len(sys.argv) > 3:  # third argument: comma-separated kernel functions of another command (fourth: its description)
        names = {k: k for k in sys.argv[3].split(",")}
        what = sys.argv[4] if len(sys.argv) > 4 else "the given command"
    for fn, launch in names.items():
        if fn in rd and fn in wr:
            fetch = 2.0 * rd[fn][1] * 1024.0 / rd[fn][0]
            write = wr[fn][1] * 1024.0 / wr[fn][0]
            res[launch] = {"kernel": fn, "launches_sampled": rd[fn][0], "fetch_bytes_per_launch": round(fetch),
                           "write_bytes_per_launch": round(write), "bytes_per_launch": round(fetch + write)}
    print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of "
                                + what + "; FETCH_SIZE x 2 (gfx950), KiB -> bytes",
                      "kernels": res}, indent=1))
