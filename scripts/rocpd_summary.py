"""Summarise rocprofv3 rocpd (sqlite) outputs into the text files kept under profiles/.

    python scripts/rocpd_summary.py stats <results.db>          # == --kernel-trace --stats summary
    python scripts/rocpd_summary.py pmc <results.db> [<results.db> ...]   # per-kernel mean counter values
    python scripts/rocpd_summary.py json <lib stamp file> [--mode f16|precise] [--into old.json] <results.db> [...]   # the GEMM classes' counter figures as
                                                                              # profiles/pmc_current.json (bench.py load_pmc)
"""
import sqlite3
import sys
from collections import defaultdict


def stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print(f"# rocprofv3 --kernel-trace --stats  ({db})")
    print(f"{'kernel':<70} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}")
    for name, calls, total, avg, pct in rows:
        print(f"{name[:70]:<70} {calls:>6} {total:>12.1f} {avg:>10.2f} {pct:>6.2f}")
    # per-shape split of templated GEMM kernels: group by (name, grid)
    print("\n# per (kernel, grid) split")
    q = "select name, grid_x, count(*), avg(duration)/1000.0, min(duration)/1000.0, max(duration)/1000.0, vgpr_count, lds_size from kernels group by name, grid_x order by sum(duration) desc"
    print(f"{'kernel':<50} {'grid_x':>9} {'calls':>6} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'vgpr':>5} {'lds':>7}")
    for name, gx, n, avg, mn, mx, vg, lds in con.execute(q):
        print(f"{name[:50]:<50} {gx:>9} {n:>6} {avg:>9.2f} {mn:>9.2f} {mx:>9.2f} {vg:>5} {lds:>7}")
    # one symbol, two problem shapes with the same grid (the persistent residual GEMM serves the attention-output
    # projection, K = 768, and FFN-2, K = 3072): split bimodal symbols like the pmc summary does, so that the per-use
    # average can be compared with bench.py's HIP-event figure for that kernel class
    by_sym = defaultdict(list)
    for name, gx, d in con.execute("select name, grid_x, duration from kernels"):
        by_sym[(name, gx)].append(d)
    first = True
    for (name, gx), ds in sorted(by_sym.items(), key=lambda kv: -sum(kv[1])):
        ds = sorted(ds)
        lo, hi = ds[len(ds) // 10], ds[-1 - len(ds) // 10]
        if lo > 0 and hi / lo > 1.6:
            cut = (lo * hi) ** 0.5
            if first:
                print("\n# bimodal symbols split by dispatch duration (geometric mean of the 10th / 90th percentile as the cut)")
                print(f"{'kernel':<58} {'calls':>6} {'avg_us':>9} {'min_us':>9} {'max_us':>9}")
                first = False
            for tag, part in (("[long]", [d for d in ds if d > cut]), ("[short]", [d for d in ds if d <= cut])):
                print(f"{(name[:50] + ' ' + tag):<58} {len(part):>6} {sum(part) / len(part) / 1000.0:>9.2f} {part[0] / 1000.0:>9.2f} {part[-1] / 1000.0:>9.2f}")


def pmc(dbs):
    agg = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    rows = []
    for db in dbs:
        con = sqlite3.connect(db)
        rows += list(con.execute("select kernel_name, grid_size_x, counter_name, value, duration from counters_collection"))
    # one kernel symbol can serve two problem shapes with the same grid (the persistent residual GEMM: K = 768 for the
    # attention-output projection, K = 3072 for FFN-2): split such a symbol into a "short" and a "long" class when its
    # dispatch durations are clearly bimodal
    by_sym = defaultdict(list)
    for name, gx, cname, val, d in rows:
        by_sym[(name, gx)].append(d)
    cut = {}
    for key, ds in by_sym.items():
        ds = sorted(ds)
        lo, hi = ds[len(ds) // 10], ds[-1 - len(ds) // 10]
        if lo > 0 and hi / lo > 1.6:
            cut[key] = (lo * hi) ** 0.5
    for name, gx, cname, val, d in rows:
        key = (name, gx)
        if key in cut:
            key = (name + (" [long]" if d > cut[(name, gx)] else " [short]"), gx)
        agg[key][cname].append(val)
        dur[key].append(d)
    counters = sorted({c for v in agg.values() for c in v})
    print("# rocprofv3 --kernel-trace --pmc  (mean per dispatch; one pass per counter group; FETCH_SIZE/WRITE_SIZE in KiB as reported)")
    print("# NOTE gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md §HBM):")
    print("#      fetch_corrected_MB = 2 * FETCH_SIZE KiB / 1024; WRITE_SIZE is uncalibrated (reported as is)")
    hdr = f"{'kernel':<52} {'grid':>8} {'n':>4} {'prof_us':>8}" + "".join(f" {c[:22]:>22}" for c in counters)
    if "FETCH_SIZE" in counters:
        hdr += f" {'fetch_corr_MB':>14}"
    derive = "GRBM_GUI_ACTIVE" in counters and "SQ_VALU_MFMA_BUSY_CYCLES" in counters
    if derive:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = shader cycles of the dispatch -> effective clock = that / wall
        # (MI355X_MICROARCH.md "DVFS give-back"); SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles per v_mfma_32x32x16 summed over the
        # 256 CUs x 4 SIMDs -> busy fraction of the matrix pipes = it / (1024 x shader cycles)
        hdr += f" {'eff_clock_GHz':>14} {'mfma_busy_%':>12}"
    print(hdr)
    for key, c in sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]])):
        name, gx = key
        if name.startswith("__amd"):
            continue
        n = max(len(v) for v in c.values())
        line = f"{(name[:44] + name[name.rfind(' ['):] if name.endswith(']') else name[:52]):<52} {gx:>8} {n:>4} {sum(dur[key]) / len(dur[key]) / 1000.0:>8.1f}"
        for cn in counters:
            v = c.get(cn, [])
            line += f" {(sum(v) / len(v) if v else float('nan')):>22.4g}"
        if "FETCH_SIZE" in counters:
            f = c.get("FETCH_SIZE", [])
            line += f" {(2 * sum(f) / len(f) / 1024 if f else float('nan')):>14.1f}"
        if derive:
            g, m = c.get("GRBM_GUI_ACTIVE", []), c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])
            cyc = sum(g) / len(g) / 8 if g else float("nan")
            us = sum(dur[key]) / len(dur[key]) / 1000.0
            line += f" {cyc / us / 1000.0:>14.3f} {(100.0 * (sum(m) / len(m)) / (1024 * cyc) if m and cyc else float('nan')):>12.2f}"
        print(line)


# kernel symbol (+ duration class for the residual GEMM, which serves two shapes) -> bench.py kernel class; X = 0 (MV_F16) / 1 (MV_F16X8)
def class_of(mode):
    x = "1" if mode == "precise" else "0"
    return ((f"gemm_pp_kernel<1, 1, {x}>", None, "gemm_qkv"), (f"gemm_pp_kernel<3, 1, {x}>", None, "gemm_ffn1_gelu"),
            (f"gemm_pp_kernel<8, 0, {x}>", "long", "gemm_ffn2"), (f"gemm_pp_kernel<8, 0, {x}>", "short", "gemm_attn_out"),
            (f"attention_v2_kernel<4, 1, {x}>", None, "attention"))  # S = 256 only: <2, 4, X> (S = 512) also runs here, for the anchor bank


def pmc_json(stamp_file, dbs, mode="f16", into=None):
    """Mean per-dispatch counters of the encoder's kernel classes at the default bench workload -> JSON with the stamp of the
    library that was profiled: traffic_bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; the gfx950 FETCH correction of
    MI355X_MICROARCH.md), mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), effective clock =
    GRBM_GUI_ACTIVE / 8 / profiled duration."""
    import json

    rows = []
    for db in dbs:
        con = sqlite3.connect(db)
        rows += list(con.execute("select kernel_name, grid_size_x, counter_name, value, duration from counters_collection"))
    by_sym = defaultdict(list)
    for name, gx, cname, val, d in rows:
        by_sym[name].append(d)
    out = {}
    for sym, dclass, cls in class_of(mode):
        names = [n for n in by_sym if sym in n]
        if not names:
            continue
        name = names[0]
        ds = sorted(by_sym[name])
        cut = None
        if dclass:
            lo, hi = ds[len(ds) // 10], ds[-1 - len(ds) // 10]
            cut = (lo * hi) ** 0.5
        vals, durs = defaultdict(list), []
        for n, gx, cname, val, d in rows:
            if n != name or (cut and ((d > cut) != (dclass == "long"))):
                continue
            vals[cname].append(val)
            durs.append(d)
        if not durs:
            continue
        mean = {k: sum(v) / len(v) for k, v in vals.items()}
        us = sum(durs) / len(durs) / 1000.0
        rec = {"profiled_avg_us": round(us, 2), "dispatches": len(durs) // max(1, len(vals))}
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            rec["traffic_bytes"] = int((2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024)
            rec["fetch_bytes_corrected"] = int(2 * mean["FETCH_SIZE"] * 1024)
            rec["write_bytes"] = int(mean["WRITE_SIZE"] * 1024)
        if "GRBM_GUI_ACTIVE" in mean:
            cyc = mean["GRBM_GUI_ACTIVE"] / 8
            rec["effective_clock_ghz"] = round(cyc / us / 1000.0, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
                rec["mfma_busy_frac"] = round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4)
        out[cls] = rec
    stamp = open(stamp_file).read().strip()
    doc = {"lib_stamp": stamp, "workload": "bench.py default (S=256, B=256, G=124, 12 layers), one batch in flight; one entry per compute dtype profiled",
           "source": "rocprofv3 --kernel-trace --pmc, one counter group per pass (scripts/gpu_pmc.sh)", "classes_by_mode": {}}
    if into:
        try:
            old = json.load(open(into))
            if old.get("lib_stamp") == stamp:  # same binary: add this mode's pass to the record
                doc["classes_by_mode"] = old.get("classes_by_mode", {})
        except (OSError, ValueError):
            pass
    doc["classes_by_mode"][mode] = out
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "json":  # json <stamp> [--mode f16|precise] [--into existing.json] <dbs...>
        args, mode, into = sys.argv[3:], "f16", None
        while args and args[0].startswith("--"):
            if args[0] == "--mode":
                mode = args[1]
            elif args[0] == "--into":
                into = args[1]
            args = args[2:]
        pmc_json(sys.argv[2], args, mode, into)
    else:
        pmc(sys.argv[2:])
