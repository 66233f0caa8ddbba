#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>/ directory (tools/profile_gpu.sh) into profiles/<tag>_summary.md:
per-kernel time table from the kernel trace and per-dispatch PMC averages for the MLP kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict


def rows(pattern):
    for f in glob.glob(pattern, recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


def dur_ms(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6


def frame_rows(pattern):
    """MLP-kernel rows of the full-frame launches only: bench.py's one-time setup renders 256 and 4096 rays before
    the steps (code-object load), which must not dilute per-launch means.  Full-frame = at least half as long as the
    longest MLP dispatch of the pass."""
    mlp = [r for r in rows(pattern) if "pny_mlp_kernel" in r.get("Kernel_Name", "") or "pny_mlp_h2_kernel" in r.get("Kernel_Name", "")]
    if not mlp:
        return []
    longest = max(dur_ms(r) for r in mlp)
    return [r for r in mlp if dur_ms(r) >= 0.5 * longest]


def main(tag):
    src = os.path.join("gpurun_out", "prof_" + tag)
    out = [f"# rocprofv3 summary `{tag}` (bench.py --steps 2 --warmup 1 --cpu-rays 0, 1x MI355X)\n"]
    # ---- kernel trace
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    trace_pat = os.path.join(src, "trace", "**", "*kernel_trace.csv")
    frame = frame_rows(trace_pat)
    for r in list(rows(trace_pat)) + [dict(r, Kernel_Name=r["Kernel_Name"].split("(")[0].split("<")[0] + ", FULL-FRAME launches only (the bench steps)") for r in frame]:
        name = r["Kernel_Name"].split("(")[0]
        dur = dur_ms(r)
        a = agg[name]
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    tot = sum(a[1] for k, a in agg.items() if "FULL-FRAME" not in k) or 1.0
    out.append("## Kernel trace (`--kernel-trace --stats`)\n")
    out.append("| kernel | calls | total ms | avg ms | min ms | max ms | % |")
    out.append("|---|---|---|---|---|---|---|")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append(f"| `{name[:70]}` | {a[0]} | {a[1]:.3f} | {a[1]/a[0]:.3f} | {a[2]:.3f} | {a[3]:.3f} | {100*a[1]/tot:.2f} |")
    # ---- PMC passes
    out.append("\n## PMC counters, full-frame pny_mlp_kernel dispatches only (separate `--pmc` passes; per-dispatch mean)\n")
    out.append("| pass | counter | mean per dispatch | dispatches |")
    out.append("|---|---|---|---|")
    for p in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(p):
            continue
        acc = defaultdict(lambda: [0, 0.0])
        for r in frame_rows(os.path.join(p, "**", "*counter_collection.csv")):
            a = acc[r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        for cname, a in sorted(acc.items()):
            out.append(f"| {os.path.basename(p)} | {cname} | {a[1]/max(a[0],1):.6g} | {a[0]} |")
    # ---- derived: the clock the chip held and the matrix pipe's busy fraction (profiles/r03_kernel_experiments.md section 1)
    vals = {}
    for p in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        acc2 = defaultdict(lambda: [0, 0.0])
        for r in frame_rows(os.path.join(p, "**", "*counter_collection.csv")):
            a = acc2[r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        for k, a in acc2.items():
            vals[k] = a[1] / max(a[0], 1)
    if frame and "GRBM_GUI_ACTIVE" in vals and "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
        avg_ms = sum(dur_ms(r) for r in frame) / len(frame)
        cyc = vals["GRBM_GUI_ACTIVE"] / 8.0
        busy = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        out.append("\n## Derived (one formula, profiles/r03_kernel_experiments.md section 1)\n")
        out.append("* cycles per launch = GRBM_GUI_ACTIVE / 8 XCDs = %.4g; sustained clock = cycles / %.3f ms (kernel trace, full-frame "
                   "launches) = **%.2f GHz** (profiled passes run 1-3 %% off the plain run)" % (cyc, avg_ms, cyc / avg_ms / 1e6))
        out.append("* matrix-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles) = %.4g / (1024 x %.4g) = **%.3f**" %
                   (vals["SQ_VALU_MFMA_BUSY_CYCLES"], cyc, busy))
        out.append("* roofline.frac = matrix-busy x clock / 2.4 GHz = %.3f (the bench line's `frac` from HIP events measures the same thing)"
                   % (busy * cyc / avg_ms / 1e6 / 2.4))
    os.makedirs("profiles", exist_ok=True)
    # fabric-side bytes per MLP launch for bench.py's roofline.traffic (MI355X_MICROARCH.md, HBM section:
    # (FETCH_SIZE + WRITE_SIZE) KB, FETCH_SIZE doubled on gfx950 for 16-B/lane streaming reads; the
    # counters sit on the L2's memory side, so Infinity-Cache hits are included: an upper bound on HBM bytes)
    import json
    per = {}
    names = set()
    for cname, pdir in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        tot = cnt = 0
        for r in frame_rows(os.path.join(src, pdir, "**", "*counter_collection.csv")):
            if r["Counter_Name"] == cname:
                tot += float(r["Counter_Value"])
                cnt += 1
                names.add(r["Kernel_Name"].split("(")[0])
        per[cname] = tot / cnt if cnt else None
    if per["FETCH_SIZE"] is not None and per["WRITE_SIZE"] is not None:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import kernel_source_fingerprint   # the sources the counted kernel is built from, as profiled
        traffic = {"tag": tag, "kernel_sources_sha256": kernel_source_fingerprint(), "kernels": sorted(names), "projected_latent": any(", true>" in n or "pny_mlp_h2_kernel" in n for n in names),
                   "f16x2": any("pny_mlp_h2_kernel" in n for n in names),
                   "fetch_kb_per_launch": per["FETCH_SIZE"], "write_kb_per_launch": per["WRITE_SIZE"],
                   "bytes_per_launch": (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, mean over the "
                             "MLP dispatches; (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (gfx950 "
                             "halves FETCH_SIZE for 16-B/lane streams); L2 memory-side requests, Infinity-Cache "
                             "hits included"}
        # bench.py reads mlp_traffic.json for the default line only: variant profiles (tag with a suffix after '_')
        # keep their own file
        tname = "mlp_traffic.json" if "_" not in tag else "mlp_traffic_%s.json" % tag
        with open(os.path.join("profiles", tname), "w") as fh:
            json.dump(traffic, fh, indent=1)
        out.append("\nroofline.traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = %.4g bytes per MLP launch "
                   "(written to profiles/%s)" % (traffic["bytes_per_launch"], tname))
    dst = os.path.join("profiles", tag + "_summary.md")
    with open(dst, "w") as fh:
        fh.write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
