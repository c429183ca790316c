#!/usr/bin/env python3
"""Condense one scripts/profile_r04.sh run (or a round-3 profile_r03.sh run) into the files that are committed under
profiles/: <tag>_kernel_stats.csv, <tag>_timings_kernel_stats.csv, <tag>_sweep_kernel_stats.csv, <tag>_all_sets_kernel_stats.csv,
<tag>_pmc_summary.txt, traffic.json (+ the bench lines the traces were taken on)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402

out, tag = sys.argv[1], sys.argv[2]
prof = os.path.join(ROOT, "profiles")


def find(d, pat):
    f = glob.glob(os.path.join(out, d, "**", pat), recursive=True)   # (a merged output directory may hold older runs too)
    return max(f, key=os.path.getmtime) if f else None


def keep_twr(src, dst):
    with open(src) as f, open(dst, "w", newline="") as o:
        r = csv.reader(f)
        w = csv.writer(o)
        w.writerow(next(r))
        for row in r:
            if any("twr::" in c for c in row):
                w.writerow(row)


for d, name in (("ktrace", ""), ("ktrace_t", "_timings"), ("ktrace_s", "_sweep"), ("ktrace_a", "_all_sets"), ("ktrace_v", "_values")):
    src = find(d, "*kernel_stats.csv")
    if not src:
        continue
    keep_twr(src, os.path.join(prof, tag + name + "_kernel_stats.csv"))
    for cand in ("bench_%s.json" % d, {"ktrace": "bench_under_rocprof.json", "ktrace_t": "bench_timings_under_rocprof.json"}.get(d, "")):
        if cand and os.path.exists(os.path.join(out, cand)):
            shutil.copy(os.path.join(out, cand), os.path.join(prof, tag + name + "_bench_under_rocprof.json"))
            break


def counters(d):
    acc = collections.defaultdict(list)
    f = find(d, "*counter_collection.csv")
    if not f:
        return {}
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "twr::" in kn:
            short = kn.split("twr::")[1].split("(")[0]
            acc[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def values_step_valu(d):
    """SQ_INSTS_VALU of the ONE-launch values-only step (eval_values_kernel), mean per dispatch."""
    c = counters(d)
    v = [val for (k, name), val in c.items() if k.startswith("eval_values_kernel") and name == "SQ_INSTS_VALU"]
    return v[0] if v else None


lines = ["# rocprofv3 --pmc passes at kernel sources %s (mean per launch; default bench = C3, 8192 problems)" % kernel_source_hash()]
allc = {}
for d in ("sq1", "sq2", "tcc1", "tcc2"):
    c = counters(d)
    allc.update(c)
    for k, v in sorted(c.items()):
        lines.append("%-22s %-26s %.6g" % (k[0], k[1], v))
lines.append("# --sets timings --batch 2048")
tim = {}
for d in ("sq1_t", "sq2_t", "tcc1_t", "tcc2_t"):
    c = counters(d)
    tim.update(c)
    for k, v in sorted(c.items()):
        lines.append("%-22s %-26s %.6g" % (k[0], k[1], v))
lines.append("# --workload sweep --batch 1024 (BASELINE C5 on one GPU: every candidate its own tables)")
swp = {}
for d in ("tcc1_s", "tcc2_s"):
    c = counters(d)
    swp.update(c)
    for k, v in sorted(c.items()):
        lines.append("%-22s %-26s %.6g" % (k[0], k[1], v))
for d in ("sq1_s",):
    c = counters(d)
    for k, v in sorted(c.items()):
        lines.append("%-22s %-26s %.6g" % (k[0], k[1], v))
lines.append("# --sets all (towr's whole default constraint list, 8192 problems)")
alls = {}
for d in ("tcc1_a", "tcc2_a"):
    c = counters(d)
    alls.update(c)
    for k, v in sorted(c.items()):
        lines.append("%-22s %-26s %.6g" % (k[0], k[1], v))
lines.append("# default bench incl. the values-only leg (eval_values_kernel resp. values_flat_kernel, 8192 problems)")
vals = {}
for d in ("sq_v",):
    c = {k: v for k, v in counters(d).items() if "values" in k[0]}
    vals.update(c)
    for k, v in sorted(c.items()):
        lines.append("%-22s %-26s %.6g" % (k[0], k[1], v))
lines.append("# derived")
for k, src in [(k_, allc) for k_ in sorted({k[0] for k in allc})] + [(k_, tim) for k_ in sorted({k[0] for k in tim})] + [(k_, vals) for k_ in sorted({k[0] for k in vals})]:
    g = lambda n: src.get((k, n))
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        lines.append("%-22s LDS bank conflict / idx active = %.3f" % (k, g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
    if g("SQ_ACTIVE_INST_VALU") is not None and g("SQ_WAVE_CYCLES"):
        lines.append("%-22s VALU active / wave cycles = %.3f, waiting %.3f, issue stall %.3f" % (
            k, g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), (g("SQ_WAIT_ANY") or 0) / g("SQ_WAVE_CYCLES"),
            (g("SQ_WAIT_INST_ANY") or 0) / g("SQ_WAVE_CYCLES")))
open(os.path.join(prof, tag + "_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")


def traffic(c):
    """HBM bytes per launch: WRITE_SIZE + 2 * FETCH_SIZE (KiB; gfx950 FETCH_SIZE reports half of wide streaming reads,
    MI355X_MICROARCH.md section HBM)."""
    o = {}
    for k in sorted({k[0] for k in c}):
        w, f = c.get((k, "WRITE_SIZE")), c.get((k, "FETCH_SIZE"))
        if w is not None and f is not None:
            o["twr::" + k] = (w + 2.0 * f) * 1024.0
    return o


json.dump({"workload": "C3", "problems_per_gpu": 8192, "kernel_source_sha256": kernel_source_hash(),
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (WRITE_SIZE + 2*FETCH_SIZE) KiB",
           "hbm_bytes_per_launch": traffic(allc),
           "timings_2048": {"hbm_bytes_per_launch": traffic(tim)},
           "sweep_1024": {"hbm_bytes_per_launch": traffic(swp)},
           "all_sets_8192": {"hbm_bytes_per_launch": traffic(alls)},
           "values_8192": {"problems_per_gpu": 8192, "valu_insts_per_step": values_step_valu("sq_v")}},
          open(os.path.join(prof, "traffic.json"), "w"), indent=1)
print("\n".join(lines[-12:]))
