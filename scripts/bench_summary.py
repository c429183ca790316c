"""One-screen summary of a bench.py JSON line.  Usage: python scripts/bench_summary.py gpurun_out/.../bench.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("headline %.3f M cb/s  %.3f ms/step  repeats %s" % (d["value"] / 1e6, d["ms_per_step"], ["%.3f" % v for v in d.get("ms_per_step_repeats", [])]))
print("  dominant %s %.3f ms frac %.3f   path frac %.3f   kernels %s" % (r["kernel"], r["kernel_ms"], r["frac"], r["path"]["frac"],
      {k: round(v, 3) for k, v in r["path"]["kernel_ms"].items()}))
for leg in ("timings_c3", "all_sets_c3"):
    if leg in d:
        t = d[leg]
        print("%s %.3f M cb/s  path %.3f  %s" % (leg, t["value"] / 1e6, t["roofline"]["path"]["frac"], {k: round(v, 3) for k, v in t["roofline"]["path"]["kernel_ms"].items()}))
if "values_c3" in d:
    v = d["values_c3"]
    print("values_c3 %.2f M cb/s  %.3f ms/step (%.3f without events)  hbm frac %.3f  %s" % (v["value"] / 1e6, v["ms_per_step"], v["ms_per_step_without_events"], v["hbm"]["frac"], {k: round(x, 3) for k, x in v["kernel_ms"].items()}))
if "scale_c5" in d:
    c = d["scale_c5"]
    print("scale_c5 %.3f M cb/s  %.1f us/step  %.2f TB/s   planner %.1f us/step (%.1f decision left on device)" % (
        c["value"] / 1e6, c["ms_per_step"] * 1e3, c["path_GBps"] / 1e3, c["planner"]["ms_per_step"] * 1e3, c["planner"].get("ms_per_step_decision_left_on_device", 0) * 1e3))
    if c.get("shard_curve"):
        for w, e in c["shard_curve"]["world"].items():
            print("  world %s: projected x%.2f  %s" % (w, e["projected_speedup_no_overhead"], [(s["rank"], s["candidates"], round(s["us_per_step"], 1), round(s["frac"], 3)) for s in e["shards"]]))
if "cpu_baseline" in d:
    print("cpu %.1f cb/s (1 core), %.0f on %d" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"]["value"], d["cpu_baseline"]["all_cores"]["cores"]))
