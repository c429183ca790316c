"""Groups the rom_kernel / dyn_kernel dispatches of a scripts/alloc_states.py run under rocprofv3 --pmc by allocation (ten evaluations
each) and prints mean duration and mean counter values per allocation.  Usage: alloc_states_summary.py <output dir>"""
import collections, csv, glob, sys
f = max(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True), key=lambda p: p)
rows = list(csv.DictReader(open(f)))
for kern in ("rom_kernel", "dyn_kernel"):
    disp = collections.OrderedDict()
    for r in rows:
        if kern in r["Kernel_Name"]:
            d = disp.setdefault(int(r["Dispatch_Id"]), {"dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(disp)
    names = sorted({k for d in disp.values() for k in d if k != "dur"})
    print(kern, "dispatches", len(ids), "counters", names)
    for a in range(len(ids) // 10):
        grp = [disp[i] for i in ids[10 * a + 2:10 * a + 10]]   # (the first two evaluations of an allocation: warm-up)
        line = "  allocation %d: %.1f us" % (a, sum(d["dur"] for d in grp) / len(grp) / 1e3)
        for n in names:
            line += "  %s %.4g" % (n.replace("TCC_", "").replace("_sum", ""), sum(d.get(n, 0.0) for d in grp) / len(grp))
        print(line)
