import sys, json, subprocess
sys.path.insert(0, ".")
import bench
for order in ((10.0, 5.0, 2.0, 0.0), (0.0, 0.0, 0.0, 0.0), (10.0, 10.0, 10.0, 10.0)):
    bench.PLACEMENT_BALLAST_GB = order
    sys.argv = ["bench.py", "--no-timings-c3", "--no-cpu-baseline", "--no-scale-c5", "--no-values-c3"]
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    print(order, round(d["value"] / 1e6, 3), [round(t["ms_per_step"], 3) for t in d["config"]["output_placement"]["tries"]], "kept", d["config"]["output_placement"]["kept"], flush=True)
