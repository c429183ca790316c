#!/usr/bin/env python3
"""A/B of library builds on one box: scripts/ab.py lib1.so[:ENV=VAL,...] lib2.so ... [-- bench.py arguments]
(kernel times of the default bench, or of the workload the extra arguments select, e.g. -- --workload sweep --batch 1024)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = sys.argv[1:]
EXTRA = []
if "--" in ARGS:
    EXTRA = ARGS[ARGS.index("--") + 1:]
    ARGS = ARGS[:ARGS.index("--")]
REPS = int(os.environ.get("AB_REPS", "2"))   # AB_REPS=5: more alternations (run-to-run noise on one box is +-2 %)
for rep in range(REPS):
    for spec in ARGS:
        lib, _, envs = spec.partition(":")
        env = dict(os.environ, TWR_AMD_LIB=os.path.join(ROOT, "towr_amd", lib))
        for kv in filter(None, envs.split(",")):
            k, v = kv.split("=")
            env[k] = v
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "3", "--no-cpu-baseline",
                            "--no-scale-c5", "--no-timings-c3"] + EXTRA, env=env, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            k = d["roofline"]["path"]["kernel_ms"]
            print("%-40s %.3f M cb/s  " % (spec, d["value"] / 1e6) + "  ".join("%s %.3f" % (n.split("::")[1][:10], v) for n, v in k.items()), flush=True)
        except Exception as e:  # noqa: BLE001
            print(spec, "failed", e, r.stderr[-400:], flush=True)
