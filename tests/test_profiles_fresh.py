"""The committed rocprofv3 summaries must belong to the kernels in this tree: profiles/traffic.json carries a hash of
the kernel sources and build flags (bench.kernel_source_hash), bench.py drops roofline.traffic when it differs -- this
test makes a stale profile a red CPU tier instead of a silent null."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_traffic_profile_matches_kernel_sources():
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        t = json.load(f)
    assert t["kernel_source_sha256"] == bench.kernel_source_hash(), \
        "kernel sources changed after profiles/ was taken: re-run scripts/profile_r05.sh (GPU box) and scripts/profile_summary.py gpurun_out/prof_r05 r05"
    assert bench.traffic_from_profile("C3", "twr::rom_kernel", t["problems_per_gpu"]) > 0
    assert bench.traffic_from_profile("C3+timings", "twr::dyn_phase_kernel", 2048) > 0
    assert sum(t["sweep_1024"]["hbm_bytes_per_launch"].values()) > 0
