"""B=1 latency of the C3 callback (device-resident and host-buffer variants) -- numbers for DESIGN.md section 6."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import towr_amd as ta
from bench import build_case, perturbed_inputs

model = ta.model_preset("anymal", "flat")
for sets, name in ((27, "hot path"), (63, "default list"), (127, "default + timings")):
    sched, params, S = build_case(ta, model, constraint_sets=sets)
    for B in (1, 64):
        batch = ta.Batch([S], [0] * B, device=0)
        xh = perturbed_inputs(S, model, B, 0).reshape(-1)
        x = torch.from_numpy(xh).cuda()
        g = torch.empty(int(batch.g_off[-1]), dtype=torch.float64, device="cuda")
        j = torch.empty(int(batch.jac_off[-1]), dtype=torch.float64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(20):
            batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
        torch.cuda.synchronize()
        n = 500
        t0 = time.perf_counter()
        for _ in range(n):
            batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
        torch.cuda.synchronize()
        dev_us = (time.perf_counter() - t0) / n * 1e6
        # one call, synchronised each time (what an Ipopt iteration sees with device-resident buffers)
        t0 = time.perf_counter()
        for _ in range(200):
            batch.eval_device(x.data_ptr(), g.data_ptr(), j.data_ptr(), ta.EVAL_BOTH, st)
            torch.cuda.synchronize()
        sync_us = (time.perf_counter() - t0) / 200 * 1e6
        for _ in range(5):
            batch.eval_host(xh)
        t0 = time.perf_counter()
        for _ in range(100):
            batch.eval_host(xh)
        host_us = (time.perf_counter() - t0) / 100 * 1e6
        px, pg, pj = batch.host_buffers()
        px[:] = xh
        for _ in range(5):
            batch.eval_host_pinned()
        t0 = time.perf_counter()
        for _ in range(100):
            batch.eval_host_pinned()
        pin_us = (time.perf_counter() - t0) / 100 * 1e6
        g2, j2 = batch.eval_host(xh)
        assert np.array_equal(g2, pg) and np.array_equal(j2, pj)
        # what the ifopt adapter pays per Ipopt callback (page-locked buffers): eval_g = values only, eval_jac_g on a known
        # x = Jacobian only, against one evaluation of both (what every new x cost before round 4)
        per = {}
        for what, flags in (("values", ta.EVAL_VALUES), ("jacobian", ta.EVAL_JACOBIAN)):
            for _ in range(5):
                batch.eval_host_pinned(flags)
            t0 = time.perf_counter()
            for _ in range(100):
                batch.eval_host_pinned(flags)
            per[what] = (time.perf_counter() - t0) / 100 * 1e6
        print("%-18s B=%-3d n=%d nnz=%d: back-to-back %.1f us/call, synchronised %.1f us/call, host buffers (H2D+eval+D2H) "
              "pageable %.1f us/call, page-locked %.1f us/call; page-locked eval_g (values only) %.1f us, eval_jac_g (Jacobian "
              "only) %.1f us" % (name, B, S.n, S.nnz, dev_us, sync_us, host_us, pin_us, per["values"], per["jacobian"]), flush=True)


# The ifopt boundary seen from the host: ANYmal, towr's default list, 19 constraint sets x 10 variable
# sets = 209 GetValues / FillJacobianBlock requests per Ipopt iteration, variable sets with the cost shape of towr's
# NodesVariables::GetValues (nodes_variables.cc:52-62: a map lookup + a vector copy per index).  Host microseconds spent
# finding out whether x moved and reading it, beside the device microseconds (twr_batch_eval_host, page-locked buffers).
import json, subprocess
from tests.test_ifopt_adapter import EXE, _build
_build()
for mode in ("push", "poll", "strict"):
    r = subprocess.run([EXE, "--quadruped", mode, "200"], capture_output=True, text=True, timeout=300)
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    print("ifopt adapter, ANYmal default list, x-change %-28s: %5.1f variable-set reads, host %7.1f us, device (upload + kernels + "
          "download) %6.1f us per iteration; one read of all of x %.1f us, i.e. round 4's read-on-every-request rule %.0f us"
          % (rep["mode"], rep["variable_set_reads_per_iteration"], rep["host_change_detection_us_per_iteration"],
             rep["device_eval_host_us_per_iteration"], rep["read_all_sets_once_us"], rep["round4_rule_us_per_iteration"]), flush=True)
