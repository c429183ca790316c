// The multi-GPU sweep for the reference's actual caller, which is C++ (fpowr/src/footstep_plan_server.cc:132-268 solves
// ONE hard-coded gait per goal; a planner that sweeps contact schedules would sit in its place): the C++ twin of
// towr_amd/dist.py + bench.py's scale_c5, through include/towr_amd.h and librccl only -- no Python, no torch.
//
//   one host thread per device ("rank"), all in one process (what a ROS node like fpowr's server would do):
//     1. ONE collective at start: ncclBroadcast of the POD robot/terrain model (sizeof(twr_model) = 216 bytes) from rank 0
//     2. every rank enumerates the same candidate list, weighs it by bytes (twr_candidate_bytes: 8 (n + m + nnz)),
//        cuts contiguous shards (twr_shard_bounds) and builds ONLY its own shard (twr_structure_create_many)
//     3. per-GPU independent batches: constraint values (+ Jacobian with --jacobian), per-candidate scores on the device
//     4. the one optional exchange of SURVEY 8e: ncclAllGather of the 16 scores per candidate (equal padded blocks),
//        arg-min on EVERY rank -> one decision, identical everywhere
//   --no-collective: the same sweep without RCCL at all -- the threads of one process share host memory, so the model is
//        a plain struct copy and the scores are copied device -> host into one table ("one process, N
//        devices").  Also the way to rehearse several ranks on ONE device (--devices 0,0).
//   --rank R --world N --id-file PATH [--run-id S]: one PROCESS per device instead (ncclCommInitRank; rank 0 writes the
//        ncclUniqueId to PATH[.S], the others wait for it) -- for launchers that start one process per GPU.  A launcher
//        should pass a fresh --run-id (its pid, a timestamp) to every rank: the id file of an EARLIER run can then never be
//        mistaken for this one's.  Without it ranks > 0 only accept a file that is younger than their own start (minus a
//        minute of launcher skew); rank 0 removes the file once ncclCommInitRank has returned (every rank has read it by then).
//   A rank that fails (out of memory, a HIP error, ...) while its peers sit in a collective: its thread aborts EVERY
//        communicator (ncclCommAbort), so the peers' collectives return an error instead of waiting for ever, and the
//        process exits non-zero.
//
// Build:  hipcc -std=c++17 -O2 -I include examples/sweep_multi_gpu.cc -L towr_amd -ltowr_amd -lrccl -pthread \
//               -Wl,-rpath,$PWD/towr_amd -o sweep_multi_gpu
// Usage:  sweep_multi_gpu [n_candidates=1024] [--devices 0,1,..] [--no-collective] [--jacobian] [--steps K]
// Prints "rank r: shard [lo, hi) ..." per rank and "best candidate <index> score <value>"; exits non-zero on any error
// or if the ranks disagree.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "towr_amd.h"

namespace {

struct Fail {
  std::string what;
};
#define TWR(call)                                                                             \
  do {                                                                                        \
    if ((call) != TWR_OK) throw Fail{std::string(#call) + ": " + twr_last_error()};           \
  } while (0)
#define HIP(call)                                                                             \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) throw Fail{std::string(#call) + ": " + hipGetErrorString(e_)};      \
  } while (0)
#define NCCL(call)                                                                            \
  do {                                                                                        \
    ncclResult_t r_ = (call);                                                                 \
    if (r_ != ncclSuccess) throw Fail{std::string(#call) + ": " + ncclGetErrorString(r_)};    \
  } while (0)

struct Options {
  int n_cand = 1024;
  std::vector<int> devices;    // one rank per entry (thread mode)
  bool collective = true, jacobian = false;
  int steps = 0;               // timed evaluations per rank after the decision (0: none)
  int fail_rank = -1;          // fault injection (tests)
  int rank = -1, world = 0;    // process mode
  std::string id_file, run_id;
};

struct Decision {
  int best = -1;
  double score = INFINITY;
  int lo = 0, hi = 0;
  double setup_s = 0, step_ms = 0;
  std::string error;
};

// the candidate list of SURVEY 8d (combo x total time x swing scale, K = 200 time nodes), the same on every rank
void Enumerate(const twr_model& model, int n_cand, std::vector<twr_schedule>& scheds, std::vector<twr_params>& params) {
  for (int combo = 0; combo < 5 && (int)scheds.size() < n_cand; ++combo)
    for (int i = 0; i < 8 && (int)scheds.size() < n_cand; ++i)
      for (int j = 0; j < 26 && (int)scheds.size() < n_cand; ++j) {
        const double T = 1.2 + 0.2 * i, scale = 0.80 + 0.016 * j;
        twr_schedule s;
        TWR(twr_gait_combo(model.n_ee, combo, T, scale, &s));
        twr_params p;
        TWR(twr_params_default(&p));
        p.dt_dynamic = p.dt_rom = T / (200 - 1.5);
        scheds.push_back(s);
        params.push_back(p);
      }
  if ((int)scheds.size() < n_cand) throw Fail{"only 1040 candidates are defined"};
}

// arg-min over the whole table in candidate order: summed inf-norm violation of the terrain (0), dynamic (1),
// range-of-motion (3) and force (4) families; NaN loses (towr_amd/dist.py best_candidate)
void ArgMin(const std::vector<double>& table, int n, Decision& d) {
  d.best = -1;
  d.score = INFINITY;
  for (int p = 0; p < n; ++p) {
    const double* s = table.data() + 16 * (size_t)p;
    double total = s[0] + s[2] + s[6] + s[8];
    if (std::isnan(total)) total = INFINITY;
    if (d.best < 0 || total < d.score) {
      d.score = total;
      d.best = p;
    }
  }
}

// One rank: device `device`, communicator `comm` (nullptr: no collective; then `shared_model` / `shared_table` are the
// process-wide host copies every thread reads / fills).
void RunRank(const Options& opt, int rank, int world, int device, ncclComm_t comm, const twr_model* shared_model,
             std::vector<double>* shared_table, Decision& out) {
  const auto t_start = std::chrono::steady_clock::now();
  // test hook: --fail-rank R makes rank R fail before its first collective (what an out-of-memory batch would do)
  if (opt.fail_rank == rank) throw Fail{"injected failure (--fail-rank)"};
  HIP(hipSetDevice(device));
  hipStream_t stream;
  HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

  // 1. the model: rank 0 owns it, everyone else receives it
  twr_model model;
  std::memset(&model, 0, sizeof(model));
  if (comm) {
    if (rank == 0) TWR(twr_model_preset(TWR_ROBOT_ANYMAL, TWR_TERRAIN_STAIRS, &model));
    void* d_model = nullptr;
    HIP(hipMalloc(&d_model, sizeof(twr_model)));
    HIP(hipMemcpyAsync(d_model, &model, sizeof(twr_model), hipMemcpyHostToDevice, stream));
    NCCL(ncclBroadcast(d_model, d_model, sizeof(twr_model), ncclUint8, 0, comm, stream));   // the ONE collective of the path
    HIP(hipMemcpyAsync(&model, d_model, sizeof(twr_model), hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    HIP(hipFree(d_model));
  } else {
    model = *shared_model;
  }

  // 2. shards by bytes; this rank's structures only
  std::vector<twr_schedule> scheds;
  std::vector<twr_params> params;
  Enumerate(model, opt.n_cand, scheds, params);
  const int threads = std::max(1, (int)std::thread::hardware_concurrency() / std::max(1, world));
  std::vector<int64_t> bytes(opt.n_cand);
  TWR(twr_candidate_bytes(&model, scheds.data(), params.data(), opt.n_cand, threads, bytes.data()));
  std::vector<double> weights(bytes.begin(), bytes.end());
  std::vector<int32_t> bounds(world + 1);
  TWR(twr_shard_bounds(weights.data(), opt.n_cand, world, bounds.data()));
  const int lo = bounds[rank], hi = bounds[rank + 1], B = hi - lo;
  std::vector<twr_structure*> structs(B, nullptr);
  TWR(twr_structure_create_many(&model, scheds.data() + lo, params.data() + lo, B, threads, structs.data()));
  std::vector<int32_t> map(B);
  for (int p = 0; p < B; ++p) map[p] = p;
  twr_batch* batch = nullptr;
  TWR(twr_batch_create(structs.data(), B, map.data(), B, device, &batch));
  std::vector<int64_t> x_off(B + 1), g_off(B + 1), j_off(B + 1);
  TWR(twr_batch_layout(batch, x_off.data(), g_off.data(), j_off.data()));

  // 3. x = the reference's initial guess of every candidate (nlp_formulation.cc:95-181), values (+ Jacobian), scores
  std::vector<double> x(x_off[B]);
  const double z0 = 0.5, lin0[3] = {0, 0, z0}, ang0[3] = {0, 0, 0}, lin1[3] = {2.0, 0, z0}, ang1[3] = {0, 0, 0};
  const double ee0[12] = {0.34, 0.19, 0, 0.34, -0.19, 0, -0.34, 0.19, 0, -0.34, -0.19, 0};
  for (int p = 0; p < B; ++p) TWR(twr_structure_initial_guess(structs[p], lin0, ang0, lin1, ang1, ee0, x.data() + x_off[p]));
  int n_max = 0;
  for (int r = 0; r < world; ++r) n_max = std::max(n_max, bounds[r + 1] - bounds[r]);
  double *d_x = nullptr, *d_g = nullptr, *d_j = nullptr, *d_block = nullptr, *d_table = nullptr;
  HIP(hipMalloc(&d_x, x.size() * sizeof(double)));
  HIP(hipMalloc(&d_g, g_off[B] * sizeof(double)));
  if (opt.jacobian || opt.steps > 0) HIP(hipMalloc(&d_j, j_off[B] * sizeof(double)));
  HIP(hipMalloc(&d_block, 16 * (size_t)n_max * sizeof(double)));     // this rank's scores, padded to the largest shard
  HIP(hipMemsetAsync(d_block, 0, 16 * (size_t)n_max * sizeof(double), stream));
  HIP(hipMemcpyAsync(d_x, x.data(), x.size() * sizeof(double), hipMemcpyHostToDevice, stream));
  out.setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  const int flags = (opt.jacobian ? TWR_EVAL_BOTH : TWR_EVAL_VALUES);
  TWR(twr_batch_eval(batch, d_x, d_g, opt.jacobian ? d_j : nullptr, flags, stream));
  TWR(twr_batch_score(batch, d_g, d_block, stream));

  // 4. one decision for all shards
  std::vector<double> table(16 * (size_t)opt.n_cand);
  if (comm) {
    HIP(hipMalloc(&d_table, 16 * (size_t)n_max * world * sizeof(double)));
    NCCL(ncclAllGather(d_block, d_table, 16 * (size_t)n_max, ncclDouble, comm, stream));
    std::vector<double> padded(16 * (size_t)n_max * world);
    HIP(hipMemcpyAsync(padded.data(), d_table, padded.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    for (int r = 0; r < world; ++r)   // drop the padding: candidate order
      std::memcpy(table.data() + 16 * (size_t)bounds[r], padded.data() + 16 * (size_t)n_max * r,
                  16 * sizeof(double) * (size_t)(bounds[r + 1] - bounds[r]));
    ArgMin(table, opt.n_cand, out);
  } else {
    // no collective: this rank's rows go straight into the process-wide host table; the caller takes the arg-min once
    // every thread has joined
    HIP(hipMemcpyAsync(shared_table->data() + 16 * (size_t)lo, d_block, 16 * sizeof(double) * (size_t)B, hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
  }
  out.lo = lo;
  out.hi = hi;

  // optional: the evaluation rate of this rank's shard (full callbacks, outputs resident in HBM)
  if (opt.steps > 0) {
    for (int i = 0; i < 3; ++i) TWR(twr_batch_eval(batch, d_x, d_g, d_j, TWR_EVAL_BOTH, stream));
    HIP(hipStreamSynchronize(stream));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < opt.steps; ++i) TWR(twr_batch_eval(batch, d_x, d_g, d_j, TWR_EVAL_BOTH, stream));
    HIP(hipStreamSynchronize(stream));
    out.step_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / opt.steps;
  }

  twr_batch_destroy(batch);
  for (twr_structure* s : structs) twr_structure_destroy(s);
  (void)hipFree(d_x); (void)hipFree(d_g); (void)hipFree(d_block);
  if (d_j) (void)hipFree(d_j);
  if (d_table) (void)hipFree(d_table);
  (void)hipStreamDestroy(stream);
}

std::vector<int> ParseList(const char* s) {
  std::vector<int> v;
  for (const char* p = s; *p;) {
    v.push_back(std::atoi(p));
    while (*p && *p != ',') ++p;
    if (*p == ',') ++p;
  }
  return v;
}

}  // namespace

int main(int argc, char** argv) {
  Options opt;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--no-collective") opt.collective = false;
    else if (a == "--jacobian") opt.jacobian = true;
    else if (a == "--devices" && i + 1 < argc) opt.devices = ParseList(argv[++i]);
    else if (a == "--steps" && i + 1 < argc) opt.steps = std::atoi(argv[++i]);
    else if (a == "--rank" && i + 1 < argc) opt.rank = std::atoi(argv[++i]);
    else if (a == "--world" && i + 1 < argc) opt.world = std::atoi(argv[++i]);
    else if (a == "--id-file" && i + 1 < argc) opt.id_file = argv[++i];
    else if (a == "--run-id" && i + 1 < argc) opt.run_id = argv[++i];
    else if (a == "--fail-rank" && i + 1 < argc) opt.fail_rank = std::atoi(argv[++i]);
    else if (a[0] != '-') opt.n_cand = std::atoi(argv[i]);
    else {
      std::fprintf(stderr, "unknown option %s\n", argv[i]);
      return 2;
    }
  }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
    std::fprintf(stderr, "no HIP device visible: towr_amd has no CPU fallback\n");
    return 3;
  }
  try {
    // ---- one process per device (a launcher started `world` of us)
    if (opt.rank >= 0) {
      if (opt.world < 1 || opt.rank >= opt.world || opt.id_file.empty()) throw Fail{"--rank needs --world and --id-file"};
      const int device = opt.devices.empty() ? opt.rank % n_dev : opt.devices[0];
      HIP(hipSetDevice(device));
      const time_t started = time(nullptr);
      const std::string id_path = opt.run_id.empty() ? opt.id_file : opt.id_file + "." + opt.run_id;
      ncclUniqueId id;
      if (opt.rank == 0) {
        (void)std::remove(id_path.c_str());   // whatever an earlier run left behind
        NCCL(ncclGetUniqueId(&id));
        std::ofstream f(id_path + ".tmp", std::ios::binary);
        f.write(reinterpret_cast<const char*>(&id), sizeof(id));
        f.close();
        if (std::rename((id_path + ".tmp").c_str(), id_path.c_str()) != 0) throw Fail{"cannot publish the RCCL id"};
      } else {
        for (int tries = 0;; ++tries) {
          struct stat st;
          // without a --run-id the name may be an earlier run's: only a file written around this run's start counts
          const bool fresh = stat(id_path.c_str(), &st) == 0 && (!opt.run_id.empty() || st.st_mtime >= started - 60);
          std::ifstream f(id_path, std::ios::binary);
          if (fresh && f.read(reinterpret_cast<char*>(&id), sizeof(id))) break;
          if (tries > 600) throw Fail{"no RCCL id after 60 s (" + id_path + ")"};
          std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
      }
      ncclComm_t comm;
      NCCL(ncclCommInitRank(&comm, opt.world, id, opt.rank));
      if (opt.rank == 0) (void)std::remove(id_path.c_str());   // init is collective: every rank has read it
      Decision d;
      try {
        RunRank(opt, opt.rank, opt.world, device, comm, nullptr, nullptr, d);
      } catch (const Fail&) {
        (void)ncclCommAbort(comm);   // peers blocked in a collective with this rank get an error, not a hang
        throw;
      }
      NCCL(ncclCommDestroy(comm));
      std::printf("rank %d: shard [%d, %d) on device %d, setup %.3f s\n", opt.rank, d.lo, d.hi, device, d.setup_s);
      std::printf("best candidate %d score %.12e\n", d.best, d.score);
      return d.best < 0 ? 4 : 0;
    }

    // ---- one process, one thread per device
    if (opt.devices.empty())
      for (int d = 0; d < n_dev; ++d) opt.devices.push_back(d);
    const int world = (int)opt.devices.size();
    for (int d : opt.devices)
      if (d < 0 || d >= n_dev) throw Fail{"device ordinal out of range"};
    std::vector<ncclComm_t> comms(world, nullptr);
    twr_model shared_model;
    std::vector<double> shared_table;
    if (opt.collective) {
      NCCL(ncclCommInitAll(comms.data(), world, opt.devices.data()));   // (RCCL refuses a device that is listed twice)
    } else {
      TWR(twr_model_preset(TWR_ROBOT_ANYMAL, TWR_TERRAIN_STAIRS, &shared_model));
      shared_table.assign(16 * (size_t)opt.n_cand, 0.0);
    }
    std::vector<Decision> dec(world);
    std::vector<std::thread> pool;
    std::atomic<bool> failed{false};
    for (int r = 0; r < world; ++r)
      pool.emplace_back([&, r] {
        try {
          RunRank(opt, r, world, opt.devices[r], comms[r], &shared_model, &shared_table, dec[r]);
        } catch (const Fail& f) {
          dec[r].error = f.what;
          // The peers may be sitting in ncclBroadcast / ncclAllGather (or in the stream synchronisation behind one) waiting
          // for this rank: abort every communicator, once, so that they come back with an error and join() returns.
          if (!failed.exchange(true))
            for (ncclComm_t c : comms)
              if (c) (void)ncclCommAbort(c);
        }
      });
    for (auto& t : pool) t.join();
    if (!failed.load())
      for (ncclComm_t c : comms)
        if (c) (void)ncclCommDestroy(c);   // (an aborted communicator is already gone)
    // the first failure is the cause; the errors of the aborted peers are its consequence
    for (int r = 0; r < world; ++r)
      if (!dec[r].error.empty() && dec[r].error.find("nccl") == std::string::npos) throw Fail{"rank " + std::to_string(r) + ": " + dec[r].error};
    for (int r = 0; r < world; ++r)
      if (!dec[r].error.empty()) throw Fail{"rank " + std::to_string(r) + ": " + dec[r].error};
    if (!opt.collective)
      for (int r = 0; r < world; ++r) ArgMin(shared_table, opt.n_cand, dec[r]);   // one table in host memory: one decision
    for (int r = 0; r < world; ++r) {
      std::printf("rank %d: shard [%d, %d) on device %d, setup %.3f s", r, dec[r].lo, dec[r].hi, opt.devices[r], dec[r].setup_s);
      if (opt.steps > 0)
        std::printf(", %.3f ms per step of %d callbacks (%.3g callbacks/s)", dec[r].step_ms, dec[r].hi - dec[r].lo,
                    (dec[r].hi - dec[r].lo) / (dec[r].step_ms * 1e-3));
      std::printf("\n");
      if (dec[r].best != dec[0].best || dec[r].score != dec[0].score) throw Fail{"the ranks disagree on the decision"};
    }
    std::printf("%s, %d rank(s): best candidate %d score %.12e\n", opt.collective ? "RCCL broadcast + all-gather" : "no collective (shared host memory)",
                world, dec[0].best, dec[0].score);
    std::printf("best candidate %d score %.12e\n", dec[0].best, dec[0].score);
    return dec[0].best < 0 ? 4 : 0;
  } catch (const Fail& f) {
    std::fprintf(stderr, "sweep_multi_gpu: %s\n", f.what.c_str());
    return 1;
  }
}
