// C ABI of libtowr_amd.so (see include/towr_amd.h).  Host runtime: handles, error
// reporting, device table upload, work-list construction and the launch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "structure.h"

namespace twr {
hipError_t launch_eval(int n_ee, int n_cu, const DynWork* dyn, int n_dyn, int dyn_map_chunks, const RomWork* rom, int n_rom, int rom_max_vals,
                       const NodeWork* node, int n_node, int node_families, const FamWork* const fam[4], const int n_fam[4],
                       const PDynWork* pdyn, int n_pdyn, int pdyn_img_cap,
                       const LocWork* ploc, int n_ploc, const RomPhaseWork* prom, int n_prom, int prom_img_cap, const double* x,
                       double* g, double* jac, double* dump, int flags, bool stream_nt, const FlatWork* flat, int n_flat, int flat_max_x,
                       hipStream_t stream, hipEvent_t* ev);
int dyn_dump_doubles();
int node_force_chunk();
hipError_t prepare_phase_kernels(int pdyn_img_cap, int prom_img_cap);
hipError_t launch_check(int n_problems, const int64_t* g_off, const int64_t* j_off, const double* g, const double* jac,
                        int32_t* status, int flags, hipStream_t stream);
hipError_t launch_score(const NodeWork* work, int n_problems, const double* g, double* scores, hipStream_t stream);
int best_max_blocks();
hipError_t launch_best(const double* scores, int n, unsigned families, double* partial, unsigned* counter, double* best, double index_offset,
                       hipStream_t stream);
hipError_t launch_contact_plan(const NodeWork* work, int n_problems, const double* x, double* out, int32_t* counts, double dt,
                               double time_horizon, int n_samples_max, int max_steps, hipStream_t stream);
hipError_t launch_planes(const double* plan, const int32_t* counts, const double* poly_xy, const int32_t* poly_start, int n_polys,
                         int n_problems, int max_steps, int n_ee, int32_t* plane_index, hipStream_t stream);
hipError_t launch_sample(const SampleWork* work, int n_work, const double* x, double* out, double dt, const double* times,
                         hipStream_t stream);
}  // namespace twr

struct twr_structure {
  twr::Structure s;
};
struct twr_terrain_grid {
  std::shared_ptr<twr::TerrainGrid> g;
};

// Makes `device` current for a scope and restores the calling thread's device afterwards: no entry point of the library
// leaves its caller on another device (one process may drive several GPUs).  The caller's error state is not touched.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  hipError_t status = hipSuccess;
  explicit DeviceScope(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) {
      status = hipSetDevice(device);
      switched = status == hipSuccess;
    }
  }
  ~DeviceScope() {
    if (switched && prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

struct twr_planes {
  int device = 0;
  std::vector<int32_t> start;      // polygon r = points [start[r], start[r+1])
  std::vector<double> world_xy;    // PlanarRegionsToPolygons output
  double* d_xy = nullptr;
  int32_t* d_start = nullptr;
  ~twr_planes() {                  // (also runs on the error paths of twr_planes_create)
    if (!d_xy && !d_start) return;
    DeviceScope on(device);
    if (d_xy) (void)hipFree(d_xy);
    if (d_start) (void)hipFree(d_start);
  }
};

struct twr_batch {
  int device = 0;
  int n_problems = 0, n_ee = 0;
  int n_dyn = 0, n_rom = 0, n_node = 0, n_cu = 0;
  int node_families = 4;                     // 2 when no problem has more than terrain-* / force-* work for the node kernel
  int rom_max_vals = 0;                      // Jacobian values of the largest rom slice (picks the copy-out length)
  int64_t cache_bytes = 0;                   // memory-side cache of the batch's device (structure.h MemorySideCacheBytes)
  bool stream_nt = false;                    // non-temporal copy-out stores in dyn_kernel / rom_kernel / the fused kernel (set by twr_batch_create)
  int dyn_map_chunks = 2;                    // 2: every dyn slice of the batch stages <= 128 doubles of x (256-byte staging maps), else 4
  std::vector<int64_t> x_off, g_off, j_off;  // n_problems+1
  std::vector<void*> blobs;                  // device blobs, one per distinct structure: addresses inside `arena`
  void* arena = nullptr;                     // ONE allocation for the tables of all structures (a sweep has a thousand
                                             // of them: one mapping with large pages instead of a thousand small ones,
                                             // one upload instead of a thousand)
  int64_t table_bytes = 0;                   // arena bytes
  int64_t dyn_layout_bytes = 0, dyn_layout_distinct_bytes = 0;   // layout tables of dyn_kernel: as built / after sharing by content
  std::vector<void*> grids;                  // device copies of the distinct gridded terrains
  // what twr_batch_sample needs of every problem (the structures need not outlive the batch)
  std::vector<uint64_t> blob_of_problem;     // device blob address
  std::vector<double> t_total;               // Spline::GetTotalTime of base-lin
  std::vector<char> sample_ok;               // polynomial counts fit the sampling kernel's LDS tables
  twr::SampleWork* d_swork = nullptr;        // work list of the last twr_batch_sample call (cached per dt / stride)
  int n_swork = 0;
  double swork_dt = 0.0;
  int64_t swork_stride = -1;
  twr::SampleWork* d_gwork = nullptr;        // work list of the last twr_batch_initial_guess call (cached per count / stride)
  int n_gwork = 0, gwork_times = -1;
  int64_t gwork_stride = -1;
  twr::DynWork* d_dyn = nullptr;
  twr::RomWork* d_rom = nullptr;
  twr::NodeWork* d_node = nullptr;
  twr::FlatWork* d_flat = nullptr;           // values-only evaluation of dynamic / rangeofmotion-*, one lane per time node:
  int flat_max_x = 0;                        // variables of the largest problem (the LDS a wave of that path stages x in)
  int n_flat = 0;                            // records, in groups of four per problem (nullptr when a problem of the batch cannot take that path)
  twr::FamWork* d_fam[4] = {nullptr, nullptr, nullptr, nullptr};   // chunk lists of node_chunk_kernel (large batches only)
  int n_fam[4] = {0, 0, 0, 0};
  // optimised-timings problems have their own work lists
  twr::PDynWork* d_pdyn = nullptr;
  int pdyn_img_cap = 0, prom_img_cap = 0;    // doubles of the LDS images of dyn_phase_kernel / rom_phase_kernel (largest pass of the batch)
  twr::LocWork* d_ploc = nullptr;
  twr::RomPhaseWork* d_prom = nullptr;
  int64_t *d_goff = nullptr, *d_joff = nullptr;  // device copies of g_off / j_off (TWR_EVAL_CHECK)
  int32_t* d_status = nullptr;                    // per-problem non-finite flags of the last checked evaluation
  double* d_dump = nullptr; // where dyn_kernel's first (empty) copy-out of every workgroup goes
  double* d_best = nullptr; // twr_batch_best: per-block results (2 doubles each) + the block counter behind them
  void* d_precs = nullptr;  // scratch: x-dependent DynLoc / RomRec records of the optimised-timings problems
  int n_pdyn = 0, n_ploc = 0, n_prom = 0;
  // lazily sized scratch for twr_batch_eval_host
  double *d_x = nullptr, *d_g = nullptr, *d_j = nullptr;
  double *p_x = nullptr, *p_g = nullptr, *p_j = nullptr;  // page-locked host buffers (twr_batch_host_buffers)
  // twr_batch_eval_host runs on a stream of the batch's own (non-blocking: it neither waits for nor holds up work the host
  // application has on the NULL stream or on other blocking streams); created on first use
  hipStream_t host_stream = nullptr;
  // optional per-kernel timing (twr_batch_profile_begin/end): 4 events per recorded eval
  std::vector<hipEvent_t> prof_events;
  int prof_capacity = 0, prof_count = 0;
};

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define TWR_HIP(call)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) throw std::runtime_error(std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

void check_params(const twr_params& p) {   // throws: what twr_structure_create rejects before it builds anything
  if (p.polys_per_swing < 1 || p.polys_per_stance_force < 1) throw std::runtime_error("polynomials per phase must be >= 1");
  if (p.constraint_sets <= 0 || (p.constraint_sets & ~TWR_SETS_EVERY))
    throw std::runtime_error("constraint_sets must be a non-empty mask of TWR_SET_* bits");
  if ((p.constraint_sets & TWR_SET_BASE_ROM) && !std::isfinite(p.base_z_init))
    throw std::runtime_error("TWR_SET_BASE_ROM needs twr_params.base_z_init (the initial base height; "
                             "base_motion_constraint.cc:51-55 reads it from the spline)");
}

void copy_set(const twr::SetInfo& s, twr_set_info* out) {
  std::memset(out, 0, sizeof(*out));
  std::strncpy(out->name, s.name.c_str(), TWR_NAME_LEN - 1);
  out->offset = s.offset;
  out->size = s.size;
  out->nnz_offset = s.nnz_offset;
  out->nnz = s.nnz;
}

}  // namespace

extern "C" {

const char* twr_last_error(void) { return g_err.c_str(); }

int twr_model_preset(int robot, int terrain, twr_model* out) {
  if (!out) return fail(TWR_ERR_INVALID, "null output");
  try {
    twr::ModelPreset(robot, terrain, out);
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_INVALID, e.what());
  }
}

int twr_params_default(twr_params* out) {
  if (!out) return fail(TWR_ERR_INVALID, "null output");
  out->dt_dynamic = 0.1;  // parameters.cc:43-50
  out->dt_rom = 0.08;
  out->duration_base_poly = 0.1;
  out->polys_per_swing = 2;
  out->polys_per_stance_force = 3;
  out->constraint_sets = TWR_SETS_HOT_PATH;
  out->reserved_ = 0;
  out->dt_base_motion = out->duration_base_poly / 4.;  // parameters.cc:51
  out->base_z_init = std::nan("");  // must be set by callers that enable TWR_SET_BASE_ROM
  return TWR_OK;
}

int twr_gait_combo(int n_ee, int combo, double t_total, double swing_scale, twr_schedule* out) {
  if (!out) return fail(TWR_ERR_INVALID, "null output");
  try {
    if (!(t_total > 0) || !(swing_scale > 0)) throw std::runtime_error("t_total and swing_scale must be positive");
    twr::GaitCombo(n_ee, combo, t_total, swing_scale, out);
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_INVALID, e.what());
  }
}

int twr_terrain_grid_create(const double* heights, int rows, int cols, twr_terrain_grid** out) {
  if (!heights || !out || rows < 1 || cols < 1) return fail(TWR_ERR_INVALID, "bad grid");
  try {
    std::unique_ptr<twr_terrain_grid> h(new twr_terrain_grid());
    h->g = std::make_shared<twr::TerrainGrid>();
    h->g->heights.assign(heights, heights + (size_t)rows * cols);
    h->g->rows = rows;
    h->g->cols = cols;
    *out = h.release();
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_INVALID, e.what());
  }
}
int twr_terrain_grid_map_create(const float* elevation, int size_x, int size_y, double resolution, double pos_x,
                                double pos_y, twr_terrain_grid** out) {
  if (!elevation || !out || size_x < 1 || size_y < 1 || !(resolution > 0) || !std::isfinite(pos_x) || !std::isfinite(pos_y))
    return fail(TWR_ERR_INVALID, "bad grid map");
  try {
    std::unique_ptr<twr_terrain_grid> h(new twr_terrain_grid());
    h->g = std::make_shared<twr::TerrainGrid>();
    h->g->grid_map = true;
    h->g->elevation.assign(elevation, elevation + (size_t)size_x * size_y);
    h->g->rows = size_x;
    h->g->cols = size_y;
    h->g->res = resolution;
    h->g->eps = resolution / 6.0;  // grid_height_map.h:25
    h->g->pos_x = pos_x;
    h->g->pos_y = pos_y;
    *out = h.release();
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_INVALID, e.what());
  }
}
void twr_terrain_grid_destroy(twr_terrain_grid* g) { delete g; }

int twr_structure_create(const twr_model* model, const twr_schedule* schedule, const twr_params* params,
                         twr_structure** out) {
  return twr_structure_create_with_grid(model, schedule, params, nullptr, out);
}

int twr_structure_create_with_grid(const twr_model* model, const twr_schedule* schedule, const twr_params* params,
                                   const twr_terrain_grid* grid, twr_structure** out) {
  if (!model || !schedule || !params || !out) return fail(TWR_ERR_INVALID, "null argument");
  try {
    std::unique_ptr<twr_structure> h(new twr_structure());
    if (grid) h->s.grid = grid->g;
    h->s.model = *model;
    h->s.schedule = *schedule;
    h->s.params = *params;
    check_params(*params);
    h->s.Build();
    *out = h.release();
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_INVALID, e.what());
  }
}

void twr_structure_destroy(twr_structure* s) { delete s; }

int twr_structure_create_many(const twr_model* model, const twr_schedule* schedules, const twr_params* params, int n,
                              int n_threads, twr_structure** out) {
  return twr_structure_create_many_with_grid(model, schedules, params, n, n_threads, nullptr, out);
}

int twr_structure_create_many_with_grid(const twr_model* model, const twr_schedule* schedules, const twr_params* params, int n,
                                        int n_threads, const twr_terrain_grid* grid, twr_structure** out) {
  if (!model || !schedules || !params || !out || n < 1) return fail(TWR_ERR_INVALID, "bad arguments");
  if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
  n_threads = std::max(1, std::min(n_threads, n));
  for (int i = 0; i < n; ++i) out[i] = nullptr;
  std::vector<std::string> errs(n_threads);
  std::atomic<int> next(0);
  auto worker = [&](int tid) {
    for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
      // the error string of the failing call lives in the worker's thread_local slot: carry it out
      if (twr_structure_create_with_grid(model, &schedules[i], &params[i], grid, &out[i]) != TWR_OK && errs[tid].empty())
        errs[tid] = "structure " + std::to_string(i) + ": " + twr_last_error();
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; ++t) {
    try {
      pool.emplace_back(worker, t);
    } catch (const std::exception&) {   // no more threads to be had: the ones that run share the work
      break;
    }
  }
  worker(0);
  for (auto& t : pool) t.join();
  for (const std::string& e : errs)
    if (!e.empty()) {
      for (int i = 0; i < n; ++i) {
        twr_structure_destroy(out[i]);
        out[i] = nullptr;
      }
      return fail(TWR_ERR_INVALID, e);
    }
  return TWR_OK;
}

// Bytes one callback of every candidate moves, 8 (n + m + nnz): the weight SURVEY 8e shards a sweep by.  Only the
// variable layout, the time tables and the CSR pattern are built (no device tables), on n_threads host threads.
int twr_candidate_bytes(const twr_model* model, const twr_schedule* schedules, const twr_params* params, int n, int n_threads,
                        int64_t* bytes) {
  if (!model || !schedules || !params || !bytes || n < 1) return fail(TWR_ERR_INVALID, "bad arguments");
  if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
  n_threads = std::max(1, std::min(n_threads, n));
  std::vector<std::string> errs(n_threads);
  std::atomic<int> next(0);
  auto worker = [&](int tid) {
    for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
      try {
        twr::Structure s;
        s.model = *model;
        s.schedule = schedules[i];
        s.params = params[i];
        check_params(params[i]);
        s.BuildSizes();
        bytes[i] = 8 * ((int64_t)s.n_vars + s.n_rows + s.nnz);
      } catch (const std::exception& e) {
        if (errs[tid].empty()) errs[tid] = "candidate " + std::to_string(i) + ": " + e.what();
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; ++t) {
    try {
      pool.emplace_back(worker, t);
    } catch (const std::exception&) {
      break;
    }
  }
  worker(0);
  for (auto& t : pool) t.join();
  for (const std::string& e : errs)
    if (!e.empty()) return fail(TWR_ERR_INVALID, e);
  return TWR_OK;
}

// Contiguous shards balanced by the prefix sum of the weights: rank r owns [bounds[r], bounds[r + 1]).  Never an empty
// shard; TWR_ERR_INVALID (on every rank alike: the arguments are the same everywhere) when there are fewer candidates
// than ranks.  The boundary is the prefix whose sum is closest to r / world of the total (ties: the earlier one).
int twr_shard_bounds(const double* weights, int n, int world, int32_t* bounds) {
  if (!weights || !bounds || n < 1) return fail(TWR_ERR_INVALID, "bad arguments");
  if (world < 1 || world > n)
    return fail(TWR_ERR_INVALID, "cannot shard " + std::to_string(n) + " candidates over " + std::to_string(world) +
                                     " ranks: every rank needs at least one");
  std::vector<double> csum(n + 1, 0.0);
  for (int i = 0; i < n; ++i) {
    if (!(weights[i] >= 0.0)) return fail(TWR_ERR_INVALID, "negative or NaN weight");
    csum[i + 1] = csum[i] + weights[i];
  }
  const double total = csum[n];
  bounds[0] = 0;
  for (int r = 1; r < world; ++r) {
    const double target = total * r / world;
    int i = (int)(std::lower_bound(csum.begin(), csum.end(), target) - csum.begin());   // first prefix >= target
    if (i > 0 && std::fabs(csum[i - 1] - target) <= std::fabs(csum[std::min(i, n)] - target)) --i;
    bounds[r] = std::min(std::max(i, bounds[r - 1] + 1), n - (world - r));
  }
  bounds[world] = n;
  return TWR_OK;
}

int twr_terrain_grid_info(const twr_terrain_grid* g, int32_t* kind, int32_t* rows_or_size_x, int32_t* cols_or_size_y,
                          double* resolution, double* pos_x, double* pos_y, const void** data) {
  if (!g || !g->g) return fail(TWR_ERR_INVALID, "null grid");
  const twr::TerrainGrid& t = *g->g;
  if (kind) *kind = t.grid_map ? 1 : 0;
  if (rows_or_size_x) *rows_or_size_x = t.rows;
  if (cols_or_size_y) *cols_or_size_y = t.cols;
  if (resolution) *resolution = t.res;
  if (pos_x) *pos_x = t.pos_x;
  if (pos_y) *pos_y = t.pos_y;
  if (data) *data = t.grid_map ? static_cast<const void*>(t.elevation.data()) : static_cast<const void*>(t.heights.data());
  return TWR_OK;
}

int twr_structure_sizes(const twr_structure* s, twr_sizes* out) {
  if (!s || !out) return fail(TWR_ERR_INVALID, "null argument");
  out->n_vars = s->s.n_vars;
  out->n_rows = s->s.n_rows;
  out->nnz = s->s.nnz;
  out->n_var_sets = (int)s->s.var_sets.size();
  out->n_con_sets = (int)s->s.con_sets.size();
  out->k_dynamic = (int)s->s.grid_dyn.size();
  out->k_rom = (int)s->s.grid_rom.size();
  return TWR_OK;
}

int twr_structure_values_items(const twr_structure* s, int32_t* n_dynamic_items, int32_t* n_rom_items, int32_t* dynamic_takes_rom,
                               int32_t* items) {
  if (!s || !n_dynamic_items || !n_rom_items || !dynamic_takes_rom) return fail(TWR_ERR_INVALID, "null argument");
  const bool on = s->s.off_flat_polys != 0;
  *n_dynamic_items = on ? (int32_t)s->s.flat_items_dyn.size() : 0;
  *n_rom_items = on ? (int32_t)s->s.flat_items_rom.size() : 0;
  *dynamic_takes_rom = on && s->s.flat_with_rom ? 1 : 0;
  if (items && on) {
    int32_t* o = items;
    for (const auto* list : {&s->s.flat_items_dyn, &s->s.flat_items_rom})
      for (const auto& it : *list) {
        int widest = 0;
        for (int sp = 0; sp < 8; ++sp) widest = std::max(widest, (int)((it.count >> (8 * sp)) & 0xFFu));
        *o++ = it.k0;
        *o++ = it.cnt;
        *o++ = it.gather ? 0 : widest;
      }
  }
  return TWR_OK;
}

int twr_structure_var_set(const twr_structure* s, int i, twr_set_info* out) {
  if (!s || !out || i < 0 || i >= (int)s->s.var_sets.size()) return fail(TWR_ERR_INVALID, "bad variable set index");
  copy_set(s->s.var_sets[i], out);
  return TWR_OK;
}

int twr_structure_con_set(const twr_structure* s, int i, twr_set_info* out) {
  if (!s || !out || i < 0 || i >= (int)s->s.con_sets.size()) return fail(TWR_ERR_INVALID, "bad constraint set index");
  copy_set(s->s.con_sets[i], out);
  return TWR_OK;
}

const int32_t* twr_structure_row_ptr(const twr_structure* s) { return s ? s->s.row_ptr.data() : nullptr; }
const int32_t* twr_structure_col_idx(const twr_structure* s) { return s ? s->s.col_idx.data() : nullptr; }

int twr_structure_bounds(const twr_structure* s, double* lower, double* upper) {
  if (!s || !lower || !upper) return fail(TWR_ERR_INVALID, "null argument");
  std::memcpy(lower, s->s.lower.data(), s->s.lower.size() * sizeof(double));
  std::memcpy(upper, s->s.upper.data(), s->s.upper.size() * sizeof(double));
  return TWR_OK;
}

int twr_structure_initial_guess(const twr_structure* s, const double init_base_lin[3], const double init_base_ang[3],
                                const double final_base_lin[3], const double final_base_ang[3],
                                const double* init_ee_pos, double* x_out) {
  if (!s || !init_base_lin || !init_base_ang || !final_base_lin || !final_base_ang || !init_ee_pos || !x_out)
    return fail(TWR_ERR_INVALID, "null argument");
  try {
    s->s.InitialGuess(init_base_lin, init_base_ang, final_base_lin, final_base_ang, init_ee_pos, x_out);
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_INVALID, e.what());
  }
}

int twr_structure_variable_bounds(const twr_structure* s, const double init_base[12], const double final_base[12],
                                  const double* init_ee_pos, double* lower, double* upper) {
  if (!s || !init_base || !final_base || !init_ee_pos || !lower || !upper) return fail(TWR_ERR_INVALID, "null argument");
  try {
    s->s.VariableBounds(init_base, final_base, init_ee_pos, lower, upper);
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_INVALID, e.what());
  }
}

int twr_batch_create(const twr_structure* const* structs, int n_structs, const int32_t* struct_of_problem,
                     int n_problems, int device, twr_batch** out) {
  if (!structs || !struct_of_problem || !out || n_structs < 1 || n_problems < 1)
    return fail(TWR_ERR_INVALID, "bad arguments");
  std::unique_ptr<twr_batch> b(new twr_batch());
  try {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
      return fail(TWR_ERR_NO_DEVICE, "no HIP device visible: towr_amd has no CPU fallback");
    if (device < 0 || device >= n_dev) return fail(TWR_ERR_INVALID, "device ordinal out of range");
    DeviceScope on(device);
    TWR_HIP(on.status);
    b->device = device;
    b->n_problems = n_problems;
    b->n_ee = structs[0]->s.n_ee;
    std::vector<const twr::TerrainGrid*> host_grids;
    std::vector<size_t> blob_off(n_structs + 1, 0);
    for (int i = 0; i < n_structs; ++i) {
      if (!structs[i]) throw std::runtime_error("null structure");
      if (structs[i]->s.n_ee != b->n_ee) throw std::runtime_error("all structures of a batch must share n_ee");
      blob_off[i + 1] = blob_off[i] + (structs[i]->s.blob.size() + 255) / 256 * 256;   // every blob starts on a 256-byte line
    }
    TWR_HIP(hipMalloc(&b->arena, blob_off[n_structs]));
    std::vector<char> host_arena(blob_off[n_structs], 0);
    for (int i = 0; i < n_structs; ++i) {
      b->blobs.push_back(static_cast<char*>(b->arena) + blob_off[i]);
      char* blob = host_arena.data() + blob_off[i];
      std::memcpy(blob, structs[i]->s.blob.data(), structs[i]->s.blob.size());
      if (structs[i]->s.grid) {  // gridded terrain: upload every distinct grid once, patch its address into the header
        const twr::TerrainGrid* tg = structs[i]->s.grid.get();
        void* dg = nullptr;
        for (size_t q = 0; q < host_grids.size(); ++q)
          if (host_grids[q] == tg) dg = b->grids[q];
        if (!dg) {
          const void* src = tg->grid_map ? (const void*)tg->elevation.data() : (const void*)tg->heights.data();
          const size_t bytes = tg->grid_map ? tg->elevation.size() * sizeof(float) : tg->heights.size() * sizeof(double);
          TWR_HIP(hipMalloc(&dg, bytes));
          TWR_HIP(hipMemcpy(dg, src, bytes, hipMemcpyHostToDevice));
          host_grids.push_back(tg);
          b->grids.push_back(dg);
        }
        reinterpret_cast<twr::DevStruct*>(blob)->grid_ptr = reinterpret_cast<uint64_t>(dg);
      }
    }
    TWR_HIP(hipMemcpy(b->arena, host_arena.data(), host_arena.size(), hipMemcpyHostToDevice));
    b->table_bytes = (int64_t)host_arena.size();
    // Layout tables of dyn_kernel (device_tables.h) are stored once per distinct CONTENT: a structure whose table is
    // byte-identical to one of an earlier structure of the batch reads that one (its own copy stays in the arena, unread).
    // Candidates of a sweep that differ in the total time only share all of them, and an evaluation then reads 24 B per
    // time node + 16 B per polynomial of such a candidate instead of ~35 KB.  The model constants (DevStruct header) are
    // taken from the first structure of the batch that has the same ones.
    std::vector<std::unordered_map<uint32_t, uint64_t>> layout_at(n_structs);   // [structure][blob offset of the table] -> device address
    std::vector<uint64_t> model_hdr(n_structs);
    {
      std::vector<const twr::Structure*> sp(n_structs);
      for (int i = 0; i < n_structs; ++i) sp[i] = &structs[i]->s;
      const twr::LayoutShare share = twr::ShareLayoutTables(sp);   // (host logic: structure.cc)
      b->dyn_layout_bytes = share.bytes_built;
      b->dyn_layout_distinct_bytes = share.bytes_distinct;
      for (int i = 0; i < n_structs; ++i) {
        const auto& tabs = structs[i]->s.dyn_layout_tables;
        for (size_t t = 0; t < tabs.size(); ++t) {
          const twr::LayoutShare::Ref& r = share.of[i][t];
          layout_at[i][tabs[t].off] = reinterpret_cast<uint64_t>(b->blobs[r.owner]) + r.off;
        }
        const twr::DevStruct* H = reinterpret_cast<const twr::DevStruct*>(host_arena.data() + blob_off[i]);
        model_hdr[i] = reinterpret_cast<uint64_t>(b->blobs[i]);
        for (int q = 0; q < i; ++q) {
          const twr::DevStruct* Q = reinterpret_cast<const twr::DevStruct*>(host_arena.data() + blob_off[q]);
          if (model_hdr[q] == reinterpret_cast<uint64_t>(b->blobs[q]) && Q->mass == H->mass && Q->gravity == H->gravity &&
              std::memcmp(Q->Ib, H->Ib, sizeof(H->Ib)) == 0) {
            model_hdr[i] = model_hdr[q];
            break;
          }
        }
      }
    }
    b->x_off.assign(n_problems + 1, 0);
    b->g_off.assign(n_problems + 1, 0);
    b->j_off.assign(n_problems + 1, 0);
    hipDeviceProp_t prop;
    TWR_HIP(hipGetDeviceProperties(&prop, device));
    b->n_cu = prop.multiProcessorCount;
    // (a compute partition -- CPX -- shows up as a device with a fraction of the chip's CUs: its share of the cache follows)
    b->cache_bytes = twr::MemorySideCacheBytes(prop.gcnArchName, prop.l2CacheSize, 1);
    std::vector<twr::DynWork> dyn;
    std::vector<twr::RomWork> rom;
    std::vector<twr::NodeWork> node;
    std::vector<twr::FlatWork> flat_items;     // groups of four records, every group of one problem (device_tables.h FlatWork)
    std::vector<int> flat_group_p;             // the problem of every group
    bool flat_ok = true;
    std::vector<twr::PDynWork> pdyn;
    std::vector<int> pdyn_first;   // first dynamic run of every optimised-timings problem (+ end)
    // dyn_phase_kernel: a pass = the time nodes whose expanded rows fit the LDS image (four at sixteen lanes each,
    // fewer when a node has more than 5120 values: the image then takes the whole 160 KB of a CU)
    std::vector<twr::LocWork> ploc;
    std::vector<twr::RomPhaseWork> prom;
    size_t prec_bytes = 0;  // offsets into the scratch buffer are stored first and rebased after hipMalloc
    std::vector<int> dyn_first, rom_first;  // first work item of every problem (+ end)
    // (families that are switched off -- twr_params.constraint_sets -- simply have no work items; problems with
    // optimised timings get PDynWork / LocWork / RomPhaseWork items instead of DynWork / RomWork)
    for (int i = 0; i < n_structs; ++i)
      if (structs[i]->s.dyn_staged_max > 128) b->dyn_map_chunks = 4;
    for (int p = 0; p < n_problems; ++p) {
      int si = struct_of_problem[p];
      if (si < 0 || si >= n_structs) throw std::runtime_error("struct_of_problem out of range");
      const twr::Structure& S = structs[si]->s;
      b->blob_of_problem.push_back(reinterpret_cast<uint64_t>(b->blobs[si]));
      {
        const twr::SampleTables* st = reinterpret_cast<const twr::SampleTables*>(
            S.blob.data() + reinterpret_cast<const twr::DevStruct*>(S.blob.data())->o_sample);
        b->t_total.push_back(st->t_total);
        b->sample_ok.push_back(st->n_base >= 0);
      }
      b->x_off[p + 1] = b->x_off[p] + S.n_vars;
      b->g_off[p + 1] = b->g_off[p] + S.n_rows;
      b->j_off[p + 1] = b->j_off[p] + S.nnz;
      const uint64_t blob = reinterpret_cast<uint64_t>(b->blobs[si]);
      dyn_first.push_back((int)dyn.size());
      rom_first.push_back((int)rom.size());
      const twr::SetInfo* dsp = S.FindSet("dynamic");
      const twr::SetInfo ds = dsp ? *dsp : twr::SetInfo();
      for (const auto& sl : S.dyn_slices) {   // fixed timings only (empty otherwise)
        twr::DynWork w;
        std::memset(&w, 0, sizeof(w));
        const auto& lay = layout_at[si];
        w.nodes_t = blob + S.off_dyn_nodes_t + sizeof(twr::DynNodeT) * (size_t)sl.k0;
        w.nodes_l = lay.at(S.off_dyn_nodes_l) + sizeof(twr::DynNodeL) * (size_t)sl.k0;
        w.sel = lay.at(S.off_dyn_sel) + sizeof(twr::DynSel) * (size_t)sl.k0 * 4;
        w.tile = lay.at(S.off_dyn_tile);   // (records are addressed through DynSel::tile)
        w.poly_t = blob + S.off_dyn_poly_t + sizeof(twr::DynPolyT) * (size_t)sl.poly0;
        w.poly_l = lay.at(S.off_dyn_poly_l) + sizeof(twr::DynPolyL) * (size_t)sl.poly0;
        w.map = lay.at(b->dyn_map_chunks == 2 ? sl.map2 : sl.map);
        w.hdr = model_hdr[si];
        w.x_off = b->x_off[p];
        w.g_off = b->g_off[p] + ds.offset + 6 * sl.k0;
        w.j_off = b->j_off[p] + S.row_ptr[ds.offset + 6 * sl.k0];
        w.cnt = sl.cnt;
        w.nvals = sl.nvals;
        dyn.push_back(w);
      }
      for (int e = 0; e < (int)S.rom_slices.size(); ++e) {   // fixed timings only (empty otherwise)
        if (S.rom_slices[e].empty()) continue;
        const twr::SetInfo& rs = *S.FindSet("rangeofmotion-" + std::to_string(e));
        for (const auto& sl : S.rom_slices[e]) {
          twr::RomWork w;
          std::memset(&w, 0, sizeof(w));
          w.nodes = blob + S.off_rom_nodes + sizeof(twr::RomNode) * (size_t)sl.k0;
          w.segs = blob + sl.segs;
          w.x_off = b->x_off[p];
          w.g_off = b->g_off[p] + rs.offset + 3 * sl.k0;
          w.j_off = b->j_off[p] + S.row_ptr[rs.offset + 3 * sl.k0];
          w.off_lin = S.off_base_lin;
          w.off_ang = S.off_base_ang;
          w.cnt = sl.cnt;
          w.nvals = sl.nvals;
          w.ee = e;
          b->rom_max_vals = std::max(b->rom_max_vals, w.nvals);
          rom.push_back(w);
        }
      }
      if (S.timings) {
        const bool have_dyn = S.FindSet("dynamic") != nullptr;
        const int Kd = (int)S.grid_dyn.size();
        const size_t loc_bytes = have_dyn ? sizeof(twr::DynLoc) * 4 * (size_t)Kd : 0;
        const size_t loc_off = prec_bytes;   // DynLoc[4 * Kd] of this problem, then its RomRec arrays
        prec_bytes += loc_bytes;
        if (have_dyn) {
          const twr::SetInfo* dset = S.FindSet("dynamic");
          const int nv = S.phase_tables.node_vals;
          const int run = std::max(1, std::min(4, (160 * 128) / nv));
          b->pdyn_img_cap = std::max(b->pdyn_img_cap, run * nv);
          pdyn_first.push_back((int)pdyn.size());
          for (int k0 = 0; k0 < Kd; k0 += run) {
            twr::PDynWork pw;
            std::memset(&pw, 0, sizeof(pw));
            const twr::PhaseTables& pt = S.phase_tables;
            const uint64_t pt_addr = blob + reinterpret_cast<const twr::DevStruct*>(S.blob.data())->o_phase;
            pw.hdr = blob;
            pw.loc = loc_off + sizeof(twr::DynLoc) * (size_t)k0;
            pw.loc_stride = (int32_t)(sizeof(twr::DynLoc) * (size_t)Kd);
            pw.shared = blob + pt.o_dyn_shared + sizeof(twr::DynShared) * (size_t)k0;
            pw.mput = blob + pt.o_mput;
            pw.fput = blob + pt.o_fput;
            pw.ee = pt_addr + offsetof(twr::PhaseTables, ee);
            pw.x_off = b->x_off[p];
            pw.g_off = b->g_off[p] + dset->offset + 6 * k0;
            pw.j_off = b->j_off[p] + dset->nnz_offset + (int64_t)k0 * nv;
            pw.cnt = std::min(run, Kd - k0);
            pw.node_vals = nv;
            pw.off_lin = S.off_base_lin;
            pw.off_ang = S.off_base_ang;
            pw.n_ee = S.n_ee;
            pw.n_mput = pt.n_mput;
            pw.n_fput = pt.n_fput;
            for (int q = 0; q < 5; ++q) pw.row_off[q] = pt.dyn_row_off[q];
            pdyn.push_back(pw);
          }
        }
        bool any_rom = false;
        for (int e = 0; e < S.n_ee; ++e) {
          const twr::SetInfo* rs = S.FindSet("rangeofmotion-" + std::to_string(e));
          if (!rs && !have_dyn) continue;
          twr::LocWork lw;
          std::memset(&lw, 0, sizeof(lw));
          lw.blob = blob;
          lw.recs = rs ? prec_bytes + 1 : 0;             // (+1: "present" marker until the buffer address is known)
          lw.dyn_loc = have_dyn ? loc_off + sizeof(twr::DynLoc) * (size_t)Kd * (size_t)e + 1 : 0;
          lw.x_off = b->x_off[p];
          lw.ee = e;
          ploc.push_back(lw);
          if (!rs) continue;
          any_rom = true;
          const int K = (int)S.grid_rom.size(), nv = S.phase_tables.rom_node_vals[e];
          const int run_max = std::max(1, std::min(16, (160 * 128) / nv));   // time nodes per pass (four lanes each)
          const int n_pass = (K + run_max - 1) / run_max;
          const int run = (K + n_pass - 1) / n_pass;                  // balanced: no short tail pass (it pays the full copy-out)
          b->prom_img_cap = std::max(b->prom_img_cap, run * nv);
          for (int k0 = 0; k0 < K; k0 += run) {
            twr::RomPhaseWork rw;
            rw.recs = prec_bytes + sizeof(twr::RomRec) * (size_t)k0;
            rw.x_off = b->x_off[p];
            rw.g_off = b->g_off[p] + rs->offset + 3 * k0;
            rw.j_off = b->j_off[p] + rs->nnz_offset + (int64_t)k0 * nv;
            rw.off_lin = S.off_base_lin;
            rw.off_ang = S.off_base_ang;
            rw.cnt = std::min(run, K - k0);
            rw.msize = S.phase_tables.msize[e];
            rw.ns = S.schedule.n_phases[e] - 1;
            rw.node_vals = nv;
            prom.push_back(rw);
          }
          prec_bytes += sizeof(twr::RomRec) * (size_t)K;
        }
        (void)any_rom;
      }
      {  // values-only work items (device_tables.h FlatWork): 64 time nodes of the dynamic / range-of-motion grid each
        if (S.timings) flat_ok = false;
        if (S.off_flat_polys) {
          const twr::DevStruct* H = reinterpret_cast<const twr::DevStruct*>(S.blob.data());
          auto items = [&](uint32_t off_nodes, const std::vector<twr::Structure::FlatItem>& list, bool dynamic) {
            for (const auto& it : list) {
              twr::FlatWork fw;
              std::memset(&fw, 0, sizeof(fw));
              fw.nodes = blob + off_nodes + sizeof(twr::FlatNode) * (size_t)it.k0;
              fw.polys = blob + S.off_flat_polys;
              fw.x_off = b->x_off[p];
              fw.g_off = b->g_off[p];
              fw.k0 = it.k0;
              fw.cnt = it.cnt;
              fw.start[0] = it.start[0];
              fw.start[1] = it.start[1];
              fw.count = it.count;
              fw.n_x = S.n_vars;
              fw.n_ee = S.n_ee;
              fw.off_lin = S.off_base_lin;
              fw.off_ang = S.off_base_ang;
              for (int e = 0; e < twr::kMaxEE; ++e) fw.row_rom[e] = S.flat_row_rom[e];
              fw.dynamic = dynamic ? 1 : 0;
              fw.gather = it.gather ? 1 : 0;
              if (dynamic) {
                fw.row_dyn = S.flat_row_dyn;
                fw.with_rom = S.flat_with_rom ? 1 : 0;
                fw.mass = H->mass;
                fw.gravity = H->gravity;
                for (int i = 0; i < 6; ++i) fw.Ib[i] = H->Ib[i];
              }
              flat_items.push_back(fw);
            }
          };
          b->flat_max_x = std::max(b->flat_max_x, S.n_vars);
          items(S.off_flat_dyn, S.flat_items_dyn, true);
          items(S.off_flat_rom, S.flat_items_rom, false);
          while (flat_items.size() % 4 != 0) {   // whole groups: empty items that still carry the problem's x (the group copies it
            twr::FlatWork fw;                    // to LDS with all its threads)
            std::memset(&fw, 0, sizeof(fw));
            fw.x_off = b->x_off[p];
            fw.n_x = S.n_vars;
            flat_items.push_back(fw);
          }
          flat_group_p.resize(flat_items.size() / 4, p);
        } else if (S.FindSet("rangeofmotion-0") || S.FindSet("dynamic")) {
          flat_ok = false;
        }
      }
      twr::NodeWork nw;
      nw.blob = blob;
      nw.x_off = b->x_off[p];
      nw.g_off = b->g_off[p];
      nw.j_off = b->j_off[p];
      node.push_back(nw);
    }
    dyn_first.push_back((int)dyn.size());
    rom_first.push_back((int)rom.size());
    // XCD-aware order.  Workgroups are dealt round-robin over the 8 XCDs and the persistent grids are
    // multiples of 8, so list position j runs on XCD j % 8.  Interleaving the problems in groups of 8
    // puts all slices of one problem on ONE XCD (same L2): its x is fetched from HBM once per kernel
    // instead of once per XCD.  (Speed only; ragged slice counts merely loosen the alignment.)
    auto interleave = [&](auto& items, const std::vector<int>& first) {
      auto src = items;
      size_t out = 0;
      for (int p0 = 0; p0 < n_problems; p0 += 8) {
        const int np = std::min(8, n_problems - p0);
        for (int s2 = 0;; ++s2) {
          bool any = false;
          for (int k = 0; k < np; ++k)
            if (first[p0 + k] + s2 < first[p0 + k + 1]) {
              items[out++] = src[first[p0 + k] + s2];
              any = true;
            }
          if (!any) break;
        }
      }
    };
    interleave(dyn, dyn_first);
    interleave(rom, rom_first);
    b->n_dyn = (int)dyn.size();
    b->n_rom = (int)rom.size();
    // Store policy of the copy-out (kernels.hip copy_out_fixed): non-temporal when the batch is SWEEP-LIKE -- fewer than four
    // problems per structure on average, so every evaluation re-reads tables (and x) that only that problem uses -- AND one
    // evaluation writes more than the device's memory-side cache holds (256 MB on an MI355X; by architecture name, structure.h), so that plain stores would flush those
    // tables out of it between two evaluations.  Measured (one box, no per-kernel events): the C5 sweep at 512 /
    // 1024 candidates 116-118 / 223-224 -> 99 / 209-211 us per step; at 256 candidates (220 MB of output, absorbed by the
    // Infinity Cache as it is) 52 -> 55 us, and 8192 problems of ONE structure lose 15 % in rom_kernel -- hence the two conditions.
    {
      std::vector<char> used(n_structs, 0);
      int n_used = 0;
      for (int p = 0; p < n_problems; ++p)
        if (!used[struct_of_problem[p]]) {
          used[struct_of_problem[p]] = 1;
          ++n_used;
        }
      const int64_t out_bytes = 8 * (b->g_off[n_problems] + b->j_off[n_problems]);
      b->stream_nt = twr::StreamNonTemporal(n_used, n_problems, out_bytes, b->cache_bytes);
#ifdef TWR_TUNING_KNOBS   // (include/towr_amd.h, "Tuning knobs")
      if (const char* e = getenv("TWR_STREAM_NT")) b->stream_nt = atoi(e) != 0;
#endif
    }
    b->n_node = (int)node.size();
    b->node_families = 2;
    for (int i = 0; i < n_structs; ++i)
      if (structs[i]->s.params.constraint_sets & ~(TWR_SET_TERRAIN | TWR_SET_DYNAMIC | TWR_SET_ROM | TWR_SET_FORCE)) b->node_families = 4;
    auto upload = [&](const void* src, size_t bytes, void** dst) {
      TWR_HIP(hipMalloc(dst, bytes));
      TWR_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    };
    upload(dyn.data(), dyn.size() * sizeof(twr::DynWork), reinterpret_cast<void**>(&b->d_dyn));
    upload(rom.data(), rom.size() * sizeof(twr::RomWork), reinterpret_cast<void**>(&b->d_rom));
    {  // one entry past the end carries the totals: a kernel reads a problem's row count as work[p + 1].g_off - work[p].g_off
       // from the work list alone (score_kernel requests g before the structure's header has arrived)
      twr::NodeWork end;
      end.blob = 0;
      end.x_off = b->x_off[b->n_problems];
      end.g_off = b->g_off[b->n_problems];
      end.j_off = b->j_off[b->n_problems];
      node.push_back(end);
    }
    upload(node.data(), node.size() * sizeof(twr::NodeWork), reinterpret_cast<void**>(&b->d_node));
    if (flat_ok && !flat_items.empty()) {
      // One problem, one XCD: workgroup r of the launch runs on XCD r modulo 8 and takes group r.  The groups of problem p go to
      // workgroups = p modulo 8: its x comes from HBM once and from that XCD's L2 for its other groups.
      std::vector<size_t> queue[8];   // queue c: the groups that go to the list positions = c modulo 8
      for (size_t i = 0; i < flat_group_p.size(); ++i) queue[flat_group_p[i] % 8].push_back(i);
      std::vector<twr::FlatWork> out;
      out.reserve(flat_items.size());
      auto emit = [&](size_t group) { out.insert(out.end(), flat_items.begin() + 4 * group, flat_items.begin() + 4 * group + 4); };
      size_t depth = 0, k = 0;
      for (const auto& q : queue) depth = std::max(depth, q.size());
      for (; k < depth; ++k) {   // whole rounds of eight; a round in which a queue has run dry ends the interleaving
        bool whole = true;
        for (const auto& q : queue) whole = whole && k < q.size();
        if (!whole) break;
        for (const auto& q : queue) emit(q[k]);
      }
      for (const auto& q : queue)
        for (size_t i = k; i < q.size(); ++i) emit(q[i]);
      b->n_flat = (int)out.size();
      upload(out.data(), out.size() * sizeof(twr::FlatWork), reinterpret_cast<void**>(&b->d_flat));
    }
    // Large batches whose node-based sets are terrain / force / splineacc / swing only: per-family chunk lists for the
    // persistent node_chunk_kernel (baseMotion and totalduration rows, and small batches -- where the fused launch or the
    // one-workgroup-per-problem kernel is as good -- stay with node_kernel).
    {
      // (hot-path batches -- terrain and force rows only -- are faster on node_kernel2: 0.041 vs 0.047 ms per 8192 C3 problems)
      // (from eight problems per CU on -- 2048 on the 256 CUs of an MI355X, where the cut-over was measured: below that the
      // one-workgroup-per-problem kernel has enough waves in flight and no persistent loop to fill)
      bool eligible = n_problems >= 8 * b->n_cu && b->node_families == 4;
      for (int i = 0; i < n_structs && eligible; ++i)
        if (structs[i]->s.params.constraint_sets & (TWR_SET_BASE_ROM | TWR_SET_TOTAL_TIME)) eligible = false;
      if (eligible) {
        std::vector<twr::FamWork> fam[4];
        for (int p = 0; p < n_problems; ++p) {
          const twr::Structure& S = structs[struct_of_problem[p]]->s;
          const twr::DevStruct* H = reinterpret_cast<const twr::DevStruct*>(S.blob.data());
          const uint64_t blob = b->blob_of_problem[p];
          auto add = [&](int f, int count, uint32_t table_off, size_t rec_bytes, int row0, int rows_per, int nnz0, int vals_per) {
            const int chunk = f == 1 ? twr::node_force_chunk() : 64;
            for (int i0 = 0; i0 < count; i0 += chunk) {
              twr::FamWork w;
              std::memset(&w, 0, sizeof(w));
              w.blob = blob;
              w.table = blob + table_off + (f == 2 ? 0 : rec_bytes * (size_t)i0);
              w.x_off = b->x_off[p];
              w.g_off = b->g_off[p] + row0 + (int64_t)rows_per * i0;
              w.j_off = b->j_off[p] + nnz0 + (int64_t)vals_per * i0;
              w.cnt = std::min(chunk, count - i0);
              w.i0 = f == 2 ? i0 : 0;
              w.aux0 = 3 * H->n_junctions;
              w.aux1 = H->off_base_ang;
              w.inv_t_swing = H->inv_t_swing;
              fam[f].push_back(w);
            }
          };
          add(0, H->n_terrain_rows, H->o_terrain_rows, sizeof(twr::TerrainRow), H->row_terrain, 1, H->nnz_terrain, 3);
          add(1, H->n_force_nodes, H->o_force_nodes, sizeof(twr::ForceNode), H->row_force, 5, H->nnz_force, 25);
          add(2, 6 * H->n_junctions, H->o_acc, sizeof(twr::AccJunction), H->row_acc, 1, H->nnz_acc, 6);
          add(3, H->n_swing_nodes, H->o_swing_nodes, sizeof(twr::SwingNode), H->row_swing, 4, H->nnz_swing, 12);
        }
        for (int f = 0; f < 4; ++f) {
          b->n_fam[f] = (int)fam[f].size();
          if (!fam[f].empty()) upload(fam[f].data(), fam[f].size() * sizeof(twr::FamWork), reinterpret_cast<void**>(&b->d_fam[f]));
        }
      }
    }
    upload(b->g_off.data(), b->g_off.size() * sizeof(int64_t), reinterpret_cast<void**>(&b->d_goff));
    upload(b->j_off.data(), b->j_off.size() * sizeof(int64_t), reinterpret_cast<void**>(&b->d_joff));
    TWR_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_dump), sizeof(double) * (size_t)twr::dyn_dump_doubles()));
    TWR_HIP(hipMemset(b->d_dump, 0, sizeof(double) * (size_t)twr::dyn_dump_doubles()));
    TWR_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_best), sizeof(double) * (2 * (size_t)twr::best_max_blocks() + 1)));
    TWR_HIP(hipMemset(b->d_best, 0, sizeof(double) * (2 * (size_t)twr::best_max_blocks() + 1)));
    TWR_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_status), sizeof(int32_t) * (size_t)n_problems));
    TWR_HIP(hipMemset(b->d_status, 0, sizeof(int32_t) * (size_t)n_problems));
    b->n_pdyn = (int)pdyn.size();
    b->n_ploc = (int)ploc.size();
    b->n_prom = (int)prom.size();
    if (!ploc.empty()) {
      TWR_HIP(hipMalloc(&b->d_precs, prec_bytes));
      const uint64_t base = reinterpret_cast<uint64_t>(b->d_precs);
      for (auto& lw : ploc) {
        if (lw.recs) lw.recs += base - 1;
        if (lw.dyn_loc) lw.dyn_loc += base - 1;
      }
      for (auto& rw : prom) rw.recs += base;
      for (auto& pw : pdyn) pw.loc += base;
      upload(ploc.data(), ploc.size() * sizeof(twr::LocWork), reinterpret_cast<void**>(&b->d_ploc));
    }
    if (!prom.empty()) upload(prom.data(), prom.size() * sizeof(twr::RomPhaseWork), reinterpret_cast<void**>(&b->d_prom));
    if (!pdyn.empty()) {
      pdyn_first.push_back((int)pdyn.size());
      // same XCD-aware order as the fixed-timing lists (all runs of one problem on one XCD)
      auto src = pdyn;
      size_t out_i = 0;
      const int np_all = (int)pdyn_first.size() - 1;
      for (int p0 = 0; p0 < np_all; p0 += 8) {
        const int np = std::min(8, np_all - p0);
        for (int s2 = 0;; ++s2) {
          bool any = false;
          for (int k = 0; k < np; ++k)
            if (pdyn_first[p0 + k] + s2 < pdyn_first[p0 + k + 1]) {
              pdyn[out_i++] = src[pdyn_first[p0 + k] + s2];
              any = true;
            }
          if (!any) break;
        }
      }
      upload(pdyn.data(), pdyn.size() * sizeof(twr::PDynWork), reinterpret_cast<void**>(&b->d_pdyn));
    }
    TWR_HIP(twr::prepare_phase_kernels(b->pdyn_img_cap, b->prom_img_cap));
    *out = b.release();
    return TWR_OK;
  } catch (const std::exception& e) {
    twr_batch_destroy(b.release());
    return fail(TWR_ERR_HIP, e.what());
  }
}

void twr_batch_destroy(twr_batch* b) {
  if (!b) return;
  DeviceScope on(b->device);
  if (b->arena) (void)hipFree(b->arena);
  for (void* d : b->grids) (void)hipFree(d);
  if (b->d_dyn) (void)hipFree(b->d_dyn);
  if (b->d_rom) (void)hipFree(b->d_rom);
  if (b->d_node) (void)hipFree(b->d_node);
  for (twr::FamWork* d : b->d_fam)
    if (d) (void)hipFree(d);
  if (b->d_pdyn) (void)hipFree(b->d_pdyn);
  if (b->d_prom) (void)hipFree(b->d_prom);
  if (b->d_ploc) (void)hipFree(b->d_ploc);
  if (b->d_precs) (void)hipFree(b->d_precs);
  if (b->d_goff) (void)hipFree(b->d_goff);
  if (b->d_joff) (void)hipFree(b->d_joff);
  if (b->d_status) (void)hipFree(b->d_status);
  if (b->d_dump) (void)hipFree(b->d_dump);
  if (b->d_flat) (void)hipFree(b->d_flat);
  if (b->d_best) (void)hipFree(b->d_best);
  if (b->d_swork) (void)hipFree(b->d_swork);
  if (b->d_gwork) (void)hipFree(b->d_gwork);
  for (hipEvent_t e : b->prof_events) (void)hipEventDestroy(e);
  if (b->d_x) (void)hipFree(b->d_x);
  if (b->d_g) (void)hipFree(b->d_g);
  if (b->d_j) (void)hipFree(b->d_j);
  if (b->p_x) (void)hipHostFree(b->p_x);
  if (b->p_g) (void)hipHostFree(b->p_g);
  if (b->p_j) (void)hipHostFree(b->p_j);
  if (b->host_stream) (void)hipStreamDestroy(b->host_stream);
  delete b;
}

int twr_batch_num_problems(const twr_batch* b) { return b ? b->n_problems : 0; }

int twr_batch_streaming_stores(const twr_batch* b) { return b && b->stream_nt ? 1 : 0; }

int twr_batch_table_bytes(const twr_batch* b, int64_t* resident, int64_t* dyn_layout, int64_t* dyn_layout_distinct) {
  if (!b) return fail(TWR_ERR_INVALID, "null batch");
  if (resident) *resident = b->table_bytes;
  if (dyn_layout) *dyn_layout = b->dyn_layout_bytes;
  if (dyn_layout_distinct) *dyn_layout_distinct = b->dyn_layout_distinct_bytes;
  return TWR_OK;
}

int twr_batch_layout(const twr_batch* b, int64_t* x_off, int64_t* g_off, int64_t* jac_off) {
  if (!b) return fail(TWR_ERR_INVALID, "null batch");
  size_t bytes = (b->n_problems + 1) * sizeof(int64_t);
  if (x_off) std::memcpy(x_off, b->x_off.data(), bytes);
  if (g_off) std::memcpy(g_off, b->g_off.data(), bytes);
  if (jac_off) std::memcpy(jac_off, b->j_off.data(), bytes);
  return TWR_OK;
}

int twr_batch_eval(twr_batch* b, const double* d_x, double* d_g, double* d_jac, int flags, void* hip_stream) {
  if (!b || !d_x) return fail(TWR_ERR_INVALID, "null argument");
  if ((flags & TWR_EVAL_BOTH) == 0) return fail(TWR_ERR_INVALID, "flags select nothing");
  if (((flags & TWR_EVAL_VALUES) && !d_g) || ((flags & TWR_EVAL_JACOBIAN) && !d_jac))
    return fail(TWR_ERR_INVALID, "missing output buffer");
  // One process may drive several devices: the launch needs the batch's device current.  The calling thread's current
  // device is restored afterwards and its error state is left alone (the launch's own status is what is returned).
  DeviceScope on(b->device);
  if (on.status != hipSuccess) return fail(TWR_ERR_HIP, "hipSetDevice failed");
  hipEvent_t* ev = nullptr;
  if (b->prof_count < b->prof_capacity) ev = b->prof_events.data() + 4 * b->prof_count++;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  hipError_t e = twr::launch_eval(b->n_ee, b->n_cu, b->d_dyn, b->n_dyn, b->dyn_map_chunks, b->d_rom, b->n_rom, b->rom_max_vals, b->d_node, b->n_node, b->node_families, b->d_fam, b->n_fam,
                                  b->d_pdyn, b->n_pdyn, b->pdyn_img_cap, b->d_ploc, b->n_ploc, b->d_prom, b->n_prom,
                                  b->prom_img_cap, d_x, d_g, d_jac, b->d_dump, flags & TWR_EVAL_BOTH, b->stream_nt, b->d_flat, b->n_flat, b->flat_max_x, stream, ev);
  if (e != hipSuccess) return fail(TWR_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  if (flags & TWR_EVAL_CHECK) {
    e = twr::launch_check(b->n_problems, b->d_goff, b->d_joff, d_g, d_jac, b->d_status, flags & TWR_EVAL_BOTH, stream);
    if (e != hipSuccess) return fail(TWR_ERR_HIP, std::string("check kernel launch: ") + hipGetErrorString(e));
  }
  return TWR_OK;
}

int twr_batch_status(twr_batch* b, int32_t* h_status, void* hip_stream) {
  if (!b || !h_status) return fail(TWR_ERR_INVALID, "null argument");
  try {
    DeviceScope on(b->device);
    TWR_HIP(on.status);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    TWR_HIP(hipMemcpyAsync(h_status, b->d_status, sizeof(int32_t) * (size_t)b->n_problems, hipMemcpyDeviceToHost, stream));
    TWR_HIP(hipStreamSynchronize(stream));
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_HIP, e.what());
  }
}

int twr_batch_profile_begin(twr_batch* b, int max_evals) {
  if (!b || max_evals < 1) return fail(TWR_ERR_INVALID, "bad arguments");
  try {
    DeviceScope on(b->device);
    TWR_HIP(on.status);
    for (hipEvent_t e : b->prof_events) (void)hipEventDestroy(e);
    b->prof_events.assign(4 * (size_t)max_evals, nullptr);
    for (auto& e : b->prof_events) TWR_HIP(hipEventCreate(&e));
    b->prof_capacity = max_evals;
    b->prof_count = 0;
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_HIP, e.what());
  }
}

int twr_batch_profile_end(twr_batch* b, double avg_ms[3], int* n_evals) {
  if (!b || !avg_ms) return fail(TWR_ERR_INVALID, "null argument");
  try {
    DeviceScope on(b->device);   // (events belong to the batch's device, like in _begin)
    TWR_HIP(on.status);
    const int n = b->prof_count;
    avg_ms[0] = avg_ms[1] = avg_ms[2] = 0.0;
    if (n > 0) {
      TWR_HIP(hipEventSynchronize(b->prof_events[4 * (size_t)n - 1]));
      for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
          float ms = 0.f;
          TWR_HIP(hipEventElapsedTime(&ms, b->prof_events[4 * (size_t)i + k], b->prof_events[4 * (size_t)i + k + 1]));
          avg_ms[k] += ms / n;
        }
    }
    if (n_evals) *n_evals = n;
    b->prof_capacity = 0;
    b->prof_count = 0;
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_HIP, e.what());
  }
}

int twr_batch_eval_host(twr_batch* b, const double* h_x, double* h_g, double* h_jac, int flags) {
  if (!b || !h_x) return fail(TWR_ERR_INVALID, "null argument");
  try {
    DeviceScope on(b->device);
    TWR_HIP(on.status);
    const size_t nx = b->x_off.back(), ng = b->g_off.back(), nj = b->j_off.back();
    // device staging buffers, each on first need (the zero-copy branch below needs none of them: a single-problem adapter
    // batch that only ever hands over its own page-locked buffers allocates nothing in HBM here)
    auto need_x = [&]() {
      if (!b->d_x) TWR_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_x), nx * sizeof(double)));
    };
    auto need_out = [&]() {
      if (!b->d_g) TWR_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_g), ng * sizeof(double)));
      if (!b->d_j) TWR_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_j), nj * sizeof(double)));
    };
    if (!b->host_stream) TWR_HIP(hipStreamCreateWithFlags(&b->host_stream, hipStreamNonBlocking));
    hipStream_t hs = b->host_stream;
    // one stream-ordered chain on the batch's own stream and a single synchronisation (with page-locked buffers the copies are DMA)
#ifdef TWR_TUNING_KNOBS   // (include/towr_amd.h, "Tuning knobs")
    static const bool zero_copy = [] { const char* e = getenv("TWR_HOST_ZERO_COPY"); return !e || atoi(e) != 0; }();
    static const bool zero_copy_x = [] { const char* e = getenv("TWR_HOST_ZERO_COPY_X"); return !e || atoi(e) != 0; }();
#else
    const bool zero_copy = true, zero_copy_x = true;
#endif
    const bool zc = zero_copy && b->p_g && h_g == b->p_g && h_jac == b->p_j && nj * sizeof(double) <= (size_t)(32u << 20);
    const double* dx = nullptr;
    if (zc && zero_copy_x && h_x == b->p_x) {   // x too: the kernels gather it straight from the page-locked buffer
      TWR_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(const_cast<double**>(&dx)), b->p_x, 0));
    } else {
      need_x();
      dx = b->d_x;
      TWR_HIP(hipMemcpyAsync(b->d_x, h_x, nx * sizeof(double), hipMemcpyHostToDevice, hs));
    }
    if (zc) {
      // The batch's own page-locked buffers, small batch (the single-problem callback of the ifopt adapter): the
      // kernels store g and the Jacobian values straight into host memory over PCIe -- coalesced 16-byte stores,
      // each value written once -- instead of HBM plus two device-to-host copies.
      double *dg = nullptr, *dj = nullptr;
      TWR_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dg), b->p_g, 0));
      TWR_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dj), b->p_j, 0));
      int rc0 = twr_batch_eval(b, dx, dg, dj, flags, hs);
      if (rc0 != TWR_OK) return rc0;
      TWR_HIP(hipStreamSynchronize(hs));
      return TWR_OK;
    }
    need_out();
    int rc = twr_batch_eval(b, dx, b->d_g, b->d_j, flags, hs);
    if (rc != TWR_OK) return rc;
    if ((flags & TWR_EVAL_VALUES) && h_g)
      TWR_HIP(hipMemcpyAsync(h_g, b->d_g, ng * sizeof(double), hipMemcpyDeviceToHost, hs));
    if ((flags & TWR_EVAL_JACOBIAN) && h_jac)
      TWR_HIP(hipMemcpyAsync(h_jac, b->d_j, nj * sizeof(double), hipMemcpyDeviceToHost, hs));
    TWR_HIP(hipStreamSynchronize(hs));
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_HIP, e.what());
  }
}

int twr_structure_sample_count(const twr_structure* s, double dt, int32_t* n_samples) {
  if (!s || !n_samples) return fail(TWR_ERR_INVALID, "null argument");
  try {
    *n_samples = s->s.SampleCount(dt);
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_INVALID, e.what());
  }
}

int twr_batch_sample(twr_batch* b, const double* d_x, double dt, double* d_out, int64_t problem_stride, void* hip_stream) {
  if (!b || !d_x || !d_out) return fail(TWR_ERR_INVALID, "null argument");
  try {
    DeviceScope on(b->device);
    TWR_HIP(on.status);
    if (!b->d_swork || b->swork_dt != dt || b->swork_stride != problem_stride) {  // (re)build the work list
      std::vector<twr::SampleWork> work;
      for (int p = 0; p < b->n_problems; ++p) {
        if (!(dt > 0)) throw std::runtime_error("dt must be positive");
        int n = 0;  // fpowr GetTrajectory: while (t <= T + 1e-5) { ...; t += dt; }
        for (double t = 0.0; t <= b->t_total[p] + 1e-5; t += dt)
          if (++n > 10000000) throw std::runtime_error("too many samples");
        if ((int64_t)n * (20 + 13 * b->n_ee) > problem_stride) throw std::runtime_error("problem_stride too small for the samples");
        if (!b->sample_ok[p]) throw std::runtime_error("too many polynomials per spline for trajectory sampling");
        for (int s0 = 0; s0 < n; s0 += 64) {
          twr::SampleWork w;
          w.blob = b->blob_of_problem[p];
          w.x_off = b->x_off[p];
          w.out_off = (int64_t)p * problem_stride;
          w.s0 = s0;
          w.cnt = std::min(64, n - s0);
          work.push_back(w);
        }
      }
      if (b->d_swork) TWR_HIP(hipFree(b->d_swork));
      b->d_swork = nullptr;
      TWR_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_swork), work.size() * sizeof(twr::SampleWork)));
      TWR_HIP(hipMemcpy(b->d_swork, work.data(), work.size() * sizeof(twr::SampleWork), hipMemcpyHostToDevice));
      b->n_swork = (int)work.size();
      b->swork_dt = dt;
      b->swork_stride = problem_stride;
    }
    hipError_t e = twr::launch_sample(b->d_swork, b->n_swork, d_x, d_out, dt, nullptr, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail(TWR_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_HIP, e.what());
  }
}

int twr_batch_initial_guess(twr_batch* b, const double* d_x, const double* d_times, int32_t n_times, double* d_out,
                            int64_t problem_stride, void* hip_stream) {
  if (!b || !d_x || !d_times || !d_out || n_times < 1) return fail(TWR_ERR_INVALID, "bad arguments");
  if ((int64_t)n_times * 49 > problem_stride) return fail(TWR_ERR_INVALID, "problem_stride too small for the records");
  try {
    DeviceScope on(b->device);
    TWR_HIP(on.status);
    if (!b->d_gwork || b->gwork_times != n_times || b->gwork_stride != problem_stride) {
      std::vector<twr::SampleWork> work;
      for (int p = 0; p < b->n_problems; ++p) {
        if (!b->sample_ok[p]) throw std::runtime_error("too many polynomials per spline for trajectory sampling");
        for (int s0 = 0; s0 < n_times; s0 += 64) {
          twr::SampleWork w;
          w.blob = b->blob_of_problem[p];
          w.x_off = b->x_off[p];
          w.out_off = (int64_t)p * problem_stride;
          w.s0 = s0;
          w.cnt = std::min(64, n_times - s0);
          work.push_back(w);
        }
      }
      if (b->d_gwork) TWR_HIP(hipFree(b->d_gwork));
      b->d_gwork = nullptr;
      TWR_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_gwork), work.size() * sizeof(twr::SampleWork)));
      TWR_HIP(hipMemcpy(b->d_gwork, work.data(), work.size() * sizeof(twr::SampleWork), hipMemcpyHostToDevice));
      b->n_gwork = (int)work.size();
      b->gwork_times = n_times;
      b->gwork_stride = problem_stride;
    }
    hipError_t e = twr::launch_sample(b->d_gwork, b->n_gwork, d_x, d_out, 0.0, d_times, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail(TWR_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_HIP, e.what());
  }
}

int twr_planes_create(const double* regions, const double* boundary_xy, const int32_t* boundary_start, int32_t n_regions,
                      int device, twr_planes** out) {
  if (!out || n_regions < 0 || (n_regions > 0 && (!regions || !boundary_start))) return fail(TWR_ERR_INVALID, "bad arguments");
  std::unique_ptr<twr_planes> pl;
  try {   // argument errors
    pl = std::make_unique<twr_planes>();
    pl->device = device;
    pl->start.assign(1, 0);
    for (int r = 0; r < n_regions; ++r) {
      if (boundary_start[r + 1] < boundary_start[r] || boundary_start[0] != 0) throw std::runtime_error("boundary_start must ascend from 0");
      // (boost::geometry::distance throws on an empty geometry; a region without boundary points cannot be the nearest)
      if (boundary_start[r + 1] == boundary_start[r]) throw std::runtime_error("planar region " + std::to_string(r) + " has no boundary points");
      pl->start.push_back(boundary_start[r + 1]);
    }
    const int n_pts = pl->start.back();
    if (n_pts > 0 && !boundary_xy) throw std::runtime_error("boundary_xy is null");
    pl->world_xy.resize(2 * (size_t)n_pts);
    for (int r = 0; r < n_regions; ++r) {   // PlanarRegionsToPolygons: tf::Matrix3x3(q) * (x, y, 0) + position
      const double* P = regions + 7 * r;
      const double x = P[3], y = P[4], z = P[5], w = P[6];
      const double d = x * x + y * y + z * z + w * w;
      if (!(d > 0)) throw std::runtime_error("zero orientation quaternion");
      const double s2 = 2.0 / d, xs = x * s2, ys = y * s2, zs = z * s2, wz = w * zs, xx = x * xs, xy = x * ys, yy = y * ys, zz = z * zs;
      const double R00 = 1.0 - (yy + zz), R01 = xy - wz, R10 = xy + wz, R11 = 1.0 - (xx + zz);
      for (int i = pl->start[r]; i < pl->start[r + 1]; ++i) {
        const double lx = boundary_xy[2 * i], ly = boundary_xy[2 * i + 1];
        pl->world_xy[2 * i] = (R00 * lx + R01 * ly + 0.0) + P[0];
        pl->world_xy[2 * i + 1] = (R10 * lx + R11 * ly + 0.0) + P[1];
      }
    }
  } catch (const std::exception& e) {
    return fail(TWR_ERR_INVALID, e.what());
  }
  try {   // device errors (the handle's destructor releases what was allocated)
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) return fail(TWR_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n_dev) return fail(TWR_ERR_INVALID, "device ordinal out of range");
    DeviceScope on(device);
    TWR_HIP(on.status);
    const int n_pts = pl->start.back();
    TWR_HIP(hipMalloc(reinterpret_cast<void**>(&pl->d_xy), std::max<size_t>(16, pl->world_xy.size() * sizeof(double))));
    TWR_HIP(hipMalloc(reinterpret_cast<void**>(&pl->d_start), pl->start.size() * sizeof(int32_t)));
    if (n_pts > 0) TWR_HIP(hipMemcpy(pl->d_xy, pl->world_xy.data(), pl->world_xy.size() * sizeof(double), hipMemcpyHostToDevice));
    TWR_HIP(hipMemcpy(pl->d_start, pl->start.data(), pl->start.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    *out = pl.release();
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_HIP, e.what());
  }
}

void twr_planes_destroy(twr_planes* planes) { delete planes; }

int twr_planes_world_xy(const twr_planes* planes, double* world_xy) {
  if (!planes || !world_xy) return fail(TWR_ERR_INVALID, "null argument");
  std::memcpy(world_xy, planes->world_xy.data(), planes->world_xy.size() * sizeof(double));
  return TWR_OK;
}

int twr_batch_contact_planes(twr_batch* b, const twr_planes* planes, const double* d_plan, const int32_t* d_counts,
                             int32_t max_steps, int32_t* d_plane_index, void* hip_stream) {
  if (!b || !planes || !d_plan || !d_counts || !d_plane_index || max_steps < 1) return fail(TWR_ERR_INVALID, "bad arguments");
  if (planes->device != b->device) return fail(TWR_ERR_INVALID, "planes and batch live on different devices");
  DeviceScope on(b->device);
  if (on.status != hipSuccess) return fail(TWR_ERR_HIP, "hipSetDevice failed");
  hipError_t e = twr::launch_planes(d_plan, d_counts, planes->d_xy, planes->d_start, (int)planes->start.size() - 1, b->n_problems,
                                    max_steps, b->n_ee, d_plane_index, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return fail(TWR_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  return TWR_OK;
}

int twr_batch_score(twr_batch* b, const double* d_g, double* d_scores, void* hip_stream) {
  if (!b || !d_g || !d_scores) return fail(TWR_ERR_INVALID, "null argument");
  DeviceScope on(b->device);
  if (on.status != hipSuccess) return fail(TWR_ERR_HIP, "hipSetDevice failed");
  hipError_t e = twr::launch_score(b->d_node, b->n_problems, d_g, d_scores, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return fail(TWR_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  return TWR_OK;
}

int twr_batch_score_best(twr_batch* b, const double* d_g, double* d_scores, uint32_t families, int64_t index_offset, double* d_best,
                         void* hip_stream) {
  if (!b || !d_g || !d_scores || !d_best) return fail(TWR_ERR_INVALID, "null argument");
  if (!(families & 0xffu) || (families & ~0xffu)) return fail(TWR_ERR_INVALID, "families must be a non-empty mask of the eight TWR_SET_* bits");
  DeviceScope on(b->device);
  if (on.status != hipSuccess) return fail(TWR_ERR_HIP, "hipSetDevice failed");
  // two launches behind one call: the scores, then the arg-min over this batch's rows.  (One launch -- the scoring kernel's
  // last workgroup taking the decision behind a block counter -- was built and measured: 1024 atomics on one address cost
  // more than the launch they save, 73 vs 63 us per planner step of 1024 candidates)
  unsigned* counter = reinterpret_cast<unsigned*>(b->d_best + 2 * (size_t)twr::best_max_blocks());
  hipError_t e = twr::launch_score(b->d_node, b->n_problems, d_g, d_scores, static_cast<hipStream_t>(hip_stream));
  if (e == hipSuccess)
    e = twr::launch_best(d_scores, b->n_problems, families, b->d_best, counter, d_best, (double)index_offset, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return fail(TWR_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  return TWR_OK;
}

int twr_batch_best(twr_batch* b, const double* d_scores, int32_t n_candidates, uint32_t families, double* d_best, void* hip_stream) {
  if (!b || !d_scores || !d_best) return fail(TWR_ERR_INVALID, "null argument");
  if (n_candidates < 1) return fail(TWR_ERR_INVALID, "twr_batch_best needs at least one candidate");
  if (!(families & 0xffu) || (families & ~0xffu)) return fail(TWR_ERR_INVALID, "families must be a non-empty mask of the eight TWR_SET_* bits");
  DeviceScope on(b->device);
  if (on.status != hipSuccess) return fail(TWR_ERR_HIP, "hipSetDevice failed");
  unsigned* counter = reinterpret_cast<unsigned*>(b->d_best + 2 * (size_t)twr::best_max_blocks());
  hipError_t e = twr::launch_best(d_scores, n_candidates, families, b->d_best, counter, d_best, 0.0, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return fail(TWR_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  return TWR_OK;
}

int twr_structure_contact_steps_max(const twr_structure* s, int32_t* max_steps) {
  if (!s || !max_steps) return fail(TWR_ERR_INVALID, "null argument");
  int n = 1;   // the first sample, then at most one footstep state per phase change of any foot
  for (int e = 0; e < s->s.n_ee; ++e) n += s->s.schedule.n_phases[e] - 1;
  *max_steps = n;
  return TWR_OK;
}

int twr_batch_contact_plan(twr_batch* b, const double* d_x, double dt, double time_horizon, double* d_out, int32_t max_steps,
                           int32_t* d_counts, void* hip_stream) {
  if (!b || !d_x || !d_out || !d_counts || max_steps < 1 || !(dt > 0)) return fail(TWR_ERR_INVALID, "bad arguments");
  try {
    DeviceScope on(b->device);
    TWR_HIP(on.status);
    int n_max = 0;
    for (int p = 0; p < b->n_problems; ++p) {
      if (!b->sample_ok[p]) throw std::runtime_error("too many polynomials per spline for trajectory sampling");
      int n = 0;  // fpowr GetTrajectory: while (t <= T + 1e-5) { ...; t += dt; }
      for (double t = 0.0; t <= b->t_total[p] + 1e-5; t += dt)
        if (++n > 10000000) throw std::runtime_error("too many samples");
      n_max = std::max(n_max, n);
    }
    hipError_t e = twr::launch_contact_plan(b->d_node, b->n_problems, d_x, d_out, d_counts, dt, time_horizon, n_max, max_steps,
                                            static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail(TWR_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_HIP, e.what());
  }
}

int twr_batch_host_buffers(twr_batch* b, double** h_x, double** h_g, double** h_jac) {
  if (!b) return fail(TWR_ERR_INVALID, "null batch");
  try {
    DeviceScope on(b->device);
    TWR_HIP(on.status);
    if (!b->p_x) {
      TWR_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->p_x), b->x_off.back() * sizeof(double), hipHostMallocDefault));
      TWR_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->p_g), b->g_off.back() * sizeof(double), hipHostMallocDefault));
      TWR_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->p_j), b->j_off.back() * sizeof(double), hipHostMallocDefault));
    }
    if (h_x) *h_x = b->p_x;
    if (h_g) *h_g = b->p_g;
    if (h_jac) *h_jac = b->p_j;
    return TWR_OK;
  } catch (const std::exception& e) {
    return fail(TWR_ERR_HIP, e.what());
  }
}

}  // extern "C"
