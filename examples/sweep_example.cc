// A contact-schedule sweep through the C ABI only (include/towr_amd.h), the way a C++ caller -- e.g. a planner in
// place of fpowr's single hard-coded gait (fpowr/src/footstep_plan_server.cc:191-200) -- would drive it:
//   enumerate gait candidates (GaitGenerator tables) -> structures (threaded) -> one batch on the GPU ->
//   initial guess as x -> constraint values + Jacobian -> per-candidate bound-violation scores ->
//   footstep plan of the best candidate (contact changes, foot positions, nearest planar region).
// Build:  hipcc -std=c++17 -I include examples/sweep_example.cc -L towr_amd -ltowr_amd -Wl,-rpath,$PWD/towr_amd
// Prints one line per step and "best candidate <index> score <value>"; exits non-zero on any error.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "towr_amd.h"

#define CHECK(call)                                                          \
  do {                                                                       \
    if ((call) != TWR_OK) {                                                  \
      std::fprintf(stderr, "%s failed: %s\n", #call, twr_last_error());      \
      return 1;                                                              \
    }                                                                        \
  } while (0)
#define HIP(call)                                                            \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      std::fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_)); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

int main(int argc, char** argv) {
  const int n_cand = argc > 1 ? std::atoi(argv[1]) : 64;   // BASELINE C4: 64 candidates
  twr_model model;
  CHECK(twr_model_preset(TWR_ROBOT_ANYMAL, TWR_TERRAIN_STAIRS, &model));

  // candidates: combo x total time x swing scale (SURVEY 8d enumeration), K = 200 time nodes each
  std::vector<twr_schedule> scheds;
  std::vector<twr_params> params;
  for (int combo = 0; combo < 5 && (int)scheds.size() < n_cand; ++combo)
    for (int i = 0; i < 8 && (int)scheds.size() < n_cand; ++i)
      for (int j = 0; j < 26 && (int)scheds.size() < n_cand; ++j) {
        const double T = 1.2 + 0.2 * i, scale = 0.80 + 0.016 * j;
        twr_schedule s;
        CHECK(twr_gait_combo(model.n_ee, combo, T, scale, &s));
        twr_params p;
        CHECK(twr_params_default(&p));
        p.dt_dynamic = p.dt_rom = T / (200 - 1.5);
        scheds.push_back(s);
        params.push_back(p);
      }
  const int B = (int)scheds.size();
  std::vector<twr_structure*> structs(B);
  CHECK(twr_structure_create_many(&model, scheds.data(), params.data(), B, 0, structs.data()));

  std::vector<int32_t> map(B);
  for (int p = 0; p < B; ++p) map[p] = p;
  twr_batch* batch = nullptr;
  CHECK(twr_batch_create(structs.data(), B, map.data(), B, 0, &batch));
  std::vector<int64_t> x_off(B + 1), g_off(B + 1), j_off(B + 1);
  CHECK(twr_batch_layout(batch, x_off.data(), g_off.data(), j_off.data()));
  std::printf("batch: %d candidates, %lld variables, %lld rows, %lld Jacobian values\n", B, (long long)x_off[B],
              (long long)g_off[B], (long long)j_off[B]);

  // x = the reference's initial guess of every candidate (nlp_formulation.cc:95-181): 2 m forward over the stairs
  std::vector<double> x(x_off[B]);
  const double z0 = 0.5, lin0[3] = {0, 0, z0}, ang0[3] = {0, 0, 0}, lin1[3] = {2.0, 0, z0}, ang1[3] = {0, 0, 0};
  const double ee0[12] = {0.34, 0.19, 0, 0.34, -0.19, 0, -0.34, 0.19, 0, -0.34, -0.19, 0};
  for (int p = 0; p < B; ++p) CHECK(twr_structure_initial_guess(structs[p], lin0, ang0, lin1, ang1, ee0, x.data() + x_off[p]));

  double *d_x, *d_g, *d_j, *d_scores;
  HIP(hipMalloc(&d_x, x.size() * sizeof(double)));
  HIP(hipMalloc(&d_g, g_off[B] * sizeof(double)));
  HIP(hipMalloc(&d_j, j_off[B] * sizeof(double)));
  HIP(hipMalloc(&d_scores, 16 * (size_t)B * sizeof(double)));
  HIP(hipMemcpy(d_x, x.data(), x.size() * sizeof(double), hipMemcpyHostToDevice));
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));

  // one NLP callback for all candidates (what Ipopt's eval_g + eval_jac_g ask for), then the scores
  CHECK(twr_batch_eval(batch, d_x, d_g, d_j, TWR_EVAL_BOTH | TWR_EVAL_CHECK, stream));
  CHECK(twr_batch_score(batch, d_g, d_scores, stream));
  std::vector<int32_t> status(B);
  CHECK(twr_batch_status(batch, status.data(), stream));
  std::vector<double> scores(16 * (size_t)B);
  HIP(hipMemcpyAsync(scores.data(), d_scores, scores.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIP(hipStreamSynchronize(stream));
  int best = -1;
  double best_score = INFINITY;
  for (int p = 0; p < B; ++p) {
    if (status[p] != 0) continue;   // NaN / Inf somewhere in g or the Jacobian of this candidate
    // inf-norm violation of the terrain (0), dynamic (1), range-of-motion (3) and force (4) families
    const double s = scores[16 * p + 0] + scores[16 * p + 2] + scores[16 * p + 6] + scores[16 * p + 8];
    if (s < best_score) {
      best_score = s;
      best = p;
    }
  }
  std::printf("best candidate %d score %.12e\n", best, best_score);
  if (best < 0) return 2;

  // footstep plan of every candidate (fpowr ExtractFootstepPlan), nearest planar region per foot in contact
  int32_t max_steps = 1;
  for (int p = 0; p < B; ++p) {
    int32_t m;
    CHECK(twr_structure_contact_steps_max(structs[p], &m));
    if (m > max_steps) max_steps = m;
  }
  const int rec = 2 + 4 * model.n_ee;
  double* d_plan;
  int32_t *d_counts, *d_planes;
  HIP(hipMalloc(&d_plan, (size_t)B * max_steps * rec * sizeof(double)));
  HIP(hipMalloc(&d_counts, B * sizeof(int32_t)));
  HIP(hipMalloc(&d_planes, (size_t)B * max_steps * model.n_ee * sizeof(int32_t)));
  CHECK(twr_batch_contact_plan(batch, d_x, 0.01, 2.0, d_plan, max_steps, d_counts, stream));
  // three planar regions along the path: ground before the stairs, the first step, the upper level (axis-aligned)
  const double regions[3 * 7] = {0.5, 0, 0, 0, 0, 0, 1, 1.25, 0, 0.2, 0, 0, 0, 1, 2.2, 0, 0.4, 0, 0, 0, 1};
  const double sq[3][4][2] = {{{-1.0, -1}, {-1.0, 1}, {0.5, 1}, {0.5, -1}}, {{-0.25, -1}, {-0.25, 1}, {0.25, 1}, {0.25, -1}},
                              {{-0.7, -1}, {-0.7, 1}, {1.0, 1}, {1.0, -1}}};
  std::vector<double> boundary;
  std::vector<int32_t> start(1, 0);
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 5; ++k) {   // closed rings: first point repeated
      boundary.push_back(sq[r][k % 4][0]);
      boundary.push_back(sq[r][k % 4][1]);
    }
    start.push_back((int32_t)boundary.size() / 2);
  }
  twr_planes* planes = nullptr;
  CHECK(twr_planes_create(regions, boundary.data(), start.data(), 3, 0, &planes));
  CHECK(twr_batch_contact_planes(batch, planes, d_plan, d_counts, max_steps, d_planes, stream));
  std::vector<double> plan((size_t)max_steps * rec);
  std::vector<int32_t> idx((size_t)max_steps * model.n_ee);
  int32_t count = 0;
  HIP(hipMemcpyAsync(plan.data(), d_plan + (size_t)best * max_steps * rec, plan.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIP(hipMemcpyAsync(idx.data(), d_planes + (size_t)best * max_steps * model.n_ee, idx.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  HIP(hipMemcpyAsync(&count, d_counts + best, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  HIP(hipStreamSynchronize(stream));
  std::printf("footstep plan of candidate %d: %d states\n", best, count);
  for (int s = 0; s < count; ++s) {
    std::printf("  t %.2f for %.2f s  contact/plane:", plan[(size_t)s * rec], plan[(size_t)s * rec + 1]);
    for (int e = 0; e < model.n_ee; ++e) std::printf(" %d/%d", (int)plan[(size_t)s * rec + 2 + e], idx[(size_t)s * model.n_ee + e]);
    std::printf("\n");
  }

  twr_planes_destroy(planes);
  twr_batch_destroy(batch);
  for (twr_structure* s : structs) twr_structure_destroy(s);
  (void)hipFree(d_x); (void)hipFree(d_g); (void)hipFree(d_j); (void)hipFree(d_scores);
  (void)hipFree(d_plan); (void)hipFree(d_counts); (void)hipFree(d_planes);
  (void)hipStreamDestroy(stream);
  return 0;
}
