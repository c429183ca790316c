/* towr_amd -- MI355X-native evaluation of towr's NLP constraint / Jacobian hot path.
 *
 * C ABI of libtowr_amd.so.  Plain pointers and sizes only; no C++/torch types.
 * Every entry point names the reference interface it replaces (file:line under
 * KaiNakamura/towr).  The reference has no C ABI: the path sits
 * behind the C++ virtual interface ifopt::ConstraintSet, so the binding a maintainer adds
 * is the small C++ adapter shown in INTEGRATION.md (towr_amd/csrc/ifopt_adapter.h).
 *
 * Conventions
 *   - all functions return TWR_OK (0) or a negative error code; twr_last_error() gives
 *     the message of the last failure on the calling thread.  No exceptions cross the ABI.
 *   - handles are thread-compatible: distinct handles may be used from distinct threads.
 *   - devices: a batch (and a twr_planes handle) lives on the device it was created for.  Every entry point makes
 *     that device current for its own HIP calls and restores the calling thread's current device before it
 *     returns; none reads or clears the thread's sticky HIP error.
 *   - "x" is the stacked ifopt variable vector in the reference order
 *        base-lin | base-ang | ee-motion_0.. | ee-force_0.. [| ee-schedule0..]   (nlp_formulation.cc:63-93)
 *     "g" are the stacked constraint values and "jac" the Jacobian non-zeros in the CSR
 *     order ifopt::Problem::EvalNonzerosOfJacobian copies out (row-major, columns
 *     ascending, explicit structural zeros kept), for the constraint sets
 *        terrain-ee-motion_e.. | dynamic | splineacc-base-lin | splineacc-base-ang |
 *        rangeofmotion-e.. | force-ee-force_e.. | swing-ee-motion_e.. | baseMotion | totalduration-e..
 *     i.e. params_.constraints_ order (parameters.cc:55-60); twr_params.constraint_sets selects
 *     which families exist (default: the four of the hot path).
 */
#ifndef TOWR_AMD_H_
#define TOWR_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TWR_MAX_EE 4
#define TWR_MAX_PHASES 32
#define TWR_NAME_LEN 40

enum { TWR_OK = 0, TWR_ERR_INVALID = -1, TWR_ERR_HIP = -2, TWR_ERR_NO_DEVICE = -3, TWR_ERR_INTERNAL = -4 };

/* RobotModel::Robot (robot_model.h:70-75) */
enum { TWR_ROBOT_MONOPED = 0, TWR_ROBOT_BIPED, TWR_ROBOT_HYQ, TWR_ROBOT_ANYMAL, TWR_ROBOT_GO1 };
/* HeightMap::TerrainID (height_map.h:79-86) */
enum { TWR_TERRAIN_FLAT = 0, TWR_TERRAIN_BLOCK, TWR_TERRAIN_STAIRS, TWR_TERRAIN_GAP, TWR_TERRAIN_SLOPE,
       TWR_TERRAIN_CHIMNEY, TWR_TERRAIN_CHIMNEY_LR,
       /* HeightMapFromCSV (include/towr/terrain/height_map_from_csv.h): a gridded terrain, heights per 0.17 m
         * cell; needs a twr_terrain_grid handle, see twr_structure_create_with_grid */
       TWR_TERRAIN_CSV_GRID,
       /* Grid (include/towr/terrain/grid_height_map.h:15-60): the perception-driven terrain fpowr hands the solver
        * (fpowr/src/footstep_plan_server.cc:155) -- the float "elevation" layer of a ROS grid_map, bilinear
        * sample, central-difference slopes over resolution/6, FLT_MAX outside the map; needs a handle made by
        * twr_terrain_grid_map_create, see twr_structure_create_with_grid */
       TWR_TERRAIN_GRID_MAP };
enum { TWR_EVAL_VALUES = 1, TWR_EVAL_JACOBIAN = 2, TWR_EVAL_BOTH = 3,
       /* also run the per-problem NaN/Inf scan over the outputs of this evaluation (twr_batch_status) */
       TWR_EVAL_CHECK = 4 };
/* Parameters::ConstraintName entries of the default list (parameters.h:139-147, parameters.cc:55-60),
 * as bits of twr_params.constraint_sets.  The sets always appear in this (the reference's) order. */
enum {
  TWR_SET_TERRAIN = 1,   /* TerrainConstraint per ee          (nlp_formulation.cc:278-289) */
  TWR_SET_DYNAMIC = 2,   /* DynamicConstraint                 (nlp_formulation.cc:237-245) */
  TWR_SET_BASE_ACC = 4,  /* SplineAccConstraint base-lin/-ang (nlp_formulation.cc:319-331) */
  TWR_SET_ROM = 8,       /* RangeOfMotionConstraint per ee    (nlp_formulation.cc:247-262) */
  TWR_SET_FORCE = 16,    /* ForceConstraint per ee            (nlp_formulation.cc:291-304) */
  TWR_SET_SWING = 32,    /* SwingConstraint per ee            (nlp_formulation.cc:306-317) */
  /* Parameters::OptimizePhaseDurations() (parameters.cc:76-80): TotalDurationConstraint per ee
   * (nlp_formulation.cc:264-276) AND, because IsOptimizeTimings() becomes true (parameters.cc:128-135),
   * the phase durations join x as variable sets ee-schedule<e> (n_phases-1 each, after ee-force_*),
   * the ee splines become PhaseSplines (spline_holder.cc:48-52): every Jacobian row of an ee spline
   * holds all variables of its set, and dynamic / rangeofmotion rows gain the duration columns. */
  TWR_SET_TOTAL_TIME = 64,
  /* BaseMotionConstraint "baseMotion" (nlp_formulation.cc:229-235, base_motion_constraint.cc:38-99): not in the
   * default list; rows come after swing-* (where a caller's constraints_.push_back(BaseRom) lands) and before
   * totalduration-*.  Needs twr_params.dt_base_motion and .base_z_init. */
  TWR_SET_BASE_ROM = 128,
  TWR_SETS_HOT_PATH = 1 | 2 | 8 | 16,
  TWR_SETS_TOWR_DEFAULT = 63,
  TWR_SETS_ALL = 127,      /* default list + OptimizePhaseDurations() */
  TWR_SETS_EVERY = 255     /* every Parameters::ConstraintName */
};

/* Robot + terrain constants: the POD "model blob" that rank 0 broadcasts over RCCL.
 * Replaces towr::RobotModel {KinematicModel, DynamicModel} + HeightMap::Ptr
 * (robot_model.h:63-83, single_rigid_body_dynamics.h:66-77, height_map.h:71-137). */
typedef struct twr_model {
  int32_t n_ee;
  int32_t terrain_id;
  double mass;
  double inertia[6];                    /* Ixx,Iyy,Izz,Ixy,Ixz,Iyz as given to SingleRigidBodyDynamics() */
  double nominal_stance[TWR_MAX_EE][3]; /* KinematicModel::GetNominalStanceInBase */
  double max_dev[3];                    /* KinematicModel::GetMaximumDeviationFromNominal */
  double gravity;                       /* 9.80665, dynamic_model.cc:37 */
  double friction;                      /* 0.5, height_map.h:136 */
  double force_limit;                   /* Parameters::force_limit_in_normal_direction_, parameters.cc:48 */
  double flat_height;                   /* FlatGround(height) */
} twr_model;

/* Contact schedule of one candidate: Parameters::ee_phase_durations_ / ee_in_contact_at_start_
 * (parameters.h:168-171). */
typedef struct twr_schedule {
  int32_t n_ee;
  int32_t n_phases[TWR_MAX_EE];
  int32_t in_contact_at_start[TWR_MAX_EE];
  double phase_durations[TWR_MAX_EE][TWR_MAX_PHASES];
} twr_schedule;

/* Discretisation parameters (parameters.cc:43-51). */
typedef struct twr_params {
  double dt_dynamic;            /* dt_constraint_dynamic_ (0.1) */
  double dt_rom;                /* dt_constraint_range_of_motion_ (0.08) */
  double duration_base_poly;    /* duration_base_polynomial_ (0.1) */
  int32_t polys_per_swing;      /* ee_polynomials_per_swing_phase_ (2) */
  int32_t polys_per_stance_force; /* force_polynomials_per_stance_phase_ (3) */
  int32_t constraint_sets;      /* TWR_SET_* mask; twr_params_default: TWR_SETS_HOT_PATH */
  int32_t reserved_;            /* must be 0 */
  double dt_base_motion;        /* dt_constraint_base_motion_ (duration_base_polynomial_/4, parameters.cc:51) */
  double base_z_init;           /* initial base height: BaseMotionConstraint bounds z to [z-0.02, z+0.1]
                                   (base_motion_constraint.cc:51-55, read from the spline at construction).
                                   twr_params_default leaves it NaN; TWR_SET_BASE_ROM is rejected until it is set. */
} twr_params;

typedef struct twr_sizes {
  int32_t n_vars, n_rows, nnz;
  int32_t n_var_sets, n_con_sets;
  int32_t k_dynamic, k_rom;     /* TimeDiscretizationConstraint::GetNumberOfNodes */
} twr_sizes;

typedef struct twr_set_info {
  char name[TWR_NAME_LEN];      /* ifopt component name, e.g. "rangeofmotion-2" */
  int32_t offset;               /* first variable index / first row */
  int32_t size;                 /* variables / rows */
  int32_t nnz_offset, nnz;      /* constraint sets only */
} twr_set_info;

typedef struct twr_terrain_grid twr_terrain_grid; /* host copy of a gridded terrain, shared by structures */
typedef struct twr_structure twr_structure; /* host: index maps + CSR pattern of one candidate */
typedef struct twr_batch twr_batch;         /* device: tables of a batch of candidates */

const char* twr_last_error(void);

/* RobotModel(Robot) + HeightMap::MakeTerrain(id) + Parameters defaults (robot_model.cc:41-68,
 * height_map.cc:37-50, parameters.cc:40-73). */
int twr_model_preset(int robot, int terrain, twr_model* out);
int twr_params_default(twr_params* out);

/* GaitGenerator::MakeGaitGenerator(n_ee)->SetCombo(combo); GetPhaseDurations(T, ee);
 * IsInContactAtStart(ee)  (gait_generator.cc:43-111, {monoped,biped,quadruped}_gait_generator.cc).
 * swing_scale multiplies every table entry whose contact state has a foot in the air before the
 * renormalisation to t_total (1.0 = reference tables); used to enumerate candidates. */
int twr_gait_combo(int n_ee, int combo, double t_total, double swing_scale, twr_schedule* out);

/* What NlpFormulation::GetVariableSets + GetConstraints + ifopt's LinkWithVariables compute once per
 * problem (nlp_formulation.cc:63-93,200-331): variable index maps, time grids, active-polynomial
 * tables and the x-independent CSR pattern of the stacked Jacobian. */
int twr_structure_create(const twr_model* model, const twr_schedule* schedule, const twr_params* params,
                         twr_structure** out);
/* Gridded terrain of HeightMapFromCSV (height_map_from_csv.h:16-27): heights[y_cell * cols + x_cell], i.e. the
 * matrix the reference fills from the CSV file; cell size 0.17 m and slope window cell/50 as in :112-115.
 * Structures keep a reference to the grid (it may be destroyed right after they are created); a batch
 * uploads every distinct grid once. */
int twr_terrain_grid_create(const double* heights, int rows, int cols, twr_terrain_grid** out);
/* The "elevation" layer of a grid_map::GridMap for TWR_TERRAIN_GRID_MAP: elevation[i + j * size_x] is cell (i, j)
 * (the column-major float matrix grid_map keeps; i runs along -x, j along -y from the corner with the largest x
 * and y), cell size `resolution`, map centre (pos_x, pos_y) = GridMap::getPosition(); the map must have start
 * index (0,0) (GridMap::convertToDefaultStartIndex()).  Sampling follows grid_map's published
 * atPosition(..., INTER_LINEAR): bilinear over the 2x2 cells around the position, nearest cell in the half-cell
 * border band, std::out_of_range (-> FLT_MAX in Grid::GetHeight) outside the map. */
int twr_terrain_grid_map_create(const float* elevation, int size_x, int size_y, double resolution, double pos_x,
                                double pos_y, twr_terrain_grid** out);
void twr_terrain_grid_destroy(twr_terrain_grid* g);
/* As twr_structure_create for model->terrain_id == TWR_TERRAIN_CSV_GRID / TWR_TERRAIN_GRID_MAP (the grid must be
 * of the matching kind). */
int twr_structure_create_with_grid(const twr_model* model, const twr_schedule* schedule, const twr_params* params,
                                   const twr_terrain_grid* grid, twr_structure** out);
void twr_structure_destroy(twr_structure* s);
/* n candidates of one robot/terrain model at once (a sweep: SURVEY.md 8e "each rank builds descriptors for its
 * shard only"): structure i from schedules[i] / params[i], built on up to n_threads host threads (<= 0: all
 * hardware threads).  On failure nothing is left allocated and out[] is all NULL. */
int twr_structure_create_many(const twr_model* model, const twr_schedule* schedules, const twr_params* params, int n,
                              int n_threads, twr_structure** out);
/* The same for a gridded terrain (model->terrain_id == TWR_TERRAIN_CSV_GRID / TWR_TERRAIN_GRID_MAP): every structure
 * shares `grid` (one device copy per batch).  This is what a sweep over the perception-driven `Grid` terrain of
 * fpowr (footstep_plan_server.cc:155) uses; grid == NULL is twr_structure_create_many. */
int twr_structure_create_many_with_grid(const twr_model* model, const twr_schedule* schedules, const twr_params* params, int n,
                                        int n_threads, const twr_terrain_grid* grid, twr_structure** out);
/* Sharding a sweep over ranks / devices (contiguous shards balanced by BYTES, not counts -- candidates are
 * ragged).  twr_candidate_bytes: bytes[i] = 8 (n + m + nnz) of candidate i, the bytes one callback of it moves; builds
 * only the variable layout, the time tables and the CSR pattern (no device tables), on n_threads host threads (<= 0:
 * all).  Every rank computes the same numbers from the same candidate list, so no exchange is needed.
 * twr_shard_bounds: bounds[0..world], rank r owns candidates [bounds[r], bounds[r+1]); the boundary of rank r is the
 * prefix whose weight sum is closest to r / world of the total; never an empty shard; TWR_ERR_INVALID when n < world
 * (on every rank alike).  The reference has no counterpart: fpowr solves one gait per goal
 * (fpowr/src/footstep_plan_server.cc:191-200). */
int twr_candidate_bytes(const twr_model* model, const twr_schedule* schedules, const twr_params* params, int n, int n_threads,
                        int64_t* bytes /* n */);
int twr_shard_bounds(const double* weights /* n */, int n, int world, int32_t* bounds /* world + 1 */);
/* What a rank needs to rebuild a grid handle it received over the wire (towr_amd/dist.py broadcast_grid): kind (0: CSV
 * heights, double [rows][cols]; 1: grid_map elevation layer, float, column-major [size_x][size_y]), the two sizes,
 * resolution and map position (grid_map only), and the cell data (`data` points into the handle; valid until destroy). */
int twr_terrain_grid_info(const twr_terrain_grid* g, int32_t* kind, int32_t* rows_or_size_x, int32_t* cols_or_size_y,
                          double* resolution, double* pos_x, double* pos_y, const void** data);
int twr_structure_sizes(const twr_structure* s, twr_sizes* out);
/* Introspection of the values-only path (TWR_EVAL_VALUES; fixed timings, at most 2046 variables): the work items a problem
 * of this structure is cut into -- at most 64 consecutive time nodes of the "dynamic" resp. range-of-motion grid whose active
 * polynomials span at most eight polynomials per ee spline (time_discretization_constraint.cc:36-58 gives the grids,
 * spline.cc:48-78 the active polynomials).  *dynamic_takes_rom is 1 when the two grids coincide and the "dynamic" items
 * evaluate the "rangeofmotion-*" rows of their time nodes as well (then *n_rom_items is 0).  `items` may be NULL; otherwise
 * it receives (first time node, time nodes, polynomials of the widest window) per item, "dynamic" items first; a grid so coarse
 * that the windows would cut over a quarter more items than 64 time nodes each give (towr's default grids) is cut at 64 time
 * nodes alone, its lanes fetch their own polynomial records, and the third number is 0.  All counts are 0 for a structure that keeps
 * the Jacobian kernels' cut. */
int twr_structure_values_items(const twr_structure* s, int32_t* n_dynamic_items, int32_t* n_rom_items, int32_t* dynamic_takes_rom,
                               int32_t* items /* [n_dynamic_items + n_rom_items][3] or NULL */);
int twr_structure_var_set(const twr_structure* s, int i, twr_set_info* out);
int twr_structure_con_set(const twr_structure* s, int i, twr_set_info* out);
/* Library-owned, valid until twr_structure_destroy: CSR row_ptr[n_rows+1], col_idx[nnz]. */
const int32_t* twr_structure_row_ptr(const twr_structure* s);
const int32_t* twr_structure_col_idx(const twr_structure* s);
/* ConstraintSet::GetBounds of the stacked sets (dynamic_constraint.cc:66-71,
 * range_of_motion_constraint.cc:71-81, force_constraint.cc:91-105, terrain_constraint.cc:72-88). */
int twr_structure_bounds(const twr_structure* s, double* lower, double* upper);
/* Initial guess of NlpFormulation::Make{Base,Endeffector,Force}Variables (nlp_formulation.cc:95-181). */
int twr_structure_initial_guess(const twr_structure* s, const double init_base_lin[3],
                                const double init_base_ang[3], const double final_base_lin[3],
                                const double final_base_ang[3], const double* init_ee_pos /* n_ee*3 */,
                                double* x_out /* n_vars */);
/* Variable bounds (x_l, x_u) that the same functions put on the node variables: start state of base
 * and feet fixed, final base state fixed in the dimensions of Parameters::bounds_final_* (parameters.cc:65-69:
 * lin pos {x,y}, lin vel, ang pos, ang vel), everything else ifopt::NoBound (+-1e20)
 * (nlp_formulation.cc:109-122,151; nodes_variables.cc:152-181).  Base states are 12 doubles
 * {lin pos, lin vel, ang pos, ang vel}. */
int twr_structure_variable_bounds(const twr_structure* s, const double init_base[12], const double final_base[12],
                                  const double* init_ee_pos /* n_ee*3 */, double* lower /* n_vars */,
                                  double* upper /* n_vars */);

/* Upload the tables of a batch: problem p uses structs[struct_of_problem[p]].  device is the HIP
 * device ordinal of this process (one process per GPU).  All structures of one batch must have the same
 * number of end-effectors (<= TWR_MAX_EE): the dynamic kernel is specialised per n_ee and a batch is one launch
 * of it; robots with different leg counts go into separate batches. */
int twr_batch_create(const twr_structure* const* structs, int n_structs, const int32_t* struct_of_problem,
                     int n_problems, int device, twr_batch** out);
void twr_batch_destroy(twr_batch* b);
int twr_batch_num_problems(const twr_batch* b);
/* Device memory of the batch's tables, in bytes (any pointer may be NULL): `resident` = what the batch holds for all its
 * structures; `dyn_layout` = the layout tables of the "dynamic" set (index maps, CSR positions: everything that does not
 * hold a time) as built, `dyn_layout_distinct` = what is left of them after the batch has merged byte-identical tables of
 * different structures (candidates of a sweep that differ only in their total time share all of them), i.e. what one
 * evaluation reads.  No reference counterpart: towr recomputes these indices inside every callback
 * (towr/src/nodes_variables_phase_based.cc:210-298, spline.cc:48-78). */
int twr_batch_table_bytes(const twr_batch* b, int64_t* resident, int64_t* dyn_layout, int64_t* dyn_layout_distinct);
/* 1 when the batch's Jacobian values leave the dynamic / range-of-motion kernels with non-temporal stores, else 0.  Chosen by
 * twr_batch_create from the shape of the batch alone (fewer than four problems per structure on average AND more than
 * 256 MB of output per evaluation: a sweep whose candidates all bring their own tables); the values written are the
 * same bits either way. */
int twr_batch_streaming_stores(const twr_batch* b);
/* Ragged layout of the batch arrays, each n_problems+1 prefix sums in units of doubles:
 * problem p owns x[x_off[p]..x_off[p+1]), g[g_off[p]..), jac[jac_off[p]..). */
int twr_batch_layout(const twr_batch* b, int64_t* x_off, int64_t* g_off, int64_t* jac_off);

/* One full NLP callback for every problem of the batch, device pointers in, device pointers out:
 *   ifopt::Problem::EvaluateConstraints(x)        -> g     (TWR_EVAL_VALUES)
 *   ifopt::Problem::EvalNonzerosOfJacobian(x,val) -> jac   (TWR_EVAL_JACOBIAN)
 * i.e. Composite::SetVariables + {Terrain,Dynamic,RangeOfMotion,Force}Constraint::
 * {GetValues, FillJacobianBlock}.  Asynchronous on `hip_stream` (hipStream_t, may be
 * NULL for the default stream); no host synchronisation, capturable in a hipGraph.  If the calling thread's current
 * device is not the batch's, the call switches to it for the launches and RESTORES the caller's device before it
 * returns; it neither reads nor clears the thread's sticky HIP error (every launch returns its own status, which is
 * what a TWR_ERR_HIP result reports).  At most ONE evaluation of a given batch may
 * be in flight at a time (batches with optimised timings keep per-batch scratch records; the profiling
 * counters are per batch too): serialise evaluations of one batch on one stream, use one batch per stream.
 * REPRODUCIBILITY: for a FIXED value of `flags` a given x gives bit-identical g / Jacobian values wherever the problem sits
 * in a batch, on every device and rank.  The kernels are instantiated per output selection, and the selections (values
 * only / Jacobian only / both) agree with each other to rounding (<= 1e-13 of the set scale), NOT bit for bit.  A sweep
 * that compares candidates across ranks or calls must therefore score all of them with the SAME flags (near-tied
 * candidates could otherwise be ranked differently); twr_batch_score / twr_batch_best are deterministic given g.
 * (Values only: a batch in which EVERY problem has fixed timings and at most 2046 variables takes the lane-per-time-node
 * kernel, see twr_structure_values_items; one problem that cannot keeps the whole batch on the values-only instantiation of
 * the Jacobian kernels.  The two agree to rounding, so the shards of one sweep -- which all take the same path -- compare
 * bit for bit, a batch of another composition to rounding.) */
int twr_batch_eval(twr_batch* b, const double* d_x, double* d_g, double* d_jac, int flags, void* hip_stream);
/* Failure detection (the reference only has Release-mode-silent asserts, spline.cc:52,65): after an evaluation with
 * TWR_EVAL_CHECK, h_status[p] has bit 0 set if a constraint value of problem p is NaN/Inf and bit 1 if a Jacobian
 * value is.  Waits for hip_stream (the stream of that evaluation). */
int twr_batch_status(twr_batch* b, int32_t* h_status /* n_problems */, void* hip_stream);
/* Measurement aid (bench.py): after _begin, the next max_evals calls of twr_batch_eval also record
 * HIP events on their launch stream around each of the three kernels (dynamic, range of motion,
 * force/terrain nodes); _end waits for the last one and returns the average duration of each. */
int twr_batch_profile_begin(twr_batch* b, int max_evals);
int twr_batch_profile_end(twr_batch* b, double avg_ms[3], int* n_evals);
/* Trajectory sampling of a batch of solutions, fpowr::GetTrajectory (fpowr/include/fpowr/footstep_plan_extractor.h:
 * 19-53): problem p's x sampled every dt while t <= T + 1e-5 (t accumulated).  One record per sample, end-effectors
 * in towr order:  [ t | base lin p v a (9) | quaternion w x y z | omega (3) | omega_dot (3) |
 *                  per ee: contact (0/1), ee-motion p v a (9), ee-force (3) ]   = 20 + 13 n_ee doubles.
 * twr_structure_sample_count gives the records per problem; problem p's records start at
 * d_out + p * problem_stride (doubles).  Asynchronous on hip_stream. */
int twr_structure_sample_count(const twr_structure* s, double dt, int32_t* n_samples);
int twr_batch_sample(twr_batch* b, const double* d_x, double dt, double* d_out, int64_t problem_stride, void* hip_stream);
/* fpowr::ExtractInitialGuess / ExtractInitialGuesses (fpowr/include/fpowr/initial_guess_extractor.h:17-48) for a batch of
 * solutions: every problem's x sampled at the n_times times d_times[] (device array, within [0, T]; the goal's
 * state_sample_times).  One record of 49 doubles per time:
 *   [ t | state (12): base-lin p, base-ang p (Euler angles), base-lin v, base-ang v (Euler rates) |
 *     controls (36): ee-motion acceleration of ee i at 3 i, twelve zeros ("joint torques"), ee-force of ee i at 24 + 3 i ]
 * problem p's records start at d_out + p * problem_stride (doubles, >= 49 n_times).  Asynchronous on hip_stream. */
int twr_batch_initial_guess(twr_batch* b, const double* d_x, const double* d_times, int32_t n_times, double* d_out,
                            int64_t problem_stride, void* hip_stream);

/* Candidate scoring of a sweep (new; the reference solves one NLP per goal and lets Ipopt judge feasibility): for
 * problem p and constraint family f -- the bit index of its TWR_SET_* flag: 0 terrain, 1 dynamic, 2 splineacc, 3
 * rangeofmotion, 4 force, 5 swing, 6 totalduration, 7 baseMotion -- over the family's rows, with the bounds of
 * ConstraintSet::GetBounds (twr_structure_bounds):
 *   d_scores[16 p + 2 f]     = max_i max(lower_i - g_i, g_i - upper_i, 0)        (inf-norm of the violation)
 *   d_scores[16 p + 2 f + 1] = sum_i max(lower_i - g_i, g_i - upper_i, 0)        (1-norm)
 * (0 for families the structure does not build; NaN if a constraint value of the family is NaN).  d_g is the output of
 * twr_batch_eval with TWR_EVAL_VALUES.  Asynchronous on hip_stream. */
int twr_batch_score(twr_batch* b, const double* d_g, double* d_scores /* 16 * n_problems */, void* hip_stream);
/* The planner's decision without leaving the device: the candidate with the smallest SUM over the chosen constraint
 * families (bit f of `families` = family f of twr_batch_score, i.e. the TWR_SET_* bit) of the inf-norm violations
 * d_scores[16 c + 2 f], c < n_candidates.  The table may be longer than this batch (after an all-gather of every rank's
 * score rows it holds the whole sweep; `b` only names the device and owns the scratch).  A NaN total loses, the first
 * index wins a tie.  d_best[0] = index (as a double), d_best[1] = its total; one 16-byte copy brings the decision to the
 * host.  Asynchronous on hip_stream, capturable; one call per batch in flight at a time.
 * Replaces the host arg-min of a sweep driver over time_discretization_constraint.cc:65-75-style constraint values
 * (fpowr runs ONE candidate, footstep_plan_server.cc:193-195; a sweep over gait_generator.cc:54-105 candidates needs the
 * choice). */
int twr_batch_best(twr_batch* b, const double* d_scores, int32_t n_candidates, uint32_t families, double* d_best /* 2 */,
                   void* hip_stream);
/* twr_batch_score and twr_batch_best over THIS batch's candidates behind one call (two stream-ordered launches):
 * d_scores as twr_batch_score, d_best[0] = index_offset + the winning problem's index in the batch
 * (index_offset = the shard's first candidate: the result is then a global candidate index), d_best[1] = its total.  A
 * multi-rank sweep all-gathers the ranks' 16-byte results (the smallest total, then the smallest index, wins) instead of
 * 128 bytes per candidate.  Asynchronous on hip_stream, capturable; shares the scratch of twr_batch_best. */
int twr_batch_score_best(twr_batch* b, const double* d_g, double* d_scores /* 16 * n_problems */, uint32_t families,
                         int64_t index_offset, double* d_best /* 2 */, void* hip_stream);
/* fpowr::ExtractFootstepPlan (fpowr/include/fpowr/footstep_plan_extractor.h:69-133) for every problem of the batch,
 * up to the nearest-plane lookup (twr_batch_contact_planes below): the solution x sampled every dt
 * (GetTrajectory, :19-53), a footstep state at the first sample and wherever HasEndEffectorContactChanged (:55-67)
 * against the previous sample; duration = time to the next footstep state, the last one lasts until time_horizon.
 * Problem p's records start at d_out + p * max_steps * (2 + 4 n_ee):
 *   [ t_global | duration | contact flag per ee | ee-motion position (3) per ee ]
 * and d_counts[p] is the number of footstep states FOUND -- never more than twr_structure_contact_steps_max, but it
 * may exceed a smaller max_steps the caller chose: then only the first min(d_counts[p], max_steps) records are
 * valid (each with its duration; the last of them lasts until the first state that was dropped).  Asynchronous on
 * hip_stream. */
int twr_structure_contact_steps_max(const twr_structure* s, int32_t* max_steps);
int twr_batch_contact_plan(twr_batch* b, const double* d_x, double dt, double time_horizon, double* d_out, int32_t max_steps,
                           int32_t* d_counts, void* hip_stream);

/* fpowr::NearestPlaneLookup (fpowr/include/fpowr/nearest_plane_lookup.h:51-85), the last step of ExtractFootstepPlan
 * (footstep_plan_extractor.h:72-73,111-118): the planar regions of the goal's terrain message as polygons in world x, y
 * (PlanarRegionsToPolygons, :20-49: boundary point (x, y, 0) rotated by the region's orientation, shifted by its
 * position) and, for every footstep state of twr_batch_contact_plan and every end-effector in contact, the index of the
 * first polygon with the smallest boost::geometry::distance to the foot's x, y (0 inside or on the boundary, else the
 * distance to the nearest boundary segment; the boundary points are walked as given, no closing edge is added -- what
 * boost does with the reference's uncorrected polygons); -1 for a foot in the air, for footstep states past d_counts[p]
 * and when there are no regions.
 *   regions:        n_regions x 7 doubles  [position x y z | orientation x y z w]   (plane_parameters)
 *   boundary_xy:    the outer_boundary points of all regions, x y each, region r = points [boundary_start[r],
 *                   boundary_start[r+1])
 *   d_plane_index:  n_problems x max_steps x n_ee int32, written by twr_batch_contact_planes from the d_out / d_counts
 *                   of twr_batch_contact_plan (same max_steps).  Asynchronous on hip_stream. */
typedef struct twr_planes twr_planes;
int twr_planes_create(const double* regions, const double* boundary_xy, const int32_t* boundary_start, int32_t n_regions,
                      int device, twr_planes** out);
void twr_planes_destroy(twr_planes* planes);
/* the polygons in world coordinates (x y per boundary point, in the order given), for inspection */
int twr_planes_world_xy(const twr_planes* planes, double* world_xy);
int twr_batch_contact_planes(twr_batch* b, const twr_planes* planes, const double* d_plan, const int32_t* d_counts,
                             int32_t max_steps, int32_t* d_plane_index, void* hip_stream);

/* Convenience for single-problem / adapter use: host buffers, synchronous (H2D, eval, D2H).  Runs on a NON-BLOCKING
 * stream the batch owns (created on first use), not on the NULL stream: it neither waits for nor holds up work the host
 * application has in flight on the NULL stream or on its own blocking streams; the call returns when its own chain is
 * done.  flags select what is evaluated and copied back (TWR_EVAL_VALUES alone moves no Jacobian over PCIe: the ifopt
 * adapter evaluates values for eval_g and adds the Jacobian only when eval_jac_g asks for it). */
int twr_batch_eval_host(twr_batch* b, const double* h_x, double* h_g, double* h_jac, int flags);
/* Page-locked host buffers owned by the batch (x, g, jac of the whole batch layout), allocated on first
 * use.  Passing exactly these pointers to twr_batch_eval_host makes the transfers DMA directly from / into
 * them (no staging through pageable memory): what an Ipopt callback should read g and the Jacobian
 * values from (the ifopt adapter does).  For batches of up to 32 MB of Jacobian values the kernels then store g and
 * the Jacobian values straight into these host buffers over PCIe (each value once, coalesced) instead of into HBM
 * followed by two copies: and gather x from them: 34 instead of 53 us for one quadruped problem (TWR_HOST_ZERO_COPY=0 switches it off). */
int twr_batch_host_buffers(twr_batch* b, double** h_x, double** h_g, double** h_jac);

/* Tuning knobs.  The DEFAULT build reads nothing from the environment: the values below are compiled in.  A build with
 * -DTWR_TUNING_KNOBS (make -C towr_amd/csrc TUNING=1) reads them once per process, for A/B measurements (scripts/ab.py,
 * DESIGN.md section 6); they never change results, only how the work is spread over the device:
 *   TWR_DYN_BPC, TWR_ROM_BPC        persistent workgroups per CU of dyn_kernel / rom_kernel (8 / 4: their LDS images fill a CU)
 *   TWR_PDYN_BPC, TWR_PROM_BPC      the same for dyn_phase_kernel / rom_phase_kernel (default: what their LDS images allow,
 *                                   at most 4 / 8)
 *   TWR_FUSED_MAX_ROM               rom slices up to which one fused launch replaces the three kernels (default: 20 rounds
 *                                   of the rom residency = 20 x TWR_ROM_BPC x number of CUs)
 *   TWR_FUSED_SPLIT                 eighths of the residency the fused launch gives the rom role when both roles do not fit
 *                                   (default: 5 up to 34/8 rounds of rom slices, 4 above; 8 = one role after the other)
 *   TWR_FUSED_GROM, TWR_FUSED_GDYN  explicit block counts of the two roles of the fused launch (experiments)
 *   TWR_STREAM_NT=0|1               overrides the store policy twr_batch_create picks (twr_batch_streaming_stores)
 *   TWR_HOST_ZERO_COPY[_X]=0        twr_batch_eval_host: copy through device buffers instead of letting the kernels store
 *                                   into (gather x from) the page-locked host buffers */

#ifdef __cplusplus
}
#endif
#endif /* TOWR_AMD_H_ */
