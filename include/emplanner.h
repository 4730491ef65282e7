/*
 * emplanner.h - C-ABI of the MI355X-native EM-Planner path-planning hot path.
 *
 * The reference (6Lackiu/EMplanner_Carla) is pure Python: its "plugin interface" for this
 * path is the module-level function surface of planner/path_planning.py and
 * planner/planning_utils.py.  Each entry point below is the batched, device-resident form of
 * one (or a chain) of those functions; the Python package `emplanner_carla_amd.planner` binds
 * them with ctypes and re-exposes the reference's names and signatures.
 * "ref:" comments cite the reference function an entry point replaces (paths relative to the
 * reference tree).
 *
 * Conventions
 *   - plain C, POD only; all floating point is IEEE-754 binary64, all indices int32.
 *   - every array is batch-major and padded: [B][max_x]... with a per-scene length array.
 *   - `where` tells whether the DATA pointers of the call are host (EMP_HOST: the library
 *     stages them through its own device buffers) or device (EMP_DEVICE: used in place; e.g.
 *     torch tensors' data_ptr()).  Parameter structs are always host memory.
 *   - functions return EMP_OK or a negative emp_error; emp_last_error() gives the text.
 *     Per-scene problems never fail a call: they are reported in `status[b]` (bit mask).
 *   - a context owns one device, its HIP stream (more streams once emp_set_pipeline is used) and its
 *     scratch buffers; calls on one context are serialised by the caller.  HIP is initialised lazily in emp_create(), so a
 *     forked planning process (ref: test_9.py:225-227 runs the planner in a child process)
 *     must create its context after the fork.
 */
#ifndef EMPLANNER_H
#define EMPLANNER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMP_ABI_VERSION 11

typedef struct emp_ctx emp_ctx;

typedef enum emp_error {
    EMP_OK = 0,
    EMP_ERR_INVALID = -1,   /* bad argument (null pointer, size out of supported range) */
    EMP_ERR_HIP = -2,       /* a HIP runtime call failed; see emp_last_error */
    EMP_ERR_NO_DEVICE = -3, /* no gfx950 device visible */
    EMP_ERR_NOMEM = -4
} emp_error;

/* Where the arrays of a call live.  EMP_HOST: ordinary host memory - the call copies in, computes, copies out and returns when
 * the outputs are there (arrays of up to 64 KB, 256 KB a direction and call, are packed into one page-locked block and cross PCIe
 * as ONE copy per direction: a call of the reference's function surface - three to twenty small arrays - costs 50 us instead of
 * 65-100).  EMP_DEVICE: device memory, used in place; the call returns at once (stream-ordered).
 * EMP_HOST_PINNED (ABI 10): page-locked host memory from emp_host_alloc.  Every entry point accepts it like EMP_HOST;
 * emp_plan_cycle additionally overlaps it (see there): inputs cross PCIe on a copy stream while the previous call computes,
 * outputs come back on a stream of their own, and with a pipeline set the call does not wait for them - emp_wait_cycle does.
 * Layout contract of an EMP_HOST_PINNED cycle: the arrays may be any page-locked memory.  Arrays of one direction that lie inside
 * ONE emp_host_alloc allocation of this context with at most 512 bytes between neighbours (alignment padding) cross PCIe as one
 * copy; for outputs that copy also writes the padding bytes between them (contents unspecified) - nothing else is ever written.
 * Any other layout (arrays allocated one by one, a smaller batch at the head of a larger slot, other data carved between two
 * outputs) is copied array by array.  Within one pipeline do not alternate EMP_DEVICE and EMP_HOST_PINNED cycles faster than the
 * pipeline depth unless the device-pointer call's inputs may be read late (the library orders the copies, not the caller's writes). */
typedef enum emp_mem { EMP_HOST = 0, EMP_DEVICE = 1, EMP_HOST_PINNED = 2 } emp_mem;

/* per-scene status bits */
enum {
    EMP_ST_DP_INFEASIBLE = 1,  /* min terminal cost > w_collision: ref prints its banner, path_planning.py:351 */
    EMP_ST_S_OUT_OF_RANGE = 2, /* cal_proj_point walked past the s_map: ref raises IndexError, path_planning.py:63 */
    EMP_ST_BOUND_INDEX = 4,    /* cal_lmin_lmax index >= n: ref raises IndexError, path_planning.py:267/272 */
    EMP_ST_QP_FAILED = 8,      /* path QP infeasible / not converged (ref ignores cvxopt's status, :211-218) */
    EMP_ST_SMOOTH_FAILED = 16, /* smoothing QP not converged */
    EMP_ST_TRUNCATED = 32      /* an output did not fit the caller's max_* capacity */
};

/* ref: keyword arguments of DP_algorithm, path_planning.py:276-279 (defaults in emp_dp_params_default) */
typedef struct emp_dp_params {
    int32_t row;            /* lateral samples  (ref default 12) */
    int32_t col;            /* longitudinal stations (ref default 6) */
    double sample_s;        /* 15  */
    double sample_l;        /* 1.5 */
    double sampling_res;    /* 2: densification step of enrich_DP_s_l */
    double w_collision;     /* 1e12 */
    double w_smooth[3];     /* 300, 1000, 5000 */
    double w_ref;           /* 20 */
} emp_dp_params;

/* ref: keyword arguments of Quadratic_planning, path_planning.py:78-81, and of cal_lmin_lmax :222 */
typedef struct emp_qp_params {
    double ds;              /* dp_sampling_res = 2 (the QP's fixed station spacing, :108) */
    double w_l, w_dl, w_ddl, w_dddl, w_centre;          /* 1000, 10000 (unused by the ref, :193), 3000, 150, 250 */
    double w_end_l, w_end_dl, w_end_ddl;                /* 40, 40, 40 */
    double host_d1, host_d2, host_w;                    /* 3, 3, 3 */
    double obs_length, obs_width;                       /* 5, 5 (ref: test_9.py:192) */
    int32_t decimate;       /* 2 (test_9.py:187) or 1 (test_7.py) */
    int32_t midpoint;       /* 1: re-interleave midpoints (test_9.py:204-210); 0: none (test_7.py) */
    int32_t use_qp;         /* 1; 0 skips the QP (test_5.py / test_6.py form) */
    int32_t reserved;       /* MUST be 0 (emp_qp_params_default sets it): every entry point that takes the struct refuses
                             * anything else with EMP_ERR_INVALID, so that a caller who did not initialise the struct finds
                             * out at once */
} emp_qp_params;

/* ref: keyword arguments of smooth_reference_line, planning_utils.py:262-264 */
typedef struct emp_smooth_params {
    double w_smooth, w_length, w_ref;                   /* 0.4, 0.3, 0.3 */
    double x_thre, y_thre;                              /* 0.2, 0.2 */
} emp_smooth_params;

/* ref: keyword arguments of speed_DP, speed_planning_test.py:101-102 */
typedef struct emp_speed_dp_params {
    double reference_speed;                             /* 50 */
    double w_cost_ref_speed, w_cost_accel, w_cost_obs;  /* 4000, 100, 1e7 */
} emp_speed_dp_params;

void emp_dp_params_default(emp_dp_params* p);
void emp_speed_dp_params_default(emp_speed_dp_params* p);
void emp_qp_params_default(emp_qp_params* p);
void emp_smooth_params_default(emp_smooth_params* p);

/* ---- context ------------------------------------------------------------------------- */
int emp_abi_version(void);
int emp_create(int device_id, emp_ctx** out);
void emp_destroy(emp_ctx* ctx);
const char* emp_last_error(const emp_ctx* ctx); /* ctx may be NULL: error of the last failed emp_create */
int emp_synchronize(emp_ctx* ctx);
/* the context's HIP stream (hipStream_t) so callers can order their own work after ours */
void* emp_stream(emp_ctx* ctx);
/* device memory helpers for callers without a HIP binding (ctypes hosts) */
/* Page-locked host memory for EMP_HOST_PINNED calls (hipHostMalloc); freed by emp_host_free or emp_destroy. */
int emp_host_alloc(emp_ctx* ctx, uint64_t bytes, void** out);
int emp_host_free(emp_ctx* ctx, void* ptr);
/* EMP_HOST_PINNED cycles: block until the outputs of the emp_plan_cycle call issued `calls_back` calls ago (0 = the latest)
 * are in the caller's host arrays.  calls_back < emp_pipeline_depth(); EMP_ERR_INVALID beyond (that call's arrays were already
 * waited for when its pool was taken over).  A call that was not an EMP_HOST_PINNED cycle has nothing to wait for: EMP_OK. */
int emp_wait_cycle(emp_ctx* ctx, int32_t calls_back);
/* The number of pipelined emp_plan_cycle calls issued on this context so far: the ticket of the latest one.  A caller that keeps
 * the ticket of a call finds it again as calls_back = emp_cycle_ticket() - ticket. */
uint64_t emp_cycle_ticket(emp_ctx* ctx);
/* Block until the host outputs of the EMP_HOST_PINNED cycle that got `ticket` are in place; a ticket older than the pipeline
 * depth has been waited for by the call that took its pool over: EMP_OK at once.  The ONE entry point that may be called from
 * another thread while a call on this context is in progress (a server thread waits for its batch while another submits): a pool
 * keeps answering for its ticket until the call that takes it over has itself waited for that ticket's outputs.  Not beside
 * emp_set_pipeline or emp_destroy. */
int emp_wait_ticket(emp_ctx* ctx, uint64_t ticket);
int emp_device_alloc(emp_ctx* ctx, uint64_t bytes, void** out);
int emp_device_free(emp_ctx* ctx, void* ptr);
int emp_copy_to_device(emp_ctx* ctx, void* dst, const void* src, uint64_t bytes);
int emp_copy_to_host(emp_ctx* ctx, void* dst, const void* src, uint64_t bytes);
/* Per-kernel timing with HIP events on the context's stream.  emp_set_timing(ctx, 1) (re)starts the
 * statistics; every launch of a named kernel ("project", "dp_edge", "dp_sweep", "dp_enrich", "path_qp",
 * "to_cartesian", "heading", ...) is then bracketed by an event pair.  emp_kernel_ms returns the MEAN
 * duration in milliseconds over the launches recorded since (it synchronises on their events), or a
 * negative value if there were none; emp_kernel_launches returns how many were recorded. */
int emp_set_timing(emp_ctx* ctx, int enabled);
/* Restrict the event pairs to ONE named kernel (NULL or "" = every kernel again).  An event pair costs a few
 * microseconds of stream time per launch, so a benchmark that needs the live duration of one kernel should not
 * pay for bracketing the others. */
int emp_set_timing_filter(emp_ctx* ctx, const char* kernel);

/* Several batches in flight: CONSECUTIVE emp_plan_cycle calls with device pointers overlap on the GPU (off by default).
 * The kernels of a cycle are bound by different things - FP64 issue (edge costs), HBM (sweep), the latency of the slowest
 * scene's chain of dependent instructions (projection, path QP, Cartesian tail, the chip mostly idle) - so one batch at a
 * time leaves issue slots empty.  mode:
 *   0                      off: one cycle at a time on emp_stream().
 *   EMP_PIPELINE_STAGED    two batches: the back stage (densified DP path, path QP, Cartesian tail) of call k runs on a
 *                          second stream while the front stage (projection, edge costs, sweep) of call k+1 runs on
 *                          emp_stream().  The front stages stay serial, so the sweep runs next to nothing but the end
 *                          of a back stage and keeps its share of the HBM roofline (0.26 ms per 4096-scene step, sweep
 *                          21 us).  A call may wait ON THE HOST for the call four back (emp_pipeline_depth).
 *   n = 2..EMP_PIPELINE_MAX  n batches on n lanes (a stream and a pool of temporaries each; ABI version 7): call k runs
 *                          whole on lane k mod n, behind everything queued on emp_stream() when it is issued, and the
 *                          dispatcher overlaps the kernels of n consecutive cycles.  Highest throughput (n = 3: 0.258 ms
 *                          per step) at the price of every kernel's own duration (the sweep: 40 us).  Wants as many
 *                          hardware queues as streams (n lanes + emp_stream()): the HIP runtime maps a process's streams
 *                          onto GPU_MAX_HW_QUEUES queues (default 4) and reads the variable when it initialises, so the
 *                          HOST PROGRAM exports e.g. GPU_MAX_HW_QUEUES=8 before it first touches HIP - the library does
 *                          not change the environment.  With fewer queues lanes share one and serialise (3 lanes on 4
 *                          queues beside torch's stream: slower than 2); beyond 7 lanes they share in any case.
 *   EMP_PIPELINE_AUTO      (ABI version 11) the library picks what this PROCESS can sustain: three lanes when every stream has a
 *                          hardware queue of its own - GPU_MAX_HW_QUEUES (the value the HIP runtime was started with; default
 *                          4; the one environment variable the library reads) >= 3 lanes + emp_stream() + this context's copy /
 *                          d2h streams (EMP_HOST_PINNED cycles) + EMP_OPT_FOREIGN_STREAMS (the streams the rest of the process
 *                          uses: default 1, the caller's own; add a gather stream, RCCL's) - else EMP_PIPELINE_STAGED (two
 *                          queues).  emp_pipeline_form reports the choice.  Measured on 4 queues (HIP's default): three lanes
 *                          0.272-0.274 ms per 4096-scene step against the staged form's 0.202-0.207; on 12: 0.187-0.191 against the same
 *                          (profiles/r06_bench_queues*.json; tests/test_gpu_fullsize.py).
 * Consequences for the caller: the outputs of a call are complete on emp_result_stream() - in lane mode the lane of the
 * LATEST emp_plan_cycle call, so ask after every call - and emp_synchronize waits for every stream; each call in flight
 * needs its OWN output buffers, and its inputs must stay unchanged until it is done.  Every other entry point first
 * lets emp_stream() wait for the cycles in flight.  Results are bit-identical to the unpipelined call. */
#define EMP_PIPELINE_AUTO (-1)
#define EMP_PIPELINE_STAGED 1
#define EMP_PIPELINE_MAX 8
int emp_set_pipeline(emp_ctx* ctx, int mode);
/* The pipeline form in force: 0 off, EMP_PIPELINE_STAGED, or the number of lanes; negative: ctx is NULL.  After an
 * emp_set_pipeline(EMP_PIPELINE_AUTO), optionally (pointers may be NULL) the hardware-queue count it saw and the streams it counted
 * beside the lanes (0 / 0 when the latest mode was set explicitly). */
int emp_pipeline_form(emp_ctx* ctx, int32_t* hw_queues, int32_t* other_streams);
void* emp_result_stream(emp_ctx* ctx);
/* How many consecutive emp_plan_cycle calls may have work in flight in the current mode (ABI version 8): the output buffers
 * of call k may be reused once call k + emp_pipeline_depth() has been ISSUED.  1 when the pipeline is off, n in lane mode,
 * 4 in staged mode: two batches overlap there, but the pools of temporaries rotate over four calls and emp_plan_cycle
 * waits ON THE HOST for the call four back (a call that finished long ago unless the host runs further ahead than that,
 * which it then may not) - a stream-side wait for the pool's previous user cost ~11 us of every 0.29 ms step. */
int emp_pipeline_depth(emp_ctx* ctx);
/* The fence of the pipelined modes (on by default): every entry point other than a pipelined emp_plan_cycle first lets
 * emp_stream() wait for the cycles in flight, so that it may read their outputs.  Switched off, such calls are queued on
 * emp_stream() at once and overlap the cycles in flight - for work that does not depend on them (the S-T speed planner of
 * the same scenes: bench.py --config cfg5); what must be ordered, the caller orders with emp_result_stream().  Their
 * temporaries are the main stream's own, so consecutive unfenced calls are still serial among themselves. */
int emp_set_fence(emp_ctx* ctx, int enabled);

/* ---- options (ABI version 9; renumbered in version 11: twelve of them) -------------------------------------------------------
 * The library reads no EMP_* environment variable.  Everything that used to be an environment switch of the development
 * builds is a per-context option here, set by the host program before the calls it should affect (a change takes effect
 * at the next call; EMP_OPT_FOREIGN_STREAMS at the next emp_set_pipeline(EMP_PIPELINE_AUTO)).  Retired in version 11, each with the
 * number that retired it: EMP_OPT_SWEEP_MARKER (always on with EMP_OPT_SWEEP_EXCLUSIVE: 0.264 against 0.32-0.35 ms per step without),
 * EMP_OPT_ENRICH_ON_FRONT (0.4 % slower, never a default), EMP_OPT_BACK_STREAM_CUS (a CU mask on the back stage never gained),
 * EMP_OPT_FUSED_COLUMNS / EMP_OPT_EDGE_COLS_PER_WAVE / EMP_OPT_SWEEP_VARIANT (the auto rules are the measured optima: HISTORY.md 3.2,
 * profiles/r05_edge/README.md), EMP_OPT_ST_ORDER (heaviest scenes first: 3 % on configs[4], never slower).  Unknown options and values out of range are
 * EMP_ERR_INVALID.  "result-affecting" options change WHICH kernel computes a stage: the two forms of a stage solve the
 * same problem with the same stopping rule but associate sums differently (path QP: ~2e-9 relative; Cartesian tail: bit
 * identical) - so they are never chosen from the batch size: a scene's result does not depend on how many scenes share
 * its call or its GPU (a rank's shard equals its slice of the one-GPU result, emplanner_carla_amd/dist.py).
 *
 *   option                          default  kind             meaning
 *   EMP_OPT_PATH_QP_FORM            0        result-affecting 0: eight scenes per wavefront (emp_qp_rows.h; up to 66 stations,
 *                                                             beyond that the one-per-wavefront kernel whatever is set);
 *                                                             1: the two-scenes-per-wavefront kernel of rounds 1-2
 *                                                             (emp_qp_wave.h: ~17 % fewer instructions on a scene's own
 *                                                             critical path - for a caller that plans a handful of scenes
 *                                                             and wants the last 10 us of latency)
 *   EMP_OPT_CARTESIAN_FORM          0        A/B (bit-ident.) 0: four scenes per wavefront; 1: one scene per wavefront
 *   EMP_OPT_SMOOTH_FORCE_FALLBACK   0        test hook        1: every smoothing QP of the Cartesian tail takes its
 *                                                             projected-gradient fallback (tests/test_gpu_fullsize.py)
 *   EMP_OPT_EDGE_BLOCK              0        tuning           threads per block of the edge-cost kernel (multiple of 64 up
 *                                                             to 1024); 0: from the lattice's LDS footprint
 *   EMP_OPT_EDGE_FORM               0        A/B (bit-ident.) edge-cost kernel of the tiled lattices (<= 32 rows): 0 = work-ring
 *                                                             form (edges with obstacles in reach are queued per wavefront and
 *                                                             scanned one edge per lane: 0.93 active lanes of a scan at 40 x 9
 *                                                             with 8 obstacles), 1 = the lockstep form of rounds 1-4 (0.49);
 *                                                             obstacle rows wider than 64 slots always take the lockstep form
 *   EMP_OPT_LANE_EDGE_ORDER         2        pipeline order   lane mode (emp_set_pipeline(n >= 2)): 1 = the edge-cost kernel of a call starts
 *                                                             when the previous call's - on another lane - is done, so that at most
 *                                                             one of them runs at a time and the sweep that follows one has a single
 *                                                             edge kernel beside it instead of two (one stream-side wait per call);
 *                                                             2 (default since ABI 11) = 1 for calls of 8192 scenes and more, where
 *                                                             one edge kernel fills the chip by itself (32 768 scenes of the 40 x 9
 *                                                             lattice: 1.47 -> 1.36 ms per step, the sweep at 0.53 of the HBM peak
 *                                                             instead of 0.31; 8192: 2 %); 0 = lanes are never ordered among each
 *                                                             other.  NOT for smaller calls: the order bounds the step from below by
 *                                                             the edge kernel's own duration (4096 scenes with every obstacle beside
 *                                                             the same columns: 0.18 -> 0.23 ms per step; 1024 scenes 0.116 -> 0.121)
 *   EMP_OPT_CYCLE_GRAPH             0        tuning           1: emp_plan_cycle on device pointers, one batch at a time (no pipeline): the
 *                                                             THIRD consecutive call with the same sizes, parameters, options and
 *                                                             pointers captures its six launches into a hipGraph, every further one
 *                                                             replays it with one hipGraphLaunch (the callers that plan one scene or a
 *                                                             few per call over and over, BASELINE configs[1]: launch latency is a
 *                                                             fifth of such a cycle).  Same kernels, same arguments: same bits.  Any
 *                                                             change of an argument, of an option or of the context's buffers drops
 *                                                             the graph; emp_set_timing and the clock probes bypass it
 *   EMP_OPT_EDGE_CLOCK_PROBE        0        measurement      1: every launch of the work-ring edge kernel records, per wavefront,
 *                                                             the 100 MHz reference counter at its first and last instruction
 *                                                             and the shader clock there (emp_edge_probe, emp_edge_clock_mhz read
 *                                                             the latest launch)
 *   EMP_OPT_SWEEP_EXCLUSIVE         0        tuning           staged pipeline, what the HBM-bound sweep of call k may run beside:
 *                                                             0 (default since round 5) = no wait: the fastest step.  With the
 *                                                             work-ring edge kernel the sweep of call k starts after call k-1's
 *                                                             path QP by itself and overlaps its Cartesian tail: 0.66-0.67 of
 *                                                             the HBM peak at 4096 scenes, 0.227 ms per step;
 *                                                             2 = it waits (stream-side) for call k-1's densification and
 *                                                             path QP: 0.70-0.72 at +6 % step time (0.241 ms) - the two
 *                                                             barrier packets, not any waiting, are what the 6 % buy;
 *                                                             1 = it waits for the whole back stage of call k-1 (0.259 ms)
 *   EMP_OPT_EDGE_AFTER_ENRICH       1        tuning           staged pipeline: 1 (default) = the edge-cost kernel of call k waits
 *                                                             (stream-side) for the densification kernel of call k-1, so that
 *                                                             the path QP behind it is dispatched BEFORE the edge kernel's
 *                                                             sixteen-wavefront blocks take the compute units.  Whichever of
 *                                                             the two starts first keeps the chip: QP first = edge 195 us, QP
 *                                                             140 us side by side; edge first = edge 147 us with the QP starved
 *                                                             to 255 us behind it, a 0.32-0.34 ms step.  Without the wait the
 *                                                             order is a race that the N > 1 step (record packing on the back
 *                                                             queue) loses: 0.305 -> 0.270 ms there, +1 % on the plain step
 *   EMP_OPT_FOREIGN_STREAMS         1        pipeline order   streams of the process OUTSIDE this context that carry GPU work while it
 *                                                             plans (the caller's own stream, a gather stream, RCCL's): what
 *                                                             emp_set_pipeline(EMP_PIPELINE_AUTO) counts beside its own
 *   EMP_OPT_SWEEP_CLOCK_PROBE       0        measurement      1: every sweep launch also records, per wavefront, the shader
 *                                                             clock ticks and the 100 MHz reference ticks it ran for
 *                                                             (emp_sweep_clock_mhz reads their ratio)                       */
typedef enum emp_option {
    EMP_OPT_PATH_QP_FORM = 0,
    EMP_OPT_CARTESIAN_FORM = 1,
    EMP_OPT_SMOOTH_FORCE_FALLBACK = 2,
    EMP_OPT_EDGE_BLOCK = 3,
    EMP_OPT_EDGE_FORM = 4,
    EMP_OPT_SWEEP_EXCLUSIVE = 5,
    EMP_OPT_EDGE_AFTER_ENRICH = 6,
    EMP_OPT_LANE_EDGE_ORDER = 7,
    EMP_OPT_CYCLE_GRAPH = 8,
    EMP_OPT_SWEEP_CLOCK_PROBE = 9,
    EMP_OPT_EDGE_CLOCK_PROBE = 10,
    EMP_OPT_FOREIGN_STREAMS = 11,
    EMP_OPT_COUNT = 12
} emp_option;
int emp_set_option(emp_ctx* ctx, int32_t option, int32_t value);
int emp_get_option(emp_ctx* ctx, int32_t option, int32_t* value);
/* With EMP_OPT_SWEEP_CLOCK_PROBE on: the shader clock the sweep launches ran at since the option was last switched on (the
 * latest 32 of them), in MHz: the ratio of shader-clock ticks to 100 MHz reference ticks between a wavefront's first and
 * last instruction, summed over all their wavefronts (synchronises on the latest launch); optionally (may be NULL) the mean
 * and the longest time a wavefront was resident, in microseconds.  Negative when nothing was recorded. */
double emp_sweep_clock_mhz(emp_ctx* ctx, double* mean_wave_us, double* max_wave_us);
/* With EMP_OPT_EDGE_CLOCK_PROBE on: the latest edge-cost launch - mean time a wavefront was resident (us), time from the first
 * wavefront's start to the last one's end (us), mean number of wavefronts resident at once (their ratio x count), wavefronts.
 * Synchronises on that launch.  Any out pointer may be NULL. */
int emp_edge_probe(emp_ctx* ctx, double* mean_wave_us, double* span_us, double* mean_resident_waves, int32_t* waves);
/* The same probe: the shader clock the latest edge-cost launch ran at, in MHz (shader-clock ticks over 100 MHz reference ticks,
 * summed over its wavefronts) - the clock the chip holds under the FP64-issue-bound kernel that is most of the step, which is what
 * a vector-issue roofline of the step must be priced at (bench.py roofline_step).  Negative when nothing was recorded. */
double emp_edge_clock_mhz(emp_ctx* ctx);
/* EMP_OPT_CYCLE_GRAPH: how many emp_plan_cycle calls of this context were served by replaying a captured graph (-1: ctx is NULL). */
int64_t emp_cycle_graph_replays(emp_ctx* ctx);
/* The same probe, per launch and averaged over the recorded launches (100 MHz reference ticks, which all wavefronts share):
 * how long after the launch's first wavefront its last wavefront started, and the time from the first wavefront's first
 * instruction to the last wavefront's last - what is left of the launch's event-measured duration is dispatch and
 * completion overhead outside any wavefront. */
int emp_sweep_probe_spans(emp_ctx* ctx, double* start_spread_us, double* first_start_to_last_end_us);

/* One fixed-stride record per scene for the multi-GPU gather (no reference counterpart: the reference plans one scene
 * per process; this is the result exchange of the batched form, emplanner_carla_amd/dist.py):
 *   rec [B][3 + col + 2*path_cap + 4*(path_cap+1)] doubles = status, traj_len, path_len, dp_rows [col],
 *   path_s [path_cap], path_l [path_cap], traj [path_cap+1][4]
 * from the outputs of emp_plan_cycle (arrays with max_pts / max_pts+1 entries per scene, path_cap <= max_pts of them
 * kept).  One launch; on_result_stream != 0 queues it on emp_result_stream(), behind the cycle that produced them. */
int emp_pack_records(emp_ctx* ctx, int32_t B, int32_t col, int32_t max_pts, int32_t path_cap, const int32_t* status,
                     const int32_t* traj_len, const int32_t* path_len, const double* dp_rows, const double* path_s,
                     const double* path_l, const double* traj, double* rec, int on_result_stream, emp_mem where);
/* The same for a consumer that only drives the controller (ref: controller/controller.py:66-71 takes the list of
 * (x, y, theta, kappa) and nothing else): rec [B][2 + 4*(path_cap+1)] doubles = status, traj_len, traj [path_cap+1][4]
 * - 94 instead of 179 doubles per scene on the 40x9 lattice (ABI version 6). */
int emp_pack_trajectory_records(emp_ctx* ctx, int32_t B, int32_t max_pts, int32_t path_cap, const int32_t* status,
                                const int32_t* traj_len, const double* traj, double* rec, int on_result_stream,
                                emp_mem where);
double emp_kernel_ms(emp_ctx* ctx, const char* kernel);
int emp_kernel_launches(emp_ctx* ctx, const char* kernel);
/* The individual durations behind emp_kernel_ms, in launch order: writes min(recorded, cap) values (milliseconds) to `ms`
 * and returns how many were recorded (negative on error). */
int emp_kernel_samples(emp_ctx* ctx, const char* kernel, double* ms, int32_t cap);

/* ---- S-L lattice DP ------------------------------------------------------------------ */
/* Layout of the materialised edge-cost tensor (two-kernel DP mode):
 *   EMP_EDGE_CANONICAL  edge[b][j-1][i][k]  k fastest: cost of row k (column j-1) -> row i (column j)
 *   EMP_EDGE_TILED      the layout the sweep kernel streams: scenes are grouped in tiles of
 *                       S = 64 / row scenes and stored [tile][j-1][k][s][i] so that one wavefront
 *                       reads 64 consecutive doubles per source row k
 * `row` may be anything in [1, 1024] (ref path_planning.py:276-279 takes any; :301-346).  Up to 32 rows the DP runs on the
 * tiled kernels; wider lattices take a generic pair of kernels (one block per scene, pair table in device memory) whose
 * tensor is the CANONICAL one whichever layout is asked for (emp_edge_tensor_elems says so), and EMP_DP_FUSED falls back
 * to the two-kernel form there.  Same arithmetic, bit for bit.  Measured (profiles/r04_wide_lattice.json, 1024 scenes, 40
 * columns): the wide edge kernel costs 0.014-0.021 ns per lattice edge against 0.013 on the 21-row tiled lattice, the wide
 * sweep 0.003-0.007 against 0.0016 (2.0-2.6 TB/s of algorithmic bytes; 0.6 TB/s at exactly 33 rows) - a 48-row lattice is
 * 1.3x the per-edge cost of the 21-row one, so the wide pair stays generic.
 * THE CAP: row > 1024 is refused with EMP_ERR_INVALID (ABI 10; 256 until ABI 9: predecessor indices of the wide sweep are
 * 16-bit now).  The reference itself takes any row count; at 1024 rows the pair table alone is 126 MB and one scene of 6
 * columns 5.2 M edges - what bounds the cap is memory per scene, not the kernels.                                        */
typedef enum emp_edge_layout { EMP_EDGE_CANONICAL = 0, EMP_EDGE_TILED = 1 } emp_edge_layout;
uint64_t emp_edge_tensor_elems(const emp_dp_params* p, int32_t B, emp_edge_layout layout);

/* ref: cal_start_cost (path_planning.py:435-514) for every row and cal_neighbor_cost (:517-585)
 * for every (column, row, row) of every scene.
 *   obs_s, obs_l [B][max_obs], n_obs [B], start [B][4] = plan_start s, l, dl, ddl
 *   start_cost [B][row]   (without the +10000 lane penalty, like the ref's function)
 *   edge       emp_edge_tensor_elems(p, B, layout) doubles                                  */
int emp_dp_edge_costs(emp_ctx* ctx, const emp_dp_params* p, int32_t B, int32_t max_obs,
                      const double* obs_s, const double* obs_l, const int32_t* n_obs, const double* start,
                      double* start_cost, double* edge, emp_edge_layout layout, emp_mem where);

typedef enum emp_dp_mode {
    EMP_DP_FUSED = 0,      /* one kernel, one block per tile of 64 / row scenes: the edge costs of four columns at a time are
                            * staged in LDS and swept in place; no HBM edge tensor (8 E bytes per scene neither written nor
                            * read).  Bit-identical results; slower than the two-kernel form at every measured batch size
                            * (0.264 against 0.181 ms per 4096 scenes 40x9, 1.42 against 1.35 ms per 32768: one block per
                            * tile leaves a CU with 8-12 wavefronts) - the form for callers who cannot afford the tensor. */
    EMP_DP_TWO_KERNEL = 1  /* edge-cost kernel writes the tiled tensor to HBM, sweep kernel streams it (the measured
                            * default of the Python layer and of bench.py) */
} emp_dp_mode;

/* ref: DP_algorithm (path_planning.py:276-375) up to and including the backtrack, batched.
 *   rows   [B][col] float64: chosen lattice row per column (float because the ref's no-obstacle
 *          bypass yields (row+1)/2-1, which is x.5 for even `row`, :363)
 *   min_cost [B] (may be NULL): minimum of the last cost column (+inf in the bypass)
 *   status [B]: EMP_ST_DP_INFEASIBLE or 0                                                    */
int emp_dp_plan(emp_ctx* ctx, const emp_dp_params* p, int32_t B, int32_t max_obs,
                const double* obs_s, const double* obs_l, const int32_t* n_obs, const double* start,
                emp_dp_mode mode, double* rows, double* min_cost, int32_t* status, emp_mem where);

/* min-plus sweep + backtrack on caller-provided costs (ref: path_planning.py:301-361), for tests
 * and for the HBM-roofline measurement.  start_cost [B][row], edge in EMP_EDGE_TILED layout (canonical beyond 32 rows). */
int emp_dp_sweep(emp_ctx* ctx, const emp_dp_params* p, int32_t B, const double* start_cost, const double* edge,
                 double* rows, double* min_cost, int32_t* status, emp_mem where);

/* ref: enrich_DP_s_l (path_planning.py:378-432) after the row->(s,l) mapping of :366-370.
 *   rows [B][col] -> path_s, path_l [B][max_pts], path_len [B]; status gets EMP_ST_TRUNCATED  */
int emp_dp_enrich(emp_ctx* ctx, const emp_dp_params* p, int32_t B, const double* rows, const double* start,
                  int32_t max_pts, double* path_s, double* path_l, int32_t* path_len, int32_t* status,
                  emp_mem where);

/* ref: enrich_DP_s_l (path_planning.py:378-432) exactly as the reference function takes it: arbitrary node
 * lists node_s, node_l [B][max_nodes] with n_nodes [B], start [B][4], resolution -> path_s, path_l, path_len */
int emp_enrich_nodes(emp_ctx* ctx, int32_t B, int32_t max_nodes, double resolution, const double* node_s,
                     const double* node_l, const int32_t* n_nodes, const double* start, int32_t max_pts,
                     double* path_s, double* path_l, int32_t* path_len, int32_t* status, emp_mem where);

/* ---- Cartesian <-> Frenet ------------------------------------------------------------ */
/* ref: cal_s_map_fun (planning_utils.py:448-472), cal_s_l_fun (:475-509) for the obstacles and the
 * planning start, cal_s_l_deri_fun (:512-588) for the planning start - the front half of one
 * motion_planning cycle (test_9.py:113-177).
 *   ref_line [B][max_ref][4] x,y,theta,kappa; n_ref [B]
 *   origin_xy, start_xy, start_v, start_a [B][2]; obs_xy [B][max_obs][2]; n_obs [B]
 *   s_map [B][max_ref]; obs_s, obs_l [B][max_obs]; begin_sl [B][2] = s,l of start_xy;
 *   start [B][4] = s, l, dl/ds, d2l/ds2 (the DP's plan_start_*)                               */
int emp_frenet_project(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_obs,
                       const double* ref_line, const int32_t* n_ref,
                       const double* origin_xy, const double* start_xy, const double* start_v,
                       const double* start_a, const double* obs_xy, const int32_t* n_obs,
                       double* s_map, double* obs_s, double* obs_l, double* begin_sl, double* start,
                       emp_mem where);

/* ref: cal_s_map_fun (planning_utils.py:448-472) alone: s_map [B][max_ref] */
int emp_s_map(emp_ctx* ctx, int32_t B, int32_t max_ref, const double* ref_line, const int32_t* n_ref,
              const double* origin_xy, double* s_map, emp_mem where);

/* ref: cal_s_l_fun (planning_utils.py:475-509) with a caller-supplied s_map: xy [B][max_pts][2] -> s, l [B][max_pts].
 * With match_index != NULL and l == NULL it is cal_projection_s_fun (:429-445): s from the given match indices. */
int emp_s_l(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_pts, const double* ref_line, const double* s_map,
            const int32_t* n_ref, const double* xy, const int32_t* n_pts, const int32_t* match_index,
            double* s, double* l, emp_mem where);

/* ref: cal_s_l_deri_fun (planning_utils.py:512-588): xy, v_xy, a_xy [B][max_pts][2], origin_xy [B][2]
 * -> out [B][max_pts][7] = l, dl/dt, ds/dt, d2l/dt2, dl/ds, d2s/dt2, d2l/ds2 */
int emp_s_l_deri(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_pts, const double* ref_line,
                 const int32_t* n_ref, const double* xy, const double* v_xy, const double* a_xy,
                 const int32_t* n_pts, const double* origin_xy, double* out, emp_mem where);

/* ref: cal_proj_point (path_planning.py:52-75; twin cal_proj_point_1, planning_utils.py:647-668): n independent
 * queries, each against its own line: s [n], pre_match_index [n] -> out [n][4] x,y,theta,kappa, index [n],
 * status [n] (EMP_ST_S_OUT_OF_RANGE where the reference raises IndexError) */
int emp_proj_point(emp_ctx* ctx, int32_t n, int32_t max_ref, const double* ref_line, const double* s_map,
                   const int32_t* n_ref, const double* s, const int32_t* pre_match_index, double* out,
                   int32_t* index, int32_t* status, emp_mem where);

/* ref: match_projection_points (planning_utils.py:364-426): [B][max_pts] points against one line per scene */
int emp_match_projection(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_pts,
                         const double* ref_line, const int32_t* n_ref, const double* xy, const int32_t* n_pts,
                         int32_t* match_index, double* proj, emp_mem where);

/* ref: find_match_points (planning_utils.py:49-182) */
int emp_find_match_points(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_pts,
                          const double* ref_line, const int32_t* n_ref, const double* xy, const int32_t* n_pts,
                          const int32_t* is_first_run, const int32_t* pre_match_index,
                          int32_t* match_index, double* proj, emp_mem where);

/* ref: cal_heading_kappa (planning_utils.py:185-228): xy [B][max_pts][2] -> theta, kappa [B][max_pts] */
int emp_heading_kappa(emp_ctx* ctx, int32_t B, int32_t max_pts, const double* xy, const int32_t* n_pts,
                      double* theta, double* kappa, emp_mem where);

/* ---- QP stages ----------------------------------------------------------------------- */
/* ref: cal_lmin_lmax (path_planning.py:222-273): station bounds from the (decimated) DP path */
int emp_lmin_lmax(emp_ctx* ctx, int32_t B, int32_t max_pts, int32_t max_obs,
                  const double* dp_s, const double* dp_l, const int32_t* n_pts,
                  const double* obs_s, const double* obs_l, const int32_t* n_obs,
                  double obs_length, double obs_width, double* l_min, double* l_max, int32_t* status,
                  emp_mem where);

/* ref: Quadratic_planning (path_planning.py:78-219).  l_min, l_max [B][max_pts], n_pts [B],
 * start_l3 [B][3] = plan_start l, dl, ddl -> qp_l, qp_dl, qp_ddl [B][max_pts]; status EMP_ST_QP_FAILED */
int emp_path_qp(emp_ctx* ctx, const emp_qp_params* q, int32_t B, int32_t max_pts,
                const double* l_min, const double* l_max, const int32_t* n_pts, const double* start_l3,
                double* qp_l, double* qp_dl, double* qp_ddl, int32_t* iters, int32_t* status, emp_mem where);

/* ref: smooth_reference_line (planning_utils.py:262-361): box-QP smoothing + heading/kappa.
 * xy [B][max_pts][2] -> out [B][max_pts][4] = x, y, theta, kappa */
int emp_smooth_line(emp_ctx* ctx, const emp_smooth_params* sp, int32_t B, int32_t max_pts,
                    const double* xy, const int32_t* n_pts, double* out, int32_t* iters, int32_t* status,
                    emp_mem where);

/* ref: cal_proj_point (path_planning.py:52-75) chained over a path as frenet_2_x_y_theta_kappa does
 * (:29-46), WITHOUT the smoothing: target_xy [B][max_pts+1][2] (first = the planning start), n_out [B] */
int emp_frenet_path_to_xy(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_pts,
                          const double* ref_line, const double* s_map, const int32_t* n_ref,
                          const double* begin_sl, const double* path_s, const double* path_l,
                          const int32_t* n_pts, double* target_xy, int32_t* n_out, int32_t* status,
                          emp_mem where);

/* ---- one whole planning cycle -------------------------------------------------------- */
/* ref: the body of motion_planning, test_9.py:113-218 (reference line already smoothed):
 * projection -> DP -> decimate -> bounds -> path QP -> midpoints -> Frenet->Cartesian -> smoothing
 * -> heading/kappa.  Inputs as emp_frenet_project.  Outputs (any may be NULL except traj/traj_len/status):
 *   dp_rows [B][col]; dp_s, dp_l [B][max_pts], dp_len [B]      (DP_algorithm's return)
 *   path_s, path_l [B][max_pts], path_len [B]                   (what test_9.py:220 sends back)
 *   traj [B][max_pts+1][4] x,y,theta,kappa, traj_len [B]        (the controller's input)
 *   status [B] bit mask                                                                       */
typedef struct emp_cycle_io {
    /* inputs */
    const double* ref_line; const int32_t* n_ref;
    const double* origin_xy; const double* start_xy; const double* start_v; const double* start_a;
    const double* obs_xy; const int32_t* n_obs;
    /* outputs */
    double* dp_rows; double* dp_s; double* dp_l; int32_t* dp_len;
    double* path_s; double* path_l; int32_t* path_len;
    double* traj; int32_t* traj_len;
    int32_t* status;
    /* optional input (ABI 2): [B][2] = distance and speed of the FIRST dynamic obstacle of each scene, NaN distance =
     * none.  ref test_9.py:137-169: it becomes three virtual static obstacles on the centre line (meet_s - 10, the
     * middle of the encounter, leave_s) unless the encounter ends beyond s = 80 m.  NULL: no dynamic obstacles. */
    const double* dyn_dis_speed;
    /* optional front end (ABI 11): the cycle starts from the GLOBAL path, as the reference's planning loop does (test_9.py:99-110:
     * find_match_points for the predicted location = start_xy, sampling, smooth_reference_line) - emp_reference_line and this
     * call as ONE call, the 51-point reference line never leaving the device.  global_path [B][max_global][4], n_global [B],
     * pre_match_index [B] in; match_index [B] (the next request's pre_match_index) and ref_status [B] (emp_reference_line's status:
     * OR it into `status`) out.  Then ref_line / n_ref are ignored and max_ref must be EMP_REF_LINE_POINTS.  A scene whose
     * reference line fails runs the cycle on an empty line, exactly as with n_ref = 2 handed to the two-call form.
     * global_path NULL: no front end (the other five fields are ignored). */
    const double* global_path; const int32_t* n_global; const int32_t* pre_match_index;
    int32_t* match_index; int32_t* ref_status;
    int32_t max_global; int32_t reserved_io;      /* reserved_io: 0 */
} emp_cycle_io;

int emp_plan_cycle(emp_ctx* ctx, const emp_dp_params* p, const emp_qp_params* q, const emp_smooth_params* sp,
                   int32_t B, int32_t max_ref, int32_t max_obs, int32_t max_pts, emp_dp_mode mode,
                   const emp_cycle_io* io, emp_mem where);

/* ---- scalar utilities of the reference that sit beside the path ----------------------- */
/* ref: cal_quintic_coefficient (planning_utils.py:671-703): bc [n][8] -> coeff [n][6] (absolute-s basis) */
int emp_quintic_coefficients(emp_ctx* ctx, int32_t n, const double* bc, double* coeff, emp_mem where);
/* ref: cal_obs_cost (path_planning.py:588-609): square_d [n][10] -> cost [n] */
int emp_obs_cost(emp_ctx* ctx, int32_t n, double w_collision, double danger_dis, double safe_dis,
                 const double* square_d, double* cost, emp_mem where);
/* ... for any number of samples per row (ABI 10; the reference loops over whatever it is handed, :601): square_d [n][samples] */
int emp_obs_cost_n(emp_ctx* ctx, int32_t n, int32_t samples, double w_collision, double danger_dis, double safe_dis,
                   const double* square_d, double* cost, emp_mem where);
/* ref: cal_start_cost (path_planning.py:435-514) and cal_neighbor_cost (:517-585) for n FREE edges (ABI 10): edges [n][8] =
 * start s, l, dl, ddl, span, end l, sample_s, 0.  The quintic runs from the start state to (end l, 0, 0) at start s + span (:475 /
 * :553) while the ten samples step by sample_s / 10 from the start (:492-493 / :565-566): on the lattice span = sample_s, a
 * caller of the drop-in functions may pass anything.  Obstacles per edge: obs_s, obs_l [n][max_obs], n_obs [n];
 * w_smooth3 [3] in HOST memory.  cost [n]. */
int emp_free_edge_costs(emp_ctx* ctx, int32_t n, int32_t max_obs, const double* edges, const double* obs_s, const double* obs_l,
                        const int32_t* n_obs, double w_collision, const double* w_smooth3, double w_ref, double* cost, emp_mem where);

/* ref: trajectory_index2s (planning_utils.py:758-780): x, y [B][max_pts] -> cumulative chord length [B][max_pts] */
int emp_trajectory_index2s(emp_ctx* ctx, int32_t B, int32_t max_pts, const double* x, const double* y,
                           const int32_t* n_pts, double* index2s, emp_mem where);
/* ref: Frenet2Cartesian (planning_utils.py:706-733; proj_only = 0) and CalcProjPoint (:736-755; proj_only = 1):
 * sl [B][max_pts][4] = s, l, dl, ddl -> out [B][max_pts][4] = x, y, heading, kappa (NaN from the first NaN s on) */
int emp_frenet2cartesian(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_pts, const double* ref_line,
                         const double* index2s, const int32_t* n_ref, const double* sl, const int32_t* n_pts,
                         double* out, int32_t* status, int32_t proj_only, emp_mem where);
/* ref: cal_dy_obs_deri (planning_utils.py:783-808): in [n][5] = l, vx, vy, heading, kappa -> out [n][3] */
int emp_dy_obs_deri(emp_ctx* ctx, int32_t n, const double* in, double* out, emp_mem where);

/* ---- front end of the cycle ----------------------------------
 * ref: motion_planning, test_9.py:99-110: find_match_points (planning_utils.py:49-182) for the predicted
 * location on the GLOBAL path [B][max_global][4] (x, y, heading, kappa; windowed search from pre_match_index
 * unless is_first_run) -> sampling (:231-259; 10 nodes back / 40 forward, shifted at the ends of the path) ->
 * smooth_reference_line (:262-361).  ref_line [B][EMP_REF_LINE_POINTS][4] is what emp_plan_cycle takes;
 * match_index [B] is the next cycle's pre_match_index.  status: EMP_ST_S_OUT_OF_RANGE where the reference raises
 * IndexError or slices past the path (fewer than 51 nodes), EMP_ST_SMOOTH_FAILED for the QP.
 * is_first_run and iters may be NULL. */
#define EMP_REF_LINE_POINTS 51
int emp_reference_line(emp_ctx* ctx, const emp_smooth_params* sp, int32_t B, int32_t max_global,
                       const double* global_path, const int32_t* n_global, const double* pred_xy,
                       const int32_t* is_first_run, const int32_t* pre_match_index, double* ref_line, int32_t* n_ref,
                       int32_t* match_index, int32_t* iters, int32_t* status, emp_mem where);

/* ---- lateral MPC controller ------------------------------------
 * ref: controller/controller.py class Lateral_MPC_controller (:65-337), `_control` from explicit inputs: the reference
 * reads the vehicle state from a live carla.Vehicle (cal_vehicle_info, :90-113); here the caller supplies
 * state [B][5] = x, y, yaw fi (rad), lateral velocity Vy, yaw rate fi_dot (rad/s) and vx [B] (the caller applies the
 * reference's |Vx| >= 0.005 clamp, :107-110).  target_path [B][max_path][4] = x, y, theta, kappa is the planner's
 * trajectory; min_index [B] the previous match (the search window is 50 points from it, :204).
 * Chain: cal_A_B_C_fun (:115-148) -> cal_error_k_fun(ts = 0.1) (:170-251) -> cal_coefficient_of_discretion_fun (:151-168)
 * -> cal_control_para_fun (:253-311): condensed MPC with N = 6 steps x P = 2 controls, box |u| <= 1.
 * steer [B] = first control (res['x'][0]); optional outputs (NULL to skip): u [B][12], e_rr [B][4], k_r [B],
 * pre_pro [B][4] = predicted x, y and projected x, y, H [B][12][12] and f [B][12] of the QP, iters [B].
 * status: EMP_ST_S_OUT_OF_RANGE for a bad min_index / empty path (IndexError in the reference), EMP_ST_QP_FAILED. */
typedef struct emp_mpc_params {
    double a, b, Cf, Cr, m, Iz;            /* (a, b, Cf, Cr, m, Iz) = vehicle_para (controller.py:132); the drivers pass
                                            * (1.015, 1.895, 1412, -148970, -82204, 1537) (test_9.py:316) in THIS order */
    double q_diag[4], f_diag[4], r;        /* controller.py:321-328: (250, 1, 50, 1), (1, 1, 1, 1), 1 */
} emp_mpc_params;
void emp_mpc_params_default(emp_mpc_params* p);
int emp_mpc_lateral(emp_ctx* ctx, const emp_mpc_params* p, int32_t B, int32_t max_path, const double* target_path,
                    const int32_t* n_path, const double* state, const double* vx, const int32_t* min_index,
                    double* steer, double* u, double* e_rr, double* k_r, int32_t* min_index_out, double* pre_pro,
                    double* H, double* f, int32_t* iters, int32_t* status, emp_mem where);

/* ref: controller/controller.py class Lateral_LQR_controller (:374-611), `_control` from explicit inputs (same inputs as
 * emp_mpc_lateral; min_index is only the fallback when no path point lies within 100 m: the search covers the whole
 * path, :518).  Chain: cal_A_B_fun (:424-455) -> LQR_fun (:457-486: bilinear discretisation, Riccati iteration until
 * max|dP| < 0.1 or 5000 sweeps) -> cal_error_k_fun(ts = 0.1) (:488-567) -> forward_control_fun (:569-583) ->
 * steer = -K e_rr + delta_f (:606; the raw command, not clipped).  p->q_diag / p->r are Q and R (:592-598: (200, 1, 50,
 * 1), 1; emp_lqr_params_default sets them), p->f_diag is unused.  Optional outputs (NULL to skip): K [B][4],
 * e_rr [B][4], k_r [B], pre_pro [B][4], sweeps [B] (Riccati sweeps performed). */
void emp_lqr_params_default(emp_mpc_params* p);
int emp_lqr_lateral(emp_ctx* ctx, const emp_mpc_params* p, int32_t B, int32_t max_path, const double* target_path,
                    const int32_t* n_path, const double* state, const double* vx, const int32_t* min_index,
                    double* steer, double* K, double* e_rr, double* k_r, int32_t* min_index_out, double* pre_pro,
                    int32_t* sweeps, int32_t* status, emp_mem where);

/* ---- S-T speed DP (BASELINE config 5; reference planner/speed_planning_test.py) ----------------
 * The S-T grid is hard-coded in the reference (40 non-uniform s samples :114, 16 t samples :116); tables are
 * [B][EMP_ST_ROWS][EMP_ST_COLS], row 0 = largest s (CalcSTCoordinate, :287-305).  Obstacle slots hold NaN when
 * absent (:43-46, :255); max_obs <= 64. */
#define EMP_ST_ROWS 40
#define EMP_ST_COLS 16

/* ref: generate_st_graph (speed_planning_test.py:38-98): dynamic obstacles [B][max_obs] (s, l, s_dot, l_dot; the scan
 * stops at the first NaN s) -> S-T segments s_in, s_out, t_in, t_out [B][max_obs], NaN = ignored */
int emp_st_graph(emp_ctx* ctx, int32_t B, int32_t max_obs, const double* obs_s, const double* obs_l,
                 const double* obs_s_dot, const double* obs_l_dot, double* s_in, double* s_out, double* t_in,
                 double* t_out, emp_mem where);

/* ref: speed_DP (speed_planning_test.py:101-188): forward sweep over the 40 x 16 grid with state-dependent edges
 * (CalcDpCost :191-231: source row 0 means "the DP origin", acceleration from the speed stored at the source node),
 * terminal node = last <=-minimum over the right column then the top row (:158-172), backtrack (:178-186).
 * The reference raises IndexError in its backtrack (float row index) and aliases its two output arrays (:156);
 * here the predecessor is an integer and speed_s / speed_t [B][16] are separate (NaN after the terminal column).
 * cost, s_dot [B][40][16] doubles and node [B][40][16] int32 are the reference's dp_st_cost / dp_st_s_dot /
 * dp_st_node; each may be NULL.  end_node [B][2] = (row, col) of the terminal node, (-1, -1) if every cost is NaN.
 * A negative w_cost_obs is refused (EMP_ERR_ARG): w ** (1.5 - d) (:281) is complex there and the reference fails on its
 * next comparison.  Scenes are independent; the order in which the library runs them (heaviest first) shows in nothing. */
int emp_speed_dp(emp_ctx* ctx, const emp_speed_dp_params* p, int32_t B, int32_t max_obs, const double* s_in,
                 const double* s_out, const double* t_in, const double* t_out, const double* plan_start_s_dot,
                 double* cost, double* s_dot, int32_t* node, int32_t* end_node, double* speed_s, double* speed_t,
                 emp_mem where);

/* ref: CalcDpCost (:191-231) / CalcObsCost (:234-271) for arbitrary edges: edges [B][n_edges][5] =
 * s_start, t_start, s_dot_start, s_end, t_end against scene b's obstacles -> total [B][n_edges] and (optional)
 * the obstacle term alone */
int emp_st_edge_costs(emp_ctx* ctx, const emp_speed_dp_params* p, int32_t B, int32_t n_edges, int32_t max_obs,
                      const double* edges, const double* s_in, const double* s_out, const double* t_in,
                      const double* t_out, double* total, double* obs, emp_mem where);

/* ref: CalcCollisionCost (:274-284): n distances -> n costs */
int emp_st_collision_cost(emp_ctx* ctx, int32_t n, double w_cost_obs, const double* min_dis, double* cost, emp_mem where);

/* ref: calc_speed_planning_start_condition (:23-35), called by the driver at test_10.py:249: the planning start's velocity and
 * acceleration (Cartesian) projected on the path tangent at the start's heading, n scenes -> s_dot [n], s_dot2 [n] */
int emp_speed_start_condition(emp_ctx* ctx, int32_t n, const double* vx, const double* vy, const double* ax, const double* ay,
                              const double* heading, double* s_dot, double* s_dot2, emp_mem where);

/* ---- S-T speed planning back end (reference planner/speed_planning_test.py:308-620) ----
 * Status bits of these four entry points (per scene; the arrays of a flagged scene are NaN): */
#define EMP_STB_RANGE 2       /* scipy interp1d bounds error / np.interp on an empty path (ValueError in the reference) */
#define EMP_STB_INDEX 4       /* IndexError in the reference (s_ub[16], dp_speed_s[16], a trajectory without NaN padding) */
#define EMP_STB_QP_FAILED 8   /* speed QP infeasible or not converged */
#define EMP_STB_NO_PROFILE 64 /* the profile / trajectory starts with NaN: nothing to work on */
#define EMP_SPEED_DP_COLS 16
#define EMP_SPEED_QP_POINTS 17
#define EMP_SPEED_DENSE_POINTS 401

/* ref: keyword arguments of speed_QP, speed_planning_test.py:410-411 */
typedef struct emp_speed_qp_params {
    double w_cost_s_dot2, w_cost_v_ref, w_cost_jerk;    /* 10, 50, 500 */
    double reference_speed;                             /* 50 */
} emp_speed_qp_params;
void emp_speed_qp_params_default(emp_speed_qp_params* p);

/* ref: generate_convex_space (:308-407).  dp_speed_s / dp_speed_t [B][16] (NaN behind the terminal column, as
 * emp_speed_dp returns them); path_index2s / path_kappa [B][max_path] with path_len[b] entries handed to the
 * reference (ascending; a zero-padded tail is recognised as in :326-330); obstacle S-T segments [B][n_slots], NaN =
 * empty.  Outputs s_lb, s_ub, s_dot_lb, s_dot_ub [B][16] (+-inf where unbounded). */
int emp_speed_convex_space(emp_ctx* ctx, int32_t B, int32_t n_slots, int32_t max_path, double max_lateral_accel,
                           const double* dp_speed_s, const double* dp_speed_t, const double* path_index2s,
                           const double* path_kappa, const int32_t* path_len, const double* s_in, const double* s_out,
                           const double* t_in, const double* t_out, double* s_lb, double* s_ub, double* s_dot_lb,
                           double* s_dot_ub, int32_t* status, emp_mem where);

/* ref: speed_QP (:410-511).  The reference's call cannot run (untransposed equality matrix, bounds never passed,
 * ub aliased to lb); this solves the problem it states: piecewise-linear acceleration through qp_size = (valid DP
 * columns) time stations dt = T / (qp_size - 1) apart, station 0 pinned to (0, plan_start_s_dot, plan_start_s_dot2),
 * station i >= 1 bounded by column i-1 of the convex space and -6 <= s_dot2 <= 4, s non-decreasing, cost
 * sum w_a s_dot2^2 + w_v (s_dot - v_ref)^2 + w_j (s_dot2_{i+1} - s_dot2_i)^2.  Outputs [B][17], NaN behind the last
 * station.  A DP profile without NaN tail is EMP_STB_INDEX, as in the reference (:435). */
int emp_speed_qp(emp_ctx* ctx, const emp_speed_qp_params* p, int32_t B, const double* plan_start_s_dot,
                 const double* plan_start_s_dot2, const double* dp_speed_s, const double* dp_speed_t, const double* s_lb,
                 const double* s_ub, const double* s_dot_lb, const double* s_dot_ub, double* qp_s, double* qp_s_dot,
                 double* qp_s_dot2, double* relative_time, int32_t* iters, int32_t* status, emp_mem where);

/* ref: increase_points (:514-566): [B][17] profiles -> [B][401] samples (the first sample lies at -dt and the last
 * interval extrapolates the one before it, as in the reference) */
int emp_speed_increase_points(emp_ctx* ctx, int32_t B, const double* s_init, const double* s_dot_init,
                              const double* s_dot2_init, const double* relative_time_init, double* s, double* s_dot,
                              double* s_dot2, double* relative_time, int32_t* status, emp_mem where);

/* ref: path_speed_merge (:569-620): speed samples [B][401] x path arrays [B][max_path] (n_init[b] entries as handed to
 * the reference, NaN behind the valid points) -> trajectory [B][7][401] = x, y, heading, kappa, speed, accel, time */
int emp_path_speed_merge(emp_ctx* ctx, int32_t B, int32_t max_path, const double* s, const double* s_dot,
                         const double* s_dot2, const double* relative_time, const double* current_time,
                         const double* path_s, const double* x_init, const double* y_init, const double* heading_init,
                         const double* kappa_init, const int32_t* n_init, double* trajectory, int32_t* status,
                         emp_mem where);

#ifdef __cplusplus
}
#endif
#endif /* EMPLANNER_H */
