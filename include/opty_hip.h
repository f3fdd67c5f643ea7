/* opty_hip.h -- C ABI of libopty_hip.so, the MI355X-native replacement for the
 * native code opty generates at run time for its direct-collocation hot path.
 *
 * What it replaces in the reference (csu-hmc/opty, paths relative to the
 * reference root):
 *
 *   - the generated per-node C function  `void eval_matrix(double matrix[R*C],
 *     double a0, ...)` (template opty/utils.py:483-498) and its Cython node
 *     loop `eval_matrix_loop` (opty/utils.py:500-529), instantiated twice per
 *     problem by ConstraintCollocator._gen_multi_arg_con_func
 *     (opty/direct_collocation.py:2373-2378) and _gen_multi_arg_con_jac_func
 *     (opty/direct_collocation.py:2798-2803);
 *   - the free-vector unpacking and output re-layout around them
 *     (opty/direct_collocation.py:2382-2446, :2816-2887, :2928-3001), which is
 *     folded into the kernels: the library consumes `free` and produces the
 *     flat `constraints(free)` / `jacobian(free)` vectors directly;
 *   - ConstraintCollocator.jacobian_indices
 *     (opty/direct_collocation.py:2450-2690).
 *
 * The per-problem device code is a gfx950 code object (.hsaco) produced by
 * opty_amd.codegen (the counterpart of ufuncify_matrix, opty/utils.py:639-928)
 * that exports the kernels `opty_con`, `opty_jac`, `opty_conjac`, `opty_uni`
 * (node-invariant sub-expressions) and, when the problem has instance
 * constraints, `opty_inst`.
 *
 * All functions return 0 on success and a non-zero code on failure;
 * opty_hip_last_error() then holds a message (thread-local).  A handle is not
 * thread-safe (neither is the reference: opty/utils.py:548, :672-675).  There
 * is no CPU fallback: without a HIP device opty_hip_create fails.
 *
 * Layouts (identical to the reference's):
 *   free : [x_0[0..N-1] ... x_{n-1}[..], u_0[0..N-1] ... u_{q-1}[..],
 *           p_0..p_{r-1}, h]                 (opty/direct_collocation.py:116-125)
 *   con  : con[j*(N-1) + i] for equation j, constraint node i, then the o
 *           instance constraints            (opty/direct_collocation.py:2446)
 *   jac  : jac[i*P + j*C + k], P = M*C, then the instance partials
 *                                           (opty/direct_collocation.py:2885-2887)
 *   rows/cols : int64 COO indices of every jac entry, same order
 *                                           (opty/direct_collocation.py:2628-2688)
 */
#ifndef OPTY_HIP_H
#define OPTY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct opty_hip_problem opty_hip_problem;

enum { OPTY_HIP_BACKWARD_EULER = 0, OPTY_HIP_MIDPOINT = 1 };
/* where the caller's `free` / output buffers live */
enum { OPTY_HIP_HOST = 0, OPTY_HIP_DEVICE = 1 };
/* selector for opty_hip_time_eval */
enum { OPTY_HIP_EVAL_CON = 0, OPTY_HIP_EVAL_JAC = 1, OPTY_HIP_EVAL_PAIR = 2,
       OPTY_HIP_EVAL_FUSED = 3,
       /* the fused kernel itself, whatever opty_hip_desc.fused_loses says
        * (measurement tools) */
       OPTY_HIP_EVAL_FUSED_KERNEL = 4 };

/* opty_hip_desc.routing.
 * CALIBRATE: at the first OPTY_HIP_EVAL_FUSED / OPTY_HIP_EVAL_JAC launch of
 *   every launch size the handle times opty_conjac, opty_con and opty_jac on
 *   its own device (hipEvents, a few launches into the caller's buffers) and
 *   from then on launches the faster of fused kernel / pair and of opty_jac /
 *   fused kernel; desc.fused_loses / desc.jac_via_fused (the launch plan's
 *   flags, measured on the tuner's box) only break ties (1 %).
 *   OPTY_HIP_ROUTING=plan in the environment keeps the flags as they are.
 * NO_JAC_KERNEL / NO_FUSED_KERNEL: that kernel of the module must never be
 *   launched (the build found it to spill vector registers):
 *   EVAL_JAC and the pair go through the fused kernel / EVAL_FUSED issues
 *   opty_con + opty_jac.  Not both. */
#define OPTY_HIP_ROUTE_CALIBRATE 1
#define OPTY_HIP_ROUTE_NO_JAC_KERNEL 2
#define OPTY_HIP_ROUTE_NO_FUSED_KERNEL 4

#define OPTY_HIP_LAYOUT_COO 0
#define OPTY_HIP_LAYOUT_CSR 1
#define OPTY_HIP_LAYOUT_SEGMENTED 2

/* strip classes a list schedule can hold (opty_hip_desc.jac_class_cost) */
#define OPTY_HIP_MAX_CLASSES 32

typedef struct opty_hip_desc {
    int64_t N;            /* collocation (time) nodes                        */
    int32_t n;            /* states                                          */
    int32_t M;            /* equations of motion                             */
    int32_t m_known;      /* known input trajectories                        */
    int32_t q;            /* unknown input trajectories                      */
    int32_t p_known;      /* known parameters                                */
    int32_t r;            /* unknown parameters                              */
    int32_t s;            /* 1 if the node time interval is free             */
    int32_t C;            /* columns of the per-node block (len(wrt))        */
    int32_t P;            /* values stored per node: M*C (the reference's dense
                             block), fewer when structural zeros are pruned  */
    int32_t method;       /* OPTY_HIP_BACKWARD_EULER / OPTY_HIP_MIDPOINT     */
    int32_t num_inst;     /* o: instance constraints                         */
    int32_t nnz_inst;     /* instance-constraint Jacobian entries            */
    int32_t num_inst_atoms; /* distinct x(t_k) atoms in instance constraints */
    int32_t jac_wgs_per_block; /* workgroups per 64-node block (opty_jac)    */
    int32_t jac_waves_per_wg;  /* 64-lane waves per such workgroup           */
    int32_t fused_wgs_per_block; /* workgroups per block of opty_conjac      */
    int32_t con_wgs_per_block; /* workgroups per block of opty_con           */
    int32_t num_uniform;  /* entries of the node-invariant table (opty_uni)  */
    int32_t uniform_dynamic; /* 1 if that table depends on `free` (r+s > 0)  */
    int32_t device;       /* HIP device ordinal                              */
    int32_t fused_waves_per_wg; /* waves per workgroup of opty_conjac         */
    int32_t con_waves_per_wg;   /* waves per workgroup of opty_con            */
    int32_t layout;       /* OPTY_HIP_LAYOUT_COO: the reference's node-major
                             order jac[i*P + e]; OPTY_HIP_LAYOUT_CSR: sorted by
                             row then column, jac[S_j*(N-1) + i*L_j + pos]
                             (needs opty_hip_set_block_pattern);
                             OPTY_HIP_LAYOUT_SEGMENTED: the block's entries in
                             three node-major segments, jac[S_g*(N-1) + i*L_g
                             + pos] (needs opty_hip_set_segments)            */
    int32_t inst_folded;  /* 1: opty_con / opty_jac / opty_conjac evaluate the
                             instance tails themselves when launched with one
                             workgroup more than the node blocks need (small
                             problems: saves the launch of opty_inst)        */
    int32_t fused_loses;  /* 1: for this problem and launch size the fused
                             kernel was MEASURED slower than opty_con followed
                             by opty_jac (launch plan): opty_hip_eval_con_jac
                             and OPTY_HIP_EVAL_FUSED issue the two launches.
                             The reference evaluates the two callbacks
                             separately in any case
                             (opty/direct_collocation.py:498-562)            */
    int32_t jac_via_fused; /* 1: the launch plan measured opty_conjac FASTER than
                             opty_jac for this problem and launch size (the
                             constraint waves ride in the shadow of the store
                             stream): OPTY_HIP_EVAL_JAC launches the fused
                             kernel, its constraint values go to a scratch
                             vector of the handle                            */
    int32_t jac_persist;  /* > 0: opty_jac is a persistent kernel with a list
                             schedule: launched with at most this many
                             one-wave workgroups (a multiple of 8; 1024 = one
                             per SIMD), each of which evaluates a list of
                             (node block, strip class) items that the library
                             computes per launch size -- longest class first
                             onto the least loaded workgroup of the block's
                             XCD -- from the class costs below.  A launch of
                             waves that each hold a SIMD alone then costs
                             about sum(wave durations) / 1024, without the idle
                             time a one-wave-per-item dispatch leaves between
                             and after the waves                              */
    int32_t fused_persist; /* the same for opty_conjac                        */
    int32_t routing;      /* OPTY_HIP_ROUTE_* bits: which of the module's
                             kernels the entry points may launch, and whether
                             the handle measures the choice itself           */
    float jac_class_cost[OPTY_HIP_MAX_CLASSES];   /* relative duration of the
                             wave of strip class g (any unit; measured by the
                             launch plan's tuner or the printer's estimate);
                             classes = jac_wgs_per_block                      */
    float fused_class_cost[OPTY_HIP_MAX_CLASSES];
} opty_hip_desc;

/* Version of this header's structs and signatures; opty_hip_abi_version()
 * returns the one the library was built from.  A client built against another
 * version must not call the library: the descriptor grew in 5, 6 and 7, and
 * opty_hip_eval_jac_persistent / opty_hip_shard_jac_to_host took their `fresh`
 * argument in 4. */
#define OPTY_HIP_ABI_VERSION 7
int opty_hip_abi_version(void);

/* (The build verification's device side -- register poisoner, instruction
 * tape -- is a library of its own: include/opty_hip_referee.h.) */

/* The list schedule the library gives a persistent kernel (opty_hip_desc.
 * jac_persist / fused_persist) for a launch over `node_blocks` 64-node blocks
 * with `classes` strip classes of relative duration class_cost[g]: host
 * arithmetic only (inspection, tests).  table[0] = workgroups npw;
 * table[1 .. npw + 1] = first item of workgroup w (workgroup w serves XCD
 * w % 8); table[npw + 2 ...] = items, (class << 24) | s for node block
 * 8 s + XCD, every workgroup's in the order in which it evaluates them.
 * `table` may be null (count only). */
int opty_hip_list_schedule(int persist, int64_t node_blocks, int classes,
                           const float *class_cost, int32_t *table,
                           int64_t capacity, int64_t *count);

/* Loads the code object and allocates the device-side state (known
 * parameters, known trajectories, staging buffers). */
int opty_hip_create(const opty_hip_desc *desc, const char *code_object_path,
                    opty_hip_problem **out);
int opty_hip_destroy(opty_hip_problem *p);

/* The null / legacy default stream (hipStreamLegacy) as a set_stream argument:
 * NULL itself means "the handle's own stream". */
#define OPTY_HIP_STREAM_LEGACY ((void *)1)

/* Use an existing hipStream_t (e.g. torch's current stream; pass
 * OPTY_HIP_STREAM_LEGACY for the default stream, whose handle is NULL) for
 * all work of this handle; NULL restores the handle's own stream.  A handle's device
 * state belongs to one stream at a time: the first evaluation after a switch
 * waits (on the host) for everything the handle enqueued on the previous
 * stream. */
int opty_hip_set_stream(opty_hip_problem *p, void *hip_stream);
int opty_hip_synchronize(opty_hip_problem *p);

/* Values of the known parameters in ConstraintCollocator.known_parameters
 * order.  Replaces the scalar by-value arguments c0.. of eval_matrix
 * (opty/utils.py:489-490) that _multi_arg_con_func passes on every call
 * (opty/direct_collocation.py:2436-2437). */
int opty_hip_set_known_parameters(opty_hip_problem *p, const double *values,
                                  int32_t count);
/* Node time interval when it is not a free variable; must be set before the
 * first evaluation (the reference passes `interval_value` per call,
 * opty/direct_collocation.py:2382-2385, :2437). */
int opty_hip_set_interval(opty_hip_problem *p, double h);
/* (m_known x N) row-major array, known_input_trajectories order: the known
 * rows _merge_fixed_free interleaves into the specifieds
 * (opty/direct_collocation.py:2891-2926). */
int opty_hip_set_known_trajectories(opty_hip_problem *p, const double *values,
                                    int32_t mem);
/* Instance constraints: free-vector index of every atom
 * (_find_closest_free_index, opty/direct_collocation.py:2169-2217) and the COO
 * rows/cols of the instance part of the Jacobian
 * (_instance_constraints_jacobian_indices, :2233-2251); host arrays. */
int opty_hip_set_instance_indices(opty_hip_problem *p,
                                  const int64_t *atom_free_index,
                                  const int64_t *rows, const int64_t *cols);

/* Pruned blocks (P < M*C) and the CSR layout: (j, k) of every stored entry
 * of a block in storage order, 2*P int32 (CSR: grouped by j, ascending
 * column within a row). */
int opty_hip_set_block_pattern(opty_hip_problem *p, const int32_t *jk);

int64_t opty_hip_num_free(const opty_hip_problem *p);
int64_t opty_hip_num_constraints(const opty_hip_problem *p);
int64_t opty_hip_nnz(const opty_hip_problem *p);

/* constraints(free): writes num_constraints doubles.  Replaces the closure
 * built by _wrap_constraint_funcs(..., 'con') (opty/direct_collocation.py:
 * 2952-2993) including _multi_arg_con_func (:2382-2446) and the compiled
 * eval_matrix_loop (opty/utils.py:500-529). */
int opty_hip_eval_con(opty_hip_problem *p, const double *free, double *con,
                      int32_t mem);
/* jacobian(free): writes nnz doubles.  Replaces the 'jac' closure of
 * _wrap_constraint_funcs and _multi_arg_con_jac_func
 * (opty/direct_collocation.py:2816-2887). */
int opty_hip_eval_jac(opty_hip_problem *p, const double *free, double *jac,
                      int32_t mem);
/* both of the above for one `free` from one kernel launch (no reference
 * counterpart: IPOPT asks for them separately, :498-525, :552-562). */
int opty_hip_eval_con_jac(opty_hip_problem *p, const double *free, double *con,
                          double *jac, int32_t mem);
/* ---- host-visible Jacobian: only what changed crosses PCIe ------------------
 * The reference returns jacobian(free) in ONE persistent array that the next
 * call overwrites (opty/direct_collocation.py:2814, :2885-2887) and whose
 * per-node block is dense -- structural zeros and node-invariant entries
 * included (:2589-2593).  opty_hip_set_varying_entries names the block entries
 * whose value can differ between two evaluations with the same known
 * parameters and node time interval (ascending, 0 <= e < P; everything else
 * is a function of those alone: literal zeros, +-1, 1/h, masses ...).
 * opty_hip_eval_jac_persistent(free, jac), both HOST, jac page-locked
 * (opty_hip_host_alloc), then computes the same values as opty_hip_eval_jac
 * but moves only the varying entries -- packed on the device, copied in chunks,
 * scattered into `jac` by a pool of host threads -- whenever `jac` is the
 * vector of the previous call and the invariant entries it holds are still
 * valid (they are re-sent after opty_hip_set_known_parameters /
 * opty_hip_set_interval).  The caller must not write into `jac` between
 * calls, and passes `fresh` != 0 on the FIRST call with an allocation (and
 * whenever it may have written into it): the library compares addresses only
 * as a second line of defence -- an allocator hands a freed block out again at
 * the same address, so address identity is not a validity token.  Node-major
 * layout only; without varying entries set it is opty_hip_eval_jac. */
int opty_hip_set_varying_entries(opty_hip_problem *p, const int32_t *entries,
                                 int32_t count);
/* Block entries that are the SAME expression as an earlier varying entry (a
 * symmetric mass matrix; its entries in the columns of the current and of the
 * adjacent node's speeds): they are not named as varying entries, do not cross
 * PCIe, and the host threads fill entry dst[k] of every node's block from
 * entry src[k] of the same block (dst ascending and not varying entries; src
 * varying entries).  Call after opty_hip_set_varying_entries, which clears the
 * list. */
int opty_hip_set_entry_copies(opty_hip_problem *p, const int32_t *dst,
                              const int32_t *src, int32_t count);
/* The same with a factor per copy: the host threads fill entry dst[k] with
 * scale[k] x entry src[k] of the same block -- varying entries that are a
 * node-invariant multiple of another varying entry (c_1 X and c_2 X over the
 * same per-node expression X: scale = c_1/c_2, computed by the caller from the
 * known parameters and set again when those change) do not cross PCIe either.
 * scale == NULL: plain copies.  A factor of exactly 1.0 reproduces the source
 * bit for bit. */
int opty_hip_set_entry_copies_scaled(opty_hip_problem *p, const int32_t *dst,
                                     const int32_t *src, const double *scale,
                                     int32_t count);
int opty_hip_eval_jac_persistent(opty_hip_problem *p, const double *free,
                                 double *jac, int32_t fresh);
/* ---- OPTY_HIP_LAYOUT_SEGMENTED: a host-visible Jacobian without a scatter ---
 * IPOPT takes the (row, col, value) triplets in any order
 * (opty/direct_collocation.py:527-562 hands it rows / cols once, :552-562 the
 * values per call).  In this layout the dense block of the reference
 * (:2589-2593) is stored as three node-major segments,
 *
 *   jac = [ seg 0 of node 0 .. N-2 | seg 1 of node 0 .. N-2 | seg 2 ... | tail ]
 *
 * order[0 .. P): the block entries (reference numbering e = j*C + k) in
 * stored order -- seg_len[0] entries that can differ between two evaluations,
 * then seg_len[1] entries that are the same expression as an entry of segment
 * 0 (copy_source[k]: its position in segment 0), then seg_len[2] entries that
 * depend on known parameters and the node time interval alone.  The kernels
 * and opty_hip_jacobian_indices follow that order; HOST evaluations move
 * segment 0 over PCIe straight into the head of the caller's vector, fill
 * segment 1 on the host from it, and -- opty_hip_eval_jac_persistent -- leave
 * segment 2 alone after the first call (re-sent after
 * opty_hip_set_known_parameters / opty_hip_set_interval and when `fresh`).
 * No read-for-ownership of the dense vector, no per-entry scatter.  Whole
 * problems only (node shards use the node-major layout). */
int opty_hip_set_segments(opty_hip_problem *p, const int32_t *order,
                          const int32_t *seg_len, const int32_t *copy_source);
/* Largest fraction of a block's stored entries for which only the varying
 * ones are moved (beyond it whole blocks are copied and no entry copies are
 * applied): 0.8, or OPTY_HIP_PACK_RATIO. */
double opty_hip_pack_ratio(void);
/* Where the host threads that scatter the varying entries run (NUMA node, -1:
 * unplaced), the NUMA node of the current device, and whether the placement
 * has been verified by measurement (the pool times its candidates when a
 * scatter ends late; OPTY_HIP_HOST_PLACEMENT=fixed disables that). */
int opty_hip_host_placement(int32_t *workers_node, int32_t *device_node,
                            int32_t *verified);
/* The same for one node shard of a problem evaluated by several GPUs: the
 * blocks of the constraint nodes [node_begin, node_end), as
 * opty_hip_eval_shard left them in device memory (d_jac_shard), go into the
 * slice [node_begin*P, node_end*P) of the dense HOST vector of the GLOBAL
 * problem (host_jac: the shared page-locked vector every rank copies its
 * shard into, SURVEY.md 8(e) "direct-to-host") -- all of them the first
 * time (and whenever `fresh` != 0: a new mapping of the shared vector, see
 * above), later only the varying entries.  Synchronous; runs on the handle's
 * stream behind the evaluation. */
int opty_hip_shard_jac_to_host(opty_hip_problem *p, const double *d_jac_shard,
                               double *host_jac, int64_t node_begin,
                               int64_t node_end, int32_t fresh);
/* Host threads of the scatter pool (per process; 0 = the default:
 * OPTY_HIP_HOST_THREADS, or min(16, hardware threads / 4 / LOCAL_WORLD_SIZE)
 * -- the ranks of a node share the cores of the NUMA node that holds the one
 * vector they all scatter into; each worker gets a core of its own there,
 * LOCAL_RANK deciding which).  The workers run on the cores of
 * the NUMA node that holds the caller's dense vector (get_mempolicy);
 * OPTY_HIP_HOST_NUMA=<node> overrides, =off leaves them where the creating
 * thread may run.  The workers never leave the affinity mask of the thread
 * that created the pool (taskset / cpuset / an OpenMP binding are honoured,
 * and the pool has at most as many workers as that mask has CPUs);
 * OPTY_HIP_HOST_AFFINITY=wide lifts that for applications whose binding pins
 * only the calling thread. */
int opty_hip_set_host_threads(int32_t count);
int opty_hip_host_threads(void);
/* NUMA node that holds the first page of a host allocation (-1: unknown). */
int opty_hip_host_numa_node(const void *ptr);

/* jacobian_indices(): writes nnz int64 rows and cols -- the closed form of
 * ConstraintCollocator.jacobian_indices (opty/direct_collocation.py:2450-2690,
 * formulas :2644-2675). */
int opty_hip_jacobian_indices(opty_hip_problem *p, int64_t *rows,
                              int64_t *cols, int32_t mem);

/* The same for a handle that evaluates the constraint nodes
 * [node_offset, node_offset + N - 1) of a larger problem with N_global nodes
 * (node sharding across GPUs): global row/col indices of the shard's values. */
int opty_hip_jacobian_indices_shard(opty_hip_problem *p, int64_t N_global,
                                    int64_t node_offset, int64_t *rows,
                                    int64_t *cols, int32_t mem);

/* Global indices of the values opty_hip_eval_shard writes for the constraint
 * nodes [node_begin, node_end) of this handle's problem: (node_end -
 * node_begin)*P int64 each. */
int opty_hip_jacobian_indices_range(opty_hip_problem *p, int64_t node_begin,
                                    int64_t node_end, int64_t *rows,
                                    int64_t *cols, int32_t mem);

/* Runs `iters` evaluations back to back on the handle's stream with device
 * buffers and returns the mean milliseconds per evaluation measured with
 * hipEvents recorded on that stream. */
int opty_hip_time_eval(opty_hip_problem *p, int32_t what, const double *free,
                       double *con, double *jac, int32_t iters,
                       float *ms_per_iter);

/* ---- node shards (one process per GPU) ----------------------
 * Evaluates the constraint nodes [node_begin, node_end) of the handle's
 * problem from the GLOBAL free vector (device memory, full length; only the
 * time-node columns [node_begin, node_end] of its trajectory rows and its
 * parameter tail contribute -- the one-node halo of
 * opty/direct_collocation.py:2411-2413; the last 64-node block may touch up
 * to 64 columns past node_end, whose values are not used).
 * Device pointers only:
 *   con : the shard's value of equation j, node i goes to
 *         con[j*con_stride + (i - node_begin)]; con_stride = N-1 with
 *         con = global_con + node_begin writes the shard in place into the
 *         global equation-major vector (opty/direct_collocation.py:2446),
 *         con_stride = node_end - node_begin gives a dense (M x nodes) block;
 *   jac : the shard's blocks, jac[(i - node_begin)*P + e]; jac =
 *         global_jac + node_begin*P is the contiguous slice
 *         [node_begin*P, node_end*P) of the global node-major vector
 *         (opty/direct_collocation.py:2885-2887).
 * `what` is OPTY_HIP_EVAL_*.  Shards cover the collocation part only: the
 * instance constraints of a problem that has them are evaluated once, by
 * whichever rank assembles the vectors, with opty_hip_eval_instance.  The CSR
 * layout is not sharded. */
int opty_hip_eval_shard(opty_hip_problem *p, int32_t what, const double *free,
                        double *con, int64_t con_stride, double *jac,
                        int64_t node_begin, int64_t node_end);
/* The instance-constraint tails from the GLOBAL free vector (device
 * pointers): the o values that follow the M*(N-1) collocation constraints
 * (opty/direct_collocation.py:2985-2991) into con_tail[0..o) and the nnz_inst
 * partials that follow the P*(N-1) block values (:2686-2688, :2253-2282) into
 * jac_tail[0..nnz_inst); either may be NULL.  What a rank of a node-sharded
 * problem calls for the tail of the vector it assembles; a no-op for problems
 * without instance constraints. */
int opty_hip_eval_instance(opty_hip_problem *p, const double *free,
                           double *con_tail, double *jac_tail);
/* opty_hip_time_eval for a shard. */
int opty_hip_time_eval_shard(opty_hip_problem *p, int32_t what,
                             const double *free, double *con,
                             int64_t con_stride, double *jac,
                             int64_t node_begin, int64_t node_end,
                             int32_t iters, float *ms_per_iter);

/* What the entry points launch for a launch of `node_count` constraint nodes
 * (opty_hip_desc.routing): *calibrated = 1 when the handle has measured that
 * launch size on its device (ms3[0..2] = per-launch ms of opty_conjac,
 * opty_con, opty_jac as measured), 0 when the launch plan's flags are still
 * in force; *fused_loses = 1: OPTY_HIP_EVAL_FUSED issues opty_con + opty_jac;
 * *jac_via_fused = 1: OPTY_HIP_EVAL_JAC launches opty_conjac.  Any output
 * pointer may be null.  The reference calls its two callbacks separately
 * (opty/direct_collocation.py:498-562): whichever kernel serves them, the
 * values are those of the same expressions. */
int opty_hip_routing(opty_hip_problem *p, int64_t node_count,
                     int32_t *calibrated, int32_t *fused_loses,
                     int32_t *jac_via_fused, float *ms3);

/* ---- objective and objective gradient -------------
 * Device counterpart of create_objective_function (opty/utils.py:329-470):
 * value = h*sum_i w_i G(node i) + b(p) and its gradient with respect to
 * free = [x rows, u rows, p], quadrature weights per opty/utils.py:419-464.
 * The code object exports `opty_objgrad` and `opty_objfin`. */
typedef struct opty_hip_objective opty_hip_objective;

typedef struct opty_hip_objective_desc {
    int64_t N;        /* collocation nodes                                   */
    int32_t n, q, r;  /* states, unknown inputs, unknown parameters          */
    int32_t device;   /* HIP device ordinal                                  */
    double h;         /* node time interval                                  */
} opty_hip_objective_desc;

int opty_hip_objective_create(const opty_hip_objective_desc *desc,
                              const char *code_object_path,
                              opty_hip_objective **out);
int opty_hip_objective_destroy(opty_hip_objective *o);
int opty_hip_objective_set_stream(opty_hip_objective *o, void *hip_stream);
/* value: one double (host memory, always); grad: (n+q)*N + r doubles in `mem`
 * memory or NULL for the value alone; free: (n+q)*N + r doubles in `mem`. */
int opty_hip_objective_eval(opty_hip_objective *o, const double *free,
                            double *value, double *grad, int32_t mem);

/* ---- plain matrix functions: the reference's plugin call shape ---------------
 * `f = ufuncify_matrix(args, expr, const=...)`, `f(result, *num_args)`
 * (opty/utils.py:639-640; argument contract :610-617, :778-807): a rows x cols
 * matrix of expressions evaluated for n independent argument rows, the
 * generated `eval_matrix` + `eval_matrix_loop` pair (opty/utils.py:483-529)
 * as one gfx950 kernel (`opty_jac` of a "matrix program", lane = evaluation
 * row, row-major (n, rows*cols) output through the same tile flush as a
 * Jacobian block) plus `opty_uni` for sub-expressions of the const
 * arguments alone. */
typedef struct opty_hip_matrix opty_hip_matrix;

typedef struct opty_hip_matrix_desc {
    int32_t num_vec;        /* vector arguments: one double per row           */
    int32_t num_const;      /* const arguments: one double per call           */
    int32_t rows, cols;     /* shape of the matrix                            */
    int32_t wgs_per_block;  /* launch geometry of opty_jac (from the emitter) */
    int32_t waves_per_wg;
    int32_t num_uniform;    /* entries of the node-invariant table            */
    int32_t device;         /* HIP device ordinal                             */
} opty_hip_matrix_desc;

int opty_hip_matrix_create(const opty_hip_matrix_desc *desc,
                           const char *code_object_path,
                           opty_hip_matrix **out);
int opty_hip_matrix_destroy(opty_hip_matrix *m);
int opty_hip_matrix_set_stream(opty_hip_matrix *m, void *hip_stream);
/* result: n*rows*cols doubles, result[i*rows*cols + r*cols + c];
 * vec_args: HOST array of num_vec pointers, each to n contiguous doubles in
 * `mem` memory (n >= 1); const_args: num_const doubles in HOST memory (passed
 * by value in the reference).  Synchronous for OPTY_HIP_HOST, enqueued on the
 * handle's stream for OPTY_HIP_DEVICE. */
int opty_hip_matrix_eval(opty_hip_matrix *m, double *result,
                         const double *const *vec_args,
                         const double *const_args, int64_t n, int32_t mem);

/* Page-locked host memory for OPTY_HIP_HOST callers: output arrays that live
 * in it (the persistent Jacobian value buffer the reference keeps,
 * opty/direct_collocation.py:2814) are copied back at full PCIe rate. */
void *opty_hip_host_alloc(size_t bytes);
int opty_hip_host_free(void *ptr);
/* Page-locks memory the caller already owns -- e.g. a shared-memory mapping
 * of the one host vector that IPOPT reads and into which every rank copies its
 * node shard over its own PCIe link ("direct-to-host"). */
int opty_hip_host_register(void *ptr, size_t bytes);
int opty_hip_host_unregister(void *ptr);
/* Device memory for callers without a HIP toolchain of their own (a plain C
 * host that node-shards a problem: opty_hip_eval_shard and the communicator
 * calls below take device pointers); `kind`: 0 host->device, 1 device->host,
 * 2 device->device, synchronous. */
void *opty_hip_device_alloc(int32_t device, size_t bytes);
int opty_hip_device_free(void *ptr);
int opty_hip_memcpy(void *dst, const void *src, size_t bytes, int32_t kind);

/* ---------------------------------------------------------------------------
 * Several GPUs, one process each: the RCCL side of a node-sharded problem
 * (BASELINE config 4).  The reference parallelises the node
 * loop over an OpenMP team (opty/utils.py:524-526); here every rank evaluates
 * its node range with opty_hip_eval_shard from the global free vector in its
 * own HBM, and these calls move (i) that vector to every rank and (ii) the
 * shards to a rank that wants whole vectors -- RCCL point-to-point over xGMI
 * on the problem handle's stream, issued by this library (librccl is loaded
 * on first use).  No call here is needed on a single GPU.
 *
 * Bring-up: rank 0 calls opty_hip_comm_unique_id and hands the
 * OPTY_HIP_COMM_ID_BYTES bytes to the other ranks by whatever means the
 * launcher offers (a file, MPI, a torch.distributed store); every rank then
 * calls opty_hip_comm_create (collective: it returns when all `world` ranks
 * have called it). */
typedef struct opty_hip_comm opty_hip_comm;
#define OPTY_HIP_COMM_ID_BYTES 128
int opty_hip_comm_unique_id(void *id_out /* OPTY_HIP_COMM_ID_BYTES bytes */);
int opty_hip_comm_create(const void *unique_id, int32_t rank, int32_t world,
                         int32_t device, opty_hip_comm **out);
int opty_hip_comm_destroy(opty_hip_comm *c);
int opty_hip_comm_rank(const opty_hip_comm *c);
int opty_hip_comm_world(const opty_hip_comm *c);
/* Broadcast of the global free vector (device memory, opty_hip_num_free
 * doubles on every rank) from rank `root`, ordered on `p`'s stream like an
 * evaluation.  A no-op in a world of one. */
int opty_hip_bcast_free(opty_hip_comm *c, opty_hip_problem *p,
                        double *free_dev, int32_t root);
/* Gather-v of node shards to rank `root`.  `bounds`: world + 1 ascending
 * constraint-node boundaries, rank g owns [bounds[g], bounds[g+1]) (shards
 * may differ in size and may be empty).  Every other rank passes what
 * opty_hip_eval_shard wrote for its range: `jac_shard` = its (b-a)*P values,
 * `con_shard` = its dense (M, b-a) block (con_stride = b-a); one grouped
 * ncclSend each.  The root passes the global vectors of the problem
 * (opty_hip_num_constraints / opty_hip_nnz doubles): the peers' Jacobian
 * slices are received IN PLACE (jac[a*P .. b*P), contiguous in the node-major
 * layout, opty/direct_collocation.py:2885-2887), their constraint blocks into
 * staging that one strided device copy per peer puts at con[j*(N-1) + a]
 * (equation-major, :2446).  The root's own shard is copied from `con_shard` /
 * `jac_shard` unless those are NULL (it evaluated in place: con_stride = N-1,
 * jac + a*P).  `what`: OPTY_HIP_EVAL_CON, _JAC or _PAIR (both).  The o
 * instance constraints are not node shards: the root evaluates them itself
 * (opty_hip_eval_instance).  Asynchronous on `p`'s stream. */
int opty_hip_gather_v(opty_hip_comm *c, opty_hip_problem *p,
                      const int64_t *bounds, const double *con_shard,
                      const double *jac_shard, double *con_global,
                      double *jac_global, int32_t root, int32_t what);

int opty_hip_device_count(void);
const char *opty_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* OPTY_HIP_H */
