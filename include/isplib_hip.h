/*
 * isplib_hip.h -- C ABI of the MI355X (gfx950) SpMM aggregation backend.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.
 * Every pointer marked [dev] is a DEVICE pointer (HBM); every entry point is
 * asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default
 * stream) and returns a FusedMM status code.  The compute entries neither
 * allocate, free nor synchronise, so they are hipGraph-capturable; those that
 * need scratch take a caller-owned workspace whose size a *_workspace_bytes
 * query reports.  The two exceptions say so where they are declared: the plan
 * builder isplib_spmm_tasks_count_hip (one stream synchronisation, once per
 * graph) and the isplib_graph_* handle, which owns device memory on the
 * caller's behalf.
 *
 * Reference interfaces replaced (paths relative to the iSpLib tree):
 *   fusedMM_csr_hip            <- fusedMM_csr, csrc/fusedMM.h:77-99, called at
 *                                 csrc/fusedmm.cpp:198; it takes the place of
 *                                 the never-linked fusedmm_cuda() stub at
 *                                 csrc/fusedmm.cpp:89-110,191-196 and of the
 *                                 thread-per-row prototype gpu/fusedmm.cu:18-51.
 *   performDummySpMM_hip       <- performDummySpMM, csrc/fusedmm.cpp:61,570.
 *   isplib_spmm_minmax_bw_hip  <- the ATen gather/mul/masked_fill/scatter_add_
 *                                 chain of FusedMM_SPMMMax/Min::backward,
 *                                 csrc/fusedmm.cpp:410-451 and 477-517.
 *   isplib_sddmm_csr_hip       <- the commented-out spmm_value_bw call,
 *                                 csrc/fusedmm.cpp:270,351 (dA for sum/mean).
 *   isplib_csr_* (graph prep)  <- torch_sparse storage getters the wrapper
 *                                 forces at isplib/__init__.py:67-73 and the two
 *                                 nnz-sized gathers it caches at :79-80,:86-99.
 *   isplib_graph_*             <- the per-graph caches of isplib/__init__.py:35-40,
 *                                 50,76-106 (class-level dicts keyed by data pointers).
 *   isplib_suggest_slices*     <- autotuner/findbestk.py, gpu/kernels/codegen.py
 *                                 (sweep a parameter, keep the fastest).
 *   fusedMM_csr_udef*_hip      <- the other message words of csrc/fusedMM.h:18-74
 *                                 (defined there, never sent by iSpLib).
 */
#ifndef ISPLIB_HIP_H
#define ISPLIB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISPLIB_HIP_ABI_VERSION 1
#define ISPLIB_MAX_SLICES 4096       /* column-slice counts accepted by the sliced / task-list entries: 1..4096 */

/* ---- FusedMM op message (values fixed by csrc/fusedMM.h:18-74) ---------- */
#define ISPLIB_VOP_COPY_RHS 0x2
#define ISPLIB_ROP_NOOP     0x00
#define ISPLIB_SOP_COPY     0x100
#define ISPLIB_VSC_MUL      0x1000
#define ISPLIB_VSC_MEAN     0x3000
#define ISPLIB_AOP_ADD      0x10000
#define ISPLIB_AOP_MAX      0x20000
#define ISPLIB_AOP_MIN      0x30000
/* the four messages the reference sends, csrc/fusedmm.cpp:168-186 */
#define ISPLIB_MSG_SPMM_SUM  (ISPLIB_VOP_COPY_RHS | ISPLIB_ROP_NOOP | ISPLIB_SOP_COPY | ISPLIB_VSC_MUL  | ISPLIB_AOP_ADD)
#define ISPLIB_MSG_SPMM_MEAN (ISPLIB_VOP_COPY_RHS | ISPLIB_ROP_NOOP | ISPLIB_SOP_COPY | ISPLIB_VSC_MEAN | ISPLIB_AOP_ADD)
#define ISPLIB_MSG_SPMM_MAX  (ISPLIB_VOP_COPY_RHS | ISPLIB_ROP_NOOP | ISPLIB_SOP_COPY | ISPLIB_VSC_MUL  | ISPLIB_AOP_MAX)
#define ISPLIB_MSG_SPMM_MIN  (ISPLIB_VOP_COPY_RHS | ISPLIB_ROP_NOOP | ISPLIB_SOP_COPY | ISPLIB_VSC_MUL  | ISPLIB_AOP_MIN)

/* ---- status codes (csrc/fusedMM.h:105-114) + one for HIP runtime errors -- */
#define ISPLIB_SUCCESS         0
#define ISPLIB_FAIL            1    /* bad argument                            */
#define ISPLIB_NOT_ENOUGH_MEM (-1)  /* workspace too small                      */
#define ISPLIB_UNDEFINED_USER_FUNCTION 64 /* a *_UDEF stage without a built-in function (csrc/fusedMM.h:113) */
#define ISPLIB_NO_OPT_IMPL     128  /* message outside the implemented set      */
#define ISPLIB_HIP_ERROR       256  /* a HIP call failed; see isplib_hip_last_error */

int         isplib_hip_abi_version(void);
const char *isplib_hip_last_error(void);   /* thread-local, "" if none */

/*
 * The one convention of this path that nothing in the reference tree pins: what an EMPTY row of a max / min SpMM holds.
 * The reference launcher pre-fills the output with lowest() / max() and the positions with nnz (csrc/fusedmm.cpp:147-150,
 * 171,177) and hands them to fusedMM_csr, whose body is not in the tree (configure:2-7).
 *   0 "zero" (default)  the row is 0 -- torch_sparse's CPU kernel, the one the iSpLib authors compared against
 *                       (isplib/__init__.py:120-128), and what oracle/ restates
 *   1 "init"            the row keeps the launcher's pre-fill, -FLT_MAX (max) / +FLT_MAX (min) -- what a body returns
 *                       that only visits stored entries
 * The positions are nnz either way; sum / mean are 0 either way.  Process-wide, every schedule; read once from the
 * environment (ISPLIB_EMPTY_ROW=init|zero) unless set here first.  A maintainer who holds the real fusedmm_cpu.a picks
 * whichever it does.
 */
int isplib_hip_set_empty_row(int init);
int isplib_hip_get_empty_row(void);

/*
 * SpMM with the reference's 20-argument FusedMM signature, device pointers.
 *
 *   z[i,:] = REDUCE_{j in [pntrb[i], pntre[i])}  val[j] * y[indx[j], :]
 *
 * REDUCE/scale from `imessage` (one of ISPLIB_MSG_SPMM_*).  Differences from
 * the host ABI, all consequences of running on the device:
 *   - z and z_arg are WRITE-ONLY: the kernel writes the value the reference
 *     gets by pre-filling z with 0 / -FLT_MAX / +FLT_MAX (csrc/fusedmm.cpp:
 *     147-152) and z_arg with nnz (:171,177) and then accumulating.  beta must
 *     be 0 (the only value the reference passes, :117).  This folds two
 *     M*K-sized fill passes into the kernel's write-back.
 *   - val may be NULL = unit weights (what isplib/__init__.py:51-57
 *     materialises as a ones vector); the stream is then never read.
 *   - MAX/MIN: strict compare in CSR order (lowest CSR position wins ties, NaN
 *     never wins); empty row -> value 0 (or the launcher's init value: isplib_hip_set_empty_row), z_arg = nnz.  z_arg (int64, same
 *     leading dimension as z) holds ABSOLUTE CSR positions; may be NULL.
 *   - x/ldx/alpha/rows/cols are accepted and ignored, as in the reference.
 * Requirements: n < 2^31, every row's degree < 2^31, ldy >= k, ldz >= k.
 * Any other message word is handed to the generic pipeline below
 * (fusedMM_csr_udef_hip with no user function).
 */
int fusedMM_csr_hip(int32_t imessage, int64_t m, int64_t n, int64_t k,
                    float alpha, int64_t nnz, int64_t rows, int64_t cols,
                    const float *val /*[dev] nnz | NULL*/,
                    const int64_t *indx /*[dev] nnz*/,
                    const int64_t *pntrb /*[dev] m*/,
                    const int64_t *pntre /*[dev] m*/,
                    const float *x /*[dev] m x ldx; only read by non-SpMM messages*/, int64_t ldx,
                    const float *y /*[dev] n x ldy*/, int64_t ldy, float beta,
                    float *z /*[dev] m x ldz*/, int64_t ldz,
                    int64_t *z_arg /*[dev] m x ldz | NULL*/, void *stream);

/*
 * The generic five-stage FusedMM pipeline (csrc/fusedMM.h:18-74) for message words other than
 * the four SpMM ones -- the SDDMM-fused patterns of the FusedMM paper (graph embedding with a
 * sigmoid or t-distribution kernel, score-then-aggregate attention).  The reference never sends
 * them (csrc/fusedmm.cpp:168-186) and the library that defines them is absent from its tree, so
 * the semantics are restated here and in oracle/fusedmm_oracle.c; nothing pins them.
 * For row i and stored entry e = (i, j), a = val[e] (1 if val is NULL), per column c:
 *   VOP  T[c] = COPY_LHS x[i,c] | COPY_RHS y[j,c] | ADD x+y | SUBL x-y | SUBR y-x | MAX | MIN
 *   ROP  s    = NOOP 1 | DOT sum x[i,c]*T[c] | ADD_LHS sum x | ADD_RHS sum T | NORML sum x^2 | NORMR sum T^2
 *   SOP  s'   = NOOP s | COPY a | UDEF f(s): C function pointers cannot cross to the device, so the
 *               user function is chosen from a built-in menu (`sop_udef`, `sop_param`)
 *   VSC  T'[c]= NOOP T[c] | MUL s'*T[c] | ADD s'+T[c] | MEAN (MUL, the row divided by max(deg,1))
 *   AOP  z[i,c] ADD += | MAX | MIN (z_arg as in fusedMM_csr_hip) T'[c]
 * VOP/ROP/VSC/AOP_UDEF and SOP_UDEF without a menu entry return ISPLIB_UNDEFINED_USER_FUNCTION,
 * flag values the header does not define ISPLIB_NO_OPT_IMPL.  k <= 1024; same write-only z,
 * beta == 0 and tie rules as fusedMM_csr_hip.  One wave per row, no atomics.
 */
enum isplib_sop_udef {
   ISPLIB_SOP_NONE = 0,
   ISPLIB_SOP_SIGMOID = 1,            /* 1 / (1 + exp(-s))                                  */
   ISPLIB_SOP_ONE_MINUS_SIGMOID = 2,  /* 1 - sigmoid(s): the gradient scale of sigmoid embedding */
   ISPLIB_SOP_TDIST = 3,              /* 1 / (1 + s): t-distribution kernel on s = |y_j - x_i|^2 */
   ISPLIB_SOP_SCALE = 4,              /* sop_param * s                                       */
   ISPLIB_SOP_EXP = 5,                /* exp(s)                                              */
   ISPLIB_SOP_LEAKY_EXP = 6           /* exp(s > 0 ? s : sop_param * s): un-normalised GAT score */
};
int fusedMM_csr_udef_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, float alpha, int64_t nnz,
                         int64_t rows, int64_t cols, const float *val /*[dev] nnz | NULL*/,
                         const int64_t *indx, const int64_t *pntrb, const int64_t *pntre,
                         const float *x /*[dev] m x ldx | NULL if unused*/, int64_t ldx,
                         const float *y /*[dev] n x ldy*/, int64_t ldy, float beta,
                         float *z /*[dev] m x ldz*/, int64_t ldz, int64_t *z_arg /*[dev] | NULL*/,
                         int sop_udef /*enum isplib_sop_udef*/, float sop_param, void *stream);

/* The same pipeline over the task plan of the SpMM (isplib_spmm_tasks_*; one wave per task, partial rows in
 * `workspace` = isplib_spmm_tasks_workspace_bytes(AOP word, n_tasks, k), folded like the SpMM's): the gathers of y
 * get the L2 affinity of the column slices.  ROP needs whole rows, so there are no column panels: choose the
 * slice count for the full width (about n*k*4 / 7 MB).  4 <= k <= 1024. */
int fusedMM_csr_udef_tasks_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz,
                               const float *val, const int64_t *indx, const int32_t *indx32 /*optional*/,
                               const int64_t *pntrb, const int64_t *pntre,
                               const float *x, int64_t ldx, int64_t n_tasks, const int32_t *task_row,
                               const int64_t *task_b, const int32_t *task_len, const int32_t *seg_off,
                               int slices, const int64_t *lane_off_host /*9, host*/,
                               const float *y, int64_t ldy, float *z, int64_t ldz, int64_t *z_arg,
                               int sop_udef, float sop_param, void *workspace, size_t workspace_bytes,
                               void *stream);

/*
 * Column-sliced SpMM: same result as fusedMM_csr_hip, faster when y does not fit an
 * XCD's 4 MiB L2 and rows are long (Reddit-like graphs).  The columns of A (= rows of
 * y) are cut into `slices` (1..ISPLIB_MAX_SLICES) equal ranges; every XCD of the MI355X
 * walks all rows of its own share of the slices, so its L2 holds 1/slices of y instead of all of
 * it; the per-slice partials (workspace) are folded in slice order by a second kernel.
 * No atomics: bitwise reproducible; max/min still resolve ties to the lowest CSR
 * position.  Requires column indices sorted within each row (torch_sparse order; the
 * CSC operands of csrc/fusedmm.cpp:285 are sorted too).
 *
 *   isplib_spmm_slices_build_hip: sliceptr[i*(slices+1)+s] = first CSR position of row
 *     i with column >= s*ceil(n/slices); once per graph, independent of k.  If
 *     unsorted_flag != NULL it receives 1 when some row is not sorted (the table must
 *     then not be used) -- the one-per-graph replacement for the per-graph caches of
 *     isplib/__init__.py:76-106.
 *   workspace: isplib_spmm_sliced_workspace_bytes(imessage, m, k, slices) bytes,
 *     256-byte aligned.
 */
size_t isplib_spmm_slices_bytes(int64_t m, int slices);
int    isplib_spmm_slices_build_hip(int64_t m, int64_t n, int64_t nnz,
                                    const int64_t *pntrb, const int64_t *pntre,
                                    const int64_t *indx, int slices,
                                    int64_t *sliceptr /*[dev] m*(slices+1)*/,
                                    int32_t *unsorted_flag /*[dev] 1 | NULL*/,
                                    void *stream);
size_t isplib_spmm_sliced_workspace_bytes(int32_t imessage, int64_t m, int64_t k, int slices);
int    fusedMM_csr_sliced_hip(int32_t imessage, int64_t m, int64_t n, int64_t k,
                              int64_t nnz, const float *val, const int64_t *indx,
                              const int64_t *pntrb, const int64_t *pntre,
                              const int64_t *sliceptr, int slices,
                              const float *y, int64_t ldy, float *z, int64_t ldz,
                              int64_t *z_arg, void *workspace,
                              size_t workspace_bytes, void *stream);

/*
 * The same computation in phases, for overlapping a collective with compute (1-D row
 * partition: the slices that fall in this rank's own shard of y need no remote data).
 * Runs the partial kernel over the slice_count slices slice_first, slice_first + 1, ... taken
 * modulo `slices` (so "every slice but mine" is one call; count may be 0) and, when `combine`
 * != 0, the fold over ALL `slices` planes afterwards.  Every
 * slice must have been covered by some phase, with the same workspace, before the
 * combining call.  `y` may differ between phases as long as y[indx[j]] addresses the right
 * row for every column of the phase's slices (e.g. a base pointer shifted onto a shard).
 */
int    fusedMM_csr_sliced_phase_hip(int32_t imessage, int64_t m, int64_t n, int64_t k,
                                    int64_t nnz, const float *val, const int64_t *indx,
                                    const int64_t *pntrb, const int64_t *pntre,
                                    const int64_t *sliceptr, int slices,
                                    int slice_first, int slice_count, int combine,
                                    const float *y, int64_t ldy, float *z, int64_t ldz,
                                    int64_t *z_arg, void *workspace,
                                    size_t workspace_bytes, void *stream);

/*
 * Task-list schedule: the most robust form of the sliced SpMM (hub rows, empty segments, short
 * rows).  The caller supplies a per-graph plan (isplib_spmm_tasks_count_hip / _fill_hip below):
 *   task t = edges [task_b[t], task_b[t] + task_len[t]) of row task_row[t], at most a few hundred
 *   edges, all in one column slice; tasks are grouped by XCD lane: lane x (blocks with
 *   blockIdx % 8 == x) runs tasks [lane_off[x], lane_off[x+1]) (lane_off: 9 HOST int64);
 *   seg_off[(s' * m) + i] = first task of row i in plan slice position s'
 *   (s' = (s % 8) * (slices / 8) + s / 8), slices*m + 1 entries.
 * Each task writes one partial row into the workspace (isplib_spmm_tasks_workspace_bytes);
 * a second kernel folds a row's partials in ascending CSR order.  Same results and conventions
 * as fusedMM_csr_hip; requires k >= 4 and n*ldy*4 <= 3.5 GiB.
 */
typedef struct isplib_task_plan_info {
   int64_t n_tasks;
   int64_t lane_off[9];     /* tasks of XCD lane x: [lane_off[x], lane_off[x+1]) */
   int32_t slices, chunk, short_row, reserved;
} isplib_task_plan_info;

/*
 * Plan builder, two calls, once per graph (independent of k):
 *   count: from the slice table (isplib_spmm_slices_build_hip) derives the number of tasks of every
 *          (slice position, row) segment -- ceil(len / chunk), rows with fewer than short_row edges
 *          kept whole on slice row % slices -- and its exclusive prefix seg_off[slices*m + 1].
 *          Fills *info (HOST) and synchronises `stream` ONCE to do so; info->lane_off cuts the task
 *          list into eight contiguous runs of equal EDGE mass (one per XCD lane).
 *   fill : writes task_row / task_b / task_len (info->n_tasks entries each).
 */
size_t isplib_spmm_tasks_plan_workspace_bytes(int64_t m, int slices);
int    isplib_spmm_tasks_count_hip(int64_t m, const int64_t *pntrb, const int64_t *pntre,
                                   const int64_t *sliceptr, int slices, int chunk, int short_row,
                                   int32_t *seg_off /*[dev] slices*m+1*/, void *workspace,
                                   size_t workspace_bytes, isplib_task_plan_info *info /*host*/,
                                   void *stream);
int    isplib_spmm_tasks_fill_hip(int64_t m, const int64_t *pntrb, const int64_t *pntre,
                                  const int64_t *sliceptr, const isplib_task_plan_info *info,
                                  const int32_t *seg_off, int32_t *task_row, int64_t *task_b,
                                  int32_t *task_len, void *stream);
size_t isplib_spmm_tasks_workspace_bytes(int32_t imessage, int64_t n_tasks, int64_t k);
/* indx32 (optional, may be NULL): the column ids of indx packed to 32 bits, once per graph, by
 * isplib_pack_indices_hip.  The task kernels then stream 4 instead of 8 bytes per stored entry -- the index
 * stream is the one operand that always comes from HBM (Reddit: 917 MB per pass; -12 % time at k=32, -4 % at
 * k=128).  indx stays the reference's int64 array and is what NULL falls back to. */
int    isplib_pack_indices_hip(int64_t nnz, const int64_t *indx, int32_t *indx32, void *stream);
int    fusedMM_csr_tasks_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz,
                             const float *val, const int64_t *indx, const int32_t *indx32,
                             const int64_t *pntrb, const int64_t *pntre,
                             int64_t n_tasks, const int32_t *task_row,
                             const int64_t *task_b, const int32_t *task_len,
                             const int32_t *seg_off, int slices,
                             const int64_t *lane_off_host /*9, host*/,
                             const float *y, int64_t ldy, float *z, int64_t ldz,
                             int64_t *z_arg, void *workspace, size_t workspace_bytes,
                             void *stream);

/*
 * The same with an epilogue fused into the fold (sum / mean only): the step right after the path in a GNN
 * layer -- GCN's D^-1/2 (A + I) D^-1/2 X without materialising edge weights (y = D^-1/2 X, self = y,
 * row_scale = D^-1/2), bias and ReLU (callers: tests/dist/gcn/pyg-sparse.py:61-62, normalize=True).
 *     out[i,c] = act( row_scale[i] * (reduce[i,c] + self[i,c]) + bias[c] )        every member optional
 */
typedef struct isplib_epilogue {
   const float *row_scale;   /* [dev] m, or NULL */
   const float *self;        /* [dev] m x ld_self, or NULL */
   int64_t      ld_self;
   const float *bias;        /* [dev] k, or NULL */
   int          relu;        /* nonzero: max(., 0) */
} isplib_epilogue;
int    fusedMM_csr_tasks_epilogue_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz,
                                      const float *val, const int64_t *indx, const int32_t *indx32,
                                      const int64_t *pntrb, const int64_t *pntre,
                                      int64_t n_tasks, const int32_t *task_row,
                                      const int64_t *task_b, const int32_t *task_len,
                                      const int32_t *seg_off, int slices,
                                      const int64_t *lane_off_host /*9, host*/,
                                      const float *y, int64_t ldy, float *z, int64_t ldz,
                                      void *workspace, size_t workspace_bytes,
                                      const isplib_epilogue *epilogue /*host, may be NULL*/,
                                      void *stream);

/*
 * The dense passes either side of the fused GCN aggregation (epilogue above with self = y, row_scale = D^-1/2), one launch each,
 * no atomics (the column sums are per-block partials folded in block order: bitwise reproducible):
 *   isplib_row_scale_hip            y[i,c] = scale[i] * x[i,c], c < k; columns k..ldy-1 of y are written 0 -- y = D^-1/2 X at
 *                                   whatever pitch the gather wants (a 33..47-column operand at 48 floats touches 2 lines
 *                                   per row instead of 2.25)
 *   isplib_masked_scale_colsum_hip  the backward's prologue, reading dz (and out) once:
 *                                   g[i,c] = out ? (out[i,c] > 0 ? dz[i,c] : 0) : dz[i,c]      (ReLU mask; out may be NULL)
 *                                   gy[i,c] = g[i,c] * (scale ? scale[i] : 1)                  (gy may be NULL; pitch as above)
 *                                   grad_bias[c] = sum_i g[i,c]                                (may be NULL; else workspace of
 *                                   isplib_masked_scale_colsum_workspace_bytes(n, k) bytes, 256-byte aligned)
 */
int    isplib_row_scale_hip(int64_t n, int64_t k, const float *x /*[dev] n x ldx*/, int64_t ldx, const float *scale /*[dev] n*/,
                            float *y /*[dev] n x ldy*/, int64_t ldy, void *stream);
size_t isplib_masked_scale_colsum_workspace_bytes(int64_t n, int64_t k);
int    isplib_masked_scale_colsum_hip(int64_t n, int64_t k, const float *dz, int64_t lddz, const float *out /*NULL: no mask*/,
                                      int64_t ldo, const float *scale /*[dev] n | NULL*/, float *gy /*[dev] n x ldgy | NULL*/,
                                      int64_t ldgy, float *grad_bias /*[dev] k | NULL*/, void *workspace, size_t workspace_bytes,
                                      void *stream);

/*
 * Stream schedule (sum / mean) -- the default on graphs with work for the whole chip.  A launch ("generation") holds only
 * waves that are resident together; every wave owns a fixed set of (virtual) rows whose running sums live in LDS, and all
 * waves walk the column slices of their own rows in step, so an XCD's L2 serves one or two slices at a time.  The plan owns
 * a COPY of the edges in the order the waves walk them, so a wave never sees a row or slice boundary.  Each of the `streams` (= 64 / lanes per row slot)
 * slots of a wave owns rows_per_wave / streams rows and a stream of 4-byte words, (local row << 24) | column, that
 * lists those rows' edges slice by slice; step i of a wave gathers word i of each of its streams in ONE full 1-KiB
 * buffer load, and every lane adds what arrives to the running sum of the row its slot is on (registers), which
 * moves to and from the slot's LDS row when the stream changes rows.  16 or 32 gathers are in flight per wave at all
 * times; there are no per-segment latency chains, no masked tails, no cross-lane reduction, no partial rows, no fold.
 * The slots of a wave own disjoint rows and a wave's LDS operations execute in order: every sum is formed in one
 * fixed order (bitwise reproducible).  The panel width is 256 / streams columns (streams = 4: 64 columns); k is
 * swept in such panels.  Weights are part of the plan (`vals`, in stream order; NULL = unit weights): a caller whose
 * weights change refreshes them through the plan's permutation.  A launch holds few, deep waves (2 or 3 per SIMD):
 * isplib_spmm_stream_geometry reports the rows per wave the kernel of a slot width is built for and the waves the
 * device holds at once (what waves_per_gen should be: persistent waves only stay on the same slices -- and the slices
 * in the L2 -- when they all start together).
 * Requirements: k >= 4 (any k: a last vector that would reach past column k is shifted back to end there; rows need
 * only 4-byte alignment), n < 2^24 and ldy < 2^22 (24-bit address arithmetic per edge), n*ldy*4 <= 3.5 GiB.
 */
typedef struct isplib_stream_plan {
   int64_t rows, cols;              /* m, n of the graph the plan was built for */
   int32_t slices, gens, waves_per_gen, rows_per_wave, streams /* 2, 4 or 8 */, chunk /* rows over this many entries were dealt to virtual rows */;
   int64_t n_steps, n_parts, n_hub;
   const int32_t *words;            /* [dev] n_steps*streams: (local row << 24) | column; padding = (own row << 24) | n */
   const float   *vals;             /* [dev] n_steps*streams weights in the same order, or NULL */
   const int64_t *wave_step_off;    /* [dev] gens*waves_per_gen + 1: first step of a wave */
   const int32_t *wave_row;         /* [dev] gens*waves_per_gen*rows_per_wave: row of the local row, -1 = unused */
   const int32_t *wave_part;        /* [dev] same shape: -1 = whole row, else partial row id */
   const int32_t *hub_row;          /* [dev] n_hub */
   const int32_t *hub_off;          /* [dev] n_hub + 1 */
   const int32_t *perm;             /* [dev] n_steps*streams: CSR position of every word, -1 = padding (for re-gathering
                                       weights; the arg candidates of max / min); NULL in plans that do not carry it -- the
                                       sum / mean kernel never reads it */
} isplib_stream_plan;
/* Native plan builder (the same construction as isplib_amd/plan.py, on the device with rocPRIM sorts): allocates the
 * plan's device arrays -- release them with isplib_stream_plan_free -- and synchronises `stream` (twice: the sizes of
 * the plan are data dependent).  streams / slices / chunk: from isplib_suggest_stream; waves_per_gen <= 0: what
 * isplib_spmm_stream_geometry reports.  val may be NULL (unit weights).  isplib_stream_plan_set_values_hip re-gathers
 * the weights (NULL: back to unit weights) when they change. */
int  isplib_stream_plan_build_hip(int64_t m, int64_t n, int64_t nnz, const int64_t *rowptr, const int64_t *col,
                                  const float *val, int streams, int slices, int chunk, int waves_per_gen,
                                  isplib_stream_plan *out /*host*/, void *stream);
int  isplib_stream_plan_set_values_hip(isplib_stream_plan *plan, const float *val /*[dev] nnz | NULL*/, void *stream);
void isplib_stream_plan_free(isplib_stream_plan *plan);
int    isplib_spmm_stream_geometry(int streams, int *rows_per_wave /*out*/, int *waves_resident /*out*/);
/* the measured rule: nonzero when the stream schedule is expected to beat the task list for an m x n, nnz-entry SpMM
 * over k columns (sum / mean), with the plan parameters to build it with (streams, column slices, hub-row chunk) */
int    isplib_suggest_stream(int64_t m, int64_t n, int64_t nnz, int64_t k, int *streams, int *slices, int *chunk);
/* the same rule told whether the plan will carry edge weights (a weighted launch reads the plan's weight stream once per
 * column panel: whole multiples of 128 columns then run on 128-column slots, streams = 2); isplib_suggest_stream is this
 * with weighted = 0 */
int    isplib_suggest_stream_weighted(int64_t m, int64_t n, int64_t nnz, int64_t k, int weighted, int *streams, int *slices, int *chunk);
/* max / min on the stream schedule, for graphs whose rows are column-sorted (ascending, duplicates allowed).  The
 * running sum becomes the best value so far and the WORD INDEX at which it was met; a second LDS plane keeps those
 * indices beside the values, and only a strictly better candidate replaces the one held.  The plan walks a row's edges
 * slice by slice and, inside a slice, in CSR order -- for a column-sorted row that IS its CSR order, so the candidate
 * that stays among equals is the one at the lowest CSR position, the reference's tie rule (csrc/fusedmm.cpp:170-178 +
 * SURVEY.md row K3).  No position travels with the gathers: only the M x K winners are translated to CSR positions,
 * through the plan's `perm` (required), when a row is written out; hub rows cut into virtual rows carry (value, CSR
 * position) pairs into their fold.  Slots of 64 columns (streams = 4) or, for k <= 32, of 32 columns (streams = 8), with
 * half the rows per wave of the sum kernel, so max / min plans are built for isplib_spmm_stream_minmax_geometry -- isplib_stream_plan_build_minmax_hip
 * does that and returns ISPLIB_FAIL for a graph with an unsorted row (use the task list) -- and are not interchangeable
 * with sum plans.  nnz < 2^31.  z_arg (may be NULL): [m][ldz] int64 CSR positions, nnz = empty row.  With z_arg = NULL the
 * launch is a values-only one: no position is tracked at all (a quarter of the loop's vector instructions), same plan,
 * same values bit for bit. */
int    isplib_spmm_stream_minmax_geometry(int streams /* 4 | 8 */, int *rows_per_wave /*out*/, int *waves_resident /*out*/);
int    isplib_suggest_stream_minmax(int64_t m, int64_t n, int64_t nnz, int64_t k, int *streams, int *slices, int *chunk);
int    isplib_stream_plan_build_minmax_hip(int64_t m, int64_t n, int64_t nnz, const int64_t *rowptr, const int64_t *col,
                                           const float *val, int streams, int slices, int chunk, int waves_per_gen,
                                           isplib_stream_plan *out /*host*/, void *stream);
size_t isplib_spmm_stream_minmax_workspace_bytes(const isplib_stream_plan *plan);
int    fusedMM_csr_stream_minmax_hip(int32_t imessage /* ISPLIB_MSG_SPMM_MAX | _MIN */, int64_t m, int64_t n, int64_t k,
                                     int64_t nnz, const int64_t *pntrb, const int64_t *pntre,
                                     const isplib_stream_plan *plan /*host*/,
                                     const float *y, int64_t ldy, float *z, int64_t ldz, int64_t *z_arg,
                                     void *workspace, size_t workspace_bytes, void *stream);
size_t isplib_spmm_stream_workspace_bytes(const isplib_stream_plan *plan);
int    fusedMM_csr_stream_hip(int32_t imessage /* ISPLIB_MSG_SPMM_SUM | _MEAN */, int64_t m, int64_t n, int64_t k,
                              int64_t nnz, const int64_t *pntrb, const int64_t *pntre,
                              const isplib_stream_plan *plan /*host*/,
                              const float *y, int64_t ldy, float *z, int64_t ldz,
                              void *workspace, size_t workspace_bytes,
                              const isplib_epilogue *epilogue /*host, may be NULL*/, void *stream);

/*
 * The two SDDMM-fused words of the generic pipeline that graph embedding runs on -- COPY_RHS|DOT|UDEF|MUL|ADD (sigmoid embedding,
 * attention scores: z_i = sum_j f(<x_i, y_j>) y_j) and SUBR|NORMR|UDEF|MUL|ADD (t-distribution: z_i = sum_j f(|y_j - x_i|^2)
 * (y_j - x_i)), f from enum isplib_sop_udef -- on the stream schedule's front end: rows of x and of z resident in LDS, a slot as
 * wide as the row (k <= 128: the reduce stage needs the whole row), four steps' dot products summed by one transposed
 * butterfly.  Reddit shape, K=128: the task-list form (fusedMM_csr_udef_tasks_hip) takes 6.3 ms.  Same result as
 * fusedMM_csr_udef_hip up to the order of the sums (bitwise reproducible from run to run); edge weights are not read (SOP is
 * the user function).  Plans: isplib_stream_plan_build_fusedmm_hip (its own geometry: isplib_fusedmm_stream_geometry; streams
 * 2 / 4 / 8 = slots of 128 / 64 / 32 columns); isplib_suggest_fusedmm_stream is the rule (0: stay on the task list);
 * workspace: isplib_spmm_stream_workspace_bytes(plan).  k a multiple of 4, ldx and ldz multiples of 4, x and z 16-byte
 * aligned, n < 2^24, n*ldy*4 <= 3.5 GiB.  Other words return ISPLIB_NO_OPT_IMPL.
 */
int    isplib_fusedmm_stream_geometry(int streams, int *rows_per_wave /*out*/, int *waves_resident /*out*/);
int    isplib_suggest_fusedmm_stream(int32_t imessage, int64_t m, int64_t n, int64_t nnz, int64_t k, int *streams, int *slices, int *chunk);
int    isplib_stream_plan_build_fusedmm_hip(int64_t m, int64_t n, int64_t nnz, const int64_t *rowptr, const int64_t *col,
                                            int streams, int slices, int chunk, int waves_per_gen,
                                            isplib_stream_plan *out /*host*/, void *stream);
int    fusedMM_csr_udef_stream_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz,
                                   const int64_t *pntrb, const int64_t *pntre, const isplib_stream_plan *plan /*host*/,
                                   const float *x /*[dev] m x ldx*/, int64_t ldx, const float *y /*[dev] n x ldy*/, int64_t ldy,
                                   float *z /*[dev] m x ldz*/, int64_t ldz, int sop_udef /*enum isplib_sop_udef*/, float sop_param,
                                   void *workspace, size_t workspace_bytes, void *stream);

/* The plain row-per-wave kernel of fusedMM_csr_hip with the rows taken in a caller-given ORDER (`row_order`: position ->
 * row, a permutation of [0, m), [dev] int32; NULL = fusedMM_csr_hip).  For operands larger than the Infinity Cache (the
 * ogbn-products shape: y = 2.5 GB) no schedule of this library reuses a gathered row -- unless rows that share neighbours
 * are worked on at the same time on the same XCD: the workgroups of an XCD walk a contiguous range of positions, so a
 * community-grouped order (isplib_amd/reorder.py: label propagation, once per graph) turns the gathers of a community's
 * rows into hits of that XCD's L2.  Nothing is moved: y is gathered and z written where they are, every row is computed
 * by the same code in the same edge order -- the result is bit for bit the same for any order.  (row_order == NULL on an
 * operand beyond the Infinity Cache, n * ldy * 4 > 256 MiB, runs k > 128 in 128-column panels -- two rows per gather: 9 % faster
 * on the ogbn-products shape, the sums of a row associated over two slots instead of one; isplib_hip_tune(0, 64) keeps the
 * index-order launch on the one-pass form a given order runs.)
 * No reference counterpart (the reference's CPU kernel walks rows in index order, csrc/fusedmm.cpp:198). */
int    fusedMM_csr_ordered_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz, const float *val,
                               const int64_t *indx, const int64_t *pntrb, const int64_t *pntre,
                               const int32_t *row_order /*[dev] m | NULL*/, const float *y, int64_t ldy, float *z,
                               int64_t ldz, int64_t *z_arg, void *stream);

/* The order for fusedMM_csr_ordered_hip, found on the device (square graphs): synchronous label propagation -- every
 * row takes the label most of its stored entries' columns carry, ties by a per-round hash of the label, at most `rounds`
 * rounds (8 is plenty; it stops when fewer than 1 % of the rows change) -- then the rows sorted by label, index order
 * inside a label.  order: [dev] m int32, position -> row.  labels (may be NULL): [dev] m int32, the label of every row.
 * One radix sort of nnz 64-bit keys per round: ~0.2 s for the ogbn-products shape, once per graph.  The result only
 * affects speed.  isplib_order_locality_hip reports what an order found: the share of stored entries whose column lies
 * within `window` positions of its row (order NULL: index order) -- a community order of a graph WITH structure lifts it
 * from ~0 to the share of edges inside communities; one that lifts nothing is not worth passing on. */
size_t isplib_community_order_workspace_bytes(int64_t m, int64_t nnz);
int    isplib_community_order_hip(int64_t m, int64_t nnz, const int64_t *rowptr, const int64_t *col, int rounds, int seed,
                                  int32_t *order, int32_t *labels /*may be NULL*/, int *rounds_run /*host, may be NULL*/,
                                  void *workspace, size_t workspace_bytes, void *stream);
int    isplib_order_locality_hip(int64_t m, int64_t nnz, const int64_t *rowptr, const int64_t *col, const int32_t *order,
                                 int64_t window, double *share /*host*/, void *workspace /*4 m + 512 bytes*/,
                                 size_t workspace_bytes, void *stream);

/* Tuning knobs of the shipped kernels, for experiments (process-wide, not thread-safe; every setting gives the same results):
 *   0  lanes per row slot of the row kernels (0 = by k)      1  0: 64-bit addressing instead of buffer descriptors
 *   2  consecutive tasks per wave of the task kernel (0=auto) 4  column-panel width of the task entries, sum / mean (64)
 *   5  the same for max / min (64)                            8  per-slice footprint of whole rows (KiB) up to which a
 *      returns ISPLIB_FAIL for an unknown key                    task plan runs in one pass (9216; 0 = always panels)
 * (the knobs of the forms that were measured slower live with them: include/isplib_hip_experimental.h) */
int isplib_hip_tune(int key, int value);

/* Warm-up hook with the reference's name; launches one empty kernel. */
void performDummySpMM_hip(int64_t flag, void *stream);

/*
 * Fused backward of SpMM-max/min (one pass, float atomics):
 *   for every (i,c) with a = arg[i,c] != nnz, j = indx[a]:
 *     grad_mat[j,c]  += (val ? val[a] : 1) * grad_out[i,c]      (if grad_mat)
 *     grad_val[a]    += mat[j,c] * grad_out[i,c]                (if grad_val)
 * grad_mat (n x k) and grad_val (nnz) are zero-filled by this call first.
 * All dense operands contiguous with leading dimension k.
 */
int isplib_spmm_minmax_bw_hip(int64_t m, int64_t n, int64_t k, int64_t nnz,
                              const int64_t *indx, const float *val,
                              const float *mat, const int64_t *arg,
                              const float *grad_out, float *grad_mat,
                              float *grad_val, void *stream);

/*
 * The same gradients WITHOUT atomics: bitwise reproducible from launch to launch, like the reference's CPU
 * scatter_add_ (csrc/fusedmm.cpp:421-446).  grad_mat: the (destination, value) pairs of all (row, feature)
 * elements are sorted by destination with a stable radix sort and every run is added up in ascending row order by one
 * thread; grad_val: every destination of row i is an entry of row i, so one thread per row updates them in feature
 * order.  workspace: isplib_spmm_minmax_bw_workspace_bytes(m, n, k) bytes, 256-byte aligned (0 = not served:
 * n*k + 1 or m*k beyond 32-bit keys -- use the atomic form above).
 */
size_t isplib_spmm_minmax_bw_workspace_bytes(int64_t m, int64_t n, int64_t k);
int isplib_spmm_minmax_bw_det_hip(int64_t m, int64_t n, int64_t k, int64_t nnz,
                                  const int64_t *indx, const float *val,
                                  const float *mat, const int64_t *arg,
                                  const float *grad_out, float *grad_mat,
                                  float *grad_val, void *workspace, size_t workspace_bytes, void *stream);

/*
 * The local half of the max / min backward under the 1-D row partition (isplib_amd/dist.py; no reference counterpart: the
 * reference has no distributed path).  After the exchange every rank holds, in global row order, the winners' columns
 * dest[i,c] (int32 global row of the dense operand, < 0 = no winner) and the weighted gradients gval[i,c] = val[arg] *
 * grad_out[i,c] of ALL rows; it keeps the destinations that are its own rows [lo, lo + n):
 *     grad_mat[d - lo, c] = sum over i, ascending, of gval[i,c] where dest[i,c] == d
 * Same stable sort + ordered run sums as isplib_spmm_minmax_bw_det_hip (no atomics, bitwise reproducible), same
 * workspace: isplib_spmm_minmax_bw_workspace_bytes(m, n, k).  grad_mat (n x k, contiguous) is zero-filled first.
 */
int isplib_scatter_rows_det_hip(int64_t m, int64_t n, int64_t k, int64_t lo,
                                const int32_t *dest /*[dev] m x k*/, const float *gval /*[dev] m x k*/,
                                float *grad_mat /*[dev] n x k*/, void *workspace, size_t workspace_bytes, void *stream);

/*
 * SDDMM-style value gradient of SpMM-sum / SpMM-mean:
 *   dval[j] = < y[indx[j], :], g[i, :] > * (mean ? 1/max(deg_i,1) : 1),  j in row i
 */
int isplib_sddmm_csr_hip(int64_t m, int64_t k, const int64_t *indx,
                         const int64_t *pntrb, const int64_t *pntre,
                         const float *y, int64_t ldy, const float *g,
                         int64_t ldg, int mean, float *dval, void *stream);
/* The same over the task plan of the SpMM (one wave per task, XCD-lane grouping -> the L2 affinity of
 * the task-list SpMM; dval is per edge, so no workspace and no combine).  4 <= k <= 1024. */
int isplib_sddmm_csr_tasks_hip(int64_t m, int64_t n, int64_t k, const int64_t *indx,
                               const int32_t *indx32 /*optional, as in fusedMM_csr_tasks_hip*/,
                               const int64_t *pntrb, const int64_t *pntre,
                               int64_t n_tasks, const int32_t *task_row,
                               const int64_t *task_b, const int32_t *task_len,
                               const int64_t *lane_off_host /*9, host*/,
                               const float *y, int64_t ldy, const float *g, int64_t ldg,
                               int mean, float *dval, void *stream);

/*
 * Graph preparation on the device (the step immediately before the path).
 *   isplib_csr_row_ids_hip : row[j] = i for j in [rowptr[i], rowptr[i+1])
 *   isplib_csr2csc_hip     : stable counting sort of the CSR entries by column:
 *        colptr[n+1], csr2csc[nnz] (CSC position -> CSR position),
 *        row_t[nnz] = row[csr2csc], and, when val_t != NULL,
 *        val_t[nnz] = (val ? val[csr2csc] : 1) / (mean_scale ? max(deg(row),1) : 1)
 *     i.e. exactly the cached operands of isplib/__init__.py:79-80 (sum) and the
 *     intended form of :86-99 (mean; csrc/fusedmm.cpp:357-364).
 *     Deterministic (no atomics decide placement).  Workspace: see query.
 */
int    isplib_csr_row_ids_hip(int64_t m, int64_t nnz, const int64_t *rowptr,
                              int64_t *row, void *stream);
size_t isplib_csr2csc_workspace_bytes(int64_t m, int64_t n, int64_t nnz);
int    isplib_csr2csc_hip(int64_t m, int64_t n, int64_t nnz,
                          const int64_t *rowptr, const int64_t *col,
                          const float *val, int mean_scale, int64_t *colptr,
                          int64_t *csr2csc, int64_t *row_t, float *val_t,
                          void *workspace, size_t workspace_bytes, void *stream);

/*
 * isplib_graph: a handle that owns what a graph needs for the fast path -- the per-graph operands the
 * stateless entries above leave to the caller (slice tables, task plans per slice count, the packed column
 * ids, the CSC operands of the backward, one grow-only workspace), built lazily on the device.  It is what
 * the reference's C++ layer (csrc/fusedmm.cpp:113-203) would hold per graph in place of the pointer-keyed
 * dicts of isplib/__init__.py:35-40.
 *   isplib_graph_create   borrows rowptr[m+1] / col[nnz] / val[nnz]|NULL (device; must outlive the handle and
 *                         must not change without isplib_graph_set_values: val is examined once, and a vector
 *                         of exact 1.0f -- what isplib/__init__.py:51-57 materialises for an unweighted graph -- is
 *                         treated as NULL)
 *   isplib_graph_spmm     z = A (x) y for one of the four SpMM words; schedule by the measured rules: the stream
 *                         schedule where isplib_suggest_stream (sum / mean) or isplib_suggest_stream_minmax (max / min,
 *                         column-sorted rows) accepts the call, else isplib_suggest_slices
 *                         (or isplib_graph_set_slices: -1 rules, 0 plain kernel, 1..4096 task list)
 *   isplib_graph_spmm_backward   dx = A^T dy (mean != 0: with weights val/max(deg,1), the mean forward's
 *                         backward, csrc/fusedmm.cpp:375); the CSC operands are built on first use
 *   isplib_suggest_slices the measured rule (0 = plain row-per-wave kernel)
 * The first call that needs a new plan (or the transpose) allocates device memory and synchronises the
 * stream once; every later call is asynchronous and allocation-free.  Not thread-safe (serialise the calls of
 * one handle), but usable on several streams: each stream it is used on gets its own workspace.
 */
typedef struct isplib_graph isplib_graph;
int  isplib_suggest_slices(int64_t m, int64_t n, int64_t nnz, int64_t k, int minmax /*nonzero for max / min*/);
int  isplib_graph_create(int64_t m, int64_t n, int64_t nnz, const int64_t *rowptr, const int64_t *col,
                         const float *val, isplib_graph **out);
int  isplib_graph_set_slices(isplib_graph *g, int slices);
/* The order in which the plain kernel takes the rows of A (order: [dev] m int32, position -> row) and of A^T (order_t:
 * [dev] n int32) -- fusedMM_csr_ordered_hip; borrowed, NULL = index order.  Without this call the handle looks for a
 * community order itself (isplib_community_order_hip, once, ~0.2 s for the ogbn-products shape) the first time a square
 * graph reaches the plain kernel with a dense operand larger than the Infinity Cache (n k 4 > 256 MiB), and keeps it
 * only if it found structure.  Speed only: any order gives the same bits. */
int  isplib_graph_set_row_order(isplib_graph *g, const int32_t *order, const int32_t *order_t);
/* New weights for the same structure (val: [dev] nnz | NULL, borrowed like create's; also to be called when the
 * CONTENTS of the array given before have changed): plans, packed column ids and the CSC structure stay; the
 * unit-weight test, the stream plans' copies of the weights and the transposed weights of the backward are refreshed
 * by the next call that needs them, on that call's stream.  Never synchronises, never frees. */
int  isplib_graph_set_values(isplib_graph *g, const float *val);
int  isplib_graph_spmm(isplib_graph *g, int32_t imessage, int64_t k, const float *y, int64_t ldy,
                       float *z, int64_t ldz, int64_t *z_arg, void *stream);
int  isplib_graph_spmm_backward(isplib_graph *g, int mean, int64_t k, const float *dy, int64_t lddy,
                                float *dx, int64_t lddx, void *stream);
/* dA[e] = <y[col[e],:], g[row(e),:]> (mean != 0: / max(deg,1)); its plan is sized for whole rows of y (the dot
 * product cannot run in column panels): isplib_suggest_slices_whole_rows */
int  isplib_suggest_slices_whole_rows(int64_t m, int64_t n, int64_t nnz, int64_t k);
int  isplib_graph_sddmm(isplib_graph *g, int mean, int64_t k, const float *y, int64_t ldy,
                        const float *gmat, int64_t ldg, float *dval, void *stream);
void isplib_graph_destroy(isplib_graph *g);

#ifdef __cplusplus
}
#endif
#endif /* ISPLIB_HIP_H */
