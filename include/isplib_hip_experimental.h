/*
 * isplib_hip_experimental.h -- entry points of libisplib_hip_exp.so: forms of the SpMM / SDDMM schedules that were built,
 * are bit-exact and tested, and were MEASURED SLOWER than what libisplib_hip.so runs by default.
 * Nothing here is needed by a binding of the reference's path (INTEGRATION.md lists what is); they stay buildable so
 * that the measurements can be repeated.  Same conventions as include/isplib_hip.h (device pointers, a stream, status
 * codes, isplib_hip_last_error of the default library); the library links against libisplib_hip.so.
 */
#ifndef ISPLIB_HIP_EXPERIMENTAL_H
#define ISPLIB_HIP_EXPERIMENTAL_H

#include "isplib_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Sweep schedule: the same SpMM with the running rows kept in LDS instead of per-task partial rows in HBM.
 * A launch ("generation") holds only waves that are resident together; every wave owns up to rows_per_wave
 * (virtual) rows, whose sums live in LDS, and walks the column slices 0, 1, 2, ... of those rows; equal edge
 * mass per wave keeps all waves of the chip on the same one or two slices at any moment, so a slice can be as
 * small as an XCD's L2 without the partial-row traffic that short tasks cost the task-list schedule.  z is written
 * once, there is no fold kernel; rows longer than the plan's chunk are cut into virtual rows owned by different
 * waves, and only their partial rows (n_parts of them) pass through the workspace.  No atomics: a row's tasks are
 * folded by one wave in ascending CSR order, results are bitwise reproducible and max/min ties go to the lowest
 * CSR position.  Requires column-sorted rows, k % 4 == 0, ldy % 4 == 0, ldz % 4 == 0, y and z 16-byte aligned,
 * n*ldy*4 <= 3.5 GiB (otherwise use fusedMM_csr_tasks_hip / fusedMM_csr_hip).
 *   plan (host struct, device arrays; isplib_amd/plan.py builds it on the device):
 *     wave w of generation g = global wave g*waves_per_gen + w;
 *     wave_row [wave][slot]  row of the slot, -1 = unused      wave_part[wave][slot]  -1 = whole row, else partial row id
 *     tasks of a wave: [wave_task_off[wave], wave_task_off[wave+1]), sorted by (slice, slot);
 *     task_b first CSR position, task_meta = (slot << 24) | edges (edges < 2^24)
 *     hub_row[h] = a row cut into chunks; its partial rows are [hub_off[h], hub_off[h+1]), in CSR order.
 *   isplib_spmm_sweep_resident_waves: waves of this kernel family the device holds at once at width k -- what
 *     waves_per_gen should not exceed (a larger generation is still correct, only slower).
 *   epilogue: as fusedMM_csr_tasks_epilogue_hip (sum / mean only), may be NULL.
 */
typedef struct isplib_sweep_plan {
   int64_t rows;                    /* m of the graph the plan was built for */
   int32_t slices, gens, waves_per_gen, rows_per_wave /* 8, 16 or 32 (max / min: <= 16) */;
   int64_t n_tasks, n_parts, n_hub;
   const int32_t *wave_row;         /* [dev] gens*waves_per_gen*rows_per_wave */
   const int32_t *wave_part;        /* [dev] same shape */
   const int64_t *wave_task_off;    /* [dev] gens*waves_per_gen + 1 */
   const int64_t *task_b;           /* [dev] n_tasks */
   const int32_t *task_meta;        /* [dev] n_tasks */
   const int32_t *hub_row;          /* [dev] n_hub */
   const int32_t *hub_off;          /* [dev] n_hub + 1 */
} isplib_sweep_plan;
int    isplib_spmm_sweep_resident_waves(int32_t imessage, int64_t k, int rows_per_wave);
size_t isplib_spmm_sweep_workspace_bytes(int32_t imessage, const isplib_sweep_plan *plan, int64_t k);
int    fusedMM_csr_sweep_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz,
                             const float *val, const int64_t *indx, const int32_t *indx32 /*optional*/,
                             const int64_t *pntrb, const int64_t *pntre,
                             const isplib_sweep_plan *plan /*host*/,
                             const float *y, int64_t ldy, float *z, int64_t ldz, int64_t *z_arg,
                             void *workspace, size_t workspace_bytes,
                             const isplib_epilogue *epilogue /*host, may be NULL*/, void *stream);

/* SDDMM over a stream plan (the dA of sum / mean: dval[e] = <y[col[e], :], g[row(e), :]>, / max(deg, 1) for mean -- the
 * call the reference leaves commented out, csrc/fusedmm.cpp:270,351).  Same edges and same gathers of y as the SpMM, so
 * the same plan (a sum / mean plan of isplib_spmm_stream_geometry WITH its perm array; weights in the plan are ignored)
 * and the same front end; the wave's rows of g sit in LDS where the SpMM keeps its accumulators, every step yields one
 * dot product per slot, and a batch's results are stored through `perm` once.  k is swept in panels of 256 / streams
 * columns; later panels add to what earlier ones stored.  Plain stores by the one owner of every edge: bitwise
 * reproducible.  k >= 4, n < 2^24, ldy < 2^22, n*ldy*4 <= 3.5 GiB, nnz < 2^31. */
int    isplib_sddmm_stream_hip(int64_t m, int64_t n, int64_t k, int64_t nnz, const int64_t *pntrb, const int64_t *pntre,
                               const isplib_stream_plan *plan /*host*/, const float *y, int64_t ldy,
                               const float *g /*[dev] m x ldg*/, int64_t ldg, int mean, float *dval /*[dev] nnz*/, void *stream);

/*
 * Hybrid form of the stream schedule (sum / mean, unit weights): north_star's "dense feature tiles staged through LDS".
 * What bounds fusedMM_csr_stream_hip is the CU's address pipeline -- every gathered row of y crosses it whether the L2
 * hits or not -- so the only gathered bytes that cost less are bytes that do not cross it.  The plan picks, per column
 * slice, the table_rows - 1 most-referenced rows of y (in-degree; the same table for every workgroup); a workgroup of
 * 8 waves (one per CU: 128 KB of row accumulators + a 32 KB table) stages the table of a slice ONCE by LDS-DMA
 * (buffer_load ... lds) and serves the edges that point into it -- the "hot" words, (local row << 24) | table row --
 * with ds_read_b128, while the remaining "cold" edges run through the gather pipeline exactly as in the stream form
 * (`cold` is a complete stream plan of those edges; its `slices` is also the number of phases the hot chunks are
 * interleaved at).  Two workgroup barriers per slice order table reuse; nothing drains the gather pipeline.  A row's
 * contributions are added in the program order of the one wave that owns it: bitwise reproducible, no atomics.
 * Same operand requirements as fusedMM_csr_stream_hip; cold.vals must be NULL (weighted graphs stay on the stream
 * form).  isplib_spmm_hybrid_geometry reports, per slot width (streams 4: 64-column panels, 8: 32-column panels), what
 * a plan must be built for: rows per wave, resident waves (waves_per_gen; a multiple of 8), table rows and the most hot
 * steps one wave may have in one slice (edges beyond it stay cold).  No reference counterpart: supersedes
 * gpu/kernels/spmm.cuh:3-23 (thread per row, straight to global memory).
 */
typedef struct isplib_hybrid_plan {
   isplib_stream_plan cold;         /* the edges that stay on the gather path; cold.slices = phases */
   int32_t table_rows;              /* rows of the per-slice LDS table; the last one is all zero (padding words point at it) */
   int32_t hot_cap;                 /* most hot steps of one wave in one slice */
   int64_t n_hot_steps;
   const int32_t *hot_rows;         /* [dev] slices*table_rows: column id of every table row; n = unused / the zero row */
   const int32_t *hot_words;        /* [dev] n_hot_steps*streams: (local row << 24) | table row, chunk by chunk */
   const int64_t *hot_step_off;     /* [dev] gens*waves_per_gen*slices + 1: first hot step of a (wave, slice) chunk */
   const int32_t *hot_perm;         /* [dev] n_hot_steps*streams: CSR position of every hot word, -1 = padding; may be NULL */
} isplib_hybrid_plan;
int    isplib_spmm_hybrid_geometry(int streams /* 4 | 8 */, int *rows_per_wave, int *waves_resident, int *table_rows, int *hot_cap);
size_t isplib_spmm_hybrid_workspace_bytes(const isplib_hybrid_plan *plan);
int    fusedMM_csr_hybrid_hip(int32_t imessage /* ISPLIB_MSG_SPMM_SUM | _MEAN */, int64_t m, int64_t n, int64_t k,
                              int64_t nnz, const int64_t *pntrb, const int64_t *pntre,
                              const isplib_hybrid_plan *plan /*host*/,
                              const float *y, int64_t ldy, float *z, int64_t ldz,
                              void *workspace, size_t workspace_bytes,
                              const isplib_epilogue *epilogue /*host, may be NULL*/, void *stream);

/* Knobs of the forms above and of the task-list SDDMM's column panels (process-wide, not thread-safe; same results):
 *   9  column-panel width of the sweep schedule: 32, 64 (default) or 128
 *  12  column-panel width of isplib_sddmm_csr_tasks_hip (0 = whole rows, the default; 64-column panels measured 3.89 ms
 *      against 3.52 whole-row at K=128): forwarded to the default library
 * (round 4's knobs 10 -- all generations of a stream pass in one launch -- and 11 -- isplib_graph_sddmm on the stream
 * plan -- were measured slower and are gone with their code paths) */
int isplib_hip_tune_experimental(int key, int value);

#ifdef __cplusplus
}
#endif
#endif /* ISPLIB_HIP_EXPERIMENTAL_H */
