/*
 * parasail_amd.h -- C ABI of libparasail_amd.so, the MI355X (gfx950) engine behind the
 * parasail-rs `Aligner::align()` hot path.
 *
 * Two groups of entry points:
 *
 *  (1) The `parasail_*` symbols that parasail-rs binds through libparasail-sys
 *      (reference file:line given per declaration).  Same names, same argument order and
 *      meaning, same ownership and error convention (NULL = failure; alignment functions
 *      never return NULL), so the Rust L2 layer links against this library unchanged
 *      (INTEGRATION.md).  Every alignment call runs its DP fill on the GPU; there is no
 *      CPU fallback -- if no HIP device is usable the call aborts with a message on stderr.
 *
 *  (2) Additive `pmx_*` batch entry points (no reference counterpart: the reference is
 *      one pair per call, src/aligner/mod.rs:397).  They take many pairs in packed
 *      buffers and are what BASELINE.json's throughput configs are measured on.
 *
 * Plain C types only; no torch / HIP types in any signature (streams are passed as void*).
 */
#ifndef PARASAIL_AMD_H
#define PARASAIL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ types ---- */

/* Layout read directly by Rust: .type_ (src/matrix/mod.rs:193), .size (:228,:257),
 * .length (:256), .matrix (:258).  [EXTERNAL] field order follows upstream parasail.h. */
typedef struct parasail_matrix {
    const char *name;
    const int *matrix;      /* length x size, row-major */
    const int *mapper;      /* 256 entries: byte -> column index */
    int size;               /* alphabet size incl. the wildcard column */
    int max;
    int min;
    int *user_matrix;       /* non-NULL for matrices owned by the caller (set_value allowed) */
    int type;               /* 0 = square, 1 = PSSM */
    int length;             /* rows: == size for square, query length for PSSM */
    const char *alphabet;
    const char *query;      /* PSSM only */
} parasail_matrix_t;

#define PARASAIL_MATRIX_TYPE_SQUARE 0
#define PARASAIL_MATRIX_TYPE_PSSM   1

/* Opaque to Rust (only passed back): src/alignment/mod.rs:55-60, src/profile/mod.rs:281-285 */
typedef struct parasail_result parasail_result_t;
typedef struct parasail_profile parasail_profile_t;
typedef parasail_profile_t parasail_profile;   /* bindgen alias used at src/profile/mod.rs:115 */
typedef parasail_matrix_t parasail_matrix;

/* src/alignment/mod.rs:368-374 reads .query/.comp/.ref_; each is a malloc'd C string that
 * Rust adopts with CString::from_raw. */
typedef struct parasail_traceback {
    char *query;
    char *comp;
    char *ref;
} parasail_traceback_t;

typedef struct parasail_cigar {
    uint32_t *seq;          /* BAM encoding: len << 4 | op, op index into "MIDNSHP=X" */
    int len;
    int beg_query;
    int beg_ref;
} parasail_cigar_t;

/* src/alignment/mod.rs:513-543 */
typedef struct parasail_result_ssw {
    uint16_t score1;
    int32_t ref_begin1;
    int32_t ref_end1;
    int32_t read_begin1;
    int32_t read_end1;
    uint32_t *cigar;
    int32_t cigarLen;
} parasail_result_ssw_t;

/* trace-table flag values: src/alignment/table.rs:127-142 */
#define PARASAIL_ZERO_MASK 120
#define PARASAIL_E_MASK    103
#define PARASAIL_F_MASK     31
#define PARASAIL_ZERO   0
#define PARASAIL_INS    1
#define PARASAIL_DEL    2
#define PARASAIL_DIAG   4
#define PARASAIL_DIAG_E 8
#define PARASAIL_INS_E  16
#define PARASAIL_DIAG_F 32
#define PARASAIL_DEL_F  64

typedef parasail_result_t *parasail_function_t(const char *s1, const int s1Len,
                                               const char *s2, const int s2Len,
                                               const int open, const int gap,
                                               const parasail_matrix_t *matrix);
typedef parasail_result_t *parasail_pfunction_t(const parasail_profile_t *profile,
                                                const char *s2, const int s2Len,
                                                const int open, const int gap);
typedef parasail_profile_t *parasail_pcreator_t(const char *s1, const int s1Len,
                                                const parasail_matrix_t *matrix);

/* --------------------------------------------------------------- dispatch ---- */
/* src/aligner/mod.rs:345 / :349 -- name grammar src/aligner/mod.rs:319-329:
 *   {nw|sg[_q{b,e,x}][_d{b,e,x}]|sw}[_trace][_stats][_table|_rowcol]_{striped|scan|diag}[_profile]_{sat|8|16|32|64}
 * A name with or without the leading "parasail_" is accepted.  Unknown name -> NULL. */
parasail_function_t  *parasail_lookup_function(const char *funcname);
parasail_pfunction_t *parasail_lookup_pfunction(const char *funcname);

/* src/aligner/mod.rs:470-481 */
parasail_result_t *parasail_nw_banded(const char *s1, const int s1Len, const char *s2, const int s2Len,
                                      const int open, const int gap, const int k,
                                      const parasail_matrix_t *matrix);
/* src/aligner/mod.rs:500-510, src/profile/mod.rs:345 */
parasail_result_ssw_t *parasail_ssw(const char *s1, const int s1Len, const char *s2, const int s2Len,
                                    const int open, const int gap, const parasail_matrix_t *matrix);
parasail_profile_t *parasail_ssw_init(const char *s1, const int s1Len,
                                      const parasail_matrix_t *matrix, const int8_t score_size);
void parasail_result_ssw_free(parasail_result_ssw_t *result);

/* ------------------------------------------------------------------ result --- */
/* src/alignment/mod.rs:64-98 */
int parasail_result_get_score(const parasail_result_t *result);
int parasail_result_get_end_query(const parasail_result_t *result);
int parasail_result_get_end_ref(const parasail_result_t *result);
int parasail_result_get_matches(const parasail_result_t *result);
int parasail_result_get_similar(const parasail_result_t *result);
int parasail_result_get_length(const parasail_result_t *result);
/* src/alignment/mod.rs:123-192: [query_len][ref_len] int32 row-major, valid until result_free */
int *parasail_result_get_score_table(const parasail_result_t *result);
int *parasail_result_get_matches_table(const parasail_result_t *result);
int *parasail_result_get_similar_table(const parasail_result_t *result);
int *parasail_result_get_length_table(const parasail_result_t *result);
/* src/alignment/mod.rs:195-288: rows have ref_len entries, cols have query_len entries */
int *parasail_result_get_score_row(const parasail_result_t *result);
int *parasail_result_get_matches_row(const parasail_result_t *result);
int *parasail_result_get_similar_row(const parasail_result_t *result);
int *parasail_result_get_length_row(const parasail_result_t *result);
int *parasail_result_get_score_col(const parasail_result_t *result);
int *parasail_result_get_matches_col(const parasail_result_t *result);
int *parasail_result_get_similar_col(const parasail_result_t *result);
int *parasail_result_get_length_col(const parasail_result_t *result);
/* src/alignment/mod.rs:291-307: [query_len][ref_len] one byte per cell */
int *parasail_result_get_trace_table(const parasail_result_t *result);
/* src/alignment/mod.rs:356-366, :400-410, :324-339 */
parasail_traceback_t *parasail_result_get_traceback(parasail_result_t *result,
        const char *seqA, int lena, const char *seqB, int lenb,
        const parasail_matrix_t *matrix, char match, char pos, char neg);
void parasail_traceback_free(parasail_traceback_t *traceback);
void parasail_traceback_generic(const char *seqA, int lena, const char *seqB, int lenb,
        const char *nameA, const char *nameB, const parasail_matrix_t *matrix,
        parasail_result_t *result, char match, char pos, char neg,
        int width, int name_width, int use_stats);
parasail_cigar_t *parasail_result_get_cigar(parasail_result_t *result,
        const char *seqA, int lena, const char *seqB, int lenb, const parasail_matrix_t *matrix);
char *parasail_cigar_decode(parasail_cigar_t *cigar);      /* malloc'd, caller frees */
void parasail_cigar_free(parasail_cigar_t *cigar);
/* src/alignment/mod.rs:422-494 */
int parasail_result_is_nw(const parasail_result_t *result);
int parasail_result_is_sg(const parasail_result_t *result);
int parasail_result_is_sw(const parasail_result_t *result);
int parasail_result_is_saturated(const parasail_result_t *result);
int parasail_result_is_banded(const parasail_result_t *result);
int parasail_result_is_scan(const parasail_result_t *result);
int parasail_result_is_striped(const parasail_result_t *result);
int parasail_result_is_diag(const parasail_result_t *result);
int parasail_result_is_blocked(const parasail_result_t *result);
int parasail_result_is_stats(const parasail_result_t *result);
int parasail_result_is_stats_table(const parasail_result_t *result);
int parasail_result_is_table(const parasail_result_t *result);
int parasail_result_is_rowcol(const parasail_result_t *result);
int parasail_result_is_stats_rowcol(const parasail_result_t *result);
int parasail_result_is_trace(const parasail_result_t *result);
/* src/alignment/mod.rs:498-504 */
void parasail_result_free(parasail_result_t *result);

/* ------------------------------------------------------------------ matrix --- */
/* src/matrix/mod.rs:40, :62, :140, :158, :188-197, :238, :281, :304 */
parasail_matrix_t *parasail_matrix_create(const char *alphabet, const int match, const int mismatch);
const parasail_matrix_t *parasail_matrix_lookup(const char *matrixname);
parasail_matrix_t *parasail_matrix_from_file(const char *filename);
parasail_matrix_t *parasail_matrix_pssm_create(const char *alphabet, const int *values, const int length);
parasail_matrix_t *parasail_matrix_convert_square_to_pssm(const parasail_matrix_t *matrix,
                                                          const char *s1, int s1Len);
parasail_matrix_t *parasail_matrix_copy(const parasail_matrix_t *matrix);
void parasail_matrix_set_value(parasail_matrix_t *matrix, int row, int col, int value);
void parasail_matrix_free(parasail_matrix_t *matrix);

/* ----------------------------------------------------------------- profile --- */
/* src/profile/mod.rs:113-277 picks one of these by (stats, ISA, width); on the GPU the ISA
 * slot is meaningless, so all ISA-suffixed names are aliases of the generic ones. */
#define PMX_DECLARE_PROFILE_CREATORS(ISA) \
    parasail_profile_t *parasail_profile_create##ISA##_sat(const char *, const int, const parasail_matrix_t *); \
    parasail_profile_t *parasail_profile_create##ISA##_8(const char *, const int, const parasail_matrix_t *);   \
    parasail_profile_t *parasail_profile_create##ISA##_16(const char *, const int, const parasail_matrix_t *);  \
    parasail_profile_t *parasail_profile_create##ISA##_32(const char *, const int, const parasail_matrix_t *);  \
    parasail_profile_t *parasail_profile_create##ISA##_64(const char *, const int, const parasail_matrix_t *);  \
    parasail_profile_t *parasail_profile_create_stats##ISA##_sat(const char *, const int, const parasail_matrix_t *); \
    parasail_profile_t *parasail_profile_create_stats##ISA##_8(const char *, const int, const parasail_matrix_t *);   \
    parasail_profile_t *parasail_profile_create_stats##ISA##_16(const char *, const int, const parasail_matrix_t *);  \
    parasail_profile_t *parasail_profile_create_stats##ISA##_32(const char *, const int, const parasail_matrix_t *);  \
    parasail_profile_t *parasail_profile_create_stats##ISA##_64(const char *, const int, const parasail_matrix_t *);
PMX_DECLARE_PROFILE_CREATORS()
PMX_DECLARE_PROFILE_CREATORS(_sse_128)
PMX_DECLARE_PROFILE_CREATORS(_avx_256)
PMX_DECLARE_PROFILE_CREATORS(_neon_128)
PMX_DECLARE_PROFILE_CREATORS(_altivec_128)
/* src/profile/mod.rs:384-390 */
void parasail_profile_free(parasail_profile_t *profile);

/* ------------------------------------------------- additive batch interface --- */

#define PMX_MODE_NW 0
#define PMX_MODE_SG 1
#define PMX_MODE_SW 2
/* semi-global free ends: q = query (s1), d = reference (s2) */
#define PMX_SG_QB 1
#define PMX_SG_QE 2
#define PMX_SG_DB 4
#define PMX_SG_DE 8
#define PMX_SG_ALL 15

#define PMX_WANT_STATS 1     /* matches / similar / length per pair */
#define PMX_WANT_CIGAR 2     /* on-device traceback, CIGAR text per pair */
#define PMX_WANT_SORTED 4    /* lengths are ragged: let the engine process pairs in length order (records stay in
                                input order).  The host-buffer entries set it themselves when it pays. */

#define PMX_FLAG_SATURATED 1 /* result record flag: the requested width overflowed */
#define PMX_FLAG_RERUN 2     /* internal: a fast kernel left its exact range; never visible to callers */

typedef struct pmx_config {
    int mode;                /* PMX_MODE_* */
    int sg_flags;            /* PMX_SG_* (ignored unless mode == SG) */
    int open;                /* positive; a gap of length k costs open + (k-1)*extend */
    int extend;
    int width;               /* 0 = sat (promote on overflow), 8, 16, 32, 64 */
    int want;                /* PMX_WANT_* */
    const parasail_matrix_t *matrix;
} pmx_config_t;

/* One record per pair, 16 bytes. */
typedef struct pmx_record {
    int32_t score;
    int32_t end_query;       /* 0-based inclusive */
    int32_t end_ref;
    int32_t flags;           /* PMX_FLAG_* */
} pmx_record_t;

typedef struct pmx_stats {
    int32_t matches, similar, length;
} pmx_stats_t;

/* Sequences are packed back to back: pair k's query is qbuf[qoff[k] .. qoff[k+1]).
 * Returns 0 on success, <0 on error (pmx_last_error() describes it). */

/* Host buffers in, host records out (H2D + kernels + D2H inside). */
int pmx_align_batch(const pmx_config_t *cfg, int64_t n,
                    const uint8_t *qbuf, const int64_t *qoff,
                    const uint8_t *rbuf, const int64_t *roff,
                    pmx_record_t *out, pmx_stats_t *stats_out /* NULL unless WANT_STATS */);

/* The same with 2-bit packed sequences (additive input form for 4-letter alphabets): base b of a buffer sits in byte b / 4 at
 * bits 2 (b % 4) and holds the index of its letter in the matrix alphabet (0..3); the offsets count BASES.  A quarter of the
 * bytes cross PCIe; a small kernel spells the letters out on the device before the usual path runs. */
int pmx_align_batch_2bit(const pmx_config_t *cfg, int64_t n,
                         const uint8_t *q2, const int64_t *qoff,
                         const uint8_t *r2, const int64_t *roff,
                         pmx_record_t *out, pmx_stats_t *stats_out);

/* Device-resident buffers (all pointers are device pointers on the current device),
 * asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream).
 * Internal scratch (length-sort permutation, retry list of the perm-table kernel) belongs to the calling
 * host thread and is reused by its next call: a thread may queue calls back to back on ONE stream; to run
 * on several streams at once, call from several threads.  Internal record flag values (PMX_FLAG_RERUN, 4)
 * never survive to the caller. */
int pmx_align_batch_device(const pmx_config_t *cfg, int64_t n,
                           const uint8_t *d_qbuf, const int64_t *d_qoff,
                           const uint8_t *d_rbuf, const int64_t *d_roff,
                           int32_t max_qlen, int32_t max_rlen,
                           pmx_record_t *d_out, pmx_stats_t *d_stats_out, void *stream);

/* One reused query profile against many references (profile arm, src/aligner/mod.rs:431-450). */
int pmx_align_profile_batch(const pmx_config_t *cfg, const parasail_profile_t *profile, int64_t n,
                            const uint8_t *rbuf, const int64_t *roff,
                            pmx_record_t *out, pmx_stats_t *stats_out);

/* The same with device-resident references (device pointers, asynchronous on `stream`). */
int pmx_align_profile_batch_device(const pmx_config_t *cfg, const parasail_profile_t *profile, int64_t n,
                                   const uint8_t *d_rbuf, const int64_t *d_roff, int32_t max_rlen,
                                   pmx_record_t *d_out, pmx_stats_t *d_stats_out, void *stream);

/* Banded batches (extension).  The reference has one banded entry -- Aligner::banded_nw -> parasail_nw_banded above: global, main
 * diagonal.  The batch form takes any mode and an optional per-pair band centre: cell (i, j) of pair k belongs to the band iff
 * |(j - i) - diag[k]| <= band (diag == NULL: 0 for every pair); cells outside it cannot be entered or left.  Only the band's
 * cells are computed (band <= 63; wider bands run the general kernel with a mask).  BASELINE config 5's "banded SW" is
 * mode = PMX_MODE_SW with diag[k] = end_ref - end_query of a first full pass, or a seed's diagonal.  Score and end positions
 * only; 32-bit lanes (cfg->width is ignored).  profile != NULL: the profile arm (qbuf / qoff are ignored). */
int pmx_align_batch_banded(const pmx_config_t *cfg, const parasail_profile_t *profile, int64_t n,
                           const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                           int32_t band, const int32_t *diag, pmx_record_t *out);
int pmx_align_batch_banded_device(const pmx_config_t *cfg, int64_t n,
                                  const uint8_t *d_qbuf, const int64_t *d_qoff,
                                  const uint8_t *d_rbuf, const int64_t *d_roff,
                                  int32_t max_qlen, int32_t max_rlen, int32_t band, const int32_t *d_diag,
                                  pmx_record_t *d_out, void *stream);
int pmx_align_profile_batch_banded_device(const pmx_config_t *cfg, const parasail_profile_t *profile, int64_t n,
                                          const uint8_t *d_rbuf, const int64_t *d_roff, int32_t max_rlen,
                                          int32_t band, const int32_t *d_diag, pmx_record_t *d_out, void *stream);

/* CIGAR text for a batch (semi-global / global / local with traceback done on the device).
 * cigar_off has n+1 entries; *cigar_buf is malloc'd by the callee and freed with pmx_free. */
int pmx_align_batch_cigar(const pmx_config_t *cfg, int64_t n,
                          const uint8_t *qbuf, const int64_t *qoff,
                          const uint8_t *rbuf, const int64_t *roff,
                          pmx_record_t *out, char **cigar_buf, int64_t *cigar_off);
void pmx_free(void *p);

/* The same with device-resident pairs (all pointers are device pointers on the current device, the offset arrays
 * start at 0), asynchronous on `stream`.  d_cigar_off receives n+1 offsets into d_cigar_text (the last one is the
 * number of text bytes the batch needs); a pair whose text would cross cigar_capacity is not written, so a caller
 * that sees d_cigar_off[n] > cigar_capacity calls again with a larger buffer.  Inside, the traceback sweep and the
 * walk of consecutive chunks overlap on two streams.  Returns <0 when the configuration has no packed-traceback
 * kernel (width 8, PSSM, open < extend, score + open beyond a byte, queries beyond 1023 symbols); the host entry
 * above handles those. */
int pmx_align_batch_cigar_device(const pmx_config_t *cfg, int64_t n,
                                 const uint8_t *d_qbuf, const int64_t *d_qoff,
                                 const uint8_t *d_rbuf, const int64_t *d_roff,
                                 int32_t max_qlen, int32_t max_rlen,
                                 pmx_record_t *d_out, char *d_cigar_text, int64_t cigar_capacity,
                                 int64_t *d_cigar_off, void *stream);

/* Score tables for a batch (extension: the reference returns one table per call, src/alignment/mod.rs:123-192).  All pointers are
 * device pointers, asynchronous on `stream`.  d_tab_off[k] (n + 1 entries) = cells before pair k's [qlen][rlen] int32 row-major
 * table in d_score_table; d_score_row (last rows) is packed like the references, d_score_col (last columns) like the queries;
 * any of the three outputs, and d_out, may be NULL.  4 bytes per cell: this is the one output of the path that is bound by HBM
 * write bandwidth. */
int pmx_align_batch_table_device(const pmx_config_t *cfg, int64_t n,
                                 const uint8_t *d_qbuf, const int64_t *d_qoff,
                                 const uint8_t *d_rbuf, const int64_t *d_roff,
                                 int32_t max_qlen, int32_t max_rlen,
                                 const int64_t *d_tab_off, int32_t *d_score_table,
                                 int32_t *d_score_row, int32_t *d_score_col,
                                 pmx_record_t *d_out, void *stream);

/* Multi-GPU (one process driving several GPUs of a node).  Pairs are independent, so the batch is cut into ndev contiguous
 * blocks with about equal numbers of cells (sum of qlen * rlen; pmx_shard_bounds_by_cells is the planner), block g runs on
 * devices[g] from its own persistent host thread, and every block's records land in `out` at its pairs' positions: input order,
 * no collective.  A device may be listed more than once.  (One process per GPU instead: shard with the same planner and gather
 * the 16-byte records, e.g. RCCL over xGMI -- see INTEGRATION.md.) */
int pmx_align_batch_multi(const pmx_config_t *cfg, int64_t n,
                          const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                          const int *devices, int ndev, pmx_record_t *out, pmx_stats_t *stats_out);
int pmx_align_profile_batch_multi(const pmx_config_t *cfg, const parasail_profile_t *profile, int64_t n,
                                  const uint8_t *rbuf, const int64_t *roff,
                                  const int *devices, int ndev, pmx_record_t *out, pmx_stats_t *stats_out);
/* bounds[0..parts]: block g = pairs [bounds[g], bounds[g+1]).  qoff == NULL: one shared query.  Pure host arithmetic. */
int pmx_shard_bounds_by_cells(int64_t n, const int64_t *qoff, const int64_t *roff, int parts, int64_t *bounds);

/* Optional: page-lock caller-owned host buffers once, so the host-buffer entries above move them at full PCIe rate. */
int pmx_host_register(void *p, size_t bytes);
int pmx_host_unregister(void *p);

/* Runtime. */
int pmx_device_count(void);
int pmx_set_device(int device);            /* per calling thread, like hipSetDevice */
const char *pmx_last_error(void);
const char *pmx_version(void);
/* Name of the kernel family the dispatcher would use for a config and size (diagnostics). */
const char *pmx_kernel_for(const pmx_config_t *cfg, int32_t max_qlen, int32_t max_rlen);
/* Name (with template shape and arithmetic variant) of the kernel the calling thread's last batch call launched. */
const char *pmx_last_kernel(void);
/* Every environment switch the library reads, one "NAME\tkind\twhat\n" line each (parasail-rs_amd/csrc/pmx_switches.h). */
const char *pmx_switches(void);
/* Deferred results (environment switch PMX_DEFER_ALIGN=1): the one-pair alignment functions (score and statistics names) queue the
 * pair and return a PENDING parasail_result_t; the first accessor (parasail_result_get_score, ...) of any pending result of that
 * thread -- or a queue of 262 144 pairs, or a call with another configuration -- runs the queue as ONE batch launch.  The mode / width
 * predicates answer at once; freeing a pending result withdraws it.  pmx_flush_deferred() runs the calling thread's queue now. */
void pmx_flush_deferred(void);
/* The 66 matrix names the reference documents (src/matrix/mod.rs:46-50), one per line, into buf (NUL-terminated, truncated to cap);
 * returns the bytes needed.  parasail_matrix_lookup() embeds blosum62 / nuc44 and resolves the others from files (pmx_last_error()
 * tells a documented name whose file is missing from an unknown name, and why a file was refused). */
int pmx_documented_matrix_names(char *buf, int cap);
/* Test hooks of the band-strip kernel (parasail-rs_amd/csrc/pmx_bstrip.hip; model: tests/bstrip_model.py): the host's window
 * predicate -- 1 and the bias / low constants of the stored form when the int16 window holds a launch of that shape, else 0 -- and
 * the lane shape (lanes per pair, offsets per lane) chosen for a band. */
int pmx_bstrip_window(int mode, int max_qlen, int max_rlen, int open, int extend, int score_min, int score_max, int capacity, int rows,
                      int double_skew, int *bias, int *low);
int pmx_bstrip_shape(int band, int *lanes_per_pair, int *offsets_per_lane);
/* Test hook of the packed global / semi-global kernels (pmx_nwsg16.hip; model: tests/nwsgv_model.c): the host's range proof --
 * the bias nb of the stored form when the int16 window holds every pair of up to max_qlen x max_rlen under this scoring in a shape of
 * shape_rows rows (0: the dispatcher's estimate before a shape is picked), else 0.  rowx: the row-offset form (every width but 8). */
int pmx_window_nwsgv(int max_qlen, int max_rlen, int msize, int score_min, int score_max, int open, int extend, int rowx, int shape_rows);

#ifdef __cplusplus
}
#endif
#endif /* PARASAIL_AMD_H */
