// pmx_common.h -- internal declarations shared by the HIP kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/parasail_amd.h"
#include "../../include/pmx_conventions.h"

#define PMX_MAX_FAST_MSIZE 32      // fast kernels stage the matrix in LDS as int16[msize*msize]

// Raise the dynamic-LDS limit of a kernel once per (kernel, device); thread-safe.
int pmx_ensure_lds_attr(const void *kernel, int bytes = 160 * 1024);

// Device-side view of a substitution matrix (built once per parasail_matrix_t, cached).
struct PmxDevMatrix {
    const int16_t *scores;   // [msize*msize], scores[qsym*msize + rsym]   (device)
    const uint8_t *mapper;   // [256] byte -> symbol index                  (device)
    int msize;
    int min, max;
};

// Batch of pairs, device-resident, packed layout of include/parasail_amd.h.
struct PmxBatch {
    const uint8_t *qbuf; const int64_t *qoff;
    const uint8_t *rbuf; const int64_t *roff;
    int64_t n;
    int max_qlen, max_rlen;
    int q_shared;            // > 0: one shared query of that many bytes at qbuf (profile arm), qoff unused
    const unsigned *perm;    // optional processing order (pmx_sort.hip): position -> pair index
    unsigned *retry_list;    // optional scratch, n entries + one int: lets the sw16 launcher use kernels that hand
    int *retry_count;        //   some pairs back for a second launch (decided on the device, no host sync)
    int q_has_wildcard;      // shared query only: it holds a letter beyond the first four of the alphabet
    int sat_above;           // sw16 only, 0 = off: scores above this set PMX_FLAG_SATURATED (width 8: 127; local H >= 0, so the
                             //   maximum H is the score and the oracle's saturation rule needs nothing else)
    int track8;              // nwsg16v only: width 8 -- track the range of H and flag pairs that leave [-128, 127]
    int *blockflag;          // nwsg16v only, optional scratch of (n + 15) / 2 ints: lets the launcher run the perm-table form first, which
                             //   marks the blocks (of 2 * 64 / G pairs) it leaves to the LDS-profile form; the walk reads the flags too
};
#define PMX_FLAG_RETRY16 4   // internal record flag: redo with the LDS-profile variant of the fast kernel

// Fast path: local alignment, score + end positions, packed int16 lanes.
// Returns 0 if launched, 1 if the shape is not supported by any instantiation (caller falls
// through to the general kernel), <0 on a HIP error.
int pmx_launch_sw16(const PmxBatch &b, const PmxDevMatrix &m, int open, int ext,
                    pmx_record_t *d_out, hipStream_t stream, const char **kernel_name);

// Traceback variant of the local kernel (pmx_sw16.hip, VAR 7): same trace layout, rows top-aligned.
int pmx_sw16_trace_plan(const PmxBatch &b, const PmxDevMatrix &m, int open, int ext, int *variant, int *Tmax, size_t *trace_bytes);
int pmx_launch_sw16_trace(int variant, const PmxBatch &b, const PmxDevMatrix &m, int open, int ext,
                          pmx_record_t *d_out, uint32_t *tbuf, int Tmax, hipStream_t stream);

// Bias of the second-generation nw/sg arithmetic (0: its exact window does not hold for this batch).
int pmx_nwsgv_bias(const PmxBatch &b, const PmxDevMatrix &m, int open, int ext, int rowx = 0, int shape_rows = 0 /* > 0: the rows of the shape that will run */);
// Packed statistics kernel (pmx_stats16p.hip): 0 launched, 1 not eligible, <0 HIP error.
int pmx_launch_stats16p(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                        pmx_record_t *d_out, pmx_stats_t *d_stats, hipStream_t stream, const char **kernel_name);

// Traceback variant of the second-generation nw/sg kernel (pmx_nwsg16.hip); the walk lives in pmx_trace16.hip.
int pmx_nwsgv_trace_plan(const PmxBatch &b, const PmxDevMatrix &m, int mode, int open, int ext,
                         int *variant, int *Tmax, size_t *trace_bytes);
int pmx_launch_nwsgv_trace(int variant, const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                           pmx_record_t *d_out, uint32_t *tbuf, int Tmax, hipStream_t stream);

// Traceback with a shared query (pmx_nwsg16q_kernel<..., TR>): statistics of the profile arm are counted along the path.
long long pmx_nwsgq_trace_round_pairs(int variant, const PmxDevMatrix &m, int mode, int sg_flags);
int pmx_nwsgq_trace_plan(const PmxBatch &b, const PmxDevMatrix &m, int mode, int open, int ext,
                         int *variant, int *Tmax, size_t *trace_bytes, int *G_out, int *R_out, int short_waves = 0);
int pmx_launch_nwsgq_trace(int variant, const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                           pmx_record_t *d_out, uint32_t *tbuf, int Tmax, hipStream_t stream);

// Shared-query variant of the local kernel (pmx_sw16q.hip); called by pmx_launch_sw16 once the skewed byte-profile
// variant's conditions hold.  0 launched, 1 not eligible, <0 HIP error.
int pmx_launch_sw16q(const PmxBatch &b, const PmxDevMatrix &m, int open, int ext,
                     pmx_record_t *d_out, hipStream_t stream, const char **kernel_name);

// Matrix-lookup variant of the local kernel for per-pair queries over large alphabets (pmx_sw16m.hip); same contract.
int pmx_launch_sw16m(const PmxBatch &b, const PmxDevMatrix &m, int open, int ext,
                     pmx_record_t *d_out, hipStream_t stream, const char **kernel_name);

int pmx_launch_sw16m_trace(int variant, const PmxBatch &b, const PmxDevMatrix &m, int open, int ext,
                           pmx_record_t *d_out, uint32_t *tbuf, int Tmax, hipStream_t stream);

// 2-bit packed input -> ASCII letters (pmx_sort.hip)
int pmx_launch_unpack2(const uint8_t *in, uint8_t *out, long long lo, long long hi, uint32_t letters, hipStream_t stream);

// Length-sorted processing order for ragged batches (pmx_sort.hip).
size_t pmx_sort_scratch_bytes(long long n);
int pmx_build_length_perm(const int64_t *d_roff, long long n, void *scratch, const unsigned **perm_out, hipStream_t stream);
// Processing order for banded batches: by the number of anti-diagonal steps of each pair's band (same scratch size).
int pmx_build_band_perm(const int64_t *d_qoff, int q_shared, const int64_t *d_roff, const int32_t *d_diag, int band, long long n,
                        void *scratch, const unsigned **perm_out, hipStream_t stream, bool by_entry_row = false);

// Fast path: global / semi-global, score + end positions, biased packed lanes (pmx_nwsg16.hip).
int pmx_launch_nwsg16(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                      pmx_record_t *d_out, hipStream_t stream, const char **kernel_name);

// Fast path with statistics (pmx_stats16.hip): matches / similar / length travel with H, E, F.
int pmx_launch_stats16(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                       pmx_record_t *d_out, pmx_stats_t *d_stats, hipStream_t stream, const char **kernel_name);

// Fast path with traceback (pmx_trace16.hip): 4-bit trace in HBM + on-device walk -> run-length ops.
int pmx_trace16_plan(const PmxBatch &b, const PmxDevMatrix &m, int mode, int open, int ext,
                     int *variant, int *Tmax, size_t *trace_bytes, bool packed_ok = true /* false: first-generation kernels only */);
// Walk over the packed records (pmx_walkp.hip).  ops: run-length BAM ops written from the END of the pair's slot backwards
// (slot = ops_off[k] .. + qlen + rlen + 1, or implicit: qoff[k] + roff[k] + k - ops_base): the forward list is the last nops[k] entries.
int pmx_launch_walkp(int gsel, int R, const PmxBatch &b, const PmxDevMatrix &m, int mode, int open, int ext, int Tmax, int top_aligned,
                     pmx_stats_t *stats_out, int row_pen, int col_pen, const uint32_t *tbuf, const pmx_record_t *recs,
                     uint32_t *ops, const int64_t *ops_off, long long ops_base, int32_t *nops, int32_t *beg, int32_t *textlen,
                     hipStream_t stream, const int *blockflag = nullptr /* per sweep block: 0 = top-aligned rows (perm-table sweep) */);
// Optional second stream for the walk (device CIGAR entry: the walk of chunk c runs beside the sweep of chunk c+1).
struct PmxWalkSplit {
    hipStream_t walk_stream;   // == the sweep's stream: no split
    hipEvent_t sweep_done;     // recorded on the sweep stream after the sweep, awaited by the walk stream
    hipEvent_t walk_done;      // recorded after the walk (may be null)
    long long ops_base;        // ops_off == nullptr: slot of pair k starts at qoff[k] + roff[k] + k - ops_base
    int32_t *textlen;          // optional, per pair: bytes of its CIGAR text
};
int pmx_launch_trace16(int variant, const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                       pmx_record_t *d_out, uint32_t *tbuf, int Tmax,
                       uint32_t *ops, const int64_t *ops_off, int32_t *nops, int32_t *beg, hipStream_t stream,
                       pmx_stats_t *stats_out = nullptr /* count the path's statistics instead of emitting ops */,
                       const PmxWalkSplit *split = nullptr);

// ---- general kernel (all modes, stats, tables, rows/cols, trace, band) -------------------
struct PmxGeneralArgs {
    // inputs
    const uint8_t *qbuf; const int64_t *qoff;     // qoff == nullptr: one shared query of shared_qlen bytes at qbuf
    const uint8_t *rbuf; const int64_t *roff;
    long long n;
    int max_rlen;             // longest reference in the launch (sizes the LDS symbol buffer)
    const int64_t *index;     // optional: block b works on pair index[b] (promotion re-runs); n = number of blocks
    int shared_qlen;
    const int16_t *scores; const uint8_t *mapper; int msize;
    int mat_rows;             // rows of `scores`: msize for a square matrix, query length for a PSSM
    int pssm;                 // 1: row of `scores` is the query position, not the query symbol
    int mode, sg_flags, open, ext;
    int band;                 // < 0: no band; else cells with |(j - i) - diag[pair]| > band are excluded (nw_banded: diag == nullptr)
    const int32_t *diag;      // optional per-pair band centre (indexed like roff)
    int bits;                 // 0/32/64: no range check; 8 or 16: report saturation of that range
    // per-pair scratch: boundary row between 64-row bands, 8 ints per reference column
    int32_t *bound; long long bound_stride;       // ints per pair
    // optional per-block scratch (max_rlen + 8 bytes each) for references that do not fit the LDS; see pmx_general_lds_fits()
    uint8_t *rs_scratch; long long rs_stride;
    // outputs (device); any may be null
    pmx_record_t *rec; pmx_stats_t *stats;
    // table-like outputs: cell offset of pair k is tab_off[k] (nullptr -> pair 0 at 0, n must be 1)
    const int64_t *tab_off;
    int32_t *score_table, *matches_table, *similar_table, *length_table;
    int8_t *trace_table;
    int trace_lds;            // set by the launcher: stage one band of trace bytes in LDS, flush coalesced
    int max_qlen;             // longest query of the launch (0: unknown); lets the launcher share one long pair among several waves
    int mw_sched_off;         // set by the launcher (multi-wave form): LDS offset of the band schedule
    // row/col outputs: row offset = roff[k], col offset = qoff[k] (or 0 for n == 1)
    int32_t *score_row, *matches_row, *similar_row, *length_row;
    int32_t *score_col, *matches_col, *similar_col, *length_col;
};
int pmx_launch_general(const PmxGeneralArgs &a, bool want_stats, hipStream_t stream);
// true if the general kernel can keep a reference of max_rlen symbols (and the matrix) in the LDS; otherwise rs_scratch is needed
static inline bool pmx_general_lds_fits(int mat_rows, int msize, int max_rlen)
{
    return ((((size_t)mat_rows * msize * 2 + 15) & ~(size_t)15) + (((size_t)max_rlen + 8 + 15) & ~(size_t)15)) <= 160 * 1024;
}

// Banded fast kernel (pmx_banded.hip): lanes over the band's diagonals, one query row per step, only the band's cells are
// computed.  0 launched, 1 not eligible (the general kernel masks instead), <0 HIP error.
int pmx_launch_banded(int mode, int sg_flags, int open, int ext, const PmxDevMatrix &m, long long n,
                      const uint8_t *qbuf, const int64_t *qoff, int q_shared, const uint8_t *rbuf, const int64_t *roff,
                      int max_qlen, int max_rlen, int band, const int32_t *diag, pmx_record_t *out, hipStream_t stream, const char **kernel_name = nullptr,
                      void *sort_scratch = nullptr /* pmx_sort_scratch_bytes(n) bytes: lets the packed forms pair up bands of equal length */,
                      unsigned *retry_list = nullptr, int *retry_count = nullptr /* [count][n entries] twice (retry_count first): lets the band-strip kernel hand pairs back */);
// Band-strip kernel (pmx_bstrip.hip): band coordinates, packed int16, alphabets of <= 4 letters (+ wildcard).  Same contract.
int pmx_launch_bstrip(int mode, int sg_flags, int open, int ext, const PmxDevMatrix &m, long long n,
                      const uint8_t *qbuf, const int64_t *qoff, int q_shared, const uint8_t *rbuf, const int64_t *roff,
                      int max_qlen, int max_rlen, int band, const int32_t *diag, pmx_record_t *out, hipStream_t stream,
                      const char **kernel_name, void *sort_scratch, unsigned *retry_list, int *retry_count);

// Score tables / last rows and columns, row by row at HBM write speed (pmx_table.hip).  0 launched, 1 not eligible, <0 HIP error.
int pmx_launch_table(int mode, int sg_flags, int open, int ext, const PmxDevMatrix &m, long long n,
                     const uint8_t *qbuf, const int64_t *qoff, int q_shared, const uint8_t *rbuf, const int64_t *roff,
                     int max_qlen, int max_rlen, const int64_t *tab_off, int32_t *table, int32_t *row_out, int32_t *col_out,
                     pmx_record_t *out, hipStream_t stream, int8_t *trace = nullptr);

// On-device traceback walk: trace tables -> run-length ops (BAM-encoded uint32 per run).
// ops_off[k] = first slot of pair k in `ops` (capacity qlen+rlen each), nops[k] = runs written.
struct PmxWalkArgs {
    const uint8_t *qbuf; const int64_t *qoff; const uint8_t *rbuf; const int64_t *roff;
    long long n; int shared_qlen;
    const uint8_t *mapper; int mode;
    const int8_t *trace_table; const int64_t *tab_off;
    const pmx_record_t *rec;
    uint32_t *ops; const int64_t *ops_off; int32_t *nops; int32_t *beg; /* 2 per pair */
};
int pmx_launch_walk(const PmxWalkArgs &a, hipStream_t stream);
int pmx_launch_cigar_textlen(const uint32_t *ops, const int64_t *ops_off, const int32_t *nops, int32_t *textlen, long long n, hipStream_t stream);
int pmx_launch_cigar_render(const uint32_t *ops, const int64_t *ops_off, const int32_t *nops,
                            const int64_t *text_off, char *text, long long n, hipStream_t stream);
// Device CIGAR entry: exclusive scan of the per-pair text lengths (n + 1 entries in, the last one ignored) into int64
// offsets (n + 1 entries out, the last one = total bytes), and the render with implicit op slots and a capacity limit.
size_t pmx_text_scan_scratch_bytes(long long n);
int pmx_launch_text_offsets(const int32_t *textlen, long long n, int64_t *text_off, void *scratch, size_t scratch_bytes, hipStream_t stream);
int pmx_launch_cigar_render_slots(const uint32_t *ops, const int64_t *qoff, const int64_t *roff, long long ops_base, const int32_t *nops,
                                  const int64_t *text_off, char *text, long long capacity, long long n, hipStream_t stream);

// One long pair (or a few) across the chip: bands of the query on different CUs, pipelined through HBM granules (pmx_long.hip).
// R = rows per lane (4 or 16).  0 launched, 1 not eligible, <0 HIP error; `scratch` holds pmx_long_scratch_bytes() bytes.
size_t pmx_long_scratch_bytes(long long n, int max_qlen, int max_rlen, int R, long long *bstride, int *nbmax);
// The first 64 bytes of `scratch` are the call's abort word: the caller zeroes them before the first launch and reads them after the
// last (non-zero: a band's bounded wait ran out -- every record of the call is marked PMX_FLAG_RERUN and has to be redone elsewhere).
int pmx_launch_long(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext, int R,
                    void *scratch, pmx_record_t *d_out, int sat_above, int force_sat, hipStream_t stream, int spin_limit = 1 << 20,
                    int chunk_cols = 16 /* boundary columns a band takes over at a time: 16 or 64 (two-column form: steps, 32 or 64) */,
                    int two_cols = 1 /* rows per lane 2 or 4: the form with two columns per step */);

// Run-time CIGAR letter convention (switch PMX_CIGAR_SWAP_ID, read per call): 1 = exchange I and D in everything handed out.
int pmx_cigar_swapped();

// Collect the indices of records whose flags intersect `mask`: list[0..*count) (device), any order.
int pmx_launch_collect_saturated(const pmx_record_t *rec, long long n, int64_t *list, int *count, int mask, hipStream_t stream);
