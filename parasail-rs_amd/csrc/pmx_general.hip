// pmx_general.hip -- the general DP kernel: every mode (nw / sg with any free-end set / sw),
// optional statistics (matches, similar, length), full tables, last row/col, trace table,
// banded nw, 32-bit arithmetic.  gfx950 only.
//
// It backs every `parasail_*` dispatch name (name grammar /root/reference/src/aligner/mod.rs:
// 319-329) that the packed-int16 fast kernels do not cover, and produces the memory layouts the
// result accessors hand out: tables [query_len][ref_len] int32 row-major
// (src/alignment/table.rs:4-9), 1-byte trace cells with the TraceFlags bit values
// (src/alignment/table.rs:127-142), rows of ref_len and cols of query_len entries
// (src/alignment/mod.rs:195-288).
//
// Mapping: one 64-lane wave per pair.  The query is cut into bands of 64 rows; inside a band
// lane l owns row band*64+l and keeps that row's state (H of the previous column, E, and their
// statistics) in registers.  The wave sweeps the band along anti-diagonals: at step t lane l
// computes column t-l, taking H/F (and stats) of the row above from lane l-1 through a
// wave shuffle.  The band's last row is parked in a per-pair boundary buffer in HBM
// (8 ints per reference column) and picked up by lane 0 of the next band.
//
// Recurrences, boundary values and every tie-break are the ones written down in
// oracle/pmx_oracle.c (the checker); they are restated here, not shared.
#include "pmx_common.h"
#include "pmx_switches.h"

#define NEG_INF (INT32_MIN / 2)
#ifndef PMX_MW_MAX_PAIRS
#define PMX_MW_MAX_PAIRS 2048   // up to this many pairs per launch, a pair gets a workgroup of waves instead of one wave (see pmx_general_mw_kernel;
                                // 3 kbp x 3 kbp pairs, profiles/bench_mw_threshold.py: 256 pairs 9.8 vs 44.8 ms, 1024: 36 vs 64 ms, 2048: 80 vs 78 ms)
#endif

#define T_INS 1
#define T_DEL 2
#define T_DIAG 4
#define T_DIAG_E 8
#define T_INS_E 16
#define T_DIAG_F 32
#define T_DEL_F 64

struct Cand { int H, i, j, M, S, L; };

// value of lane-1 (DPP wave_shr:1; lane 0 keeps its own value, which the caller overrides)
__device__ __forceinline__ int lane_up(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xF, 0xF, false); }

// true if a should replace b under "larger H, then smaller j, then smaller i"
__device__ __forceinline__ bool better_sw(const Cand &a, const Cand &b)
{
    if (a.H != b.H) return a.H > b.H;
    if (a.j != b.j) return a.j < b.j;
    return a.i < b.i;
}
__device__ __forceinline__ bool better_minj(const Cand &a, const Cand &b)
{
    if (a.H != b.H) return a.H > b.H;
    return a.j < b.j;
}
__device__ __forceinline__ bool better_mini(const Cand &a, const Cand &b)
{
    if (a.H != b.H) return a.H > b.H;
    return a.i < b.i;
}
__device__ __forceinline__ Cand shfl_cand(const Cand &c, int off)
{
    Cand o;
    o.H = __shfl_xor(c.H, off, 64); o.i = __shfl_xor(c.i, off, 64); o.j = __shfl_xor(c.j, off, 64);
    o.M = __shfl_xor(c.M, off, 64); o.S = __shfl_xor(c.S, off, 64); o.L = __shfl_xor(c.L, off, 64);
    return o;
}

template <bool STATS, bool OUT>
__global__ __launch_bounds__(64)
void pmx_general_kernel(const PmxGeneralArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int16_t *mat = reinterpret_cast<int16_t *>(lds);
    const int lane = threadIdx.x;
    const long long pair = a.index ? a.index[blockIdx.x] : (long long)blockIdx.x;
    const int msize = a.msize;

    for (int i = lane; i < a.mat_rows * msize; i += 64) mat[i] = a.scores[i];

    long long qb, rb; int ql, rl;
    if (a.qoff) { qb = a.qoff[pair]; ql = (int)(a.qoff[pair + 1] - qb); }
    else { qb = 0; ql = a.shared_qlen; }
    rb = a.roff[pair]; rl = (int)(a.roff[pair + 1] - rb);
    const uint8_t *q = a.qbuf + qb, *r = a.rbuf + rb;
    // mapped reference symbols live in LDS for the whole pair (the sweep reads one per lane and step;
    // fetching them from HBM inside the loop made every step wait for two dependent global loads)
    // (a reference too long for the LDS -- beyond ~160 k symbols -- is mapped into a per-pair HBM scratch instead and read through
    //  the caches: slower per step, but no length limit; the reference has none either)
    unsigned char *rs_lds = lds + (((size_t)a.mat_rows * msize * 2 + 15) & ~(size_t)15);
    unsigned char *rs = a.rs_scratch ? a.rs_scratch + (long long)blockIdx.x * a.rs_stride : rs_lds;
    for (int j = lane; j < rl; j += 64) rs[j] = a.mapper[r[j]];
    for (int j = rl + lane; j < rl + 4; j += 64) rs[j] = 0;
    if (a.rs_scratch) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent"); }
    __syncthreads();
    // optional staging area for one band of trace bytes (64 rows x rl), flushed with wide stores
    unsigned char *tstage = rs_lds + (((size_t)a.max_rlen + 8 + 15) & ~(size_t)15);
    const bool tlds = OUT && a.trace_table && a.trace_lds;
    const long long tab0 = a.tab_off ? a.tab_off[pair] : 0;
    const long long row0 = rb;                                // row outputs packed like the references (a one-pair call has rb == 0;
    const long long col0 = a.qoff ? qb : pair * (long long)ql;   //  a one-pair CHUNK of a batch keeps its place), columns like the queries
    int32_t *bound = a.bound + (long long)blockIdx.x * a.bound_stride;

    const int mode = a.mode, open = a.open, ext = a.ext, band_w = a.band;
    const int band_d = a.diag ? a.diag[pair] : 0;             // band centre: cells with |(j - i) - band_d| > band_w are excluded
    const bool s1_beg = mode == PMX_MODE_SG && (a.sg_flags & PMX_SG_QB);
    const bool s1_end = mode == PMX_MODE_SG && (a.sg_flags & PMX_SG_QE);
    const bool s2_beg = mode == PMX_MODE_SG && (a.sg_flags & PMX_SG_DB);
    const bool s2_end = mode == PMX_MODE_SG && (a.sg_flags & PMX_SG_DE);
    const bool col_pen = mode == PMX_MODE_NW || (mode == PMX_MODE_SG && !s1_beg);  // H(i,-1) penalised
    const bool row_pen = mode == PMX_MODE_NW || (mode == PMX_MODE_SG && !s2_beg);  // H(-1,j) penalised

    Cand best_sw = {NEG_INF, 0, 0, 0, 0, 0};      // sw: global best
    Cand best_row = {NEG_INF, 0, 0, 0, 0, 0};     // sg: best of the last row
    Cand best_col = {NEG_INF, 0, 0, 0, 0, 0};     // sg: best of the last column
    Cand corner = {NEG_INF, 0, 0, 0, 0, 0};
    int hmin = 0, hmax = 0;

    const int nbands = (ql + 63) / 64;
    for (int bandi = 0; bandi < nbands; ++bandi) {
        const int i = bandi * 64 + lane;
        const bool row_ok = i < ql;
        const int qsym = row_ok ? a.mapper[q[i]] : 0;
        const int16_t *mrow = mat + (a.pssm ? (row_ok ? i : 0) : qsym) * msize;
        // left boundary H(i,-1) and the diagonal seed H(i-1,-1)
        int leftH = col_pen ? -(open + i * ext) : 0;
        int leftM = 0, leftS = 0, leftL = col_pen ? i + 1 : 0;
        int diagH = (i == 0) ? 0 : (col_pen ? -(open + (i - 1) * ext) : 0);
        int diagM = 0, diagS = 0, diagL = (i == 0) ? 0 : (col_pen ? i : 0);
        if (row_ok) { hmin = min(hmin, leftH); }
        int E = NEG_INF, EM = 0, ES = 0, EL = 0;
        // what this lane hands to the lane below: H(i,j), F(i,j) and stats
        int oH = NEG_INF, oF = NEG_INF, oHM = 0, oHS = 0, oHL = 0, oFM = 0, oFS = 0, oFL = 0;

        int8_t *tdst = nullptr; unsigned char *ts = nullptr;
        if (tlds) {
            tdst = a.trace_table + tab0 + (long long)bandi * 64 * rl;
            ts = tstage + ((uintptr_t)tdst & 15);          // same alignment mod 16 as the destination
        }
        // lane 0 reads the previous band's last row one column ahead of its use
        int pb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (lane == 0 && bandi > 0) {
            pb[0] = bound[0]; pb[1] = bound[1];
            if (STATS) { for (int x = 2; x < 8; ++x) pb[x] = bound[x]; }
        }
        // Banded: only the columns some row of this 64-row band can reach are swept -- rows 64 b .. 64 b + 63 see columns
        // [64 b + d - w, 64 b + 63 + d + w]; everything left and right of that range is outside the band for every row of the band,
        // so the sweep may start there with "minus infinity" to its left.  Work per pair: qlen x (2 w + 190) instead of qlen x rlen.
        int jlo = 0, jhi = rl - 1, pjhi = rl - 1;            // this band's column range; the previous band's last column
        if (band_w >= 0) {
            jlo = max(0, bandi * 64 + band_d - band_w);
            jhi = min(rl - 1, bandi * 64 + 63 + band_d + band_w);
            pjhi = min(rl - 1, bandi * 64 - 1 + band_d + band_w);
            if (jlo > 0) {                                    // the column left of the range is outside the band for all 64 rows ...
                leftH = NEG_INF; leftM = leftS = leftL = 0;
                diagH = NEG_INF; diagM = diagS = diagL = 0;
                if (lane == 0 && bandi == 0) {                // ... but row -1 is the boundary row, which keeps its values
                    diagH = row_pen ? -(open + (jlo - 1) * ext) : 0;
                    diagL = row_pen ? jlo : 0;
                }
                if (lane == 0 && bandi > 0 && jlo - 1 <= pjhi) {   // ... and (64 b - 1, jlo - 1) may lie inside the band: lane 0's first diagonal source
                    diagH = bound[8LL * (jlo - 1) + 0];
                    if (STATS) { diagM = bound[8LL * (jlo - 1) + 2]; diagS = bound[8LL * (jlo - 1) + 3]; diagL = bound[8LL * (jlo - 1) + 4]; }
                }
            }
            if (lane == 0 && bandi > 0 && jlo > 0) {          // the read-ahead of the previous band's row starts at jlo
                const bool in = jlo <= pjhi;
                pb[0] = in ? bound[8LL * jlo + 0] : NEG_INF; pb[1] = in ? bound[8LL * jlo + 1] : NEG_INF;
                if (STATS) { for (int x = 2; x < 8; ++x) pb[x] = in ? bound[8LL * jlo + x] : 0; }
            }
        }
        // two-stage LDS pipeline: symbol of column j+2, score of column j+1
        int sym_n = rs[max(0, min(rl, jlo + 1 - lane))];
        int s_n = mrow[rs[max(0, min(rl, jlo - lane))]];

        const int steps = jhi + 1 + 63;
        // score only, no band, every lane on a real cell away from the last column: the 25-instruction step (see do_step_lean below)
        const bool lean_band = !STATS && !OUT && band_w < 0 && a.bits != 8 && a.bits != 16 && bandi + 1 < nbands;
        for (int t = jlo; t < steps; ++t) {
            if (lean_band && t >= 63 && t <= rl - 2) {
                const int j = t - lane;
                const int s = s_n;
                s_n = mrow[sym_n];
                sym_n = rs[j + 2];
                int upH = lane_up(oH), upF = lane_up(oF);
                if (lane == 0) {
                    if (bandi == 0) { upH = row_pen ? -(open + j * ext) : 0; upF = NEG_INF; }
                    else { upH = pb[0]; upF = pb[1]; pb[0] = bound[8LL * (j + 1) + 0]; pb[1] = bound[8LL * (j + 1) + 1]; }
                }
                const int F = max(upH - open, upF - ext);
                E = max(leftH - open, E - ext);
                int H = max(diagH + s, max(E, F));
                if (mode == PMX_MODE_SW) {
                    H = max(H, 0);
                    if (H > best_sw.H) { best_sw.H = H; best_sw.i = i; best_sw.j = j; }
                }
                diagH = upH; leftH = H; oH = H; oF = F;
                if (lane == 63) { bound[8LL * j + 0] = H; bound[8LL * j + 1] = F; }
                continue;
            }
            const int j = t - lane;
            const int s = s_n;                       // score for column j
            const int rsym_cur = rs[max(0, min(rl, j))];
            s_n = mrow[sym_n];
            sym_n = rs[max(0, min(rl, j + 2))];
            // --- values of the row above for column j (produced one step ago by lane-1) ---
            int upH = lane_up(oH), upF = lane_up(oF);
            int upHM = 0, upHS = 0, upHL = 0, upFM = 0, upFS = 0, upFL = 0;
            if (STATS) {
                upHM = lane_up(oHM); upHS = lane_up(oHS); upHL = lane_up(oHL);
                upFM = lane_up(oFM); upFS = lane_up(oFS); upFL = lane_up(oFL);
            }
            const bool active = row_ok && j >= jlo && j <= jhi;
            if (lane == 0 && j <= jhi) {
                if (bandi == 0) {
                    upH = row_pen ? -(open + j * ext) : 0;
                    upF = NEG_INF;
                    upHM = upHS = 0; upHL = row_pen ? j + 1 : 0;
                    upFM = upFS = upFL = 0;
                    hmin = min(hmin, upH);
                } else {
                    upH = pb[0]; upF = pb[1];
                    if (STATS) { upHM = pb[2]; upHS = pb[3]; upHL = pb[4]; upFM = pb[5]; upFS = pb[6]; upFL = pb[7]; }
                    if (j + 1 < rl) {
                        if (j + 1 <= pjhi) {
                            pb[0] = bound[8LL * (j + 1) + 0]; pb[1] = bound[8LL * (j + 1) + 1];
                            if (STATS) { for (int x = 2; x < 8; ++x) pb[x] = bound[8LL * (j + 1) + x]; }
                        } else {                              // (banded) the previous band never reached that column: outside the band
                            pb[0] = NEG_INF; pb[1] = NEG_INF;
                            if (STATS) { for (int x = 2; x < 8; ++x) pb[x] = 0; }
                        }
                    }
                }
            }
            if (active) {
                const int rsym = rsym_cur;
                int T = 0;
                int F, FM, FS, FL;
                {
                    const int F_opn = upH - open, F_ext = upF - ext;
                    if (F_opn > F_ext) { F = F_opn; FM = upHM; FS = upHS; FL = upHL + 1; T |= T_DIAG_F; }
                    else { F = F_ext; FM = upFM; FS = upFS; FL = upFL + 1; T |= T_DEL_F; }
                    if (F < NEG_INF) F = NEG_INF;
                }
                {
                    const int E_opn = leftH - open, E_ext = E - ext;
                    if (E_opn > E_ext) { E = E_opn; EM = leftM; ES = leftS; EL = leftL + 1; T |= T_DIAG_E; }
                    else { E = E_ext; EL = EL + 1; T |= T_INS_E; }
                    if (E < NEG_INF) E = NEG_INF;
                }
                const int H_dag = diagH + s;
                int H, HM, HS, HL;
                if (H_dag >= E && H_dag >= F) {
                    H = H_dag; HM = diagM + (qsym == rsym); HS = diagS + (s > 0); HL = diagL + 1; T |= T_DIAG;
                } else if (F >= E) {
                    H = F; HM = FM; HS = FS; HL = FL; T |= T_DEL;
                } else {
                    H = E; HM = EM; HS = ES; HL = EL; T |= T_INS;
                }
                if (mode == PMX_MODE_SW && H <= 0) {
                    H = 0; HM = HS = HL = 0; T &= ~(T_INS | T_DEL | T_DIAG);
                }
                if (band_w >= 0 && (j - i - band_d > band_w || j - i - band_d < -band_w)) {
                    H = NEG_INF; E = NEG_INF; F = NEG_INF; HM = HS = HL = 0;
                }
                hmax = max(hmax, H);
                if (band_w < 0) hmin = min(hmin, H);

                if (OUT) {
                    const long long c = tab0 + (long long)i * rl + j;
                    if (a.score_table) a.score_table[c] = H;
                    if (STATS) {
                        if (a.matches_table) a.matches_table[c] = HM;
                        if (a.similar_table) a.similar_table[c] = HS;
                        if (a.length_table) a.length_table[c] = HL;
                    }
                    if (tlds) ts[lane * rl + j] = (unsigned char)T;
                    else if (a.trace_table) a.trace_table[c] = (int8_t)T;
                    if (i == ql - 1) {
                        if (a.score_row) a.score_row[row0 + j] = H;
                        if (STATS) {
                            if (a.matches_row) a.matches_row[row0 + j] = HM;
                            if (a.similar_row) a.similar_row[row0 + j] = HS;
                            if (a.length_row) a.length_row[row0 + j] = HL;
                        }
                    }
                    if (j == rl - 1) {
                        if (a.score_col) a.score_col[col0 + i] = H;
                        if (STATS) {
                            if (a.matches_col) a.matches_col[col0 + i] = HM;
                            if (a.similar_col) a.similar_col[col0 + i] = HS;
                            if (a.length_col) a.length_col[col0 + i] = HL;
                        }
                    }
                }
                const Cand c = {H, i, j, HM, HS, HL};
                if (mode == PMX_MODE_SW) { if (better_sw(c, best_sw)) best_sw = c; }
                else {
                    if (i == ql - 1 && j == rl - 1) corner = c;
                    if (i == ql - 1 && s2_end && c.H > best_row.H) best_row = c;   // j ascends: first max kept
                    if (j == rl - 1 && s1_end && c.H > best_col.H) best_col = c;   // i ascends: first max kept
                }
                // state for the next column / the lane below
                diagH = upH; diagM = upHM; diagS = upHS; diagL = upHL;
                leftH = H; leftM = HM; leftS = HS; leftL = HL;
                oH = H; oF = F; oHM = HM; oHS = HS; oHL = HL; oFM = FM; oFS = FS; oFL = FL;
                if (lane == 63 && bandi + 1 < nbands) {
                    bound[8LL * j + 0] = H; bound[8LL * j + 1] = F;
                    if (STATS) {
                        bound[8LL * j + 2] = HM; bound[8LL * j + 3] = HS; bound[8LL * j + 4] = HL;
                        bound[8LL * j + 5] = FM; bound[8LL * j + 6] = FS; bound[8LL * j + 7] = FL;
                    }
                }
            }
        }
        if (tlds) {   // flush the staged band: rows of a band are contiguous in the row-major table
            __syncthreads();
            const int nbytes = min(64, ql - bandi * 64) * rl;
            const int head = min(nbytes, (int)((16 - ((uintptr_t)tdst & 15)) & 15));
            if (lane < head) tdst[lane] = (int8_t)ts[lane];
            int k = head + lane * 16;
            for (; k + 16 <= nbytes; k += 64 * 16)
                *reinterpret_cast<int4 *>(tdst + k) = *reinterpret_cast<const int4 *>(ts + k);
            const int tail0 = head + ((nbytes - head) / 16) * 16;
            for (int b = tail0 + lane; b < nbytes; b += 64) tdst[b] = (int8_t)ts[b];
            __syncthreads();
        }
        // the next band's lane 0 reads what this band's lane 63 stored: drain the stores, drop stale L1 lines
        if (bandi + 1 < nbands) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }

    // ---- wave reduction ---------------------------------------------------------------
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Cand o = shfl_cand(best_sw, off); if (better_sw(o, best_sw)) best_sw = o;
        o = shfl_cand(best_row, off); if (better_minj(o, best_row)) best_row = o;
        o = shfl_cand(best_col, off); if (better_mini(o, best_col)) best_col = o;
        o = shfl_cand(corner, off); if (o.H > corner.H) corner = o;
        hmin = min(hmin, __shfl_xor(hmin, off, 64));
        hmax = max(hmax, __shfl_xor(hmax, off, 64));
    }
    if (lane == 0) {
        Cand res;
        if (mode == PMX_MODE_SW) res = best_sw;
        else if (mode == PMX_MODE_NW || (!s1_end && !s2_end)) { res = corner; res.i = ql - 1; res.j = rl - 1; }   // (also when a band excludes the corner: -inf there)
        else {
            res = best_row;                                   // NEG_INF when the ref end is not free
            if (s1_end && best_col.H > res.H) res = best_col; // last column must be strictly better
        }
        pmx_record_t rec;
        rec.score = res.H; rec.end_query = res.i; rec.end_ref = res.j; rec.flags = 0;
        if (a.bits == 8 && (hmax > 127 || hmin < -128)) rec.flags |= PMX_FLAG_SATURATED;
        if (a.bits == 16 && (hmax > 32767 || hmin < -32768)) rec.flags |= PMX_FLAG_SATURATED;
        if (a.rec) a.rec[pair] = rec;
        if (STATS && a.stats) { pmx_stats_t st = {res.M, res.S, res.L}; a.stats[pair] = st; }
    }
}


// MW (few long pairs): the workgroup has several waves and they share ONE pair -- wave w takes the 64-row bands w, w + W, ... of
// the query and the bands run as a software pipeline: band b starts 66 (+ its column offset) steps after band b - 1, so what its
// lane 0 reads from the boundary buffer was written by band b - 1's lane 63 in an earlier step; every step ends with a
// workgroup barrier (that is the whole synchronisation: no flags, no spinning), the start step of every band is computed up front.
// A single long pair -- Aligner::align() on two 10 kbp sequences -- no longer crawls through 157 bands on one wave.
// (The one-wave-per-pair kernel above keeps its own copy of the sweep: this kernel run with a single wave is a third slower on
// batches -- 2000 banded pairs 5.0 vs 7.0 ms -- because every step passes through the schedule test and a barrier.  The band set-up
// and both steps are written out inside the step loop: as lambdas they left the end-cell candidates in scratch memory.)
template <bool STATS, bool OUT>
__global__ __launch_bounds__(1024)
void pmx_general_mw_kernel(const PmxGeneralArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int16_t *mat = reinterpret_cast<int16_t *>(lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = (int)(blockDim.x >> 6);
    const int nthr = (int)blockDim.x, tid = threadIdx.x;
    const long long pair = a.index ? a.index[blockIdx.x] : (long long)blockIdx.x;
    const int msize = a.msize;

    for (int i = tid; i < a.mat_rows * msize; i += nthr) mat[i] = a.scores[i];

    long long qb, rb; int ql, rl;
    if (a.qoff) { qb = a.qoff[pair]; ql = (int)(a.qoff[pair + 1] - qb); }
    else { qb = 0; ql = a.shared_qlen; }
    rb = a.roff[pair]; rl = (int)(a.roff[pair + 1] - rb);
    const uint8_t *q = a.qbuf + qb, *r = a.rbuf + rb;
    // mapped reference symbols live in LDS for the whole pair (the sweep reads one per lane and step;
    // fetching them from HBM inside the loop made every step wait for two dependent global loads)
    // (a reference too long for the LDS -- beyond ~160 k symbols -- is mapped into a per-pair HBM scratch instead and read through
    //  the caches: slower per step, but no length limit; the reference has none either)
    unsigned char *rs_lds = lds + (((size_t)a.mat_rows * msize * 2 + 15) & ~(size_t)15);
    unsigned char *rs = a.rs_scratch ? a.rs_scratch + (long long)blockIdx.x * a.rs_stride : rs_lds;
    for (int j = tid; j < rl; j += nthr) rs[j] = a.mapper[r[j]];
    for (int j = rl + tid; j < rl + 4; j += nthr) rs[j] = 0;
    if (a.rs_scratch) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent"); }
    __syncthreads();
    // optional staging area for one band of trace bytes (64 rows x rl), flushed with wide stores
    unsigned char *tstage = rs_lds + (((size_t)a.max_rlen + 8 + 15) & ~(size_t)15);
    const bool tlds = false;                                   // (no LDS staging of trace bytes in this form)
    const long long tab0 = a.tab_off ? a.tab_off[pair] : 0;
    const long long row0 = rb;                                // row outputs packed like the references (a one-pair call has rb == 0;
    const long long col0 = a.qoff ? qb : pair * (long long)ql;   //  a one-pair CHUNK of a batch keeps its place), columns like the queries
    int32_t *bound = a.bound + (long long)blockIdx.x * a.bound_stride;

    const int mode = a.mode, open = a.open, ext = a.ext, band_w = a.band;
    const int band_d = a.diag ? a.diag[pair] : 0;             // band centre: cells with |(j - i) - band_d| > band_w are excluded
    const bool s1_beg = mode == PMX_MODE_SG && (a.sg_flags & PMX_SG_QB);
    const bool s1_end = mode == PMX_MODE_SG && (a.sg_flags & PMX_SG_QE);
    const bool s2_beg = mode == PMX_MODE_SG && (a.sg_flags & PMX_SG_DB);
    const bool s2_end = mode == PMX_MODE_SG && (a.sg_flags & PMX_SG_DE);
    const bool col_pen = mode == PMX_MODE_NW || (mode == PMX_MODE_SG && !s1_beg);  // H(i,-1) penalised
    const bool row_pen = mode == PMX_MODE_NW || (mode == PMX_MODE_SG && !s2_beg);  // H(-1,j) penalised

    Cand best_sw = {NEG_INF, 0, 0, 0, 0, 0};      // sw: global best
    Cand best_row = {NEG_INF, 0, 0, 0, 0, 0};     // sg: best of the last row
    Cand best_col = {NEG_INF, 0, 0, 0, 0, 0};     // sg: best of the last column
    Cand corner = {NEG_INF, 0, 0, 0, 0, 0};
    int hmin = 0, hmax = 0;

    const int nbands = (ql + 63) / 64;
    // ---- per-band state (one band at a time per wave) ----
    int bandi = 0, i = 0, qsym = 0; bool row_ok = false;
    const int16_t *mrow = mat;
    int leftH = 0, leftM = 0, leftS = 0, leftL = 0, diagH = 0, diagM = 0, diagS = 0, diagL = 0;
    int E = NEG_INF, EM = 0, ES = 0, EL = 0;
    int oH = NEG_INF, oF = NEG_INF, oHM = 0, oHS = 0, oHL = 0, oFM = 0, oFS = 0, oFL = 0;
    int8_t *tdst = nullptr; unsigned char *ts = nullptr;
    int pb0 = 0, pb1 = 0, pb2 = 0, pb3 = 0, pb4 = 0, pb5 = 0, pb6 = 0, pb7 = 0;      // (named scalars: an array captured by the lambdas lands in scratch)
    int jlo = 0, jhi = rl - 1, pjhi = rl - 1, sym_n = 0, s_n = 0;
    // column range of band b (banded calls sweep only what the band's rows can reach)
    auto band_range = [&](int b, int &lo, int &hi) __attribute__((always_inline)) {
        lo = 0; hi = rl - 1;
        if (band_w >= 0) { lo = max(0, b * 64 + band_d - band_w); hi = min(rl - 1, b * 64 + 63 + band_d + band_w); }
    };
    // Score only, no band, every lane on a real cell away from the last column (steps 63 .. rlen - 2 of a band that is not the
    // last): none of the checked step's questions is open -- 25 instructions instead of ~100 (20 kbp x 20 kbp 358 -> 218 ms).
    // (Measured and dropped: fetching the row above 64 columns at a time one block ahead, 218 -> 267 ms; a barrier only every 16th
    // step, 218 -> 207 ms but slower banded batches: sixteen waves on one CU are bound by its four SIMDs, not by the barrier.)
    const bool lean_kind = !STATS && !OUT && band_w < 0 && a.bits != 8 && a.bits != 16;
    // start step of every band: 66 steps (+ the shift of its column range) behind the band above, and not before the wave
    // that owns it has finished its previous band
    int *sched = reinterpret_cast<int *>(lds + a.mw_sched_off);        // [nbands] start steps, then the end step
    if (tid == 0) {
        int end = 0;
        for (int b = 0; b < nbands; ++b) {
            int lo, hi, plo = 0, phi;
            band_range(b, lo, hi);
            if (b) band_range(b - 1, plo, phi);
            int st = b ? sched[b - 1] + 66 + max(0, lo - plo) : 0;
            if (b >= W) { int l2, h2; band_range(b - W, l2, h2); st = max(st, sched[b - W] + (h2 + 1 + 63 - l2)); }
            sched[b] = st;
            end = max(end, st + (hi + 1 + 63 - lo));
        }
        sched[nbands] = end;
    }
    __syncthreads();
    const int end = sched[nbands];
    int cb = wave, t = 0, t_end = 0, start = cb < nbands ? sched[cb] : 0x7fffffff;
    for (int gs = 0; gs < end; ++gs) {
        if (gs == start) {
            const int b = cb;
            bandi = b;
            i = bandi * 64 + lane;
            row_ok = i < ql;
            qsym = row_ok ? a.mapper[q[i]] : 0;
            mrow = mat + (a.pssm ? (row_ok ? i : 0) : qsym) * msize;
            // left boundary H(i,-1) and the diagonal seed H(i-1,-1)
            leftH = col_pen ? -(open + i * ext) : 0;
            leftM = 0; leftS = 0; leftL = col_pen ? i + 1 : 0;
            diagH = (i == 0) ? 0 : (col_pen ? -(open + (i - 1) * ext) : 0);
            diagM = 0; diagS = 0; diagL = (i == 0) ? 0 : (col_pen ? i : 0);
            if (row_ok) { hmin = min(hmin, leftH); }
            E = NEG_INF; EM = 0; ES = 0; EL = 0;
            // what this lane hands to the lane below: H(i,j), F(i,j) and stats
            oH = NEG_INF; oF = NEG_INF; oHM = 0; oHS = 0; oHL = 0; oFM = 0; oFS = 0; oFL = 0;

            tdst = nullptr; ts = nullptr;
            if (tlds) {
                tdst = a.trace_table + tab0 + (long long)bandi * 64 * rl;
                ts = tstage + ((uintptr_t)tdst & 15);          // same alignment mod 16 as the destination
            }
            // lane 0 reads the previous band's last row one column ahead of its use
            pb0 = pb1 = pb2 = pb3 = pb4 = pb5 = pb6 = pb7 = 0;
            if (lane == 0 && bandi > 0) {
                pb0 = bound[0]; pb1 = bound[1];
                if (STATS) { pb2 = bound[2]; pb3 = bound[3]; pb4 = bound[4]; pb5 = bound[5]; pb6 = bound[6]; pb7 = bound[7]; }
            }
            // Banded: only the columns some row of this 64-row band can reach are swept -- rows 64 b .. 64 b + 63 see columns
            // [64 b + d - w, 64 b + 63 + d + w]; everything left and right of that range is outside the band for every row of the band,
            // so the sweep may start there with "minus infinity" to its left.  Work per pair: qlen x (2 w + 190) instead of qlen x rlen.
            jlo = 0; jhi = rl - 1; pjhi = rl - 1;                // this band's column range; the previous band's last column
            if (band_w >= 0) {
                jlo = max(0, bandi * 64 + band_d - band_w);
                jhi = min(rl - 1, bandi * 64 + 63 + band_d + band_w);
                pjhi = min(rl - 1, bandi * 64 - 1 + band_d + band_w);
                if (jlo > 0) {                                    // the column left of the range is outside the band for all 64 rows ...
                    leftH = NEG_INF; leftM = leftS = leftL = 0;
                    diagH = NEG_INF; diagM = diagS = diagL = 0;
                    if (lane == 0 && bandi == 0) {                // ... but row -1 is the boundary row, which keeps its values
                        diagH = row_pen ? -(open + (jlo - 1) * ext) : 0;
                        diagL = row_pen ? jlo : 0;
                    }
                    if (lane == 0 && bandi > 0 && jlo - 1 <= pjhi) {   // ... and (64 b - 1, jlo - 1) may lie inside the band: lane 0's first diagonal source
                        diagH = bound[8LL * (jlo - 1) + 0];
                        if (STATS) { diagM = bound[8LL * (jlo - 1) + 2]; diagS = bound[8LL * (jlo - 1) + 3]; diagL = bound[8LL * (jlo - 1) + 4]; }
                    }
                }
                if (lane == 0 && bandi > 0 && jlo > 0) {          // the read-ahead of the previous band's row starts at jlo
                    const bool in = jlo <= pjhi;
                    pb0 = in ? bound[8LL * jlo + 0] : NEG_INF; pb1 = in ? bound[8LL * jlo + 1] : NEG_INF;
                    if (STATS) { pb2 = in ? bound[8LL * jlo + 2] : 0; pb3 = in ? bound[8LL * jlo + 3] : 0; pb4 = in ? bound[8LL * jlo + 4] : 0;
                                 pb5 = in ? bound[8LL * jlo + 5] : 0; pb6 = in ? bound[8LL * jlo + 6] : 0; pb7 = in ? bound[8LL * jlo + 7] : 0; }
                }
            }
            // two-stage LDS pipeline: symbol of column j+2, score of column j+1
            sym_n = rs[max(0, min(rl, jlo + 1 - lane))];
            s_n = mrow[rs[max(0, min(rl, jlo - lane))]];
            t = jlo; t_end = jhi + 1 + 63;
        }
        if (gs >= start) {
            if (lean_kind && bandi + 1 < nbands && t >= 63 && t <= rl - 2) {
                const int j = t - lane;
                const int s = s_n;
                s_n = mrow[sym_n];
                sym_n = rs[j + 2];
                int upH = lane_up(oH), upF = lane_up(oF);
                if (lane == 0) {
                    if (bandi == 0) { upH = row_pen ? -(open + j * ext) : 0; upF = NEG_INF; }
                    else { upH = pb0; upF = pb1; pb0 = bound[8LL * (j + 1) + 0]; pb1 = bound[8LL * (j + 1) + 1]; }
                }
                const int F = max(upH - open, upF - ext);
                E = max(leftH - open, E - ext);
                int H = max(diagH + s, max(E, F));
                if (mode == PMX_MODE_SW) {
                    H = max(H, 0);
                    if (H > best_sw.H) { best_sw.H = H; best_sw.i = i; best_sw.j = j; }     // (j ascends in a lane: the first maximum is kept)
                }
                diagH = upH; leftH = H; oH = H; oF = F;
                if (lane == 63) { bound[8LL * j + 0] = H; bound[8LL * j + 1] = F; }
            } else {
                const int j = t - lane;
                const int s = s_n;                       // score for column j
                const int rsym_cur = rs[max(0, min(rl, j))];
                s_n = mrow[sym_n];
                sym_n = rs[max(0, min(rl, j + 2))];
                // --- values of the row above for column j (produced one step ago by lane-1) ---
                int upH = lane_up(oH), upF = lane_up(oF);
                int upHM = 0, upHS = 0, upHL = 0, upFM = 0, upFS = 0, upFL = 0;
                if (STATS) {
                    upHM = lane_up(oHM); upHS = lane_up(oHS); upHL = lane_up(oHL);
                    upFM = lane_up(oFM); upFS = lane_up(oFS); upFL = lane_up(oFL);
                }
                const bool active = row_ok && j >= jlo && j <= jhi;
                if (lane == 0 && j <= jhi) {
                    if (bandi == 0) {
                        upH = row_pen ? -(open + j * ext) : 0;
                        upF = NEG_INF;
                        upHM = upHS = 0; upHL = row_pen ? j + 1 : 0;
                        upFM = upFS = upFL = 0;
                        hmin = min(hmin, upH);
                    } else {
                        upH = pb0; upF = pb1;
                        if (STATS) { upHM = pb2; upHS = pb3; upHL = pb4; upFM = pb5; upFS = pb6; upFL = pb7; }
                        if (j + 1 < rl) {
                            if (j + 1 <= pjhi) {
                                pb0 = bound[8LL * (j + 1) + 0]; pb1 = bound[8LL * (j + 1) + 1];
                                if (STATS) { pb2 = bound[8LL * (j + 1) + 2]; pb3 = bound[8LL * (j + 1) + 3]; pb4 = bound[8LL * (j + 1) + 4];
                                             pb5 = bound[8LL * (j + 1) + 5]; pb6 = bound[8LL * (j + 1) + 6]; pb7 = bound[8LL * (j + 1) + 7]; }
                            } else {                              // (banded) the previous band never reached that column: outside the band
                                pb0 = NEG_INF; pb1 = NEG_INF;
                                if (STATS) { pb2 = pb3 = pb4 = pb5 = pb6 = pb7 = 0; }
                            }
                        }
                    }
                }
                if (active) {
                    const int rsym = rsym_cur;
                    int T = 0;
                    int F, FM, FS, FL;
                    {
                        const int F_opn = upH - open, F_ext = upF - ext;
                        if (F_opn > F_ext) { F = F_opn; FM = upHM; FS = upHS; FL = upHL + 1; T |= T_DIAG_F; }
                        else { F = F_ext; FM = upFM; FS = upFS; FL = upFL + 1; T |= T_DEL_F; }
                        if (F < NEG_INF) F = NEG_INF;
                    }
                    {
                        const int E_opn = leftH - open, E_ext = E - ext;
                        if (E_opn > E_ext) { E = E_opn; EM = leftM; ES = leftS; EL = leftL + 1; T |= T_DIAG_E; }
                        else { E = E_ext; EL = EL + 1; T |= T_INS_E; }
                        if (E < NEG_INF) E = NEG_INF;
                    }
                    const int H_dag = diagH + s;
                    int H, HM, HS, HL;
                    if (H_dag >= E && H_dag >= F) {
                        H = H_dag; HM = diagM + (qsym == rsym); HS = diagS + (s > 0); HL = diagL + 1; T |= T_DIAG;
                    } else if (F >= E) {
                        H = F; HM = FM; HS = FS; HL = FL; T |= T_DEL;
                    } else {
                        H = E; HM = EM; HS = ES; HL = EL; T |= T_INS;
                    }
                    if (mode == PMX_MODE_SW && H <= 0) {
                        H = 0; HM = HS = HL = 0; T &= ~(T_INS | T_DEL | T_DIAG);
                    }
                    if (band_w >= 0 && (j - i - band_d > band_w || j - i - band_d < -band_w)) {
                        H = NEG_INF; E = NEG_INF; F = NEG_INF; HM = HS = HL = 0;
                    }
                    hmax = max(hmax, H);
                    if (band_w < 0) hmin = min(hmin, H);

                    if (OUT) {
                        const long long c = tab0 + (long long)i * rl + j;
                        if (a.score_table) a.score_table[c] = H;
                        if (STATS) {
                            if (a.matches_table) a.matches_table[c] = HM;
                            if (a.similar_table) a.similar_table[c] = HS;
                            if (a.length_table) a.length_table[c] = HL;
                        }
                        if (tlds) ts[lane * rl + j] = (unsigned char)T;
                        else if (a.trace_table) a.trace_table[c] = (int8_t)T;
                        if (i == ql - 1) {
                            if (a.score_row) a.score_row[row0 + j] = H;
                            if (STATS) {
                                if (a.matches_row) a.matches_row[row0 + j] = HM;
                                if (a.similar_row) a.similar_row[row0 + j] = HS;
                                if (a.length_row) a.length_row[row0 + j] = HL;
                            }
                        }
                        if (j == rl - 1) {
                            if (a.score_col) a.score_col[col0 + i] = H;
                            if (STATS) {
                                if (a.matches_col) a.matches_col[col0 + i] = HM;
                                if (a.similar_col) a.similar_col[col0 + i] = HS;
                                if (a.length_col) a.length_col[col0 + i] = HL;
                            }
                        }
                    }
                    const Cand c = {H, i, j, HM, HS, HL};
                    if (mode == PMX_MODE_SW) { if (better_sw(c, best_sw)) best_sw = c; }
                    else {
                        if (i == ql - 1 && j == rl - 1) corner = c;
                        if (i == ql - 1 && s2_end && c.H > best_row.H) best_row = c;   // j ascends: first max kept
                        if (j == rl - 1 && s1_end && c.H > best_col.H) best_col = c;   // i ascends: first max kept
                    }
                    // state for the next column / the lane below
                    diagH = upH; diagM = upHM; diagS = upHS; diagL = upHL;
                    leftH = H; leftM = HM; leftS = HS; leftL = HL;
                    oH = H; oF = F; oHM = HM; oHS = HS; oHL = HL; oFM = FM; oFS = FS; oFL = FL;
                    if (lane == 63 && bandi + 1 < nbands) {
                        bound[8LL * j + 0] = H; bound[8LL * j + 1] = F;
                        if (STATS) {
                            bound[8LL * j + 2] = HM; bound[8LL * j + 3] = HS; bound[8LL * j + 4] = HL;
                            bound[8LL * j + 5] = FM; bound[8LL * j + 6] = FS; bound[8LL * j + 7] = FL;
                        }
                    }
                }
            }
            if (++t == t_end) { cb += W; start = cb < nbands ? sched[cb] : 0x7fffffff; }
        }
        __syncthreads();                                   // (stores of this step are visible to the whole workgroup after it)
    }

    // ---- wave reduction ---------------------------------------------------------------
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Cand o = shfl_cand(best_sw, off); if (better_sw(o, best_sw)) best_sw = o;
        o = shfl_cand(best_row, off); if (better_minj(o, best_row)) best_row = o;
        o = shfl_cand(best_col, off); if (better_mini(o, best_col)) best_col = o;
        o = shfl_cand(corner, off); if (o.H > corner.H) corner = o;
        hmin = min(hmin, __shfl_xor(hmin, off, 64));
        hmax = max(hmax, __shfl_xor(hmax, off, 64));
    }
    {                                                          // wave 0 collects the other waves' candidates
        Cand *cw = reinterpret_cast<Cand *>(lds + a.mw_sched_off + ((size_t)(nbands + 1) * 4 + 15) / 16 * 16);      // [W][4]
        int *mm = reinterpret_cast<int *>(cw + 4 * W);                                                       // [W][2]
        if (lane == 0) { cw[4 * wave + 0] = best_sw; cw[4 * wave + 1] = best_row; cw[4 * wave + 2] = best_col; cw[4 * wave + 3] = corner;
                         mm[2 * wave] = hmin; mm[2 * wave + 1] = hmax; }
        __syncthreads();
        if (tid == 0) {
            for (int w2 = 1; w2 < W; ++w2) {
                if (better_sw(cw[4 * w2 + 0], best_sw)) best_sw = cw[4 * w2 + 0];
                if (better_minj(cw[4 * w2 + 1], best_row)) best_row = cw[4 * w2 + 1];
                if (better_mini(cw[4 * w2 + 2], best_col)) best_col = cw[4 * w2 + 2];
                if (cw[4 * w2 + 3].H > corner.H) corner = cw[4 * w2 + 3];
                hmin = min(hmin, mm[2 * w2]); hmax = max(hmax, mm[2 * w2 + 1]);
            }
        }
    }
    if (tid == 0) {
        Cand res;
        if (mode == PMX_MODE_SW) res = best_sw;
        else if (mode == PMX_MODE_NW || (!s1_end && !s2_end)) { res = corner; res.i = ql - 1; res.j = rl - 1; }   // (also when a band excludes the corner: -inf there)
        else {
            res = best_row;                                   // NEG_INF when the ref end is not free
            if (s1_end && best_col.H > res.H) res = best_col; // last column must be strictly better
        }
        pmx_record_t rec;
        rec.score = res.H; rec.end_query = res.i; rec.end_ref = res.j; rec.flags = 0;
        if (a.bits == 8 && (hmax > 127 || hmin < -128)) rec.flags |= PMX_FLAG_SATURATED;
        if (a.bits == 16 && (hmax > 32767 || hmin < -32768)) rec.flags |= PMX_FLAG_SATURATED;
        if (a.rec) a.rec[pair] = rec;
        if (STATS && a.stats) { pmx_stats_t st = {res.M, res.S, res.L}; a.stats[pair] = st; }
    }
}

#include <mutex>
#include <set>
#include <utility>
int pmx_ensure_lds_attr(const void *kernel, int bytes)
{
    static std::mutex mx;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return -(int)e;
    std::lock_guard<std::mutex> lk(mx);
    if (done.count({kernel, dev})) return 0;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return -(int)e;
    done.insert({kernel, dev});
    return 0;
}

int pmx_launch_general(const PmxGeneralArgs &a_in, bool want_stats, hipStream_t stream)
{
    if (a_in.n <= 0) return 0;
    PmxGeneralArgs a = a_in;
    const size_t mat_bytes = ((size_t)a.mat_rows * a.msize * 2 + 15) & ~(size_t)15;
    size_t lds = mat_bytes + (((size_t)a.max_rlen + 8 + 15) & ~(size_t)15);
    if (lds > 160 * 1024) {                       // the reference does not fit the LDS: the caller's HBM scratch holds the mapped symbols
        if (!a.rs_scratch || mat_bytes > 160 * 1024) return 1;
        lds = mat_bytes;
    } else a.rs_scratch = nullptr;
    // Few long pairs: several waves share a pair (see MW in the kernel).  W = bands of the longest query, at most 16.
    const int nbands_max = (a.max_qlen + 63) / 64;
    int W = 1;
    // (a handful of pairs: always -- nothing else would fill the chip; up to PMX_MW_MAX_PAIRS: when a band's sweep is long against the
    //  66-step stagger of the pipeline, i.e. unbanded pairs with references of >= 1024 symbols)
    const bool few = a.n <= 64 || (a.n <= PMX_MW_MAX_PAIRS && a.band < 0 && a.max_rlen >= 1024);
    if (few && nbands_max >= 3 && nbands_max <= 8192 && !pmx_env("PMX_GENERAL_ONE_WAVE")) W = nbands_max < 16 ? nbands_max : 16;
    if (W > 1) {
        a.mw_sched_off = (int)((lds + 15) & ~(size_t)15);
        lds = (size_t)a.mw_sched_off + (((size_t)(nbands_max + 1) * 4 + 15) & ~(size_t)15) + (size_t)W * (4 * sizeof(int) * 6 + 8) + 64;
        if (lds > 160 * 1024) W = 1;
    }
    a.trace_lds = 0;
    if (W == 1) {
        lds = mat_bytes + (a.rs_scratch ? 0 : (((size_t)a.max_rlen + 8 + 15) & ~(size_t)15));
        if (!a.rs_scratch && a.trace_table && lds + (size_t)64 * a.max_rlen + 32 <= 96 * 1024) { a.trace_lds = 1; lds += (size_t)64 * a.max_rlen + 32; }
    }
    const bool out = a.score_table || a.trace_table || a.score_row || a.score_col ||
                     a.matches_table || a.similar_table || a.length_table;
    {
        const void *fns[8] = {(const void *)&pmx_general_kernel<true, true>, (const void *)&pmx_general_kernel<true, false>,
                              (const void *)&pmx_general_kernel<false, true>, (const void *)&pmx_general_kernel<false, false>,
                              (const void *)&pmx_general_mw_kernel<true, true>, (const void *)&pmx_general_mw_kernel<true, false>,
                              (const void *)&pmx_general_mw_kernel<false, true>, (const void *)&pmx_general_mw_kernel<false, false>};
        for (const void *f : fns) { const int rc = pmx_ensure_lds_attr(f); if (rc) return rc; }
    }
    dim3 grid((unsigned)a.n), block(64 * W);
    if (W > 1) {
        if (want_stats) {
            if (out) hipLaunchKernelGGL((pmx_general_mw_kernel<true, true>), grid, block, lds, stream, a);
            else     hipLaunchKernelGGL((pmx_general_mw_kernel<true, false>), grid, block, lds, stream, a);
        } else {
            if (out) hipLaunchKernelGGL((pmx_general_mw_kernel<false, true>), grid, block, lds, stream, a);
            else     hipLaunchKernelGGL((pmx_general_mw_kernel<false, false>), grid, block, lds, stream, a);
        }
    } else if (want_stats) {
        if (out) hipLaunchKernelGGL((pmx_general_kernel<true, true>), grid, block, lds, stream, a);
        else     hipLaunchKernelGGL((pmx_general_kernel<true, false>), grid, block, lds, stream, a);
    } else {
        if (out) hipLaunchKernelGGL((pmx_general_kernel<false, true>), grid, block, lds, stream, a);
        else     hipLaunchKernelGGL((pmx_general_kernel<false, false>), grid, block, lds, stream, a);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// ---- traceback walk on the device: one lane per pair ---------------------------------------
// Restates oracle/pmx_oracle.c:orc_walk (checker) for the product path; emits run-length BAM ops
// in reverse, then reverses in place.  '=' 7, 'X' 8, 'I' 1, 'D' 2.
#define OP_I 1u
#define OP_D 2u
#define OP_EQ 7u
#define OP_X 8u
// state INS (E, consumes a reference character) prints 'D'; state DEL (F, consumes a query
// character) prints 'I' (SAM sense with query = s1, reference = s2).
#define OP_FOR_INS_STATE PMX_BAM_OP_FOR_INS_STATE    // include/pmx_conventions.h
#define OP_FOR_DEL_STATE PMX_BAM_OP_FOR_DEL_STATE    // include/pmx_conventions.h

__global__ void pmx_walk_kernel(const PmxWalkArgs a)
{
    const long long pair = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= a.n) return;
    long long qb; int ql;
    if (a.qoff) { qb = a.qoff[pair]; ql = (int)(a.qoff[pair + 1] - qb); } else { qb = 0; ql = a.shared_qlen; }
    const long long rb = a.roff[pair]; const int rl = (int)(a.roff[pair + 1] - rb);
    const uint8_t *q = a.qbuf + qb, *r = a.rbuf + rb;
    const int8_t *tr = a.trace_table + (a.tab_off ? a.tab_off[pair] : 0);
    uint32_t *ops = a.ops + a.ops_off[pair];
    const pmx_record_t rec = a.rec[pair];
    int i = rec.end_query, j = rec.end_ref, n = 0;
    uint32_t cur_op = 0, cur_len = 0;
    int where = T_DIAG;
    auto emit = [&](uint32_t op) {
        if (op == cur_op) { ++cur_len; }
        else { if (cur_len) ops[n++] = (cur_len << 4) | cur_op; cur_op = op; cur_len = 1; }
    };
    if (a.mode == PMX_MODE_SG) {
        if (i + 1 == ql) { for (int k = rl - 1; k > j; --k) emit(OP_FOR_INS_STATE); }
        else if (j + 1 == rl) { for (int k = ql - 1; k > i; --k) emit(OP_FOR_DEL_STATE); }
    }
    while (i >= 0 || j >= 0) {
        if (i < 0) { if (a.mode == PMX_MODE_SW) break; emit(OP_FOR_INS_STATE); --j; continue; }
        if (j < 0) { if (a.mode == PMX_MODE_SW) break; emit(OP_FOR_DEL_STATE); --i; continue; }
        const int t = tr[(long long)i * rl + j];
        if (where == T_DIAG) {
            if (t & T_DIAG) { emit(a.mapper[q[i]] == a.mapper[r[j]] ? OP_EQ : OP_X); --i; --j; }
            else if (t & T_INS) where = T_INS;
            else if (t & T_DEL) where = T_DEL;
            else break;
        } else if (where == T_INS) {
            emit(OP_FOR_INS_STATE);
            if (t & T_DIAG_E) where = T_DIAG;
            --j;
        } else {
            emit(OP_FOR_DEL_STATE);
            if (t & T_DIAG_F) where = T_DIAG;
            --i;
        }
    }
    if (cur_len) ops[n++] = (cur_len << 4) | cur_op;
    for (int k = 0; k < n / 2; ++k) { uint32_t tmp = ops[k]; ops[k] = ops[n - 1 - k]; ops[n - 1 - k] = tmp; }
    a.nops[pair] = n;
    a.beg[2 * pair] = i + 1; a.beg[2 * pair + 1] = j + 1;
}

int pmx_launch_walk(const PmxWalkArgs &a, hipStream_t stream)
{
    if (a.n <= 0) return 0;
    const unsigned blocks = (unsigned)((a.n + 63) / 64);
    hipLaunchKernelGGL(pmx_walk_kernel, dim3(blocks), dim3(64), 0, stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

__global__ void pmx_collect_saturated_kernel(const pmx_record_t *rec, long long n, int64_t *list, int *count, int mask)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (rec[i].flags & mask)) list[atomicAdd(count, 1)] = i;
}

int pmx_launch_collect_saturated(const pmx_record_t *rec, long long n, int64_t *list, int *count, int mask, hipStream_t stream)
{
    if (n <= 0) return 0;
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int), stream);
    if (e != hipSuccess) return -(int)e;
    hipLaunchKernelGGL(pmx_collect_saturated_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rec, n, list, count, mask);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// CIGAR text on the device: pass 1 measures each pair's text, pass 2 writes it at the caller's offsets
// ("<len><op>" per run, ops = BAM codes, "MIDNSHP=X").
__device__ __forceinline__ int pmx_digits(uint32_t v) { int d = 1; while (v >= 10) { v /= 10; ++d; } return d; }
__global__ void pmx_cigar_textlen_kernel(const uint32_t *ops, const int64_t *ops_off, const int32_t *nops, int32_t *textlen, long long n)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t *src = ops + ops_off[k];
    int len = 0;
    for (int t = 0; t < nops[k]; ++t) len += pmx_digits(src[t] >> 4) + 1;
    textlen[k] = len;
}
// `swap`: the run-time convention switch PMX_CIGAR_SWAP_ID (include/pmx_conventions.h) -- letters I and D exchanged in the text
__global__ void pmx_cigar_render_kernel(const uint32_t *ops, const int64_t *ops_off, const int32_t *nops,
                                        const int64_t *text_off, char *text, long long n, int swap)
{
    const char *letters = swap ? "MDINSHP=X" : "MIDNSHP=X";
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t *src = ops + ops_off[k];
    char *dst = text + text_off[k];
    for (int t = 0; t < nops[k]; ++t) {
        const uint32_t o = src[t];
        uint32_t v = o >> 4;
        const int d = pmx_digits(v);
        for (int x = d - 1; x >= 0; --x) { dst[x] = (char)('0' + v % 10); v /= 10; }
        dst[d] = letters[o & 0xF];
        dst += d + 1;
    }
}
int pmx_launch_cigar_textlen(const uint32_t *ops, const int64_t *ops_off, const int32_t *nops, int32_t *textlen, long long n, hipStream_t stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pmx_cigar_textlen_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, ops, ops_off, nops, textlen, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
int pmx_launch_cigar_render(const uint32_t *ops, const int64_t *ops_off, const int32_t *nops,
                            const int64_t *text_off, char *text, long long n, hipStream_t stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pmx_cigar_render_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, ops, ops_off, nops, text_off, text, n,
                       pmx_cigar_swapped());
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}


// ---- device CIGAR entry: text offsets and render with implicit op slots ---------------------------------
// Eight lanes per pair: every lane renders one run per round ("<len><op>"), its text position is the group's running
// base plus a prefix sum of the widths over the eight lanes.  A pair whose text would cross `capacity` is skipped
// (the caller sees text_off[n] > capacity).
__global__ void pmx_cigar_render_slots_kernel(const uint32_t *ops, const int64_t *qoff, const int64_t *roff, long long ops_base,
                                              const int32_t *nops, const int64_t *text_off, char *text, long long capacity, long long n, int swap)
{
    const char *letters = swap ? "MDINSHP=X" : "MIDNSHP=X";
    const long long k = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int l = threadIdx.x & 7;
    if (k >= n) return;
    if (text_off[k + 1] > capacity) return;
    const int cnt = nops[k];
    // the walk fills a pair's slot of qlen + rlen + 1 entries from its end backwards: the forward list is its last cnt entries
    const uint32_t *src = ops + (qoff[k + 1] + roff[k + 1] + k + 1 - ops_base) - cnt;
    long long base = text_off[k];
    for (int t0 = 0; t0 < cnt; t0 += 8) {
        const int t = t0 + l;
        const uint32_t o = t < cnt ? src[t] : 0u;
        uint32_t v = o >> 4;
        const int d = pmx_digits(v);
        const int w = t < cnt ? d + 1 : 0;
        int inc = w;
#pragma unroll
        for (int s = 1; s < 8; s <<= 1) { const int up = __shfl_up(inc, s, 8); if (l >= s) inc += up; }
        if (t < cnt) {
            char *dst = text + base + inc - w;
            for (int x = d - 1; x >= 0; --x) { dst[x] = (char)('0' + v % 10); v /= 10; }
            dst[d] = letters[o & 0xF];
        }
        base += __shfl(inc, 7, 8);
    }
}

int pmx_launch_cigar_render_slots(const uint32_t *ops, const int64_t *qoff, const int64_t *roff, long long ops_base, const int32_t *nops,
                                  const int64_t *text_off, char *text, long long capacity, long long n, hipStream_t stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pmx_cigar_render_slots_kernel, dim3((unsigned)((n * 8 + 255) / 256)), dim3(256), 0, stream,
                       ops, qoff, roff, ops_base, nops, text_off, text, capacity, n, pmx_cigar_swapped());
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
