// pmx_walkp.hip -- on-device traceback walk over the packed 4-bit trace records the second-generation sweeps write
// (pmx_nwsg16v / pmx_nwsg16q / pmx_nwsg16m / pmx_sw16 VAR 7 / pmx_sw16m with TR).  gfx950 only.
//
// What the reference does per pair on the host inside libparasail (parasail_result_get_cigar / the statistics of the
// one optimal path; /root/reference/src/alignment/mod.rs:390-419, :79-98) runs here for a whole batch: one lane per
// pair, from the captured end position back to the beginning, emitting run-length ops (BAM codes) or counting
// matches / similar / length.
//
// Trace records: R rows of one lane at one step, two pairs: [pair A: R/2 bytes][pair B: R/2 bytes][pad to D dwords],
// byte = a row pair, the even row in the high nibble; nibble = ND NDL EO FO (see pmx_nwsg16.hip).  Records are
// lane-major: the steps of one lane are contiguous, so a path that runs along a row or a diagonal reads one lane's
// stream downwards -- neighbouring cells of a path sit in neighbouring records of one cache line.
// The walk is a chain of dependent reads, so it is made short and wide: EIGHT lanes per pair.  In every iteration lane x
// probes the x-th cell ahead on the current kind of move (diagonal: (i-x, j-x); inside an E gap: (i, j-1-x); inside an F gap:
// (i-1-x, j)); the eight probes are one round trip to memory for the whole wave (one coalesced line per group), a ballot
// turns them into a bit mask, and count-trailing-zeros says how far the path runs straight: up to eight cells per iteration
// instead of one.  No LDS windows, 48 VGPRs: thousands of waves stay resident beside the sweep of the next chunk.
#include "pmx_common.h"
#include <cstdlib>

#define OP_I 1u
#define OP_D 2u
#define OP_EQ 7u
#define OP_X 8u
#define OP_FOR_INS_STATE PMX_BAM_OP_FOR_INS_STATE    // include/pmx_conventions.h
#define OP_FOR_DEL_STATE PMX_BAM_OP_FOR_DEL_STATE    // include/pmx_conventions.h

#define LG 8                     // lanes per pair

// ST (statistics instead of run-length ops) and SW (local alignment: the path ends where the score is used up) are template
// parameters: the walk is VALU-bound beside the sweep it overlaps with, and each mode carries the other modes' bookkeeping otherwise
// (score lookups and prefix sums only SW / ST need, run merging and text lengths only the op mode needs).
template <int G, int R, bool ST, bool SW>
__global__ __launch_bounds__(256)
void pmx_walkp_kernel(const uint8_t *__restrict__ qbuf, const int64_t *__restrict__ qoff, int q_shared,
                      const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff,
                      long long n, const unsigned *__restrict__ perm,
                      const uint8_t *__restrict__ mapper, const int16_t *__restrict__ scores, int msize, int open, int ext,
                      int mode, int Tmax, int top_aligned, const int *__restrict__ blockflag /* per sweep block: 0 = its rows are top-aligned */,
                      pmx_stats_t *__restrict__ stats_out, int row_pen, int col_pen,
                      const uint32_t *__restrict__ tbuf, const pmx_record_t *__restrict__ recs,
                      uint32_t *__restrict__ ops, const int64_t *__restrict__ ops_off, long long ops_base,
                      int32_t *__restrict__ nops, int32_t *__restrict__ beg, int32_t *__restrict__ textlen)
{
    constexpr int QP = G * R, NPW = 2 * (64 / G);
    constexpr int RB = (R + 1) / 2;                    // bytes per pair in a record (odd R: the last row has a byte of its own)
    constexpr int D = (2 * RB + 3) / 4;                // dwords per record
    static_assert(D <= 5, "record layout");
    __shared__ unsigned char s_map[256];
    __shared__ int16_t s_scores[PMX_MAX_FAST_MSIZE * PMX_MAX_FAST_MSIZE];
    for (int x = threadIdx.x; x < 256; x += blockDim.x) s_map[x] = mapper[x];
    for (int x = threadIdx.x; x < msize * msize; x += blockDim.x) s_scores[x] = scores[x];
    __syncthreads();

    const int lane = threadIdx.x & 63, l = lane & (LG - 1), gbase = lane & ~(LG - 1);
    const long long pos = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / LG;      // position in processing order (= trace order)
    bool active = pos < n;
    const long long pair = active ? (perm ? (long long)perm[pos] : pos) : 0;
    const long long qb = q_shared ? 0 : qoff[pair], rb = roff[pair];
    const int ql = q_shared ? q_shared : (int)(qoff[pair + 1] - qb), rl = (int)(roff[pair + 1] - rb);
    const uint8_t *q = qbuf + qb, *r = rbuf + rb;
    const long long area = pos / NPW; const int slot = (int)(pos % NPW);
    const uint32_t *tb = tbuf + (size_t)area * Tmax * (64 * D);
    const int P = (top_aligned || (blockflag && blockflag[area] == 0)) ? 0 : QP - ql;
    constexpr bool st = ST, sw = SW;
    const long long slot_lo = ops_off ? ops_off[pair] : qb + rb + pair - ops_base;
    const int slot_cap = ql + rl + 1;
    // ops are written from the end of the pair's slot backwards (the walk runs from the end of the alignment to its
    // beginning): the forward list is ops[slot_lo + slot_cap - cnt .. slot_lo + slot_cap)
    uint32_t *o_end = st ? nullptr : ops + slot_lo + slot_cap;
    pmx_record_t rec; rec.score = 0; rec.end_query = -1; rec.end_ref = -1; rec.flags = 0;
    if (active) rec = recs[pair];
    // ---- group-uniform state (every lane of the group computes the same values) ----
    int i = rec.end_query, j = rec.end_ref, cnt = 0;
    uint32_t cur_op = 0, cur_len = 0;
    int tlen = 0;
    int nM = 0, nS = 0, nL = 0;
    auto digits = [](uint32_t v) -> int { return v < 10 ? 1 : v < 100 ? 2 : v < 1000 ? 3 : v < 10000 ? 4 : v < 100000 ? 5 : 10; };
    auto flush = [&]() { if (cur_len) { ++cnt; if (l == 0) o_end[-cnt] = (cur_len << 4) | cur_op; tlen += digits(cur_len) + 1; } };
    auto add_run = [&](uint32_t op, int len) {
        if (len <= 0) return;
        if (st) { nL += len; return; }
        if (op == cur_op) cur_len += (uint32_t)len;
        else { flush(); cur_op = op; cur_len = (uint32_t)len; }
    };
    if (active && mode == PMX_MODE_SG && !st) {           // the unaligned tail beyond (end_query, end_ref): end gaps
        if (i + 1 == ql) add_run(OP_FOR_INS_STATE, rl - 1 - j);
        else if (j + 1 == rl) add_run(OP_FOR_DEL_STATE, ql - 1 - i);
    }
    int where = 0;                   // 0 DIAG, 1 INS, 2 DEL
    int rem = rec.score;             // local alignment: value of the current H / E / F cell; the path starts where it is used up
    auto group_bits = [&](bool pred) -> unsigned { return (unsigned)(__builtin_amdgcn_ballot_w64(pred) >> gbase) & 0xFFu; };

    while (__builtin_amdgcn_ballot_w64(active) != 0) {
        if (active && (i < 0 || j < 0)) {                 // one sequence is used up: the rest of the other is one gap run
            if (!sw) {
                if (i < 0 && j >= 0 && !(st && !row_pen)) add_run(OP_FOR_INS_STATE, j + 1), j = -1;
                else if (j < 0 && i >= 0 && !(st && !col_pen)) add_run(OP_FOR_DEL_STATE, i + 1), i = -1;
            }
            active = false;
        }
        if (active && where == 0 && sw && rem <= 0) active = false;            // ZERO cell
        // ---- probe: lane x looks at the x-th cell ahead on the current kind of move; ONE round trip to memory per iteration ----
        const int pi_ = where == 0 ? i - l : where == 1 ? i : i - 1 - l;
        const int pj_ = where == 0 ? j - l : where == 1 ? j - 1 - l : j;
        const int er = pi_ + P;
        const bool cell = active && pj_ >= 0 && er >= 0 && (where != 0 || pi_ >= 0);
        uint32_t w = 0; int a = 0, bsym = 0;
        const int g = er / R, k = er - g * R;
        const int bb = (slot & 1) * RB + (k >> 1);
        if (cell) {
            w = tb[((size_t)((slot >> 1) * G + g) * Tmax + (size_t)(pj_ + g)) * D + (bb >> 2)];
            if (where == 0) { a = q[pi_]; bsym = r[pj_]; }
        }
        const int nib = cell ? (int)((w >> (8 * (bb & 3) + ((k & 1) ? 0 : 4))) & 0xFu) : 0;
        if (where == 0) { a = s_map[a]; bsym = s_map[bsym]; }
        const int sc = ((ST || SW) && where == 0 && cell) ? (int)s_scores[a * msize + bsym] : 0;
        // group masks
        const unsigned m_cell = group_bits(cell);
        const unsigned m_diag = group_bits(cell && !(nib & 8));
        const unsigned m_eq = group_bits(a == bsym), m_sim = ST ? group_bits(sc > 0) : 0u;
        const unsigned m_eo = group_bits(cell && (nib & 2)), m_fo = group_bits(cell && (nib & 1));
        if (active) {
            if (where == 0) {
                int c = __builtin_ctz(~m_diag | 0x100u);                       // leading diagonal cells
                bool zero_stop = false;
                if (sw) {                                                       // the path ends where the score is used up
                    int inc = sc;
#pragma unroll
                    for (int s = 1; s < LG; s <<= 1) { const int up = __shfl_up(inc, s, LG); if (l >= s) inc += up; }
                    const unsigned m_stop = group_bits(rem - (inc - sc) <= 0);  // the value is used up before cell x: a ZERO cell
                    const int cz = __builtin_ctz(m_stop | 0x100u);
                    if (cz <= c && cz < LG) { zero_stop = true; c = cz; }
                    const int tot = __shfl(inc, gbase + (c > 0 ? c - 1 : 0), 64);
                    if (c > 0) rem -= tot;
                }
                const unsigned cm = (1u << c) - 1u;
                if (st) { nM += __builtin_popcount(m_eq & cm); nS += __builtin_popcount(m_sim & cm); }
                if (ST) nL += c;
                else if (c > 0) {
                    // The window's = / X runs leave in ONE pass, a lane per run (no loop over the runs: eight pairs walk in one wave, and
                    // the loop ran for the pair with the most runs).  Run starts: cell 0 and every cell that differs from the one before.
                    // The pending run (cur_op, cur_len) absorbs run 0 if it has the same op and leaves as soon as another run follows;
                    // the runs between the first and the last are complete (at most 8 cells: one digit each); the last one is the new
                    // pending run.  All counts are group-uniform; only the stores are per lane.
                    const unsigned cmw = (1u << c) - 1u, w = m_eq & cmw;
                    const unsigned starts = ((w ^ (w << 1)) & cmw) | 1u;
                    const int nr = __builtin_popcount(starts);
                    const uint32_t op0 = (w & 1u) ? OP_EQ : OP_X;
                    const bool same = cur_len != 0 && op0 == cur_op;
                    const unsigned rest = starts & ~1u;
                    const int s1 = rest ? __builtin_ctz(rest) : c;                 // end of run 0
                    const int slast = 31 - __builtin_clz(starts);                  // start of the last run
                    const uint32_t lenA = same ? cur_len + (uint32_t)s1 : cur_len;
                    const bool emitA = same ? nr >= 2 : cur_len != 0;
                    int basec = cnt;
                    if (emitA) { ++basec; if (l == 0) o_end[-basec] = (lenA << 4) | cur_op; tlen += digits(lenA) + 1; }
                    const int r0 = same ? 1 : 0;
                    const int nfull = nr - 1 - r0 > 0 ? nr - 1 - r0 : 0;
                    if ((starts >> l) & 1u) {
                        const int r = __builtin_popcount(starts & ((1u << l) - 1u));
                        if (r >= r0 && r <= nr - 2) {
                            const unsigned nx = starts >> (l + 1);                  // (not 0: a later run exists)
                            const int e = l + 1 + __builtin_ctz(nx | 0x100u);
                            o_end[-(basec + (r - r0) + 1)] = ((uint32_t)(e - l) << 4) | (((w >> l) & 1u) ? OP_EQ : OP_X);
                        }
                    }
                    tlen += 2 * nfull; cnt = basec + nfull;
                    if (nr == 1 && same) cur_len = lenA;
                    else { cur_op = ((w >> (c - 1)) & 1u) ? OP_EQ : OP_X; cur_len = (uint32_t)(c - slast); }
                }
                i -= c; j -= c;
                if (c < LG) {
                    if (zero_stop) active = false;
                    else if (m_cell >> c & 1) {                                 // a cell that did not come from the diagonal
                        const int nc = __shfl(nib, gbase + c, 64);
                        where = (nc & 4) ? 1 : 2;
                    }
                    // else: a boundary (i < 0 or j < 0), handled at the top of the next iteration
                }
            } else if (where == 1) {
                // ops for columns j, j-1, ...: the gap goes on while E of that column did not open (EO of the column before it)
                const unsigned stopm = m_eo | (~m_cell & 0xFFu);               // lane x: EO(i, j-1-x) set, or column j-1-x < 0
                const int mrun = __builtin_ctz(stopm | 0x100u);
                if (mrun == LG) { add_run(OP_FOR_INS_STATE, LG); j -= LG; rem += LG * ext; }
                else {
                    add_run(OP_FOR_INS_STATE, mrun + 1);
                    j -= mrun + 1;
                    if (m_eo >> mrun & 1) { where = 0; rem += open + mrun * ext; } else rem += (mrun + 1) * ext;     // else: column -1 reached
                }
            } else {
                const unsigned stopm = m_fo | (~m_cell & 0xFFu);               // lane x: FO(i-1-x, j) set, or no such row
                const int mrun = __builtin_ctz(stopm | 0x100u);
                const int limit = i + 1 < LG ? i + 1 : LG;                      // rows i .. 0 are all there is
                if (mrun >= limit) { add_run(OP_FOR_DEL_STATE, limit); i -= limit; rem += limit * ext; }
                else {
                    add_run(OP_FOR_DEL_STATE, mrun + 1);
                    i -= mrun + 1;
                    if (m_fo >> mrun & 1) { where = 0; rem += open + mrun * ext; } else rem += (mrun + 1) * ext;
                }
            }
        }
    }
    if (pos >= n || l != 0) return;
    if (st) { pmx_stats_t r3; r3.matches = nM; r3.similar = nS; r3.length = nL; stats_out[pair] = r3; return; }
    if (cur_len) { ++cnt; o_end[-cnt] = (cur_len << 4) | cur_op; tlen += digits(cur_len) + 1; }
    if (textlen) textlen[pair] = tlen;
    nops[pair] = cnt;
    beg[2 * pair] = i + 1; beg[2 * pair + 1] = j + 1;
}

// variant: the packed sweep's lane-group size index (0..3 -> G = 8, 16, 32, 64); rows per lane R = 16 or 10
int pmx_launch_walkp(int gsel, int R, const PmxBatch &b, const PmxDevMatrix &m, int mode, int open, int ext, int Tmax, int top_aligned,
                     pmx_stats_t *stats_out, int row_pen, int col_pen, const uint32_t *tbuf, const pmx_record_t *recs,
                     uint32_t *ops, const int64_t *ops_off, long long ops_base, int32_t *nops, int32_t *beg, int32_t *textlen,
                     hipStream_t stream, const int *blockflag)
{
    if (b.n <= 0) return 0;
    const dim3 grid((unsigned)((b.n * LG + 255) / 256)), block(256);
#define WALKP4(GG, RR, STV, SWV) hipLaunchKernelGGL((pmx_walkp_kernel<GG, RR, STV, SWV>), grid, block, 0, stream, \
        b.qbuf, b.qoff, b.q_shared, b.rbuf, b.roff, (long long)b.n, b.perm, m.mapper, m.scores, m.msize, open, ext, mode, Tmax, top_aligned, blockflag, \
        stats_out, row_pen, col_pen, tbuf, recs, ops, ops_off, ops_base, nops, beg, textlen)
#define WALKP(GG, RR) do { const bool st_ = stats_out != nullptr, sw_ = mode == PMX_MODE_SW; \
        if (st_) { if (sw_) WALKP4(GG, RR, true, true); else WALKP4(GG, RR, true, false); } \
        else { if (sw_) WALKP4(GG, RR, false, true); else WALKP4(GG, RR, false, false); } } while (0)
    if (R == 16) {
        switch (gsel) {
        case 0: WALKP(8, 16); break;
        case 1: WALKP(16, 16); break;
        case 2: WALKP(32, 16); break;
        default: WALKP(64, 16); break;
        }
    } else if (R == 20) {
        if (gsel != 1) return 1;
        WALKP(16, 20);
    } else if (R == 19) {
        if (gsel != 1) return 1;
        WALKP(16, 19);
    } else if (R == 10) {
        switch (gsel) {
        case 1: WALKP(16, 10); break;
        case 2: WALKP(32, 10); break;
        default: return 1;
        }
    } else return 1;
#undef WALKP
#undef WALKP4
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
