// pmx_trace16.hip -- fast kernel for global / semi-global alignment WITH traceback, for the batch
// CIGAR entry (BASELINE config 4: `sg_trace_striped_16` + CIGAR, /root/reference/src/alignment/mod.rs:
// 390-419 for the single-pair counterpart).  gfx950 only.
//
// Mapping: the strip-systolic layout of pmx_nwsg16.hip (G lanes per pair, R query rows per lane in
// VGPRs, one reference column per step, DPP skew, LDS query profile, virtual rows / columns that
// reproduce the boundary conditions), but ONE pair per slot and unpacked 16-bit lanes:
//
//   * All DP arithmetic is full-rate 32-bit-encoded VOP2 on the low 16 bits of a VGPR
//     (v_add_u16 / v_sub_u16 / v_max_u16, values biased by 32768): 8 instructions per cell.
//   * The four traceback decisions of a cell are four v_cmp_lt_u16 (VOPC -> VCC), each followed by
//     v_addc_co_u32 plane, plane, plane, vcc  -- "shift the lane's trace word left and insert the
//     bit" in one VOP2.  After R = 8 rows the word holds eight 4-bit cells:
//        bit3 ND  : H came from E or F            (T < H)
//        bit2 NDL : not from F                    (F < H)      [ND && !NDL -> DEL, ND && NDL -> INS]
//        bit1 EO  : E of the NEXT column opened   (E - ext < H - open)
//        bit0 FO  : F of the NEXT row opened      (F - ext < H - open)
//     Ties resolve exactly as in the oracle: diag, then F, then E; "open" only when strictly greater.
//   * One coalesced dword store per lane and step: 4 bits per cell in HBM instead of the reference
//     layout's byte per cell (which only the single-pair API still materialises).
//   * pmx_walk16_kernel (one lane per pair) walks the nibbles from the captured end position and
//     emits run-length BAM ops.
#include "pmx_common.h"
#include "pmx_switches.h"
#include <cstdlib>

#define TB 32768                 // bias of the u16 lanes
#define TNEG (-16384)            // finite "-inf": loses against every value of the exact window
#define OP_I 1u
#define OP_D 2u
#define OP_EQ 7u
#define OP_X 8u
#define OP_FOR_INS_STATE PMX_BAM_OP_FOR_INS_STATE    // include/pmx_conventions.h
#define OP_FOR_DEL_STATE PMX_BAM_OP_FOR_DEL_STATE    // include/pmx_conventions.h

__device__ __forceinline__ unsigned a16(unsigned a, unsigned b) { unsigned r; asm("v_add_u16_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned s16(unsigned a, unsigned b) { unsigned r; asm("v_sub_u16_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned m16(unsigned a, unsigned b) { unsigned r; asm("v_max_u16_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// plane = plane * 2 + (a < b)   (unsigned 16-bit compare)
__device__ __forceinline__ void push_lt(unsigned &plane, unsigned a, unsigned b)
{
    asm("v_cmp_lt_u16_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(plane) : "v"(a), "v"(b) : "vcc");
}

template <int G>
__device__ __forceinline__ unsigned t_shift_up(unsigned x, unsigned neutral, int g)
{
    if (G <= 16) {
        int r = __builtin_amdgcn_update_dpp((int)neutral, (int)x, 0x111, 0xF, 0xF, false);
        if (G < 16) r = (g == 0) ? (int)neutral : r;
        return (unsigned)r;
    } else {
        int r = __builtin_amdgcn_update_dpp((int)neutral, (int)x, 0x138, 0xF, 0xF, false);
        if (G < 64) r = (g == 0) ? (int)neutral : r;
        return (unsigned)r;
    }
}

template <int G, int R, bool SW>
__global__ __launch_bounds__(64)
void pmx_trace16_kernel(const uint8_t *__restrict__ qbuf, const int64_t *__restrict__ qoff,
                        const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff,
                        long long n, const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap,
                        int msize, int open, int ext, int RP, int Tmax,
                        int col_pen, int row_pen, int s1_end, int s2_end,
                        pmx_record_t *__restrict__ out, uint32_t *__restrict__ tbuf)
{
    static_assert(R == 8 || R == 16, "one trace dword per 8 rows");
    constexpr int QP = G * R;
    constexpr int NP = 64 / G;             // pairs per wave
    constexpr int TW = R / 8;              // trace dwords per lane and step
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int lane = threadIdx.x;
    const int g = lane % G;
    const int slot = lane / G;
    const int MS1 = msize + 1;
    const int PROF_STRIDE = MS1 * QP * 2;

    int16_t *prof = reinterpret_cast<int16_t *>(lds);
    unsigned char *rsym = lds + NP * PROF_STRIDE;
    int16_t *mat = reinterpret_cast<int16_t *>(rsym + NP * RP);
    unsigned char *map = reinterpret_cast<unsigned char *>(mat + msize * msize);
    int *ptab = reinterpret_cast<int *>(map + 256 + ((4 - ((msize * msize * 2) & 3)) & 3));

    const long long pair0 = (long long)blockIdx.x * NP;
    for (int i = lane; i < msize * msize; i += 64) mat[i] = gmat[i];
    for (int i = lane; i < 256; i += 64) map[i] = gmap[i];
    if (lane < NP) {
        long long pi = pair0 + lane; if (pi >= n) pi = n - 1;
        const long long qb = qoff[pi], rb = roff[pi];
        ptab[4 * lane + 0] = (int)(qb - qoff[pair0]);
        ptab[4 * lane + 1] = (int)(qoff[pi + 1] - qb);
        ptab[4 * lane + 2] = (int)(rb - roff[pair0]);
        ptab[4 * lane + 3] = (int)(roff[pi + 1] - rb);
    }
    __syncthreads();
    const uint8_t *qbase = qbuf + qoff[pair0];
    const uint8_t *rbase = rbuf + roff[pair0];

    int max_rlen = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) max_rlen = max(max_rlen, ptab[4 * p + 3]);

    // reference symbols: G-1 virtual columns in front, pad symbol behind
    for (int p = 0; p < NP; ++p) {
        const int rl = ptab[4 * p + 3];
        const uint8_t *rp = rbase + ptab[4 * p + 2];
        for (int j0 = 0; j0 < RP; j0 += 64 * 4) {
            unsigned char raw[4]; bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 64 + lane, jj = j - (G - 1);
                ok[u] = j < RP && jj >= 0 && jj < rl;
                raw[u] = ok[u] ? rp[jj] : (unsigned char)0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 64 + lane;
                if (j < RP) rsym[p * RP + j] = ok[u] ? map[raw[u]] : (unsigned char)msize;
            }
        }
    }
    // extended profile: P virtual rows on top, then the qlen real rows
    const int vrow_score = row_pen ? TNEG : 0, vcol_score = col_pen ? TNEG : 0;
    for (int p = 0; p < NP; ++p) {
        const int P = QP - ptab[4 * p + 1];
        const uint8_t *qp = qbase + ptab[4 * p + 0];
        int16_t *pp = prof + p * (PROF_STRIDE / 2);
        for (int er0 = 0; er0 < QP; er0 += 64 * 4) {
            int qs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int er = er0 + u * 64 + lane;
                qs[u] = (er < QP && er >= P) ? (int)qp[er - P] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int er = er0 + u * 64 + lane;
                if (er < QP) {
                    const int q0 = qs[u] < 0 ? -1 : map[qs[u]];
                    for (int sym = 0; sym < msize; ++sym)
                        pp[sym * QP + er] = (int16_t)((q0 < 0) ? vrow_score : mat[q0 * msize + sym]);
                    pp[msize * QP + er] = (int16_t)((q0 < 0) ? 0 : vcol_score);
                }
            }
        }
    }
    __syncthreads();

    // ---- per-lane state --------------------------------------------------------------------
    const unsigned short *profL = reinterpret_cast<const unsigned short *>(lds + slot * PROF_STRIDE) + g * R;
    const unsigned char *rs = rsym + slot * RP + (G - 1) - g;
    const int ql = ptab[4 * slot + 1], rl = ptab[4 * slot + 3];
    const int P = QP - ql;
    const unsigned vOpen = (unsigned)open, vExt = (unsigned)ext;

    auto left_h = [&](int er) -> unsigned {
        const int i = er - P;
        return (unsigned)(TB + ((i >= 0 && col_pen) ? -(open + i * ext) : 0));
    };
    unsigned HA[R], HB[R], E[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { HA[k] = left_h(g * R + k); HB[k] = HA[k]; E[k] = HA[k] - vOpen; }
    unsigned Hout = HA[R - 1];
    unsigned Fout;
    {
        const int i = (g + 1) * R - P;
        Fout = (unsigned)(TB + ((i >= 0 && col_pen) ? -(open + i * ext) : -open));
    }
    unsigned diag0 = (g == 0) ? (unsigned)TB : left_h(g * R - 1);

    int res = TB, bestrow = 0, bestrowj = 0, bestcol = 0, bestcoli = 0;
    // local alignment: running best of this lane, the column where it was first exceeded, the strip at that column
    unsigned swbest = (unsigned)TB, swcol = 0u, swsave[R];
#pragma unroll
    for (int k = 0; k < R; ++k) swsave[k] = (unsigned)TB;
    uint32_t *tw = tbuf + ((size_t)blockIdx.x * Tmax) * (64 * TW) + lane * TW;

    auto load_scores = [&](int sym, unsigned (&w)[R]) {
        const unsigned short *sp = profL + sym * QP;
#pragma unroll
        for (int k = 0; k < R; ++k) w[k] = sp[k];
    };
    auto step = [&](const unsigned (&Hold)[R], unsigned (&Hnew)[R], const unsigned (&w)[R], int t) {
        const unsigned Hin = t_shift_up<G>(Hout, (unsigned)TB, g);
        unsigned F = t_shift_up<G>(Fout, (unsigned)(TB + TNEG), g);
        unsigned plane[TW];
#pragma unroll
        for (int x = 0; x < TW; ++x) plane[x] = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const unsigned d = (k == 0) ? diag0 : Hold[k - 1];
            const unsigned Tt = a16(d, w[k]);
            unsigned H = m16(m16(Tt, E[k]), F);
            if (SW) H = m16(H, (unsigned)TB);          // local: floor at zero (the walk stops where the score is used up)
            push_lt(plane[k / 8], Tt, H);          // ND
            push_lt(plane[k / 8], F, H);           // NDL
            const unsigned Ho = s16(H, vOpen);
            const unsigned Ee = s16(E[k], vExt), Fe = s16(F, vExt);
            push_lt(plane[k / 8], Ee, Ho);         // EO (next column)
            push_lt(plane[k / 8], Fe, Ho);         // FO (next row)
            E[k] = m16(Ee, Ho);
            F = m16(Fe, Ho);
            Hnew[k] = H;
        }
        diag0 = Hin;
        Hout = Hnew[R - 1];
        Fout = F;
#pragma unroll
        for (int x = 0; x < TW; ++x) tw[(size_t)t * (64 * TW) + x] = plane[x];

        // ---- captures ----
        const int jcol = t - g;
        if (SW) {
            unsigned cm = Hnew[0] & 0xFFFFu;
#pragma unroll
            for (int k = 1; k < R; ++k) cm = m16(cm, Hnew[k]) & 0xFFFFu;
            const bool imp = cm > swbest;          // strictly greater: the first column reaching a value keeps it
            swbest = imp ? cm : swbest;
            swcol = imp ? (unsigned)jcol : swcol;
#pragma unroll
            for (int k = 0; k < R; ++k) swsave[k] = imp ? Hnew[k] : swsave[k];
            return;
        }
        const int hl = (int)(Hout & 0xFFFF);
        if (jcol == rl - 1) res = hl;
        if (s2_end && jcol >= 0 && jcol < rl && hl > bestrow) { bestrow = hl; bestrowj = jcol; }
        if (s1_end && jcol == rl - 1) {
            int cm = 0, cr = 0;
#pragma unroll
            for (int k = R - 1; k >= 0; --k) {
                const int er = g * R + k;
                const int v = (er >= P) ? (int)(Hnew[k] & 0xFFFF) : 0;
                if (v >= cm) { cm = v; cr = er; }          // descending k with >= : smallest row among equals
            }
            if (cm > bestcol) { bestcol = cm; bestcoli = cr; }
        }
    };

    const int T = (max_rlen + G - 1 + 1) & ~1;
    unsigned w0[R], w1[R];
    load_scores(rs[0], w0);
    int ns = rs[1];
    for (int t = 0; t < T; t += 2) {
        load_scores(ns, w1);
        ns = rs[t + 2];
        __builtin_amdgcn_sched_barrier(0);
        step(HA, HB, w0, t);
        __builtin_amdgcn_sched_barrier(0);
        load_scores(ns, w0);
        ns = rs[t + 3];
        __builtin_amdgcn_sched_barrier(0);
        step(HB, HA, w1, t + 1);
        __builtin_amdgcn_sched_barrier(0);
    }

    if (SW) {
        int krow = 0;
#pragma unroll
        for (int k = R - 1; k >= 0; --k) if ((swsave[k] & 0xFFFFu) == swbest) krow = k;
        unsigned long long key = ((unsigned long long)swbest << 32) | ((unsigned long long)(0xFFFFu - (swcol & 0xFFFFu)) << 16) |
                                 (unsigned long long)(0xFFFFu - (unsigned)(g * R + krow));
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off, 64);
            key = o > key ? o : key;
        }
        if (g == 0) {
            const long long pi = pair0 + slot;
            if (pi < n) {
                pmx_record_t rec; rec.flags = 0;
                rec.score = (int)(key >> 32) - TB;
                rec.end_ref = 0xFFFF - (int)((key >> 16) & 0xFFFF);
                rec.end_query = 0xFFFF - (int)(key & 0xFFFF) - P;
                if (rec.score == 0) { rec.end_query = 0; rec.end_ref = 0; }      // all-zero table: (0, 0) like the oracle
                out[pi] = rec;
            }
        }
        return;
    }
    // ---- combine (same rules as pmx_nwsg16.hip / the oracle) ---------------------------------
    unsigned key = ((unsigned)bestcol << 16) | (0xFFFFu - (unsigned)bestcoli);
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) {
        const unsigned o = __shfl_xor(key, off, 64);
        key = o > key ? o : key;
    }
    const int lastlane = slot * G + G - 1;
    const int resL = __shfl(res, lastlane, 64), browL = __shfl(bestrow, lastlane, 64), browjL = __shfl(bestrowj, lastlane, 64);
    if (g == 0) {
        const long long pi = pair0 + slot;
        if (pi < n) {
            pmx_record_t rec; rec.flags = 0;
            if (!s1_end && !s2_end) { rec.score = resL - TB; rec.end_query = ql - 1; rec.end_ref = rl - 1; }
            else {
                int best = -2147483647 - 1, ei = 0, ej = 0;
                if (s2_end) { best = browL - TB; ei = ql - 1; ej = browjL; }
                if (s1_end) {
                    const int cv = (int)(key >> 16) - TB;
                    if (cv > best) { best = cv; ei = (int)(0xFFFFu - (key & 0xFFFFu)) - P; ej = rl - 1; }
                }
                rec.score = best; rec.end_query = ei; rec.end_ref = ej;
            }
            out[pi] = rec;
        }
    }
}

// ---- walk over the 4-bit trace ---------------------------------------------------------------
// PACKED: the layout written by pmx_nwsg16v_kernel<G,16,true> (two pairs per slot, 16 bytes per lane and step).
template <int G, int R, bool PACKED>
__global__ void pmx_walk16_kernel(const uint8_t *qbuf, const int64_t *qoff, const uint8_t *rbuf, const int64_t *roff,
                                  long long n, const uint8_t *mapper, const int16_t *scores, int msize, int open, int ext,
                                  int mode, int Tmax, int stage /* LDS bytes reserved for each of the block's queries / references, 0 = none */,
                                  int top_aligned /* query rows start at the first lane's first register (local kernel) */,
                                  pmx_stats_t *stats_out /* != nullptr: count matches / similar / length along the path instead of emitting ops */,
                                  int row_pen, int col_pen /* stats: the begin gaps along the reference / query are part of the alignment */,
                                  const uint32_t *tbuf, const pmx_record_t *recs,
                                  uint32_t *ops, const int64_t *ops_off /* nullptr: slot of pair k starts at qoff[k] + roff[k] + k - ops_base */,
                                  long long ops_base, int32_t *nops, int32_t *beg, int32_t *textlen /* optional: bytes of the CIGAR text */)
{
    extern __shared__ unsigned char w_lds[];
    constexpr int QP = G * R, NP = (PACKED ? 2 : 1) * (64 / G), TW = PACKED ? 4 : R / 8;
    // mapper and matrix in LDS: the walk is a chain of dependent loads, keep all but the trace fetch short
    __shared__ unsigned char s_map[256];
    __shared__ int16_t s_scores[PMX_MAX_FAST_MSIZE * PMX_MAX_FAST_MSIZE];
    for (int x = threadIdx.x; x < 256; x += blockDim.x) s_map[x] = mapper[x];
    for (int x = threadIdx.x; x < msize * msize; x += blockDim.x) s_scores[x] = scores[x];
    // The block's sequences are contiguous in the packed buffers: stage them in LDS with coalesced loads, so
    // that the only scattered global access left per step is the trace nibble.
    const long long p0 = (long long)blockIdx.x * blockDim.x;
    const long long p1 = p0 + blockDim.x < n ? p0 + blockDim.x : n;
    const long long qlo = qoff[p0], rlo = roff[p0];
    if (stage) {
        const int qn = (int)(qoff[p1] - qlo), rn = (int)(roff[p1] - rlo);
        for (int x = threadIdx.x; x < qn; x += blockDim.x) w_lds[x] = qbuf[qlo + x];
        for (int x = threadIdx.x; x < rn; x += blockDim.x) w_lds[stage + x] = rbuf[rlo + x];
    }
    __syncthreads();
    const long long pair = p0 + threadIdx.x;
    if (pair >= n) return;
    const long long qb = qoff[pair], rb = roff[pair];
    const int ql = (int)(qoff[pair + 1] - qb), rl = (int)(roff[pair + 1] - rb);
    const uint8_t *q = qbuf + qb, *r = rbuf + rb;
    const unsigned char *sq = w_lds + (qb - qlo), *sr = w_lds + stage + (rb - rlo);
    const long long block = pair / NP; const int slot = (int)(pair % NP);
    // PACKED: records of 16 bytes per lane and step, lane-major (a lane's steps are contiguous): [A rows 0-7][A rows 8-15][B rows 0-7]
    // [B rows 8-15], byte = a row pair, the even row in the high nibble.  A walk moves along a row or a diagonal, i.e. down one
    // lane's steps: the lane keeps a window of 8 consecutive records (one 128-byte line of that stream) in LDS and goes back to
    // memory only when the path leaves it.  Unpacked (first generation): R / 8 words per lane and step, step-major.
    const uint32_t *tb = tbuf + (size_t)block * Tmax * (64 * TW);
    const int P = top_aligned ? 0 : QP - ql;
    __shared__ uint4 s_win[PACKED ? 64 * 8 : 1];
    int w_stream = -1, w_t0 = -1;
    auto ldw = [&](int i, int j) -> uint32_t {         // the trace word holding cell (i, j); 0 outside i >= -1, j >= 0
        if (i < -1 || j < 0) return 0u;
        const int er = i + P;
        if (er < 0) return 0u;
        const int g = er / R, k = er % R;
        if (!PACKED) return tb[(size_t)(j + g) * (64 * TW) + (slot * G + g) * TW + (k / 8)];
        const int stream = (slot >> 1) * G + g, t = j + g, t0 = t & ~7;
        const int lx = threadIdx.x;
        if (stream != w_stream || t0 != w_t0) {
            const uint4 *src = reinterpret_cast<const uint4 *>(tb + ((size_t)stream * Tmax + t0) * 4);
            uint4 v[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) v[x] = src[x];
#pragma unroll
            for (int x = 0; x < 8; ++x) s_win[lx * 8 + ((x ^ lx) & 7)] = v[x];
            w_stream = stream; w_t0 = t0;
        }
        const uint32_t *rec4 = reinterpret_cast<const uint32_t *>(&s_win[lx * 8 + (((t & 7) ^ lx) & 7)]);
        return rec4[(slot & 1) * 2 + (k / 8)];
    };
    auto nibof = [&](uint32_t w, int i) -> unsigned {
        const int k8 = ((i + P) % R) % 8;
        return PACKED ? (w >> (8 * (k8 >> 1) + ((k8 & 1) ? 0 : 4))) & 0xFu : (w >> (28 - 4 * k8)) & 0xFu;
    };
    auto nib = [&](int i, int j) -> unsigned { return nibof(ldw(i, j), i); };
    const bool st = stats_out != nullptr;
    uint32_t *o = st ? nullptr : ops + (ops_off ? ops_off[pair] : qb + rb + pair - ops_base);
    const pmx_record_t rec = recs[pair];
    int i = rec.end_query, j = rec.end_ref, cnt = 0;
    uint32_t cur_op = 0, cur_len = 0;
    int tlen = 0;                    // text bytes: digits of every run length + one letter
    auto digits = [](uint32_t v) -> int { return v < 10 ? 1 : v < 100 ? 2 : v < 1000 ? 3 : v < 10000 ? 4 : v < 100000 ? 5 : 10; };
    int nM = 0, nS = 0, nL = 0;      // statistics of the path = the coupled stats tables of the reference (same decisions, same ties)
    auto emit = [&](uint32_t op) {
        if (st) { ++nL; return; }
        if (op == cur_op) ++cur_len;
        else { if (cur_len) { o[cnt++] = (cur_len << 4) | cur_op; tlen += digits(cur_len) + 1; } cur_op = op; cur_len = 1; }
    };
    if (mode == PMX_MODE_SG && !st) {
        if (i + 1 == ql) { for (int k = rl - 1; k > j; --k) emit(OP_FOR_INS_STATE); }
        else if (j + 1 == rl) { for (int k = ql - 1; k > i; --k) emit(OP_FOR_DEL_STATE); }
    }
    int where = 0;   // 0 DIAG, 1 INS, 2 DEL
    const bool sw = mode == PMX_MODE_SW;
    int rem = rec.score;       // local alignment: value of the current H / E / F cell; the path starts where it is used up
    while (i >= 0 || j >= 0) {
        if (i < 0) { if (sw || (st && !row_pen)) break; emit(OP_FOR_INS_STATE); --j; continue; }
        if (j < 0) { if (sw || (st && !col_pen)) break; emit(OP_FOR_DEL_STATE); --i; continue; }
        if (where == 0) {
            if (sw && rem <= 0) break;                       // ZERO cell
            const unsigned t = nib(i, j);
            if (!(t & 8u)) {
                const int a = s_map[stage ? sq[i] : q[i]], b = s_map[stage ? sr[j] : r[j]];
                emit(a == b ? OP_EQ : OP_X);
                if (st) { nM += a == b; nS += s_scores[a * msize + b] > 0; }
                if (sw) rem -= s_scores[a * msize + b];
                --i; --j;
            }
            else if (!(t & 4u)) where = 2;
            else where = 1;
        } else if (where == 1) {
            emit(OP_FOR_INS_STATE);
            if (j > 0 && (nib(i, j - 1) & 2u)) { where = 0; rem += open; } else rem += ext;
            --j;
        } else {
            emit(OP_FOR_DEL_STATE);
            if (nib(i - 1, j) & 1u) { where = 0; rem += open; } else rem += ext;
            --i;
        }
    }
    if (st) { pmx_stats_t r3; r3.matches = nM; r3.similar = nS; r3.length = nL; stats_out[pair] = r3; return; }
    if (cur_len) { o[cnt++] = (cur_len << 4) | cur_op; tlen += digits(cur_len) + 1; }
    if (textlen) textlen[pair] = tlen;
    for (int k = 0; k < cnt / 2; ++k) { const uint32_t tmp = o[k]; o[k] = o[cnt - 1 - k]; o[cnt - 1 - k] = tmp; }
    nops[pair] = cnt;
    beg[2 * pair] = i + 1; beg[2 * pair + 1] = j + 1;
}

// ------------------------------------------------------------------------ host side ----
// LDS bytes per side for staging a 64-pair block's queries / references in the walk (0: too long, read them from HBM)
static int walk_stage_bytes(const PmxBatch &b)
{
    const long long m = 64LL * (b.max_qlen > b.max_rlen ? b.max_qlen : b.max_rlen);
    return (m <= 60 * 1024) ? (int)((m + 15) / 16 * 16) : 0;
}

template <int G, int R, bool SW>
static int launch_trace(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                        pmx_record_t *d_out, uint32_t *tbuf, int Tmax,
                        uint32_t *ops, const int64_t *ops_off, int32_t *nops, int32_t *beg, pmx_stats_t *stats_out, hipStream_t stream)
{
    constexpr int QP = G * R, NP = 64 / G;
    const int RP = ((b.max_rlen + 2 * (G - 1) + 4 + 7) / 4) * 4;
    const size_t lds = (size_t)NP * (m.msize + 1) * QP * 2 + (size_t)NP * RP +
                       (size_t)m.msize * m.msize * 2 + 256 + 4 + (size_t)NP * 16;
    if (lds > 160 * 1024) return 1;
    { const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_trace16_kernel<G, R, SW>)); if (rc) return rc; }
    const bool sg = mode == PMX_MODE_SG;
    const int col_pen = SW ? 0 : !(sg && (sg_flags & PMX_SG_QB)), row_pen = SW ? 0 : !(sg && (sg_flags & PMX_SG_DB));
    const int s1_end = sg && (sg_flags & PMX_SG_QE), s2_end = sg && (sg_flags & PMX_SG_DE);
    const long long blocks = (b.n + NP - 1) / NP;
    hipLaunchKernelGGL((pmx_trace16_kernel<G, R, SW>), dim3((unsigned)blocks), dim3(64), lds, stream,
                       b.qbuf, b.qoff, b.rbuf, b.roff, (long long)b.n, m.scores, m.mapper,
                       m.msize, open, ext, RP, Tmax, col_pen, row_pen, s1_end ? 1 : 0, s2_end ? 1 : 0, d_out, tbuf);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e;
    const int stage = walk_stage_bytes(b);
    { const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_walk16_kernel<G, R, false>), 128 * 1024); if (rc) return rc; }
    hipLaunchKernelGGL((pmx_walk16_kernel<G, R, false>), dim3((unsigned)((b.n + 63) / 64)), dim3(64), 2 * (size_t)stage, stream,
                       b.qbuf, b.qoff, b.rbuf, b.roff, (long long)b.n, m.mapper, m.scores, m.msize, open, ext, mode, Tmax, stage, 0, stats_out, row_pen, col_pen,
                       (const uint32_t *)tbuf, (const pmx_record_t *)d_out, ops, ops_off, 0LL, nops, beg, (int32_t *)nullptr);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// Picks the instantiation; *trace_bytes / *Tmax tell the caller how much trace scratch to provide
// (call once with tbuf == nullptr to size it).  Returns 1 when the batch is not eligible.
int pmx_trace16_plan(const PmxBatch &b, const PmxDevMatrix &m, int mode, int open, int ext,
                     int *variant, int *Tmax, size_t *trace_bytes, bool packed_ok)
{
    if (pmx_env("PMX_NO_FAST_TRACE")) return 1;
    if (mode != PMX_MODE_NW && mode != PMX_MODE_SG && mode != PMX_MODE_SW) return 1;
    if (m.msize > PMX_MAX_FAST_MSIZE - 1) return 1;
    if (open < ext || open < 0 || ext < 0 || open > 4096) return 1;
    if (b.max_rlen > 30000 || b.q_shared) return 1;
    if (packed_ok && !pmx_env("PMX_TRACE16_GEN1") && pmx_nwsgv_trace_plan(b, m, mode, open, ext, variant, Tmax, trace_bytes) == 0) {
        *variant += 10;            // packed traceback of the second-generation nw/sg kernel
        return 0;
    }
    if (packed_ok && mode == PMX_MODE_SW && pmx_sw16_trace_plan(b, m, open, ext, variant, Tmax, trace_bytes) == 0) {
        *variant += 20;            // packed traceback of the local kernel
        return 0;
    }
    const long long lo = mode == PMX_MODE_SW ? -(2LL * open + 2LL * ext + (m.min < 0 ? -m.min : 0))
                                             : -(3LL * open + (long long)(b.max_qlen + b.max_rlen + 2) * ext + (m.min < 0 ? -m.min : 0));
    const long long hi = (long long)(b.max_qlen < b.max_rlen ? b.max_qlen : b.max_rlen) * (m.max > 0 ? m.max : 0) + (m.max > 0 ? m.max : 0);
    if (lo < -15000 || hi > 15000) return 1;
    int G, R;
    if (b.max_qlen <= 32 * 8 - 1) { *variant = 0; G = 32; R = 8; }
    else if (b.max_qlen <= 64 * 8 - 1) { *variant = 1; G = 64; R = 8; }
    else if (b.max_qlen <= 64 * 16 - 1) { *variant = 2; G = 64; R = 16; }
    else return 1;
    const int NP = 64 / G;
    *Tmax = (b.max_rlen + G - 1 + 1) & ~1;
    const long long blocks = (b.n + NP - 1) / NP;
    *trace_bytes = (size_t)blocks * (size_t)*Tmax * 64 * (R / 8) * sizeof(uint32_t);
    return 0;
}

int pmx_launch_trace16(int variant, const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                       pmx_record_t *d_out, uint32_t *tbuf, int Tmax,
                       uint32_t *ops, const int64_t *ops_off, int32_t *nops, int32_t *beg, hipStream_t stream, pmx_stats_t *stats_out,
                       const PmxWalkSplit *split)
{
    const bool sw = mode == PMX_MODE_SW;
    const bool sg_ = mode == PMX_MODE_SG;
    const int col_pen_ = sw ? 0 : !(sg_ && (sg_flags & PMX_SG_QB)), row_pen_ = sw ? 0 : !(sg_ && (sg_flags & PMX_SG_DB));
    if (variant >= 10) {
        const int top = variant >= 20 ? 1 : 0;
        // (Tmax is a multiple of 16 -- the plans round it -- so a walk window never crosses into the next lane's stream)
        int rc = top ? pmx_launch_sw16_trace(variant - 20, b, m, open, ext, d_out, tbuf, Tmax, stream)
                     : pmx_launch_nwsgv_trace(variant - 10, b, m, mode, sg_flags, open, ext, d_out, tbuf, Tmax, stream);
        if (rc) return rc;
        hipStream_t wstream = stream;
        if (split && split->walk_stream != stream) {      // the walk runs beside the next chunk's sweep
            hipError_t e1 = hipEventRecord(split->sweep_done, stream);
            if (e1 == hipSuccess) e1 = hipStreamWaitEvent(split->walk_stream, split->sweep_done, 0);
            if (e1 != hipSuccess) return -(int)e1;
            wstream = split->walk_stream;
        }
        const long long ops_base = split ? split->ops_base : 0;
        int32_t *textlen = split ? split->textlen : nullptr;
        // (Tmax is a multiple of 16 -- the plans round it -- so a walk window never crosses into the next lane's stream)
        rc = pmx_launch_walkp((variant % 10) & 3, 16, b, m, mode, open, ext, Tmax, top, stats_out, row_pen_, col_pen_,
                              (const uint32_t *)tbuf, (const pmx_record_t *)d_out, ops, ops_off, ops_base, nops, beg, textlen, wstream,
                              (!top && variant - 10 < 4) ? b.blockflag : nullptr);      // (the nwsgv shapes: their launcher fills the flags)
        if (rc) return rc;
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && split && split->walk_done) e = hipEventRecord(split->walk_done, wstream);
        return e == hipSuccess ? 0 : -(int)e;
    }
    if (split) return 1;                 // the split form is only built for the packed (second-generation) sweeps
#define LT(GG, RR) (sw ? launch_trace<GG, RR, true>(b, m, mode, sg_flags, open, ext, d_out, tbuf, Tmax, ops, ops_off, nops, beg, stats_out, stream) \
                       : launch_trace<GG, RR, false>(b, m, mode, sg_flags, open, ext, d_out, tbuf, Tmax, ops, ops_off, nops, beg, stats_out, stream))
    switch (variant) {
    case 0: return LT(32, 8);
    case 1: return LT(64, 8);
    case 2: return LT(64, 16);
    }
#undef LT
    return 1;
}
