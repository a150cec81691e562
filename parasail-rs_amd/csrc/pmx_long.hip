// pmx_long.hip -- ONE long pair across the chip (score and end positions; all modes).  gfx950 only.
//
// `Aligner::align()` has no length limit (/root/reference/src/aligner/mod.rs:397-430, and :454-456 "for aligning large
// sequences"): one call on 20 kbp x 20 kbp is 4e8 cells.  The batch kernels give a pair one wave (or the waves of one workgroup:
// pmx_general_mw_kernel, 1.6 GCUPS on one CU with 255 CUs idle).  Here the query is cut into BANDS of 64 R rows; band b is ONE wave
// (one 64-thread workgroup, so the bands spread over the CUs) that sweeps the whole reference with the strip-systolic mapping of
// the packed kernels -- lane g keeps R consecutive rows in VGPRs, works on column t - g at step t, F runs down the R rows
// inside the lane and crosses to lane g + 1 through one DPP wave_shr:1 per value and step -- in 32-bit lanes (no score limit).
// Band b + 1 needs, per column, the H and F that leave band b's last row: lane 63 of band b stores them as ONE naturally
// aligned 8-byte granule per column with a write-through (sc1) store; band b + 1 loads 64 granules at a time (one per lane,
// coalesced, sc1), polls until none of them is the fill pattern the buffer was initialised with, and feeds them to its lane 0
// by rotating the 64 values one lane per step (wave_shl:1) -- the rotating register IS the `old` operand of the systolic
// wave_shr, which lane 0 keeps.  A granule is data and tag at once (MI355X_MICROARCH.md, hand-off by data-tagged granules: a
// single aligned 8-byte sc1 store is observed whole), so there is no flag, no fence and no barrier anywhere.  The bands form a
// software pipeline across the chip: band b + 1 finishes 150-200 steps (20-25 us) after band b -- 63 of them are the lanes' skew, the
// rest the chunk it takes over at a time plus the chunk it prefetches ahead (round 4, measured: time of one call = columns x
// 117 ns + bands x 22 us for 256-row bands; placing consecutive bands on one XCD or polling faster moves that by < 3 %, so it is
// not memory latency; profiles/r04/long_shapes.txt).  A consumer only ever waits for a producer
// with a LOWER workgroup index.  Forward progress rests on ONE assumption about the hardware that HIP does not promise: the
// workgroups of a grid are dispatched in index order (per XCD), so the lowest unfinished band of every grid is resident and
// depends on finished bands only -- also when two such grids share the chip (two host threads, each with its own scratch and
// stream: tests/test_gpu_long.py runs that).  The wait is therefore BOUNDED: a band that polls `spin_limit` times (seconds; a
// legitimate wait is the producer's next 64 columns, microseconds), or sees the launch's abort word set, sets the word and
// stops waiting for good (it finishes its sweep on whatever the buffer holds); every band downstream finds the word (or runs
// into the limit itself) and does the same, the finalize kernel marks the launch's records, and the host (`long_batch`) redoes
// the call on the per-pair kernels and says so in pmx_last_error().
// Reference symbols take the same road: 64 mapped symbols per lane-parallel load, rotated to lane 0 and handed down the lanes
// with the wave -- no reference in LDS, no length limit.  The LDS holds the band's query profile only (int16 [symbol][row]).
//
// End positions by the oracle's rules (oracle/pmx_oracle.c): sw -- first maximum in column-major order: every lane keeps its
// running best, the step at which it was first strictly exceeded and its strip at that step; the bands' winners are merged by
// (score, column, row) in a finalize kernel.  nw -- the corner.  sg -- first maximum of the last row by ascending column,
// then the last column (smallest row) only if strictly greater; free begins are boundary values.
#include "pmx_common.h"
#include <type_traits>

#define LONG_NEG (-(1 << 30))
#define LONG_PAD (-16384)                              // profile entry of padding rows / columns
#define LONG_SENT 0x8080808080808080ull                // fill pattern of the boundary buffer: low word is below every live value

struct PmxLongArgs {
    const uint8_t *qbuf; const int64_t *qoff; int q_shared;
    const uint8_t *rbuf; const int64_t *roff;
    long long n; int nbmax;
    const int16_t *scores; const uint8_t *mapper; int msize;
    int sg_flags, open, ext;
    unsigned long long *bound; long long bstride;      // granules per (pair, band) boundary
    int *cand;                                         // 8 ints per (pair, band)
    pmx_record_t *out; int sat_above; int force_sat;
    int *abort_word;                                   // device word, zero before the launch: a band gave up waiting
    int spin_limit;                                    // polls of one wait before a band gives up
};

__device__ __forceinline__ int dpp_wave_shr(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, 0x138 /*wave_shr:1*/, 0xF, 0xF, false); }
// (lane 63 has no source and gets 0: what enters there reaches lane 0 only after 64 shifts, and every rotating register is reloaded by then;
//  without an `old` operand the result is not tied to the source's register -- one v_mov_b32 less per shift)
__device__ __forceinline__ int dpp_wave_shl(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x130 /*wave_shl:1*/, 0xF, 0xF, true); }

template <int R, int MODE, int CH /* columns per boundary chunk: 64, or 16 (lanes 0 .. 15 load; a band runs ~50 steps closer behind the one above) */>
__global__ __launch_bounds__(64)
void pmx_long32_kernel(PmxLongArgs a)
{
    constexpr int BR = 64 * R;
    constexpr bool SW = MODE == PMX_MODE_SW, SG = MODE == PMX_MODE_SG;
    // SKEW (global / semi-global, round 4): every value of cell (i, j) is kept + (i + j) * ext, so neither gap needs its subtraction:
    // E(j) = max(E(j - 1), X(j - 1)), F(i) = max(F(i - 1), X(i - 1)) with X = H - (open - ext); the diagonal step crosses a row and a
    // column: + 2 ext, folded into the profile (score + open + ext).  5 instructions per cell instead of 7; boundaries and granules are in
    // the same form (the offset is global), captures take it off.  Local alignment keeps the plain form: its zero floor is per cell.
    constexpr bool SKEW = !SW;
    const int lane = threadIdx.x;
    const long long pair = blockIdx.x / a.nbmax;
    const int band = (int)(blockIdx.x % a.nbmax);
    const long long qb = a.q_shared ? 0 : a.qoff[pair], rb = a.roff[pair];
    const int ql = a.q_shared ? a.q_shared : (int)(a.qoff[pair + 1] - qb), rl = (int)(a.roff[pair + 1] - rb);
    const int NB = (ql + BR - 1) / BR;
    if (band >= NB) return;
    const bool lastband = band == NB - 1;
    const uint8_t *q = a.qbuf + qb, *r = a.rbuf + rb;
    const int msize = a.msize, open = a.open, ext = a.ext;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int16_t *prof = reinterpret_cast<int16_t *>(lds);                  // [(msize + 1) * BR]: row msize = the pad symbol
    int16_t *mat = prof + (msize + 1) * BR;                            // [msize * msize]
    unsigned char *map = reinterpret_cast<unsigned char *>(mat + msize * msize);
    for (int i = lane; i < msize * msize; i += 64) mat[i] = a.scores[i];
    for (int i = lane; i < 256; i += 64) map[i] = a.mapper[i];
    __syncthreads();
    for (int row = lane; row < BR; row += 64) {                        // the profile carries score + open (the strips carry H - open)
        const int i = band * BR + row;
        const int qs = i < ql ? (int)map[q[i]] : -1;
        for (int sym = 0; sym < msize; ++sym) prof[sym * BR + row] = (int16_t)(qs < 0 ? LONG_PAD : mat[qs * msize + sym] + open + (SKEW ? ext : 0));
        prof[msize * BR + row] = (int16_t)LONG_PAD;
    }
    __syncthreads();

    const bool pen_col = MODE == PMX_MODE_NW || (SG && !(a.sg_flags & PMX_SG_QB));      // column -1 (query begin) is penalised
    const bool pen_row = MODE == PMX_MODE_NW || (SG && !(a.sg_flags & PMX_SG_DB));      // row -1 (reference begin) is penalised
    auto left = [&](int i) -> int { return i < 0 ? 0 : (pen_col ? -(open + i * ext) : 0); };      // H(i, -1); H(-1, -1) = 0
    auto top = [&](int j) -> int { return pen_row ? -(open + j * ext) : 0; };                      // H(-1, j)

    const int i0 = band * BR + lane * R;
    int HA[R], HB[R], E[R], hs[R], lc[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { HA[k] = left(i0 + k) - open + (SKEW ? (i0 + k) * ext : 0); HB[k] = HA[k]; E[k] = LONG_NEG; hs[k] = 0; lc[k] = LONG_NEG; }
    int diag = left(i0 - 1) - open + (SKEW ? (i0 - 1) * ext : 0);
    int Hout = 0, Fout = 0;
    int best = SW ? -open - 1 : 0, bcol = 0;
    // sg, reference end free: the lane and register that hold the query's last row (last band only)
    const int gstar = ((ql - 1) % BR) / R, kstar = (ql - 1) % R;
    int rbest = LONG_NEG, rcol = 0;
    const bool row_track = SG && (a.sg_flags & PMX_SG_DE) && lastband;

    const unsigned char *prof_lane = lds + lane * (R * 2);
    const int RU = (rl + 63) & ~63, T = RU + 64;
    const unsigned long long *bin = band ? a.bound + ((size_t)pair * a.nbmax + band - 1) * a.bstride : nullptr;
    unsigned long long *bout = lastband ? nullptr : a.bound + ((size_t)pair * a.nbmax + band) * a.bstride;

    // chunks of 64 columns: mapped reference symbols (as profile row offsets) and, below band 0, the boundary granules
    int symch = 0, Hb = 0, Fb = 0, symcur = msize * (BR * 2);
    int nraw = 0; unsigned long long ngran = 0;
    bool dead = false;                                 // wave-uniform: this band gave up waiting (or found the abort word)
    const int spin_limit = a.spin_limit;
    int *const abort_word = a.abort_word;
    auto prefetch_sym = [&](int base) { const int c = base + lane; nraw = c < rl ? (int)r[c] : -1; };
    auto prefetch_bound = [&](int base) { if (bin && (CH == 64 || lane < CH)) ngran = __hip_atomic_load(bin + base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // symbols of columns [base, base + 64) take over the rotating register (one step BEFORE the first of them is worked on:
    // a step reads the profile row of the NEXT step's symbol ahead of its own arithmetic)
    auto reload_sym = [&](int base) {
        symch = (nraw < 0 ? msize : (int)map[nraw]) * (BR * 2);
        if (base + 64 < T) prefetch_sym(base + 64);
    };
    auto reload_bound = [&](int base) {
        if (bin && base < RU) {                            // (the chunk behind the reference is padding on both sides: never written, never needed)
            // The tight poll is what every band's pipeline lag is made of; the limit and the abort word are looked at once per 64 polls.
            // A band that gives up does NOT leave (an exit in the middle of the sweep cost global alignment 70 % -- 4.2 -> 7.1 ms at
            // 20 kbp -- whatever the poll looked like): it stops waiting and finishes its sweep on whatever the buffer holds, as does
            // every band that finds the abort word; nothing of the launch counts then (finalize), and nothing waits any more.
            for (int spins = 0; !dead && __builtin_amdgcn_ballot_w64(ngran == LONG_SENT) != 0; spins += 64) {
                for (int k = 0; k < 64 && __builtin_amdgcn_ballot_w64(ngran == LONG_SENT) != 0; ++k) {      // the producer is not that far yet
                    __builtin_amdgcn_s_sleep(8);
                    if (CH == 64 || lane < CH) ngran = __hip_atomic_load(bin + base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (__builtin_amdgcn_ballot_w64(ngran == LONG_SENT) == 0) break;
                if (spins >= spin_limit || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (lane == 0) __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dead = true;
                }
            }
            Hb = (int)(unsigned)(ngran & 0xFFFFFFFFu); Fb = (int)(unsigned)(ngran >> 32);
        } else if (bin) {
            Hb = 0; Fb = 0;
        } else {
            Hb = top(base + lane) - open + (SKEW ? (base + lane) * ext : 0);      // H(-1, j) - open; F(0, j) = max(-inf, H(-1, j) - open)
            Fb = SW ? 0 : Hb;
        }
        if (base + CH < T) prefetch_bound(base + CH);
    };
    // the symbol chain moves one lane per step whatever the DP does (lanes that have not started yet hand garbage on that
    // the front of real symbols overwrites before it is used); the profile row of the symbol is read one step ahead
    auto advance = [&](int (&w)[R / 2]) {
        const int sy = dpp_wave_shr(symch, symcur);
        symch = dpp_wave_shl(symch);
        symcur = sy;
        if (R == 2) w[0] = *reinterpret_cast<const int *>(prof_lane + sy);
        else if (R == 4) { const int2 v = *reinterpret_cast<const int2 *>(prof_lane + sy); w[0] = v.x; w[1] = v.y; }
        else {
#pragma unroll
            for (int x = 0; x < R / 8; ++x) {
                const int4 v = *reinterpret_cast<const int4 *>(prof_lane + sy + 16 * x);
                w[4 * x] = v.x; w[4 * x + 1] = v.y; w[4 * x + 2] = v.z; w[4 * x + 3] = v.w;
            }
        }
    };

    // one step; EDGE: lanes whose column t - g is outside [0, rl) exist (fill / drain), and the last column is captured
    auto step = [&](const int (&Hold)[R], int (&Hnew)[R], const int (&w)[R / 2], int (&wn)[R / 2], int t, auto edge, auto last_of_chunk) {
        constexpr bool EDGE = decltype(edge)::value;
        // (a new 64-column chunk of symbols can only begin behind the LAST step of a boundary chunk: the other steps do not test for it --
        //  a wave alone on its SIMD issues one instruction of any kind per 4 cycles, so the two scalar instructions of the test count)
        if (decltype(last_of_chunk)::value && ((t + 1) & 63) == 0) reload_sym(t + 1);
        advance(wn);                                       // the next step's scores, in flight while this step computes
        __builtin_amdgcn_sched_barrier(0);
        const int Hin = dpp_wave_shr(Hb, Hout), Fin = dpp_wave_shr(Fb, Fout);
        Hb = dpp_wave_shl(Hb); Fb = dpp_wave_shl(Fb);
        if (!EDGE || t >= lane) {
            int F = Fin, d = diag;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int s = (k & 1) ? (w[k / 2] >> 16) : (int)(short)(w[k / 2] & 0xFFFF);
                const int Tt = d + s;
                const int En = SKEW ? max(E[k], Hold[k]) : max(E[k] - ext, Hold[k]);
                int H = max(max(Tt, En), F);
                const int Ho = H - (SKEW ? open - ext : open);
                E[k] = En;
                F = SW ? max(max(F - ext, Ho), 0) : SKEW ? max(F, Ho) : max(F - ext, Ho);
                d = Hold[k];
                Hnew[k] = Ho;
            }
            diag = Hin; Hout = Hnew[R - 1]; Fout = F;
            if (SW) {
                int cm = Hnew[0];
#pragma unroll
                for (int k = 1; k < R; ++k) cm = max(cm, Hnew[k]);
                if (cm > best) {
                    best = cm; bcol = t;
#pragma unroll
                    for (int k = 0; k < R; ++k) hs[k] = Hnew[k];
                }
            } else {
                if (EDGE && t - lane == rl - 1) {
#pragma unroll
                    for (int k = 0; k < R; ++k) lc[k] = Hnew[k];
                }
                if (row_track) {
                    int hr = Hnew[0];
#pragma unroll
                    for (int k = 1; k < R; ++k) hr = (kstar == k) ? Hnew[k] : hr;
                    if (SKEW) hr -= (t - lane) * ext;      // (one row, many columns: the column part of the offset comes off)
                    if (lane == gstar && (!EDGE || t - lane < rl) && hr > rbest) { rbest = hr; rcol = t - lane; }
                }
            }
        }
        if (bout && lane == 63 && (!EDGE || t >= 63))
            __hip_atomic_store(bout + (t - 63), ((unsigned long long)(unsigned)Fout << 32) | (unsigned)Hout, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    };

    const int tB = rl >= 129 ? ((rl - 1) & ~63) : 64;     // [64, tB): every lane's column is inside the reference and left of its last column
    int w0[R / 2], w1[R / 2];
    prefetch_sym(0); prefetch_bound(0);
    reload_sym(0);
    advance(w0);
    for (int base = 0; base < T; base += CH) {          // (lanes CH .. 63 never hold the fill pattern: `ngran` starts at 0 there)
        reload_bound(base);
        if (base >= 64 && base < tB) {
            for (int t = base; t < base + CH - 2; t += 2) {
                step(HA, HB, w0, w1, t, std::false_type(), std::false_type());
                step(HB, HA, w1, w0, t + 1, std::false_type(), std::false_type());
            }
            step(HA, HB, w0, w1, base + CH - 2, std::false_type(), std::false_type());
            step(HB, HA, w1, w0, base + CH - 1, std::false_type(), std::true_type());
        } else {
            for (int t = base; t < base + CH - 2; t += 2) {
                step(HA, HB, w0, w1, t, std::true_type(), std::false_type());
                step(HB, HA, w1, w0, t + 1, std::true_type(), std::false_type());
            }
            step(HA, HB, w0, w1, base + CH - 2, std::true_type(), std::false_type());
            step(HB, HA, w1, w0, base + CH - 1, std::true_type(), std::true_type());
        }
    }

    // ---- this band's candidates -----------------------------------------------------------------------------------------
    int *cand = a.cand + ((size_t)pair * a.nbmax + band) * 8;
    if (SW) {
        int kf = 0;
#pragma unroll
        for (int k = R - 1; k >= 0; --k) if (hs[k] == best) kf = k;
        int sc = best + open, col = bcol - lane, row = i0 + kf;
        if (row >= ql) { sc = -1; col = 0x7FFFFFFF; row = 0x7FFFFFFF; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const int os = __shfl_xor(sc, off, 64), oc = __shfl_xor(col, off, 64), orow = __shfl_xor(row, off, 64);
            const bool take = os > sc || (os == sc && (oc < col || (oc == col && orow < row)));
            if (take) { sc = os; col = oc; row = orow; }
        }
        if (lane == 0) { cand[0] = sc; cand[1] = col; cand[2] = row; }
    } else {
        // last column: this band's best over its rows, smallest row first
        int sc = LONG_NEG, row = 0x7FFFFFFF;
#pragma unroll
        for (int k = R - 1; k >= 0; --k) {
            const int v = lc[k] + open - (SKEW ? ext + (i0 + k + rl - 1) * ext : 0);      // (the offset of cell (i0 + k, rl - 1) comes off)
            if (i0 + k < ql && v >= sc) { sc = v; row = i0 + k; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const int os = __shfl_xor(sc, off, 64), orow = __shfl_xor(row, off, 64);
            const bool take = os > sc || (os == sc && orow < row);
            if (take) { sc = os; row = orow; }
        }
        if (lane == 0) { cand[3] = sc; cand[4] = row; }
        if (lastband && lane == gstar) {
            int corner = lc[0];
#pragma unroll
            for (int k = 1; k < R; ++k) corner = (kstar == k) ? lc[k] : corner;
            cand[7] = corner + open - (SKEW ? ext + (ql - 1 + rl - 1) * ext : 0);
            cand[5] = rbest + open - (SKEW ? ext + (ql - 1) * ext : 0); cand[6] = rcol;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// TWO COLUMNS PER STEP (round 4).  One wave per SIMD cannot issue a step's dependent chain any faster than its latency: the one-column
// kernel above takes ~117 ns per step whether a lane holds 2 or 4 rows (46 instructions, 6 cycles each).  Here a lane works on the
// R x 2 block of columns 2 (t - g), 2 (t - g) + 1 per step: the block's cells form a wavefront of R + 1 levels instead of the 2 R of
// two steps, so the same instructions overlap, and everything that is per step (hand-off, symbol and boundary rotation, the
// producer's store, loop control) is shared by two columns.  The lanes' skew stays 64 STEPS -- a band starts 64 steps after the one
// above, which is now 128 columns but the same time.  Same arithmetic, same granules (two per step and lane 63), same captures.
// CH = steps per boundary chunk (2 CH columns; lanes 0 .. CH-1 load two granules each).
template <int R, int MODE, int CH>
__global__ __launch_bounds__(64)
void pmx_long32_kernel_c2(PmxLongArgs a)
{
    constexpr int BR = 64 * R;
    constexpr bool SW = MODE == PMX_MODE_SW, SG = MODE == PMX_MODE_SG;
    // SKEW (global / semi-global, round 4): every value of cell (i, j) is kept + (i + j) * ext, so neither gap needs its subtraction:
    // E(j) = max(E(j - 1), X(j - 1)), F(i) = max(F(i - 1), X(i - 1)) with X = H - (open - ext); the diagonal step crosses a row and a
    // column: + 2 ext, folded into the profile (score + open + ext).  5 instructions per cell instead of 7; boundaries and granules are in
    // the same form (the offset is global), captures take it off.  Local alignment keeps the plain form: its zero floor is per cell.
    constexpr bool SKEW = !SW;
    const int lane = threadIdx.x;
    const long long pair = blockIdx.x / a.nbmax;
    const int band = (int)(blockIdx.x % a.nbmax);
    const long long qb = a.q_shared ? 0 : a.qoff[pair], rb = a.roff[pair];
    const int ql = a.q_shared ? a.q_shared : (int)(a.qoff[pair + 1] - qb), rl = (int)(a.roff[pair + 1] - rb);
    const int NB = (ql + BR - 1) / BR;
    if (band >= NB) return;
    const bool lastband = band == NB - 1;
    const uint8_t *q = a.qbuf + qb, *r = a.rbuf + rb;
    const int msize = a.msize, open = a.open, ext = a.ext;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int16_t *prof = reinterpret_cast<int16_t *>(lds);                  // [(msize + 1) * BR]: row msize = the pad symbol
    int16_t *mat = prof + (msize + 1) * BR;                            // [msize * msize]
    unsigned char *map = reinterpret_cast<unsigned char *>(mat + msize * msize);
    for (int i = lane; i < msize * msize; i += 64) mat[i] = a.scores[i];
    for (int i = lane; i < 256; i += 64) map[i] = a.mapper[i];
    __syncthreads();
    for (int row = lane; row < BR; row += 64) {                        // the profile carries score + open (the strips carry H - open)
        const int i = band * BR + row;
        const int qs = i < ql ? (int)map[q[i]] : -1;
        for (int sym = 0; sym < msize; ++sym) prof[sym * BR + row] = (int16_t)(qs < 0 ? LONG_PAD : mat[qs * msize + sym] + open + (SKEW ? ext : 0));
        prof[msize * BR + row] = (int16_t)LONG_PAD;
    }
    __syncthreads();

    const bool pen_col = MODE == PMX_MODE_NW || (SG && !(a.sg_flags & PMX_SG_QB));
    const bool pen_row = MODE == PMX_MODE_NW || (SG && !(a.sg_flags & PMX_SG_DB));
    auto left = [&](int i) -> int { return i < 0 ? 0 : (pen_col ? -(open + i * ext) : 0); };
    auto top = [&](int j) -> int { return pen_row ? -(open + j * ext) : 0; };

    const int i0 = band * BR + lane * R;
    int H[R], E[R], hs[R], lc[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { H[k] = left(i0 + k) - open + (SKEW ? (i0 + k) * ext : 0); E[k] = LONG_NEG; hs[k] = 0; lc[k] = LONG_NEG; }
    int diag = left(i0 - 1) - open + (SKEW ? (i0 - 1) * ext : 0);
    int Hout0 = 0, Hout1 = 0, Fout0 = 0, Fout1 = 0;
    int best = SW ? -open - 1 : 0, bcol = 0;
    const int gstar = ((ql - 1) % BR) / R, kstar = (ql - 1) % R;
    int rbest = LONG_NEG, rcol = 0;
    const bool row_track = SG && (a.sg_flags & PMX_SG_DE) && lastband;

    const unsigned char *prof_lane = lds + lane * (R * 2);
    const int RU = (rl + 127) & ~127, CP = RU / 2, T = CP + 64;        // column pairs; steps
    const unsigned long long *bin = band ? a.bound + ((size_t)pair * a.nbmax + band - 1) * a.bstride : nullptr;
    unsigned long long *bout = lastband ? nullptr : a.bound + ((size_t)pair * a.nbmax + band) * a.bstride;

    // chunks of 64 steps: mapped reference symbols (two profile row offsets per lane, 16 bits each); chunks of CH steps: granules
    const int padoff = msize * (BR * 2);
    int symch = 0, symcur = padoff | (padoff << 16);
    int Hb0 = 0, Fb0 = 0, Hb1 = 0, Fb1 = 0;
    int nraw0 = 0, nraw1 = 0; unsigned long long ngA = 0, ngB = 0;
    bool dead = false;
    const int spin_limit = a.spin_limit;
    int *const abort_word = a.abort_word;
    auto prefetch_sym = [&](int base) { const int c = 2 * (base + lane); nraw0 = c < rl ? (int)r[c] : -1; nraw1 = c + 1 < rl ? (int)r[c + 1] : -1; };
    auto load_gran = [&](int base) {
        if (CH == 64 || lane < CH) {
            ngA = __hip_atomic_load(bin + 2 * (base + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ngB = __hip_atomic_load(bin + 2 * (base + lane) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto prefetch_bound = [&](int base) { if (bin) load_gran(base); };
    auto reload_sym = [&](int base) {
        const int o0 = (nraw0 < 0 ? msize : (int)map[nraw0]) * (BR * 2), o1 = (nraw1 < 0 ? msize : (int)map[nraw1]) * (BR * 2);
        symch = o0 | (o1 << 16);
        if (base + 64 < T) prefetch_sym(base + 64);
    };
    auto waiting = [&]() -> bool { return __builtin_amdgcn_ballot_w64(ngA == LONG_SENT || ngB == LONG_SENT) != 0; };
    auto reload_bound = [&](int base) {
        if (bin && base < CP) {
            // (bounded wait: see pmx_long32_kernel)
            for (int spins = 0; !dead && waiting(); spins += 64) {
                for (int k = 0; k < 64 && waiting(); ++k) {
                    __builtin_amdgcn_s_sleep(8);
                    load_gran(base);
                }
                if (!waiting()) break;
                if (spins >= spin_limit || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (lane == 0) __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dead = true;
                }
            }
            Hb0 = (int)(unsigned)(ngA & 0xFFFFFFFFu); Fb0 = (int)(unsigned)(ngA >> 32);
            Hb1 = (int)(unsigned)(ngB & 0xFFFFFFFFu); Fb1 = (int)(unsigned)(ngB >> 32);
        } else if (bin) {
            Hb0 = 0; Fb0 = 0; Hb1 = 0; Fb1 = 0;
        } else {
            Hb0 = top(2 * (base + lane)) - open + (SKEW ? 2 * (base + lane) * ext : 0); Hb1 = top(2 * (base + lane) + 1) - open + (SKEW ? (2 * (base + lane) + 1) * ext : 0);
            Fb0 = SW ? 0 : Hb0; Fb1 = SW ? 0 : Hb1;
        }
        if (base + CH < T) prefetch_bound(base + CH);
    };
    auto advance = [&](int (&wa)[R / 2], int (&wb)[R / 2]) {
        const int sy = dpp_wave_shr(symch, symcur);
        symch = dpp_wave_shl(symch);
        symcur = sy;
        const unsigned char *pa = prof_lane + (sy & 0xFFFF), *pb = prof_lane + ((unsigned)sy >> 16);
        if (R == 2) { wa[0] = *reinterpret_cast<const int *>(pa); wb[0] = *reinterpret_cast<const int *>(pb); }
        else { const int2 va = *reinterpret_cast<const int2 *>(pa), vb = *reinterpret_cast<const int2 *>(pb); wa[0] = va.x; wa[1] = va.y; wb[0] = vb.x; wb[1] = vb.y; }
    };

    auto step = [&](const int (&wa)[R / 2], const int (&wb)[R / 2], int (&na)[R / 2], int (&nb)[R / 2], int t, auto edge, auto last_of_chunk) {
        constexpr bool EDGE = decltype(edge)::value;
        if (decltype(last_of_chunk)::value && ((t + 1) & 63) == 0) reload_sym(t + 1);
        advance(na, nb);
        __builtin_amdgcn_sched_barrier(0);
        const int Hin0 = dpp_wave_shr(Hb0, Hout0), Fin0 = dpp_wave_shr(Fb0, Fout0);
        const int Hin1 = dpp_wave_shr(Hb1, Hout1), Fin1 = dpp_wave_shr(Fb1, Fout1);
        Hb0 = dpp_wave_shl(Hb0); Fb0 = dpp_wave_shl(Fb0); Hb1 = dpp_wave_shl(Hb1); Fb1 = dpp_wave_shl(Fb1);
        if (!EDGE || t >= lane) {
            int N0[R], N1[R];
            int F0 = Fin0, F1 = Fin1, d0 = diag, d1 = Hin0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                {   // column 2 (t - g), row k
                    const int s = (k & 1) ? (wa[k / 2] >> 16) : (int)(short)(wa[k / 2] & 0xFFFF);
                    const int Tt = d0 + s;
                    const int En = SKEW ? max(E[k], H[k]) : max(E[k] - ext, H[k]);
                    const int Hh = max(max(Tt, En), F0);
                    const int Ho = Hh - (SKEW ? open - ext : open);
                    E[k] = En;
                    F0 = SW ? max(max(F0 - ext, Ho), 0) : SKEW ? max(F0, Ho) : max(F0 - ext, Ho);
                    d0 = H[k];
                    N0[k] = Ho;
                }
                {   // column 2 (t - g) + 1, row k
                    const int s = (k & 1) ? (wb[k / 2] >> 16) : (int)(short)(wb[k / 2] & 0xFFFF);
                    const int Tt = d1 + s;
                    const int En = SKEW ? max(E[k], N0[k]) : max(E[k] - ext, N0[k]);
                    const int Hh = max(max(Tt, En), F1);
                    const int Ho = Hh - (SKEW ? open - ext : open);
                    E[k] = En;
                    F1 = SW ? max(max(F1 - ext, Ho), 0) : SKEW ? max(F1, Ho) : max(F1 - ext, Ho);
                    d1 = N0[k];
                    N1[k] = Ho;
                }
            }
#pragma unroll
            for (int k = 0; k < R; ++k) H[k] = N1[k];
            diag = Hin1; Hout0 = N0[R - 1]; Hout1 = N1[R - 1]; Fout0 = F0; Fout1 = F1;
            if (SW) {
                int c0 = N0[0], c1 = N1[0];
#pragma unroll
                for (int k = 1; k < R; ++k) { c0 = max(c0, N0[k]); c1 = max(c1, N1[k]); }
                if (max(c0, c1) > best) {
                    if (c0 > best) {
                        best = c0; bcol = 2 * t;
#pragma unroll
                        for (int k = 0; k < R; ++k) hs[k] = N0[k];
                    }
                    if (c1 > best) {
                        best = c1; bcol = 2 * t + 1;
#pragma unroll
                        for (int k = 0; k < R; ++k) hs[k] = N1[k];
                    }
                }
            } else {
                const int j0 = 2 * (t - lane);
                if (EDGE && (j0 == rl - 1 || j0 + 1 == rl - 1)) {
#pragma unroll
                    for (int k = 0; k < R; ++k) lc[k] = (j0 == rl - 1) ? N0[k] : N1[k];
                }
                if (row_track) {
                    int h0 = N0[0], h1 = N1[0];
#pragma unroll
                    for (int k = 1; k < R; ++k) { h0 = (kstar == k) ? N0[k] : h0; h1 = (kstar == k) ? N1[k] : h1; }
                    if (SKEW) { h0 -= j0 * ext; h1 -= (j0 + 1) * ext; }      // (one row, many columns: the column part of the offset comes off)
                    if (lane == gstar) {
                        if ((!EDGE || j0 < rl) && h0 > rbest) { rbest = h0; rcol = j0; }
                        if ((!EDGE || j0 + 1 < rl) && h1 > rbest) { rbest = h1; rcol = j0 + 1; }
                    }
                }
            }
        }
        if (bout && lane == 63 && (!EDGE || t >= 63)) {
            __hip_atomic_store(bout + 2 * (t - 63), ((unsigned long long)(unsigned)Fout0 << 32) | (unsigned)Hout0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(bout + 2 * (t - 63) + 1, ((unsigned long long)(unsigned)Fout1 << 32) | (unsigned)Hout1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    int wa0[R / 2], wb0[R / 2], wa1[R / 2], wb1[R / 2];
    prefetch_sym(0); prefetch_bound(0);
    reload_sym(0);
    advance(wa0, wb0);
    for (int base = 0; base < T; base += CH) {
        reload_bound(base);
        if (base >= 64 && 2 * (base + CH) < rl) {          // every lane's two columns are inside the reference and left of its last column
            for (int t = base; t < base + CH - 2; t += 2) {
                step(wa0, wb0, wa1, wb1, t, std::false_type(), std::false_type());
                step(wa1, wb1, wa0, wb0, t + 1, std::false_type(), std::false_type());
            }
            step(wa0, wb0, wa1, wb1, base + CH - 2, std::false_type(), std::false_type());
            step(wa1, wb1, wa0, wb0, base + CH - 1, std::false_type(), std::true_type());
        } else {
            for (int t = base; t < base + CH - 2; t += 2) {
                step(wa0, wb0, wa1, wb1, t, std::true_type(), std::false_type());
                step(wa1, wb1, wa0, wb0, t + 1, std::true_type(), std::false_type());
            }
            step(wa0, wb0, wa1, wb1, base + CH - 2, std::true_type(), std::false_type());
            step(wa1, wb1, wa0, wb0, base + CH - 1, std::true_type(), std::true_type());
        }
    }

    // ---- this band's candidates (as pmx_long32_kernel; a lane's column of step t is 2 (t - lane) [+ 1]) -----------------------
    int *cand = a.cand + ((size_t)pair * a.nbmax + band) * 8;
    if (SW) {
        int kf = 0;
#pragma unroll
        for (int k = R - 1; k >= 0; --k) if (hs[k] == best) kf = k;
        int sc = best + open, col = bcol - 2 * lane, row = i0 + kf;
        if (row >= ql) { sc = -1; col = 0x7FFFFFFF; row = 0x7FFFFFFF; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const int os = __shfl_xor(sc, off, 64), oc = __shfl_xor(col, off, 64), orow = __shfl_xor(row, off, 64);
            const bool take = os > sc || (os == sc && (oc < col || (oc == col && orow < row)));
            if (take) { sc = os; col = oc; row = orow; }
        }
        if (lane == 0) { cand[0] = sc; cand[1] = col; cand[2] = row; }
    } else {
        int sc = LONG_NEG, row = 0x7FFFFFFF;
#pragma unroll
        for (int k = R - 1; k >= 0; --k) {
            const int v = lc[k] + open - (SKEW ? ext + (i0 + k + rl - 1) * ext : 0);      // (the offset of cell (i0 + k, rl - 1) comes off)
            if (i0 + k < ql && v >= sc) { sc = v; row = i0 + k; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const int os = __shfl_xor(sc, off, 64), orow = __shfl_xor(row, off, 64);
            const bool take = os > sc || (os == sc && orow < row);
            if (take) { sc = os; row = orow; }
        }
        if (lane == 0) { cand[3] = sc; cand[4] = row; }
        if (lastband && lane == gstar) {
            int corner = lc[0];
#pragma unroll
            for (int k = 1; k < R; ++k) corner = (kstar == k) ? lc[k] : corner;
            cand[7] = corner + open - (SKEW ? ext + (ql - 1 + rl - 1) * ext : 0);
            cand[5] = rbest + open - (SKEW ? ext + (ql - 1) * ext : 0); cand[6] = rcol;
        }
    }
}

// merges the bands' candidates of every pair into its record
__global__ void pmx_long_finalize_kernel(PmxLongArgs a, int mode, int R)
{
    const long long pair = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= a.n) return;
    const int BR = 64 * R;
    const long long qb = a.q_shared ? 0 : a.qoff[pair];
    const int ql = a.q_shared ? a.q_shared : (int)(a.qoff[pair + 1] - qb), rl = (int)(a.roff[pair + 1] - a.roff[pair]);
    const int NB = (ql + BR - 1) / BR;
    const int *cand = a.cand + (size_t)pair * a.nbmax * 8;
    pmx_record_t rec; rec.flags = 0;
    if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {       // a band gave up: nothing of this launch counts
        rec.score = 0; rec.end_query = 0; rec.end_ref = 0; rec.flags = PMX_FLAG_RERUN;
        a.out[pair] = rec;
        return;
    }
    if (mode == PMX_MODE_SW) {
        int sc = -1, col = 0x7FFFFFFF, row = 0x7FFFFFFF;
        for (int b = 0; b < NB; ++b) {
            const int os = cand[8 * b], oc = cand[8 * b + 1], orow = cand[8 * b + 2];
            if (os > sc || (os == sc && (oc < col || (oc == col && orow < row)))) { sc = os; col = oc; row = orow; }
        }
        rec.score = sc; rec.end_query = row; rec.end_ref = col;
        if (sc > a.sat_above) rec.flags |= PMX_FLAG_SATURATED;
    } else {
        const int *last = cand + 8 * (NB - 1);
        const bool s1_end = mode == PMX_MODE_SG && (a.sg_flags & PMX_SG_QE), s2_end = mode == PMX_MODE_SG && (a.sg_flags & PMX_SG_DE);
        if (!s1_end && !s2_end) { rec.score = last[7]; rec.end_query = ql - 1; rec.end_ref = rl - 1; }
        else {
            int sc = LONG_NEG * 2 + 1, eq = 0, er = 0;
            if (s2_end) { sc = last[5]; eq = ql - 1; er = last[6]; }
            if (s1_end) {
                int cs = LONG_NEG * 2 + 1, crow = 0;
                for (int b = 0; b < NB; ++b) if (cand[8 * b + 3] > cs) { cs = cand[8 * b + 3]; crow = cand[8 * b + 4]; }
                if (cs > sc) { sc = cs; eq = crow; er = rl - 1; }
            }
            rec.score = sc; rec.end_query = eq; rec.end_ref = er;
        }
        if (a.force_sat) rec.flags |= PMX_FLAG_SATURATED;
    }
    a.out[pair] = rec;
}

size_t pmx_long_scratch_bytes(long long n, int max_qlen, int max_rlen, int R, long long *bstride, int *nbmax)
{
    const int BR = 64 * R;
    *nbmax = (max_qlen + BR - 1) / BR;
    *bstride = *nbmax > 1 ? (((long long)max_rlen + 127) & ~127LL) + 128 : 0;   // one band per pair: nothing is handed on (the two-column form pads to 128 columns)
    return (size_t)n * (size_t)*nbmax * ((size_t)*bstride * 8 + 32) + 64;      // + the abort word (the first 64 bytes of the scratch)
}

// 0 launched, 1 not eligible, <0 HIP error.  `scratch` = pmx_long_scratch_bytes() bytes.
int pmx_launch_long(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext, int R,
                    void *scratch, pmx_record_t *d_out, int sat_above, int force_sat, hipStream_t stream, int spin_limit, int chunk_cols, int two_cols)
{
    if (b.perm || m.msize > 64) return 1;
    if (m.max + open + ext > 32000 || m.min + open < -16000 || open < 0 || ext < 0) return 1;      // int16 profile entries (score + open [+ ext])
    if ((long long)(b.max_qlen + b.max_rlen) * (long long)(ext > open ? ext : open) > (1LL << 29)) return 1;   // boundary values stay above LONG_NEG
    PmxLongArgs a;
    a.qbuf = b.qbuf; a.qoff = b.qoff; a.q_shared = b.q_shared; a.rbuf = b.rbuf; a.roff = b.roff; a.n = b.n;
    a.scores = m.scores; a.mapper = m.mapper; a.msize = m.msize; a.sg_flags = sg_flags; a.open = open; a.ext = ext;
    long long bstride = 0; int nbmax = 0;
    const size_t bytes = pmx_long_scratch_bytes(b.n, b.max_qlen, b.max_rlen, R, &bstride, &nbmax);
    a.nbmax = nbmax; a.bstride = bstride;
    // scratch: [abort word, 64 bytes -- zeroed by the caller before the first launch of a call][boundary granules][candidates]
    a.abort_word = reinterpret_cast<int *>(scratch);
    a.spin_limit = spin_limit;
    a.bound = reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned char *>(scratch) + 64);
    const size_t bound_bytes = (size_t)b.n * nbmax * (size_t)bstride * 8;
    a.cand = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(scratch) + 64 + bound_bytes);
    a.out = d_out; a.sat_above = sat_above; a.force_sat = force_sat;
    (void)bytes;
    hipError_t e = bound_bytes ? hipMemsetAsync(a.bound, 0x80, bound_bytes, stream) : hipSuccess;
    if (e != hipSuccess) return -(int)e;
    const int BR = 64 * R;
    const size_t lds = (size_t)(m.msize + 1) * BR * 2 + (size_t)m.msize * m.msize * 2 + 256 + 16;
    if (lds > 160 * 1024) return 1;
    const long long blocks = b.n * nbmax;
    if (blocks <= 0 || blocks > 0x7FFFFFFFLL) return 1;
#define LONG_LAUNCH(RR, MM, CC) do { \
        const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_long32_kernel<RR, MM, CC>)); if (rc) return rc; \
        hipLaunchKernelGGL((pmx_long32_kernel<RR, MM, CC>), dim3((unsigned)blocks), dim3(64), lds, stream, a); } while (0)
#define LONG_MODES(RR, CC) do { \
        if (mode == PMX_MODE_SW) LONG_LAUNCH(RR, PMX_MODE_SW, CC); else if (mode == PMX_MODE_SG) LONG_LAUNCH(RR, PMX_MODE_SG, CC); else LONG_LAUNCH(RR, PMX_MODE_NW, CC); } while (0)
#define LONG_LAUNCH2(RR, MM, CC) do { \
        const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_long32_kernel_c2<RR, MM, CC>)); if (rc) return rc; \
        hipLaunchKernelGGL((pmx_long32_kernel_c2<RR, MM, CC>), dim3((unsigned)blocks), dim3(64), lds, stream, a); } while (0)
#define LONG_MODES2(RR, CC) do { \
        if (mode == PMX_MODE_SW) LONG_LAUNCH2(RR, PMX_MODE_SW, CC); else if (mode == PMX_MODE_SG) LONG_LAUNCH2(RR, PMX_MODE_SG, CC); else LONG_LAUNCH2(RR, PMX_MODE_NW, CC); } while (0)
    const bool c16 = chunk_cols == 16;
    if (two_cols && (R == 2 || R == 4)) {                  // (chunk_cols counts STEPS here: 32 or 64, two columns each)
        const bool c32 = chunk_cols != 64;
        if (R == 2) { if (c32) LONG_MODES2(2, 32); else LONG_MODES2(2, 64); }
        else { if (c32) LONG_MODES2(4, 32); else LONG_MODES2(4, 64); }
    }
    else if (R == 2) { if (c16) LONG_MODES(2, 16); else LONG_MODES(2, 64); }
    else if (R == 4) { if (c16) LONG_MODES(4, 16); else LONG_MODES(4, 64); }
    else if (R == 16) LONG_MODES(16, 64);
    else return 1;
#undef LONG_MODES2
#undef LONG_LAUNCH2
#undef LONG_MODES
#undef LONG_LAUNCH
    e = hipGetLastError();
    if (e != hipSuccess) return -(int)e;
    hipLaunchKernelGGL(pmx_long_finalize_kernel, dim3((unsigned)((b.n + 63) / 64)), dim3(64), 0, stream, a, mode, R);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
