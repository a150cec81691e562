// pmx_stats16.hip -- fast kernel for global / semi-global alignment WITH statistics (matches,
// similar, length), score + end positions + stats per pair.  BASELINE config 3:
// `nw_stats_striped_profile_16`, one reused query against many references
// (/root/reference/src/aligner/mod.rs:431-450, stats getters src/alignment/mod.rs:79-98).  gfx950 only.
//
// Strip-systolic layout (G lanes per pair, R query rows per lane in VGPRs, one reference column per
// step, one-step DPP skew), one pair per slot, unpacked lanes:
//   * DP values: full-rate VOP2 on the low 16 bits (v_add_u16 / v_sub_u16 / v_max_u16), biased by 32768.
//   * The three statistics travel with H, E and F exactly as in the oracle (coupled tables, same
//     tie-breaks: diag, then F, then E; "open" only when strictly greater).  matches and similar share
//     one register (M | S << 16) so one add and one select serve both; length has its own.
//     Every select is v_cmp_lt_u16 (VOPC -> VCC) followed by v_cndmask_b32_e32 on VCC.
//   * The query profile in LDS has two planes: score (u16) and increment (match | similar << 16).
//     With a shared query (profile arm) it is built once per 4-wave workgroup.
//   * The query is top-aligned; lane 0 gets the top boundary (H, F and their stats) arithmetically,
//     the left boundary is reproduced by the G-1 virtual columns in front of the reference
//     (score -inf + F chain when penalised, score 0 + diagonal when free; length increments are
//     switched off on virtual columns).  Needs open >= extend >= 1.
#include "pmx_common.h"
#include "pmx_switches.h"
#include <cstdlib>

#define SB 32768
#define SNEG (-16384)

__device__ __forceinline__ unsigned sa16(unsigned a, unsigned b) { unsigned r; asm("v_add_u16_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned ss16(unsigned a, unsigned b) { unsigned r; asm("v_sub_u16_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned sm16(unsigned a, unsigned b) { unsigned r; asm("v_max_u16_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// (x0, x1) = (a < b) ? (t0, t1) : (f0, f1)      -- 16-bit unsigned compare, 32-bit selects
__device__ __forceinline__ void sel2_lt(unsigned &x0, unsigned &x1, unsigned a, unsigned b,
                                        unsigned t0, unsigned t1, unsigned f0, unsigned f1)
{
    asm("v_cmp_lt_u16_e32 vcc, %2, %3\n\tv_cndmask_b32_e32 %0, %6, %4, vcc\n\tv_cndmask_b32_e32 %1, %7, %5, vcc"
        : "=&v"(x0), "=&v"(x1) : "v"(a), "v"(b), "v"(t0), "v"(t1), "v"(f0), "v"(f1) : "vcc");
}

__device__ __forceinline__ unsigned s_up(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x138, 0xF, 0xF, false); }
template <int G>
__device__ __forceinline__ unsigned s_shift_up(unsigned x)
{
    if (G <= 16) return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x111, 0xF, 0xF, false);
    return s_up(x);
}

struct SCand { int H, i, j; unsigned MS, L; };

template <int G, int R, int WAVES, bool SW>
__global__ __launch_bounds__(64 * WAVES)
void pmx_stats16_kernel(const uint8_t *__restrict__ qbuf, const int64_t *__restrict__ qoff,
                        const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff,
                        long long n, const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap,
                        int msize, int open, int ext, int RP, int q_shared,
                        int col_pen, int row_pen, int s1_end, int s2_end,
                        const unsigned *__restrict__ perm,
                        pmx_record_t *__restrict__ out, pmx_stats_t *__restrict__ stats_out)
{
    constexpr int QP = G * R;
    constexpr int NPW = 64 / G;                 // pairs per wave
    constexpr int NP = NPW * WAVES;             // pairs per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane % G, slot = wave * NPW + lane / G;
    const int MS1 = msize + 1;
    const int NPROF = q_shared ? 1 : NP;

    // LDS carve: [inc u32 NPROF*MS1*QP][score u16 NPROF*MS1*QP][rsym NP*RP][mat][map][ptab]
    unsigned *pinc = reinterpret_cast<unsigned *>(lds);
    unsigned short *psc = reinterpret_cast<unsigned short *>(pinc + NPROF * MS1 * QP);
    unsigned char *rsym = reinterpret_cast<unsigned char *>(psc + NPROF * MS1 * QP);
    int16_t *mat = reinterpret_cast<int16_t *>(rsym + NP * RP);
    unsigned char *map = reinterpret_cast<unsigned char *>(mat + msize * msize);
    long long *ptab = reinterpret_cast<long long *>(map + 256 + ((8 - ((msize * msize * 2) & 7)) & 7));

    const long long pair0 = (long long)blockIdx.x * NP;
    for (int i = tid; i < msize * msize; i += 64 * WAVES) mat[i] = gmat[i];
    for (int i = tid; i < 256; i += 64 * WAVES) map[i] = gmap[i];
    if (tid < NP) {
        long long pos = pair0 + tid; if (pos >= n) pos = n - 1;
        const long long pi = perm ? (long long)perm[pos] : pos;
        const long long qb = q_shared ? 0 : qoff[pi], rb = roff[pi];
        ptab[5 * tid + 0] = qb;
        ptab[5 * tid + 1] = q_shared ? q_shared : (qoff[pi + 1] - qb);
        ptab[5 * tid + 2] = rb;
        ptab[5 * tid + 3] = roff[pi + 1] - rb;
        ptab[5 * tid + 4] = (pair0 + tid < n) ? pi : -1;
    }
    __syncthreads();
    const uint8_t *qbase = qbuf;
    const uint8_t *rbase = rbuf;

    for (int p = 0; p < NP; ++p) {
        const int rlp = (int)ptab[5 * p + 3];
        const uint8_t *rp = rbase + ptab[5 * p + 2];
        for (int j0 = 0; j0 < RP; j0 += 64 * WAVES * 4) {
            unsigned char raw[4]; bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 64 * WAVES + tid, jj = j - (G - 1);
                ok[u] = j < RP && jj >= 0 && jj < rlp;
                raw[u] = ok[u] ? rp[jj] : (unsigned char)0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 64 * WAVES + tid;
                if (j < RP) rsym[p * RP + j] = ok[u] ? map[raw[u]] : (unsigned char)msize;
            }
        }
    }
    const int vcol_score = col_pen ? SNEG : 0;
    for (int p = 0; p < NPROF; ++p) {
        const int qlp = (int)ptab[5 * p + 1];
        const uint8_t *qp = qbase + ptab[5 * p + 0];
        for (int er = tid; er < QP; er += 64 * WAVES) {
            const int q0 = (er < qlp) ? (int)map[qp[er]] : -1;
            for (int sym = 0; sym < msize; ++sym) {
                const int s = (q0 < 0) ? 0 : mat[q0 * msize + sym];
                psc[(p * MS1 + sym) * QP + er] = (unsigned short)s;
                pinc[(p * MS1 + sym) * QP + er] = (q0 < 0) ? 0u : (unsigned)(q0 == sym) | ((unsigned)(s > 0) << 16);
            }
            psc[(p * MS1 + msize) * QP + er] = (unsigned short)((q0 < 0) ? 0 : vcol_score);
            pinc[(p * MS1 + msize) * QP + er] = 0u;
        }
    }
    __syncthreads();

    // ---- per-lane state --------------------------------------------------------------------
    const int pslot = q_shared ? 0 : slot;
    const unsigned short *scL = psc + pslot * MS1 * QP + g * R;
    const unsigned *incL = pinc + pslot * MS1 * QP + g * R;
    const unsigned char *rs = rsym + slot * RP + (G - 1) - g;
    const int ql = (int)ptab[5 * slot + 1], rl = (int)ptab[5 * slot + 3];
    const unsigned vOpen = (unsigned)open, vExt = (unsigned)ext;
    const int gL = (ql - 1) / R, kL = (ql - 1) % R;       // owner of the last query row

    unsigned HA[R], HB[R], MSA[R], MSB[R], LA[R], LB[R], E[R], EMS[R], EL[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int i = g * R + k;
        HA[k] = (unsigned)(SB + (col_pen ? -(open + i * ext) : 0)); HB[k] = HA[k];
        MSA[k] = MSB[k] = 0u;
        LA[k] = LB[k] = col_pen ? (unsigned)(i + 1) : 0u;
        E[k] = HA[k] - vOpen; EMS[k] = 0u; EL[k] = LA[k] + 1u;
    }
    unsigned Hout = HA[R - 1], HMSout = 0u, HLout = LA[R - 1];
    unsigned Fout, FMSout = 0u, FLout;
    {
        const int i = (g + 1) * R;                 // first row of the lane below, at a virtual column
        Fout = (unsigned)(SB + (col_pen ? -(open + i * ext) : -open));
        FLout = col_pen ? (unsigned)(i + 1) : 1u;
    }
    unsigned diag0 = (g == 0) ? (unsigned)SB : (unsigned)(SB + (col_pen ? -(open + (g * R - 1) * ext) : 0));
    unsigned diagMS0 = 0u, diagL0 = (g == 0) ? 0u : (col_pen ? (unsigned)(g * R) : 0u);

    SCand corner = {-(1 << 30), 0, 0, 0u, 0u}, brow = corner, bcol = corner;
    // local: running best of the lane, first column that reached it, H / stats strips at that column
    unsigned swbest = (unsigned)SB, swcol = 0u, svH[R], svMS[R], svL[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { svH[k] = (unsigned)SB; svMS[k] = 0u; svL[k] = 0u; }

    auto load_scores = [&](int sym, unsigned (&w)[R], unsigned (&wi)[R]) {
#pragma unroll
        for (int k = 0; k < R; ++k) { w[k] = scL[sym * QP + k]; wi[k] = incL[sym * QP + k]; }
    };
    auto step = [&](const unsigned (&Hold)[R], const unsigned (&MSold)[R], const unsigned (&Lold)[R],
                    unsigned (&Hnew)[R], unsigned (&MSnew)[R], unsigned (&Lnew)[R],
                    const unsigned (&w)[R], const unsigned (&wi)[R], int t) {
        const int jcol = t - g;
        const unsigned linc = (jcol >= 0 && jcol < rl) ? 1u : 0u;
        unsigned Hin = s_shift_up<G>(Hout), HMSin = s_shift_up<G>(HMSout), HLin = s_shift_up<G>(HLout);
        unsigned F = s_shift_up<G>(Fout), FMS = s_shift_up<G>(FMSout), FL = s_shift_up<G>(FLout);
        if (g == 0) {                            // top boundary of column t
            const unsigned topH = (unsigned)(SB + (row_pen ? -(open + t * ext) : 0));
            const unsigned topL = row_pen ? (unsigned)(t + 1) : 0u;
            Hin = topH; HMSin = 0u; HLin = topL;
            F = topH - vOpen; FMS = 0u; FL = topL + 1u;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const unsigned d = (k == 0) ? diag0 : Hold[k - 1];
            const unsigned dMS = (k == 0) ? diagMS0 : MSold[k - 1];
            const unsigned dL = (k == 0) ? diagL0 : Lold[k - 1];
            const unsigned Tt = sa16(d, w[k]);
            const unsigned TMS = dMS + wi[k], TL = dL + linc;
            unsigned H = sm16(sm16(Tt, E[k]), F);
            unsigned xMS, xL, hMS, hL;
            sel2_lt(xMS, xL, F, H, EMS[k], EL[k], FMS, FL);        // not from F -> E's stats, else F's
            sel2_lt(hMS, hL, Tt, H, xMS, xL, TMS, TL);             // not diagonal -> gap stats, else diagonal's
            if (SW) {                                              // local: H <= 0 restarts the alignment
                unsigned zMS, zL;
                sel2_lt(zMS, zL, (unsigned)SB, H, hMS, hL, 0u, 0u);
                hMS = zMS; hL = zL;
                H = sm16(H, (unsigned)SB);
            }
            const unsigned Ho = ss16(H, vOpen), Ee = ss16(E[k], vExt), Fe = ss16(F, vExt);
            unsigned eMS, eL, fMS, fL;
            sel2_lt(eMS, eL, Ee, Ho, hMS, hL, EMS[k], EL[k]);      // E opened from H
            sel2_lt(fMS, fL, Fe, Ho, hMS, hL, FMS, FL);            // F opened from H
            EMS[k] = eMS; EL[k] = eL + 1u;
            FMS = fMS; FL = fL + 1u;
            E[k] = sm16(Ee, Ho);
            F = sm16(Fe, Ho);
            Hnew[k] = H; MSnew[k] = hMS; Lnew[k] = hL;
        }
        diag0 = Hin; diagMS0 = HMSin; diagL0 = HLin;
        Hout = Hnew[R - 1]; HMSout = MSnew[R - 1]; HLout = Lnew[R - 1];
        Fout = F; FMSout = FMS; FLout = FL;

        // ---- captures ----
        if (SW) {
            unsigned cm = Hnew[0] & 0xFFFFu;
#pragma unroll
            for (int k = 1; k < R; ++k) cm = sm16(cm, Hnew[k]) & 0xFFFFu;
            const bool imp = cm > swbest;
            swbest = imp ? cm : swbest;
            swcol = imp ? (unsigned)jcol : swcol;
#pragma unroll
            for (int k = 0; k < R; ++k) { svH[k] = imp ? Hnew[k] : svH[k]; svMS[k] = imp ? MSnew[k] : svMS[k]; svL[k] = imp ? Lnew[k] : svL[k]; }
            return;
        }
        if (jcol >= 0 && jcol < rl) {
            if (g == gL && (s2_end || jcol == rl - 1)) {
                unsigned h = 0, ms = 0, l = 0;
#pragma unroll
                for (int k = 0; k < R; ++k) if (k == kL) { h = Hnew[k]; ms = MSnew[k]; l = Lnew[k]; }
                const int hv = (int)(h & 0xFFFF);
                if (jcol == rl - 1) { corner.H = hv; corner.i = ql - 1; corner.j = jcol; corner.MS = ms; corner.L = l; }
                if (s2_end && hv > brow.H) { brow.H = hv; brow.i = ql - 1; brow.j = jcol; brow.MS = ms; brow.L = l; }
            }
            if (s1_end && jcol == rl - 1) {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int i = g * R + k, hv = (int)(Hnew[k] & 0xFFFF);
                    if (i < ql && hv > bcol.H) { bcol.H = hv; bcol.i = i; bcol.j = jcol; bcol.MS = MSnew[k]; bcol.L = Lnew[k]; }
                }
            }
        }
    };

    int max_rlen = 0;
#pragma unroll
    for (int p = 0; p < NPW; ++p) max_rlen = max(max_rlen, (int)ptab[5 * (wave * NPW + p) + 3]);
    const int T = (max_rlen + G - 1 + 1) & ~1;
    unsigned w0[R], wi0[R], w1[R], wi1[R];
    load_scores(rs[0], w0, wi0);
    int ns = rs[1];
    for (int t = 0; t < T; t += 2) {
        load_scores(ns, w1, wi1);
        ns = rs[t + 2];
        __builtin_amdgcn_sched_barrier(0);
        step(HA, MSA, LA, HB, MSB, LB, w0, wi0, t);
        __builtin_amdgcn_sched_barrier(0);
        load_scores(ns, w0, wi0);
        ns = rs[t + 3];
        __builtin_amdgcn_sched_barrier(0);
        step(HB, MSB, LB, HA, MSA, LA, w1, wi1, t + 1);
        __builtin_amdgcn_sched_barrier(0);
    }

    if (SW) {
        int krow = 0; unsigned kMS = svMS[0], kL = svL[0];
#pragma unroll
        for (int k = R - 1; k >= 0; --k) if ((svH[k] & 0xFFFFu) == swbest) { krow = k; kMS = svMS[k]; kL = svL[k]; }
        const unsigned long long mykey = ((unsigned long long)swbest << 32) | ((unsigned long long)(0xFFFFu - (swcol & 0xFFFFu)) << 16) |
                                         (unsigned long long)(0xFFFFu - (unsigned)(g * R + krow));
        unsigned long long key = mykey;
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off, 64);
            key = o > key ? o : key;
        }
        const unsigned long long win = __ballot(key == mykey);
        const unsigned long long slotmask = (G == 64) ? ~0ULL : (((1ULL << G) - 1ULL) << ((lane / G) * G));
        const int wl = __builtin_ctzll(win & slotmask);
        const unsigned wMS = __shfl(kMS, wl, 64), wL = __shfl(kL, wl, 64);
        if (g == 0) {
            const long long pi = ptab[5 * slot + 4];
            if (pi >= 0) {
                pmx_record_t rec; rec.flags = 0;
                rec.score = (int)(key >> 32) - SB;
                rec.end_ref = 0xFFFF - (int)((key >> 16) & 0xFFFF);
                rec.end_query = 0xFFFF - (int)(key & 0xFFFF);
                pmx_stats_t st; st.matches = (int)(wMS & 0xFFFF); st.similar = (int)(wMS >> 16); st.length = (int)wL;
                if (rec.score == 0) { rec.end_query = 0; rec.end_ref = 0; st.matches = st.similar = st.length = 0; }
                out[pi] = rec; stats_out[pi] = st;
            }
        }
        return;
    }
    // ---- combine: last-column candidates over the slot (value desc, row asc), then the oracle's rule ----
    unsigned key = ((unsigned)(bcol.H < 0 ? 0 : bcol.H) << 16) | (0xFFFFu - (unsigned)bcol.i);
    unsigned best = key;
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) {
        const unsigned o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    const unsigned long long win = __ballot(best == key && bcol.H >= 0);
    // lanes of this slot only
    const unsigned long long slotmask = (G == 64) ? ~0ULL : (((1ULL << G) - 1ULL) << ((lane / G) * G));
    const int wl = (win & slotmask) ? __builtin_ctzll(win & slotmask) : (lane / G) * G;
    SCand bc;
    bc.H = __shfl(bcol.H, wl, 64); bc.i = __shfl(bcol.i, wl, 64); bc.j = __shfl(bcol.j, wl, 64);
    bc.MS = __shfl(bcol.MS, wl, 64); bc.L = __shfl(bcol.L, wl, 64);
    const int ll = (lane / G) * G + gL;
    SCand co, br;
    co.H = __shfl(corner.H, ll, 64); co.i = __shfl(corner.i, ll, 64); co.j = __shfl(corner.j, ll, 64);
    co.MS = __shfl(corner.MS, ll, 64); co.L = __shfl(corner.L, ll, 64);
    br.H = __shfl(brow.H, ll, 64); br.i = __shfl(brow.i, ll, 64); br.j = __shfl(brow.j, ll, 64);
    br.MS = __shfl(brow.MS, ll, 64); br.L = __shfl(brow.L, ll, 64);
    if (g == 0) {
        const long long pi = ptab[5 * slot + 4];
        if (pi >= 0) {
            SCand res;
            if (!s1_end && !s2_end) res = co;
            else {
                res.H = -(1 << 30); res.i = res.j = 0; res.MS = res.L = 0;
                if (s2_end) res = br;
                if (s1_end && bc.H > res.H) res = bc;
            }
            pmx_record_t rec; rec.score = res.H - SB; rec.end_query = res.i; rec.end_ref = res.j; rec.flags = 0;
            out[pi] = rec;
            pmx_stats_t st; st.matches = (int)(res.MS & 0xFFFF); st.similar = (int)(res.MS >> 16); st.length = (int)res.L;
            stats_out[pi] = st;
        }
    }
}

// ------------------------------------------------------------------------ host side ----
template <int G, int R, int WAVES, bool SW>
static int launch_stats(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                        pmx_record_t *d_out, pmx_stats_t *d_stats, hipStream_t stream)
{
    constexpr int QP = G * R, NP = (64 / G) * WAVES;
    const int RP = ((b.max_rlen + 2 * (G - 1) + 4 + 7) / 4) * 4;
    const int nprof = b.q_shared ? 1 : NP;
    const size_t lds = (size_t)nprof * (m.msize + 1) * QP * 6 + (size_t)NP * RP +
                       (size_t)m.msize * m.msize * 2 + 256 + 8 + (size_t)NP * 40;
    if (lds > 160 * 1024) return 1;
    { const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_stats16_kernel<G, R, WAVES, SW>)); if (rc) return rc; }
    const bool sg = mode == PMX_MODE_SG;
    const int col_pen = SW ? 0 : !(sg && (sg_flags & PMX_SG_QB)), row_pen = SW ? 0 : !(sg && (sg_flags & PMX_SG_DB));
    const int s1_end = sg && (sg_flags & PMX_SG_QE), s2_end = sg && (sg_flags & PMX_SG_DE);
    const long long blocks = (b.n + NP - 1) / NP;
    hipLaunchKernelGGL((pmx_stats16_kernel<G, R, WAVES, SW>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, stream,
                       b.qbuf, b.qoff, b.rbuf, b.roff, (long long)b.n, m.scores, m.mapper,
                       m.msize, open, ext, RP, b.q_shared, col_pen, row_pen, s1_end ? 1 : 0, s2_end ? 1 : 0,
                       b.perm, d_out, d_stats);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// 0 launched, 1 not eligible (caller uses the general kernel), <0 HIP error
int pmx_launch_stats16(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                       pmx_record_t *d_out, pmx_stats_t *d_stats, hipStream_t stream, const char **kernel_name)
{
    if (pmx_env("PMX_NO_FAST_STATS")) return 1;
    if (mode != PMX_MODE_NW && mode != PMX_MODE_SG && mode != PMX_MODE_SW) return 1;
    if (m.msize > PMX_MAX_FAST_MSIZE - 1) return 1;
    if (ext < 1 || open < ext || open > 4096) return 1;
    if (b.max_rlen > 30000 || b.max_qlen + b.max_rlen > 60000) return 1;
    const long long lo = mode == PMX_MODE_SW ? -(2LL * open + 2LL * ext + (m.min < 0 ? -m.min : 0))
                                             : -(3LL * open + (long long)(b.max_qlen + b.max_rlen + 2) * ext + (m.min < 0 ? -m.min : 0));
    const long long hi = (long long)(b.max_qlen < b.max_rlen ? b.max_qlen : b.max_rlen) * (m.max > 0 ? m.max : 0) + (m.max > 0 ? m.max : 0);
    if (lo < -15000 || hi > 15000) return 1;
    const int q = b.max_qlen;
    const int W = b.q_shared ? 4 : 1;      // a shared query profile is built once per 4-wave workgroup
#define TRYS(GG, RR, NAME)                                                      \
    if (q <= (GG) * (RR)) {                                                     \
        int rc = mode == PMX_MODE_SW                                                                        \
            ? (W == 4 ? launch_stats<GG, RR, 4, true>(b, m, mode, sg_flags, open, ext, d_out, d_stats, stream)  \
                      : launch_stats<GG, RR, 1, true>(b, m, mode, sg_flags, open, ext, d_out, d_stats, stream)) \
            : (W == 4 ? launch_stats<GG, RR, 4, false>(b, m, mode, sg_flags, open, ext, d_out, d_stats, stream) \
                      : launch_stats<GG, RR, 1, false>(b, m, mode, sg_flags, open, ext, d_out, d_stats, stream)); \
        if (rc <= 0) { if (kernel_name) *kernel_name = NAME; return rc; }       \
    }
    if (b.n <= 64 && W == 1) {          // one pair or a handful (Aligner::align()): all 64 lanes on the pair, few rows per lane
        TRYS(64, 3, "pmx_stats16_kernel<64,3>")
        TRYS(64, 5, "pmx_stats16_kernel<64,5>")
    }
    TRYS(16, 10, "pmx_stats16_kernel<16,10>")
    TRYS(32, 8, "pmx_stats16_kernel<32,8>")
    TRYS(64, 5, "pmx_stats16_kernel<64,5>")
    TRYS(64, 8, "pmx_stats16_kernel<64,8>")
    TRYS(64, 16, "pmx_stats16_kernel<64,16>")
#undef TRYS
    return 1;
}
