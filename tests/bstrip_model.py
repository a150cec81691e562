"""CPU model of the band-strip kernel's arithmetic (parasail-rs_amd/csrc/pmx_bstrip.hip) -- TEST INFRASTRUCTURE.

The kernel computes only the cells of a band |(j - i) - diag| <= k, in band coordinates (row i, offset d = j - i - (diag - k)):
every value is kept in a STORED form (true value + skew + bias) inside the int16 window [1024, 31743] in which
v_pk_maximum3_f16 orders bit patterns like integers; boundary conditions are produced by the ordinary recurrences from
"virtual" columns left of the matrix (selector bytes 0x00 / 0xFF of v_perm_b32), a per-row E input at the band's left edge and
the initial state; cells beyond the band are guarded (never updated).  This file restates that arithmetic cell for cell in plain
Python integers -- same initial state, same byte scores, same selectors, same captures and tie rules, with every intermediate
checked against the window -- so that the design (not the lane mapping) can be compared with the oracle on the CPU tier
(tests/test_bstrip_model.py), and so that the host's admissibility predicate (`window_ok`, mirrored by pmx_bstrip_window_ok in
the library) is exercised at its edges without a GPU.
"""
NW, SG, SW = 0, 1, 2
S1_BEG, S1_END, S2_BEG, S2_END = 1, 2, 4, 8
NEG = -(2 ** 31) // 2          # ORC_NEG_INF = B_NEG

WIN_LO, WIN_HI = 1024, 31743


class WindowError(Exception):
    pass


def variant_of(mode, double_skew):
    """(alpha, beta) of the skew sigma = alpha * tau + beta * d, in units of ext."""
    if mode == SW:
        return 1, 0                  # row offset only: the zero floor is one value per row
    return (2, 1) if double_skew else (1, 1)


def plan(mode, m, n, open_, ext, smin, smax, k, diag, cap, double_skew):
    """Host-side geometry and constants of one pair; None when the band misses the matrix."""
    W = 2 * k + 1
    assert cap >= W
    j0 = diag - k
    if j0 > n - 1 or diag + k < -(m - 1):
        return None
    i_s = max(0, -j0 - W + 1)
    i_e = min(m - 1, n - 1 - j0)
    if i_e < i_s:
        return None
    return dict(W=W, j0=j0, i_s=i_s, i_e=i_e)


def bias_and_low(mode, sg_flags, m, n, open_, ext, smin, smax, k, cap, rows, double_skew):
    """Bias B (stored = true + sigma + B) and the LOW constant, or None when the window cannot hold the batch.
    Conservative bounds over every pair of a launch with query <= m, reference <= n, `rows` row steps at most."""
    a, b = variant_of(mode, double_skew)
    Cg = open_ - ext
    if open_ < ext or ext < 0 or Cg > 120:
        return None
    OB = Cg + a * ext
    if smin + OB < 0 or smax + OB > 254:
        return None
    LOW = WIN_LO + 2 * max(open_, ext) + 8
    L = min(m, n)
    if mode == SW:
        lo_true = 0
        hi_true = max(0, smax) * L
    else:
        # any in-band cell is reached from a boundary cell by a run of diagonal steps
        lo_true = -(open_ + max(m, n) * ext) + min(0, smin) * L - open_
        hi_true = max(0, smax) * L
    sig_hi = (a * (rows + 2) + b * (cap + 2)) * ext
    sig_lo = -(a + b) * ext * 2
    # lowest stored value of a live cell must clear LOW by the gap constants; pad rows sink by ext per row (F chain), at most `rows`
    need_lo = LOW + 2 * open_ + 300 + rows * ext
    B = need_lo - (lo_true + sig_lo)
    top = hi_true + sig_hi + B + 256
    if top > WIN_HI - 8:
        return None
    return B, LOW


def window_ok(mode, sg_flags, m, n, open_, ext, smin, smax, k, cap, rows, double_skew):
    return bias_and_low(mode, sg_flags, m, n, open_, ext, smin, smax, k, cap, rows, double_skew) is not None


def align(mode, sg_flags, q, r, open_, ext, mat, k, diag, cap, double_skew=False, B_LOW=None, check=True):
    """q, r: lists of letter indices 0..3 (4 = wildcard: not representable, caller must not pass it for r);
    mat[a][b]: 5 x 5 ints.  Returns (score, end_query, end_ref) by the kernel's arithmetic."""
    m, n = len(q), len(r)
    sw, sg = mode == SW, mode == SG
    s1_beg = sg and (sg_flags & S1_BEG); s1_end = sg and (sg_flags & S1_END)
    s2_beg = sg and (sg_flags & S2_BEG); s2_end = sg and (sg_flags & S2_END)
    col_pen = mode == NW or (sg and not s1_beg)
    row_pen = mode == NW or (sg and not s2_beg)
    smin = min(min(row[:4]) for row in mat[:5]); smax = max(max(row[:4]) for row in mat[:5])
    g = plan(mode, m, n, open_, ext, smin, smax, k, diag, cap, double_skew)
    if g is None:
        if sw or (sg and (s1_end or s2_end)):
            return NEG, 0, 0
        return NEG, m - 1, n - 1
    W, j0, i_s, i_e = g["W"], g["j0"], g["i_s"], g["i_e"]
    rows = i_e - i_s + 1
    if B_LOW is None:
        B_LOW = bias_and_low(mode, sg_flags, m, n, open_, ext, smin, smax, k, cap, rows, double_skew)
    if B_LOW is None:
        raise WindowError("window does not hold")
    B, LOW = B_LOW
    a, b = variant_of(mode, double_skew)
    alpha, beta = a * ext, b * ext
    Cg = open_ - ext
    OB = Cg + alpha
    ESUB = beta == 0              # E pays the subtraction (sw)
    FSUB = alpha == beta          # F pays the subtraction (nw/sg, single skew)

    def chk(v):
        if check and not (WIN_LO <= v <= WIN_HI):
            raise WindowError("value %d left the window" % v)
        return v

    def sigma(tau, d):
        return alpha * tau + beta * d

    def Br(j):
        return -(open_ + j * ext) if row_pen else 0

    def Bc(i):
        if i < 0:
            return 0
        return -(open_ + i * ext) if col_pen else 0

    def target(tau):          # stored value of the boundary column (column -1) in row tau (sigma of its own offset)
        i = i_s + tau
        d_b = -1 - i - j0
        return Bc(i) + sigma(tau, d_b) + B

    def K_E(tau):             # level of the virtual cells (columns <= -2) of row tau: next row's column -1 gets + 255
        return target(tau + 1) - 255 + Cg

    def sel_of(j):
        if j >= n or j <= -2:
            return 0x0C
        if j == -1:
            return 0x0C if (sw or col_pen) else 0x0D
        return r[j]

    def byte_of(i, s):
        if s == 0x0C:
            return 0
        if s == 0x0D:
            return 255
        if i < 0 or i >= m:
            return 0
        v = mat[q[i]][s] + OB
        assert 0 <= v <= 254
        return v

    # ---- initial state (row tau = -1): Hx = X form of H one row above the first, Fn = max(F, X) of that row ----
    def init_of(d):
        jp = i_s + j0 + d - 1                              # column of offset d in the row above the first one
        if sw:
            return chk(B + sigma(-1, d) - Cg), LOW         # true 0 everywhere above / left of the matrix; F below the floor
        if jp <= -2:
            return (LOW if col_pen else chk(K_E(-1) - Cg)), LOW
        if i_s == 0:
            true = 0 if jp == -1 else Br(jp)
            hx = chk(true + sigma(-1, d) + B - Cg)
            return hx, hx                                  # boundary row: F(-1, .) = -inf, so max(F, X) = X
        if jp != -1:
            return LOW, LOW                                # a real column above the band's first row: outside the band
        true = Bc(i_s - 1)
        hx = chk(true + sigma(-1, d) + B - Cg)
        # the F chain down column -1 carries the penalised boundary: F = H there, so max(F, X) = H
        return hx, (chk(true + sigma(-1, d) + B) if col_pen else hx)
    Hx = [0] * W; Fn = [0] * W
    for d in range(W):
        Hx[d], Fn[d] = init_of(d)
    Fedge = init_of(W)[1]                                  # F input of the band's last offset: the boundary row in the first row, nothing after

    best = None                # sw: (H, j, i) by true value
    brow = (NEG, 0, 0); bcol = (NEG, 0, 0); corner = NEG
    for tau in range(rows):
        i = i_s + tau
        jL = i + j0
        # E entering the band's first cell
        Zpe = B + sigma(tau, 0) + ext                      # sw: stored zero of this row (beta = 0) + ext: the floor rides on the E chain
        if sw:
            E = chk(Zpe)
        elif jL <= -1:
            E = LOW if col_pen else chk(K_E(tau))
        elif jL == 0:
            E = chk(target(tau) + beta - open_)            # E of column 0 opened from the boundary column's cell
        else:
            E = LOW
        newH = list(Hx); newF = list(Fn)
        for d in range(W):
            j = jL + d
            byte = byte_of(i, sel_of(j))
            T = chk(Hx[d] + byte)
            Fin = Fn[d + 1] if d + 1 < W else Fedge
            Fe = chk(Fin - ext) if FSUB else Fin
            Ee = chk(E - ext) if ESUB else E
            H = max(T, Ee, Fe)
            X = chk(H - Cg)
            E = max(Ee, X, Zpe) if sw else max(Ee, X)
            newF[d] = max(Fe, X)
            newH[d] = X
            chk(H)
            # ---- captures (true values) ----
            if 0 <= j < n:
                true = H - sigma(tau, d) - B
                if sw:
                    true = max(true, 0)                    # the floor sits in F: H >= Z by construction, but be explicit
                    cand = (true, j, i)
                    if best is None or cand[0] > best[0] or (cand[0] == best[0] and (cand[1], cand[2]) < (best[1], best[2])):
                        best = cand
                else:
                    if i == m - 1 and j == n - 1:
                        corner = true
                    if i == m - 1 and s2_end and (true > brow[0] or (true == brow[0] and j < brow[2])):
                        brow = (true, i, j)
                    if j == n - 1 and s1_end and (true > bcol[0] or (true == bcol[0] and i < bcol[1])):
                        bcol = (true, i, j)
        Hx, Fn = newH, newF
        Fedge = LOW
    if sw:
        if best is None:
            return NEG, 0, 0
        if best[0] == 0:
            dlo, dhi = diag - k, diag + k
            j = max(0, dlo); i = max(0, j - dhi)
            return 0, i, j
        return best[0], best[2], best[1]
    if mode == NW or not (s1_end or s2_end):
        return corner, m - 1, n - 1
    res = brow
    if s1_end and bcol[0] > res[0]:
        res = bcol
    return res
