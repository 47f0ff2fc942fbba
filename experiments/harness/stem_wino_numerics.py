#!/usr/bin/env python3
"""The 7x7 stride-2 stem as four stride-1 phase convolutions (the filter zero-padded to 8x8, every phase a 4x4 filter) in
Winograd F(2x2, 4x4) form: is it exact enough for the 1e-4 bar, and what do the transforms look like?

    python experiments/harness/stem_wino_numerics.py

out[y][x] = sum_{ky,kx,c} w[ky][kx][c] in[2y + ky - 3][2x + kx - 3]            (superpoint.py:12, padding 3)
Pad the filter with a zero row / column in FRONT: w8[ky + 1][kx + 1] = w[ky][kx].  With ky8 = 2u + a (u = 0..3, a = 0, 1):
input row 2y + ky8 - 4 = 2(y + u - 2) + a  ->  phase image P_a[i] = in[2i + a], row i = y + u - 2.  So
out[y][x] = sum_{a,b,c} sum_{u,v<4} w8[2u + a][2v + b][c] P_ab,c[y + u - 2][x + v - 2]: four 4x4 correlations.
F(2,4): 5 interpolation points; 25 multiplications per 2x2 outputs and (phase, channel) instead of 64."""
import numpy as np

def winograd_matrices(points):
    """Cook-Toom F(m=2, r=4) for n = 5 points (the last one infinity): AT [2 x 5], G [5 x 4], BT [5 x 5], in float64."""
    from fractions import Fraction
    import itertools
    pts = [Fraction(p) for p in points]          # 4 finite points + infinity
    n, m, r = 5, 2, 4
    # polynomial (Toom-Cook) construction via Vandermonde matrices
    def vand(rows, cols, inf_col):
        M = [[(p ** j) for j in range(cols)] for p in pts]
        M.append([Fraction(0)] * (cols - 1) + [Fraction(1)])
        return M
    A = vand(n, m, True)      # n x m   (evaluation of the output polynomial ... transposed use below)
    G = vand(n, r, True)      # n x r
    # B^T from the interpolation: solve so that  Y = A^T [(G g) * (B^T d)]  equals the correlation
    # Use the standard identity: B^T = inverse of the n x n Vandermonde V (points, infinity) transposed appropriately, scaled.
    V = [[(p ** j) for j in range(n)] for p in pts] + [[Fraction(0)] * (n - 1) + [Fraction(1)]]
    import sympy
    Vs = sympy.Matrix(V)
    BT = (Vs.inv()).T          # 5 x 5
    # scale rows so that G carries the denominators (common practice): here keep G = evaluation, and fold N_i = prod_{j != i}(p_i - p_j) into G
    Gs = sympy.Matrix(G)
    for i in range(4):
        Ni = 1
        for j in range(4):
            if j != i:
                Ni *= (pts[i] - pts[j])
        Gs[i, :] = Gs[i, :] / Ni
        BT[i, :] = BT[i, :] * Ni
    AT = sympy.Matrix(A).T
    return np.array(AT.tolist(), float), np.array(Gs.tolist(), float), np.array(BT.tolist(), float)

def check(points, rng):
    AT, G, BT = winograd_matrices(points)
    # 1-D sanity in float64: y = AT [(G g) * (BT d)] == correlation
    g = rng.standard_normal(4); d = rng.standard_normal(5)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([np.dot(g, d[0:4]), np.dot(g, d[1:5])])
    assert np.allclose(y, ref, atol=1e-12), (y, ref)
    return AT, G, BT

def stem_direct(x, w):
    """x [3,H,W] float64, w [64,3,7,7] -> [64,H/2,W/2] (stride 2, padding 3)."""
    C, H, W = x.shape
    xp = np.zeros((C, H + 6, W + 6)); xp[:, 3:-3, 3:-3] = x
    out = np.zeros((w.shape[0], H // 2, W // 2))
    for ky in range(7):
        for kx in range(7):
            patch = xp[:, ky:ky + H:2, kx:kx + W:2]
            out += np.einsum('nc,chw->nhw', w[:, :, ky, kx], patch)
    return out

def stem_wino(x, w, AT, G, BT, dtype):
    C, H, W = x.shape
    Ho, Wo = H // 2, W // 2
    w8 = np.zeros((w.shape[0], C, 8, 8)); w8[:, :, 1:, 1:] = w
    xp = np.zeros((C, H + 8, W + 8)); xp[:, 4:-4, 4:-4] = x          # in[r] at xp[r + 4]
    out = np.zeros((w.shape[0], Ho, Wo), dtype)
    U = {}
    for a in range(2):
        for b in range(2):
            g = w8[:, :, a::2, b::2]                                  # [64, 3, 4, 4]: g[u][v] = w8[2u + a][2v + b]
            U[a, b] = np.einsum('iu,ncuv,jv->ncij', G, g, G).astype(dtype)
    AT32, BT32 = AT.astype(dtype), BT.astype(dtype)
    for ty in range(Ho // 2):
        for tx in range(Wo // 2):
            M = np.zeros((w.shape[0], 5, 5), dtype)
            for a in range(2):
                for b in range(2):
                    # P_ab[i][j] = in[2i + a][2j + b]; rows i = y + u - 2 for y = 2ty, 2ty+1, u = 0..3 -> i = 2ty - 2 .. 2ty + 2
                    i0, j0 = 2 * ty - 2, 2 * tx - 2
                    d = xp[:, 2 * i0 + a + 4:2 * (i0 + 5) + a + 4:2, 2 * j0 + b + 4:2 * (j0 + 5) + b + 4:2].astype(dtype)   # [3,5,5]
                    V = np.einsum('iu,cuv,jv->cij', BT32, d, BT32).astype(dtype)
                    M += np.einsum('ncij,cij->nij', U[a, b], V).astype(dtype)
            Y = np.einsum('yi,nij,xj->nyx', AT32, M, AT32)
            out[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = Y
    return out

def main():
    rng = np.random.default_rng(0)
    x = rng.random((3, 32, 48))                                      # frames are in [0, 1]
    w = rng.standard_normal((64, 3, 7, 7)) * np.sqrt(2.0 / 147)      # He-scaled, as synth.make_state_dict
    ref = stem_direct(x, w)
    scale = np.abs(ref).max()
    for name, pts in (("0, 1, -1, 2, inf", (0, 1, -1, 2)), ("0, 1, -1, 1/2, inf", (0, 1, -1, "1/2")), ("0, 1/2, -1/2, 1, inf", (0, "1/2", "-1/2", 1)), ("0,1,-1,-2", (0, 1, -1, -2))):
        AT, G, BT = check(pts, rng)
        o64 = stem_wino(x, w, AT, G, BT, np.float64)
        o32 = stem_wino(x.astype(np.float32), w, AT, G, BT, np.float32)
        d32 = stem_direct(x.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64))
        print("%-22s fp64 err %.2e   fp32 err %.2e (of scale %.2f; relative %.2e)" % (name, np.abs(o64 - ref).max(), np.abs(o32 - ref).max(), scale, np.abs(o32 - ref).max() / scale))
    print("AT", AT, "G", G, "BT", BT, sep="\n")

if __name__ == "__main__":
    main()
