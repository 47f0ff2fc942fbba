# numerics of Winograd F(4x4,3x3) in fp32 vs F(2x2,3x3) and direct fp32, against fp64 direct.
import numpy as np, torch, torch.nn.functional as F
torch.manual_seed(0)
def wino_mats_43(variant):
    if variant == "std":   # points 0, 1, -1, 2, -2, inf (Lavin)
        BT = np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],dtype=np.float64)
        G = np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],dtype=np.float64)
        AT = np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],dtype=np.float64)
    else:  # points 0, 1, -1, 1/2, -1/2, inf  -- derive numerically
        pts = [0.0, 1.0, -1.0, 0.5, -0.5]
        return cook_toom(pts, 4, 3)
    return BT, G, AT
def cook_toom(pts, m, r):
    # returns BT (n x n), G (n x r), AT (m x n) with n = m + r - 1, last point = infinity; exact in float64 via fractions
    from fractions import Fraction as Fr
    n = m + r - 1
    p = [Fr(x).limit_denominator(64) for x in pts]
    assert len(p) == n - 1
    # AT[i][j] = p_j^i (j < n-1); inf column: 1 at i = m-1
    AT = [[(p[j] ** i) for j in range(n - 1)] + [Fr(1 if i == m - 1 else 0)] for i in range(m)]
    # G[j][k] = p_j^k / N_j, N_j = prod_{l != j}(p_j - p_l); inf row: [0,0,1]
    G = []
    for j in range(n - 1):
        Nj = Fr(1)
        for l in range(n - 1):
            if l != j: Nj *= (p[j] - p[l])
        G.append([p[j] ** k / Nj for k in range(r)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    # BT: rows j<n-1: coefficients of M_j(x) = prod_{l != j}(x - p_l) ; last row: coefficients of M(x) = prod_l (x - p_l)
    def polymul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for k, y in enumerate(b): out[i + k] += x * y
        return out
    BT = []
    for j in range(n - 1):
        poly = [Fr(1)]
        for l in range(n - 1):
            if l != j: poly = polymul(poly, [-p[l], Fr(1)])
        BT.append(poly + [Fr(0)] * (n - len(poly)))
    poly = [Fr(1)]
    for l in range(n - 1): poly = polymul(poly, [-p[l], Fr(1)])
    BT.append(poly)
    f = lambda M: np.array([[float(x) for x in row] for row in M], dtype=np.float64)
    return f(BT), f(G), f(AT)

def check(BT, G, AT, m):
    # 1-D identity check
    r = 3; n = m + r - 1
    d = np.random.randn(n); g = np.random.randn(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    return np.abs(y - ref).max()

def wino_conv(x, w, BT, G, AT, m, scale=None):
    # x [C,H,W] fp32 with H,W multiples of m (pad 1 added here); w [O,C,3,3] float64. fp32 arithmetic in V, M, Y.
    C, H, W = x.shape; O = w.shape[0]; n = m + 2
    U = np.einsum('ij,ocjk,lk->ocil', G, w, G)            # fp64 on the host
    if scale is not None:  # row/col scaling to balance magnitudes: U' = U * s_i s_l ; V' = V / (s_i s_l)
        U = U * scale[None, None, :, None] * scale[None, None, None, :]
    U = U.astype(np.float32)
    xp = np.zeros((C, H + 2, W + 2), np.float32); xp[:, 1:-1, 1:-1] = x
    th, tw = H // m, W // m
    out = np.zeros((O, H, W), np.float32)
    BTf = BT.astype(np.float32); ATf = AT.astype(np.float32)
    if scale is not None:
        BTf = (BT / scale[:, None]).astype(np.float32)
    tiles = np.zeros((C, th, tw, n, n), np.float32)
    for i in range(th):
        for j in range(tw):
            tiles[:, i, j] = xp[:, i*m:i*m+n, j*m:j*m+n]
    V = np.einsum('ij,ctujk,lk->ctuil', BTf, tiles, BTf).astype(np.float32)   # (numpy accumulates in fp32 for fp32 inputs)
    # M: sum over c sequentially in fp32 like an fmaf chain
    M = np.zeros((O, th, tw, n, n), np.float32)
    for c in range(C):
        M += U[:, c][:, None, None] * V[c][None]
    Y = np.einsum('ij,otujk,lk->otuil', ATf, M, ATf).astype(np.float32)
    for i in range(th):
        for j in range(tw):
            out[:, i*m:(i+1)*m, j*m:(j+1)*m] = Y[:, i, j]
    return out

np.random.seed(0)
C, O, H, W = 128, 128, 24, 24
# activations like post-ReLU maps; weights He-scaled like synth
x = np.maximum(np.random.randn(C, H, W), 0).astype(np.float32)
w = (np.random.randn(O, C, 3, 3) * np.sqrt(2.0 / (C * 9)))
ref = F.conv2d(torch.from_numpy(x.astype(np.float64))[None], torch.from_numpy(w), padding=1)[0].numpy()
d32 = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w.astype(np.float32)), padding=1)[0].numpy()
sc = np.abs(ref).max()
print("scale", sc, "direct fp32 max err", np.abs(d32 - ref).max(), "rel", np.abs(d32 - ref).max() / sc)
BT2 = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],dtype=np.float64)
G2 = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],dtype=np.float64)
AT2 = np.array([[1,1,1,0],[0,1,-1,-1]],dtype=np.float64)
y = wino_conv(x, w, BT2, G2, AT2, 2)
print("F(2,3) max err", np.abs(y - ref).max(), "rel", np.abs(y - ref).max() / sc, "rms", np.sqrt(np.mean((y-ref)**2))/sc)
for name, pts in (("std 0,1,-1,2,-2", [0,1,-1,2,-2]), ("0,1,-1,1/2,-1/2", [0,1,-1,.5,-.5]), ("0,1,-1,1/2,-2", [0,1,-1,.5,-2]), ("0,1,-1,2,-1/2", [0,1,-1,2,-.5])):
    BT, G, AT = cook_toom(pts, 4, 3)
    print(name, "identity err", check(BT, G, AT, 4))
    y = wino_conv(x, w, BT, G, AT, 4)
    print("  F(4,3) max err", np.abs(y - ref).max(), "rel", np.abs(y - ref).max() / sc, "rms", np.sqrt(np.mean((y-ref)**2))/sc)
    print("  BT\n", BT, "\n  AT\n", AT)
