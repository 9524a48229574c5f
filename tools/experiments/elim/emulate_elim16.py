"""Lane-level numpy emulation of the 4x4-blocked Cholesky + inverse of a 16 x 16 Hermitian block on one wave
(the index bookkeeping of the MFMA form sketched in docs/HISTORY.md section 10.10): checks L L^H = D and W = L^-1."""
import numpy as np

rng = np.random.default_rng(0)
A = rng.standard_normal((16, 16)) + 1j * rng.standard_normal((16, 16))
D0 = A @ A.conj().T + 16 * np.eye(16)

lanes = [(l & 15, l >> 4) for l in range(64)]          # (li, g)
ROW = lambda g, v: g + 4 * v                           # HPX_ACC_ROW

# accumulator-layout registers: X[l][v] = M[ROW(g, v)][li]
Dacc = np.zeros((64, 4), complex)
Racc = np.zeros((64, 4), complex)
for l, (li, g) in enumerate(lanes):
    for v in range(4):
        r, c = ROW(g, v), li
        Dacc[l, v] = D0[r, c] if r >= c else np.conj(D0[c, r])      # lower from LDS, mirrored above the diagonal
        Racc[l, v] = 1.0 if r == c else 0.0
Lout = np.zeros((16, 16), complex)
Wout = np.zeros((16, 16), complex)


def mfma(Aop, Bop, Cacc):
    """C[l][v] += sum_k A[ROW(g,v)][k] B[k][li]; lane l supplies A[li][g] = Aop[l], B[g][li] = Bop[l]."""
    Am = np.zeros((16, 4), complex)
    Bm = np.zeros((4, 16), complex)
    for l, (li, g) in enumerate(lanes):
        Am[li, g] = Aop[l]
        Bm[g, li] = Bop[l]
    P = Am @ Bm
    out = Cacc.copy()
    for l, (li, g) in enumerate(lanes):
        for v in range(4):
            out[l, v] += P[ROW(g, v), li]
    return out


for kb in range(4):
    C0 = 4 * kb
    colblk = np.zeros((16, 4), complex)       # LDS: D[r][C0 + m]
    rowblk = np.zeros((4, 16), complex)       # LDS: R[C0 + m][c]
    for l, (li, g) in enumerate(lanes):
        if C0 <= li < C0 + 4:
            for v in range(4):
                colblk[ROW(g, v), li - C0] = Dacc[l, v]
        rowblk[g, li] = Racc[l, kb]           # row C0 + g = ROW(g, kb)
    # every lane: 4 x 4 Cholesky + inverse of the diagonal block (redundantly)
    d4 = colblk[C0:C0 + 4, :]
    L4 = np.zeros((4, 4), complex)
    inv = np.zeros(4)
    for j in range(4):
        t = d4[j, j].real - sum(abs(L4[j, m]) ** 2 for m in range(j))
        assert t > 0
        inv[j] = 1.0 / np.sqrt(t)
        L4[j, j] = t * inv[j]
        for i in range(j + 1, 4):
            L4[i, j] = (d4[i, j] - sum(L4[i, m] * np.conj(L4[j, m]) for m in range(j))) * inv[j]
    X4 = np.zeros((4, 4), complex)
    for j in range(4):
        X4[j, j] = inv[j]
        for i in range(j + 1, 4):
            X4[i, j] = -inv[i] * sum(L4[i, m] * X4[m, j] for m in range(j, i))
    assert np.allclose(X4 @ L4, np.eye(4))
    Pop = np.zeros(64, complex)
    Wop = np.zeros(64, complex)
    for l, (li, g) in enumerate(lanes):
        p = sum(colblk[li, m] * np.conj(X4[g, m]) for m in range(g + 1))         # P[li][g]
        if li >= C0:
            Lout[li, C0 + g] = p if li >= C0 + g else 0.0
        Pop[l] = p if li >= C0 + 4 else 0.0                                       # rows below the block only
        w = sum(X4[g, m] * rowblk[m, li] for m in range(g + 1))                   # W[C0 + g][li]
        Wout[C0 + g, li] = w if li <= C0 + g else 0.0
        Wop[l] = w
    Dacc = mfma(-Pop, np.conj(Pop), Dacc)
    Racc = mfma(-Pop, Wop, Racc)

Lref = np.linalg.cholesky(D0)
print("L err", np.abs(Lout - Lref).max(), " W err", np.abs(Wout - np.linalg.inv(Lref)).max())
assert np.abs(Lout - Lref).max() < 1e-12 and np.abs(Wout - np.linalg.inv(Lref)).max() < 1e-12
print("ok")
