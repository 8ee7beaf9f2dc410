"""Experiment: can an fp32 LU inverse written out by hand reproduce ``torch.inverse`` (LAPACK getrf + getri inside
the host's BLAS library) bit for bit on 4x4 camera matrices?  Candidates: unblocked right-looking LU with partial pivoting (getf2) and the
recursive splitting of LAPACK >= 3.6 (getrf2), each followed by getri's sequence (invert U, then solve X L = U^-1 column by column from
the right), with and without fused multiply-adds.  Prints, per candidate, the fraction of matrices whose 16 entries are all equal to
torch's.  CPU only; run on every host of interest (the answer depends on the BLAS build and the CPU it dispatches for)."""
import sys

import numpy as np
import torch

f32 = np.float32


def fma(a, b, c, fused):
    if fused:
        return f32(np.float64(a) * np.float64(b) + np.float64(c))
    return f32(f32(a * b) + c)


def lu_unblocked(A, fused):
    A = A.copy()
    n = 4
    piv = list(range(n))
    for j in range(n):
        p = j + int(np.argmax(np.abs(A[j:, j])))
        if p != j:
            A[[j, p]] = A[[p, j]]
            piv[j], piv[p] = piv[p], piv[j]
        r = f32(1.0) / A[j, j]
        for i in range(j + 1, n):
            A[i, j] = f32(A[i, j] * r)
        for i in range(j + 1, n):
            for k in range(j + 1, n):
                A[i, k] = fma(-A[i, j], A[j, k], A[i, k], fused)
    return A, piv


def lu_divide(A, fused):
    """As lu_unblocked but dividing by the pivot instead of multiplying by its reciprocal (getf2 does so for tiny pivots only)."""
    A = A.copy()
    n = 4
    piv = list(range(n))
    for j in range(n):
        p = j + int(np.argmax(np.abs(A[j:, j])))
        if p != j:
            A[[j, p]] = A[[p, j]]
            piv[j], piv[p] = piv[p], piv[j]
        for i in range(j + 1, n):
            A[i, j] = f32(A[i, j] / A[j, j])
        for i in range(j + 1, n):
            for k in range(j + 1, n):
                A[i, k] = fma(-A[i, j], A[j, k], A[i, k], fused)
    return A, piv


def getri(LU, piv, fused):
    n = 4
    U = np.triu(LU).astype(f32)
    # trtri (unblocked, upper, non-unit): column by column
    Ui = U.copy()
    for j in range(n):
        Ui[j, j] = f32(1.0) / Ui[j, j]
        ajj = -Ui[j, j]
        # x = Ui[:j, :j] @ Ui[:j, j]  (trmv with the upper triangle, rows ascending, on the ORIGINAL column)
        x = Ui[:j, j].copy()
        for i in range(j):
            acc = f32(0.0)
            for k in range(i, j):
                acc = fma(Ui[i, k], x[k], acc, fused)
            Ui[i, j] = f32(acc * ajj)
    # solve inv(A) L = inv(U): columns from the right
    X = Ui.copy()
    L = np.tril(LU, -1).astype(f32)
    for j in range(n - 2, -1, -1):
        for k in range(j + 1, n):
            for i in range(n):
                X[i, j] = fma(-X[i, k], L[k, j], X[i, j], fused)
    # undo the row interchanges as column swaps
    P = np.zeros((n, n), f32)
    for r, c in enumerate(piv):
        P[r, c] = 1
    return (X @ P).astype(f32)


def main():
    g = torch.Generator().manual_seed(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    mats = []
    for _ in range(n):
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
        m = torch.eye(4)
        m[:3, :3] = q
        m[:3, 3] = torch.randn(3, generator=g)
        mats.append(m)
    M = torch.stack(mats)
    want = torch.inverse(M).numpy()
    print(torch.__config__.show().split("\n")[0:8])
    for name, lu in (("getf2 (reciprocal pivot)", lu_unblocked), ("getf2 (divide)", lu_divide)):
        for fused in (False, True):
            same = 0
            worst = 0.0
            for A, w in zip(M.numpy(), want):
                LU, piv = lu(A.astype(f32), fused)
                X = getri(LU, piv, fused)
                same += int(np.array_equal(X, w))
                worst = max(worst, float(np.abs(X - w).max()))
            print(f"{name:28s} fused={fused!s:5s}: {same / n * 100:6.2f} % of {n} matrices bit-identical to torch.inverse, worst |diff| {worst:.2e}")


if __name__ == "__main__":
    main()
