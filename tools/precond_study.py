"""would a workgroup-local block preconditioner make the distance PCG need fewer passes?  CPU study on the ORACLE's system.

    python tools/precond_study.py [N] [F] [model] [rows per block]

The oracle's assembled distance system (orc_debug_dist_system: diag, rhs, mat-vec) is extracted exactly with 27 colour probes (every column of a row
lies within +-1 voxel per axis, so neighbours have distinct (i, j, k) mod 3), the reference's damping `H.diagonal() += damping * H.diagonal()`
(PsOptimizer.cpp:103-105) is applied, and Eigen's CG loop (tolerance eps_f32) is run in double with: Jacobi (the reference); Neumann-series polynomials
of the block-local and of the whole off-diagonal part; the EXACT inverse of the block-diagonal part (contiguous blocks of band rows = what one workgroup
of the persistent solve holds in LDS) -- the best any workgroup-local preconditioner could do.  Test infrastructure: imports the oracle.
"""
import sys, os, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from psgradientsdf_amd import capi, synth
from oracle import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
model = sys.argv[3] if len(sys.argv) > 3 else "SH1"
bs = int(sys.argv[4]) if len(sys.argv) > 4 else 1344
mid = {"SH1": capi.SH1, "SH2": capi.SH2, "LED": capi.LED}[model]
sc = synth.make_scene(N=N, F=F, W=320, H=240, model=model)
st = capi.default_settings(mid)
orc = oracle.Oracle(sc, sc.K, st, threads=32)
orc.load_scene(sc); orc.init_albedo(); orc.normalize_weights()
band = orc.download_band().astype(np.int64); S = band.size
i, j, k = band % N, (band // N) % N, band // (N * N)
col = (i % 3) + 3 * (j % 3) + 9 * (k % 3)
lin2row = -np.ones(N ** 3, np.int64); lin2row[band] = np.arange(S)
rows, cols, vals = [], [], []
for c in range(27):
    _, rhs, y = orc.debug_dist_system((col == c).astype(np.float32))
    nz = np.nonzero(y)[0]
    di = ((c % 3 - i[nz] % 3) + 1) % 3 - 1; dj = (((c // 3) % 3 - j[nz] % 3) + 1) % 3 - 1; dk = ((c // 9 - k[nz] % 3) + 1) % 3 - 1
    tr = lin2row[(i[nz] + di) + N * (j[nz] + dj) + N * N * (k[nz] + dk)]
    assert (tr >= 0).all()
    rows.append(nz); cols.append(tr); vals.append(y[nz].astype(np.float64))
A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(S, S))
x = np.random.default_rng(0).standard_normal(S).astype(np.float32)
_, _, y = orc.debug_dist_system(x)
print(f"{model} {N}^3 x {F}: S = {S}, {A.nnz / S:.1f} non-zeros per row, extraction error {abs(A @ x.astype(np.float64) - y).max() / abs(y).max():.1e}")
A = (A + float(st.damping) * sp.diags(A.diagonal())).tocsr()
b = rhs.astype(np.float64); d = A.diagonal()
off = np.asarray(abs(A).sum(1)).ravel() - d
print(f"damping {st.damping}: sum|off-diagonal| / diagonal: median {np.median(off / d):.2f}, max {(off / d).max():.1f}")


def pcg(Minv, tol=float(np.finfo(np.float32).eps), maxit=200):
    x = np.zeros(S); r = b.copy(); thr = tol * tol * (b @ b); z = Minv(r); p = z.copy(); rz = r @ z; it = 0
    while it < maxit:
        t = A @ p; a = rz / (p @ t); x += a * p; r -= a * t; it += 1
        if r @ r < thr: break
        z = Minv(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return it


blk = np.arange(S) // bs
Ac = A.tocoo(); offm = Ac.row != Ac.col; loc = (blk[Ac.row] == blk[Ac.col]) & offm
Nloc = sp.csr_matrix((Ac.data[loc], (Ac.row[loc], Ac.col[loc])), shape=(S, S))
Nall = sp.csr_matrix((Ac.data[offm], (Ac.row[offm], Ac.col[offm])), shape=(S, S))
print(f"blocks of {bs} contiguous band rows hold {abs(Nloc).sum() / abs(Nall).sum():.0%} of the off-diagonal mass")
print(f"Jacobi (the reference)                      {pcg(lambda r: r / d):3d} passes")


def neumann(Nm, deg):
    def f(r):
        t = r / d; acc = t.copy()
        for _ in range(deg):
            t = -(Nm @ t) / d; acc += t
        return acc
    return f


for nm, Nm in (("block-local", Nloc), ("whole matrix", Nall)):
    for deg in (1, 2):
        print(f"Neumann degree {deg}, {nm:12s}            {pcg(neumann(Nm, deg)):3d} passes  ({deg} extra mat-vec per pass)")
lu = spl.splu((Nloc + sp.diags(d)).tocsc())
print(f"EXACT inverse of the block-diagonal part    {pcg(lu.solve):3d} passes")
