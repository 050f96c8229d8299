"""Test infrastructure: the numpy statement of tridiag_dc (ground-fusion2_amd/csrc/gfbe_marg.hip, round 6) — Cuppen's divide & conquer for a
symmetric tridiagonal matrix with Gu / Eisenstat vectors and LAPACK's deflation rules, bottom-up merges of leaves of size <= 2, the
secular equation solved per root by the two-pole rational iteration from the nearer pole with a bisection safeguard. Written BEFORE the
device code, phase for phase the same structure (tear at every even index, rank sort, sequential deflation scan with recorded
rotations, root = origin pole + offset, zhat from the computed roots), so that the algorithm's numerics are pinned on the CPU against
numpy.linalg.eigh (tests/test_oracle_numpy.py::test_divide_and_conquer_tridiagonal) and the device only has to reproduce it."""
import numpy as np

EPS = 2.220446049250313e-16


def leaf_eig(d, e):
    n = len(d)
    if n == 1:
        return d.copy(), np.ones((1, 1))
    a, b, c = d[0], e[0], d[1]
    # symmetric 2x2: rotation
    if b == 0.0:
        lam = np.array([a, c]); V = np.eye(2)
    else:
        th = (c - a) / (2.0 * b)
        t = (1.0 if th >= 0 else -1.0) / (abs(th) + np.sqrt(th * th + 1.0))
        cs = 1.0 / np.sqrt(t * t + 1.0); sn = t * cs
        lam = np.array([a - t * b, c + t * b])
        V = np.array([[cs, sn], [-sn, cs]])
    if lam[0] > lam[1]:
        lam = lam[::-1].copy(); V = V[:, ::-1].copy()
    return lam, V


def secular_root(i, k, dl, z2, rho):
    """root i of 1 + rho sum z2_j / (dl_j - lam) in (dl_i, dl_{i+1}) (i < k-1) or (dl_{k-1}, dl_{k-1} + rho sum z2).
    returns (origin index o, mu) with lam = dl[o] + mu; differences dl_j - lam = (dl_j - dl_o) - mu."""
    if i < k - 1:
        gap = dl[i + 1] - dl[i]
        # sign of f at the midpoint decides the origin
        mid = 0.5 * gap
        dj = dl - dl[i]
        f = 1.0 + rho * np.sum(z2 / (dj - mid))
        if f >= 0.0:      # root in the left half: origin i, mu in (0, gap/2]
            o = i; lo, hi = 0.0, mid
        else:
            o = i + 1; lo, hi = -mid, 0.0
        delta = dl - dl[o]
        mu = 0.5 * (lo + hi)
        for it in range(60):
            den = delta - mu
            t = z2 / den
            psi = rho * np.sum(t[: i + 1]); phi = rho * np.sum(t[i + 1:])
            dpsi = rho * np.sum(t[: i + 1] / den[: i + 1]); dphi = rho * np.sum(t[i + 1:] / den[i + 1:])
            g = 1.0 + psi + phi
            err = 8.0 * EPS * (1.0 + abs(psi) + abs(phi)) + abs(mu) * (dpsi + dphi) * EPS
            if abs(g) <= err:
                break
            if g > 0.0: hi = mu
            else: lo = mu
            # two-pole model through (psi, dpsi) with pole delta_i and (phi, dphi) with pole delta_{i+1}
            Di, Dj = delta[i] - mu, delta[i + 1] - mu
            S = dpsi * Di * Di; s = psi - dpsi * Di
            R = dphi * Dj * Dj; r = phi - dphi * Dj
            cst = 1.0 + s + r
            # cst + S/(Di - x) + R/(Dj - x) = 0 with x = step from mu:   cst (Di-x)(Dj-x) + S (Dj-x) + R (Di-x) = 0
            a = cst; b = -(cst * (Di + Dj) + S + R); c = cst * Di * Dj + S * Dj + R * Di
            if a == 0.0:
                x = c / -b if b != 0 else 0.0
            else:
                disc = b * b - 4 * a * c
                if disc < 0: disc = 0.0
                sq = np.sqrt(disc)
                # the root between the poles: Di < 0 < Dj shifted... pick the root x with Di < x < Dj (Di negative side)
                q = -0.5 * (b + (sq if b >= 0 else -sq))
                x1 = q / a; x2 = c / q if q != 0 else x1
                x = x1 if (Di < x1 < Dj) else x2
            new = mu + x
            if not (lo < new < hi) or not np.isfinite(new):
                new = 0.5 * (lo + hi)
            if new == mu:
                break
            mu = new
        return o, mu
    else:
        o = k - 1
        delta = dl - dl[o]
        lo, hi = 0.0, rho * np.sum(z2)
        # f(hi) >= 0
        mu = 0.5 * (lo + hi)
        for it in range(80):
            den = delta - mu
            t = z2 / den
            psi = rho * np.sum(t); dpsi = rho * np.sum(t / den)
            g = 1.0 + psi
            err = 8.0 * EPS * (1.0 + abs(psi)) + abs(mu) * dpsi * EPS
            if abs(g) <= err:
                break
            if g > 0.0: hi = mu
            else: lo = mu
            # one-pole model on the last pole + constant for the rest: psi ~ s + S / (D - x)
            D = delta[o] - mu      # negative
            S = dpsi * D * D; s = psi - dpsi * D
            cst = 1.0 + s
            x = D + S / cst if cst != 0 else 0.0      # cst + S/(D - x) = 0 -> x = D + S/cst
            new = mu + x
            if not (lo < new < hi) or not np.isfinite(new):
                new = 0.5 * (lo + hi)
            if new == mu:
                break
            mu = new
        return o, mu


def merge(lam1, Q1, lam2, Q2, beta):
    n1, n2 = len(lam1), len(lam2)
    m = n1 + n2
    rho = abs(beta)
    sgn = 1.0 if beta >= 0 else -1.0
    D = np.concatenate([lam1, lam2])
    z = np.concatenate([Q1[-1, :], sgn * Q2[0, :]])
    Q = np.zeros((m, m)); Q[:n1, :n1] = Q1; Q[n1:, n1:] = Q2
    # normalise z: |z|^2 = 2
    zn = np.linalg.norm(z)
    z = z / zn; rho = rho * zn * zn
    order = np.argsort(D, kind="stable")
    D = D[order]; z = z[order]; Q = Q[:, order]
    tol = 8.0 * EPS * max(np.abs(D).max(), np.abs(z).max() * 1.0)      # (dlaed2: 8 eps max(|d|max, |z|max))
    defl = np.zeros(m, bool)
    if rho * np.abs(z).max() <= tol:
        return D, Q
    for j in range(m):
        if rho * abs(z[j]) <= tol:
            defl[j] = True
    # close poles: sequential scan over the non-deflated ones
    prev = -1
    for j in range(m):
        if defl[j]:
            continue
        if prev >= 0:
            s_, c_ = z[prev], z[j]
            tau = np.hypot(c_, s_)
            t = D[j] - D[prev]
            c_ /= tau; s_ = -s_ / tau
            if abs(t * c_ * s_) <= tol:
                z[j] = tau; z[prev] = 0.0
                qp, qj = Q[:, prev].copy(), Q[:, j].copy()
                Q[:, prev] = c_ * qp + s_ * qj
                Q[:, j] = -s_ * qp + c_ * qj
                dp, dj = D[prev], D[j]
                D[prev] = dp * c_ * c_ + dj * s_ * s_
                D[j] = dp * s_ * s_ + dj * c_ * c_
                defl[prev] = True
                # (dlaed2 re-sorts the deflated value; here D[prev] stays in place: order of deflated entries does not matter)
        prev = j
    idx = np.nonzero(~defl)[0]
    k = len(idx)
    lam = D.copy(); Qn = Q.copy()
    if k == 0:
        return lam, Qn
    dl = D[idx]; zl = z[idx]; z2 = zl * zl
    if k == 1:
        lam[idx[0]] = dl[0] + rho * z2[0]
        return lam, Qn
    orig = np.zeros(k, int); mu = np.zeros(k)
    for i in range(k):
        orig[i], mu[i] = secular_root(i, k, dl, z2, rho)
    # lam_i - dl_j = (dl[orig_i] - dl_j) + mu_i
    diff = (dl[orig][:, None] - dl[None, :]) + mu[:, None]      # [i, j] = lam_i - dl_j
    # Gu-Eisenstat: zhat_j^2 = prod_i (lam_i - dl_j) / prod_{i != j} (dl_i - dl_j) / rho  -> positive
    zhat = np.zeros(k)
    for j in range(k):
        p = diff[j, j] if False else 1.0
        # interleave ratios to avoid over/underflow
        num = diff[:, j]
        den = dl - dl[j]
        prod = num[j]      # lam_j - dl_j  (paired with nothing)
        for i in range(k):
            if i != j:
                prod *= num[i] / den[i]
        zhat[j] = np.sqrt(abs(prod) / rho) * (1.0 if zl[j] >= 0 else -1.0)
    V = zhat[None, :] / (-diff)      # [i, j] = zhat_j / (dl_j - lam_i)
    V /= np.linalg.norm(V, axis=1)[:, None]
    Qk = Q[:, idx] @ V.T
    Qn[:, idx] = Qk
    lam[idx] = dl[orig] + mu
    return lam, Qn


def dc_eig(d, e):
    n = len(d)
    d = d.astype(float).copy(); e = e.astype(float).copy()
    # bottom-up: leaves of size 2 (last may be 1); tear at the boundaries
    bounds = list(range(0, n, 2)) + [n]
    blocks = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        blocks.append([a, b])
    # tearing: subtract |beta| from the diagonal entries next to each boundary
    for (a, b) in blocks[1:]:
        beta = e[a - 1]
        d[a - 1] -= abs(beta); d[a] -= abs(beta)
    eigs = []
    for (a, b) in blocks:
        eigs.append(leaf_eig(d[a:b], e[a:b - 1]))
    while len(blocks) > 1:
        nb, ne = [], []
        for q in range(0, len(blocks) - 1, 2):
            (a, b), (b2, c) = blocks[q], blocks[q + 1]
            lam, Q = merge(eigs[q][0], eigs[q][1], eigs[q + 1][0], eigs[q + 1][1], e[b - 1])
            nb.append([a, c]); ne.append((lam, Q))
        if len(blocks) % 2:
            nb.append(blocks[-1]); ne.append(eigs[-1])
        blocks, eigs = nb, ne
    return eigs[0]


def check(d, e, name):
    n = len(d)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    lam, V = dc_eig(d, e)
    nrm = np.abs(T).max()
    res = np.abs(T @ V - V * lam[None, :]).max() / nrm
    orth = np.abs(V.T @ V - np.eye(n)).max()
    w = np.linalg.eigvalsh(T)
    ev = np.abs(np.sort(lam) - w).max() / nrm
    print("%-28s n=%3d residual %.2e orth %.2e eigval %.2e" % (name, n, res, orth, ev))
    return max(res, orth, ev)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    worst = 0
    for n in (1, 2, 3, 5, 8, 17, 43, 86, 87, 120, 246):
        worst = max(worst, check(rng.normal(size=n), rng.normal(size=n - 1), "random"))
    for n in (86, 120):
        # graded: from a 1e14-conditioned SPD matrix
        U = np.linalg.qr(rng.normal(size=(n, n)))[0]
        s = 10.0 ** rng.uniform(-8, 6, n); s[:4] = 1e-10 * rng.uniform(0.1, 1, 4)
        A = (U * s) @ U.T
        import scipy.linalg as sl
        Hh = sl.hessenberg(A)
        d = np.diag(Hh).copy(); e = np.diag(Hh, -1).copy()
        worst = max(worst, check(d, e, "graded 1e14 (hessenberg)"))
    n = 86
    worst = max(worst, check(np.ones(n), np.zeros(n - 1), "identity"))
    worst = max(worst, check(2 * np.ones(n), -np.ones(n - 1), "laplacian"))
    d = np.ones(n); e = 1e-9 * np.ones(n - 1)
    worst = max(worst, check(d, e, "nearly identity"))
    e = rng.normal(size=n - 1); e[::7] = 0.0
    worst = max(worst, check(rng.normal(size=n), e, "split"))
    # Wilkinson W21+
    m = 10; d = np.abs(np.arange(-m, m + 1)).astype(float); e = np.ones(2 * m)
    worst = max(worst, check(d, e, "wilkinson"))
    d = np.concatenate([np.linspace(1, 2, 40), np.linspace(1, 2, 40) + 1e-13]); e = 1e-7 * rng.normal(size=79)
    worst = max(worst, check(d, e, "clustered"))
    print("worst", worst)
