"""Second, independent restatement of the four semantics in plain Python, used only to
cross-check the C oracle on small cases.  Deliberately written differently from oracle/aligner_oracle.c:
dict-of-cells storage, explicit visiting-order penalty carry, generic max()."""
EPS = 2.0 ** -52
T, L, D, B = 0, 1, 2, 3


def _pick(top, left, diag, with_beginning):
    m = max(max(top, left), diag)
    if with_beginning and m == 0:
        return m, B
    if abs(m - top) < EPS:
        return m, T
    if abs(m - left) < EPS:
        return m, L
    return m, D


def _walk(Dm, q, t, cy, cx, qa, ta, blank=98):
    while True:
        d = Dm[(cy, cx)]
        if d == B:
            break
        if d == T:
            qa.append(blank); ta.append(t[cy - 1]); cy -= 1
        elif d == L:
            qa.append(q[cx - 1]); ta.append(blank); cx -= 1
        else:
            qa.append(q[cx - 1]); ta.append(t[cy - 1]); cx -= 1; cy -= 1
    return cy, cx


def core(q, t, dele, ext, S, local):
    N, M = len(q), len(t)
    H = {(y, x): 0.0 for y in range(M + 1) for x in range(N + 1)}
    Dm = {k: B for k in H}
    if not local:
        for x in range(1, N + 1):
            H[(0, x)] = -x * dele; Dm[(0, x)] = L
        for y in range(1, M + 1):
            H[(y, 0)] = -y * dele; Dm[(y, 0)] = T
        H[(0, N)] = -(N + 1) * dele
        H[(M, 0)] = -(M + 1) * dele
    p = dele
    for x in range(1, N + 1):
        for y in range(1, M + 1):
            v, d = _pick(H[(y - 1, x)] - p, H[(y, x - 1)] - p, H[(y - 1, x - 1)] + S[t[y - 1]][q[x - 1]], local)
            p = ext if d != B else dele
            H[(y, x)] = v; Dm[(y, x)] = d
    if not local:
        qa, ta = [q[-1]], [t[-1]]
        cy, cx = _walk(Dm, q, t, M, N, qa, ta)
        return dict(H=H, D=Dm, qa=qa[::-1], ta=ta[::-1], f=0.0, score=H[(M, N)], coords=((1, N), (1, M)))
    best = (0, 0)
    for y in range(M + 1):
        for x in range(N + 1):
            if H[(y, x)] > H[best]:
                best = (y, x)
    my, mx = best
    if my == 0 or mx == 0:
        return dict(H=H, D=Dm, panic=True)
    qa, ta = [q[mx - 1]], [t[my - 1]]
    cy, cx = _walk(Dm, q, t, my, mx, qa, ta)
    return dict(H=H, D=Dm, qa=qa[::-1], ta=ta[::-1], f=H[best], score=H[best],
                coords=((cx + 1, mx + 1), (cy + 1, my + 1)))


def legacy(q, t, dele, S, local):
    N, M = len(q), len(t)
    H = {(y, x): 0 for y in range(M + 1) for x in range(N + 1)}
    Dm = {k: B for k in H}
    if not local:
        for x in range(1, N + 1):
            H[(0, x)] = -x * dele; Dm[(0, x)] = L
        for y in range(1, M + 1):
            H[(y, 0)] = -y * dele; Dm[(y, 0)] = T
        H[(M, 0)] = -(M + 1) * dele
        H[(0, N)] = -(N + 1) * dele
    mf, mxx, myy = 0, 0, 0
    for x in range(1, N + 1):
        for y in range(1, M + 1):
            top, left = H[(y - 1, x)] - dele, H[(y, x - 1)] - dele
            diag = H[(y - 1, x - 1)] + int(S[t[y - 1]][q[x - 1]])
            m = max(top, left, diag, 0) if local else max(top, left, diag)
            H[(y, x)] = m
            if local and m == 0:
                Dm[(y, x)] = B
            elif m == top:
                Dm[(y, x)] = T
            elif m == left:
                Dm[(y, x)] = L
            else:
                Dm[(y, x)] = D
            if local and m >= mf:
                mf, mxx, myy = m, x - 1, y - 1
    if not local:
        qa, ta = [q[-1]], [t[-1]]
        _walk(Dm, q, t, M - 1, N - 1, qa, ta)
        return dict(H=H, D=Dm, qa=qa[::-1], ta=ta[::-1], score=H[(M, N)])
    qa, ta = [q[mxx]], [t[myy]]
    _walk(Dm, q, t, myy, mxx, qa, ta)
    return dict(H=H, D=Dm, qa=qa[::-1], ta=ta[::-1], score=mf, end=(myy + 1, mxx + 1))


def pwm(seq, dele, ext, M):
    """PWMAligner::perform_alignment (pwm/mod.rs:29-126), independent restatement: M is a 4 x W nested list."""
    Q, W = len(seq), len(M[0])
    H = {(y, x): 0.0 for y in range(Q + 1) for x in range(W + 1)}
    Dm = {k: B for k in H}
    p = dele
    for x in range(1, W + 1):
        for y in range(1, Q + 1):
            v, d = _pick(H[(y - 1, x)] - p, H[(y, x - 1)] - p, H[(y - 1, x - 1)] + M[seq[y - 1]][x - 1], True)
            p = ext if d != B else dele
            H[(y, x)] = v; Dm[(y, x)] = d
    best = (0, 0)
    for y in range(Q + 1):
        for x in range(W + 1):
            if H[(y, x)] > H[best]:
                best = (y, x)
    cy, cx = best
    numbered, qal = [], []
    while Dm[(cy, cx)] != B:
        d = Dm[(cy, cx)]
        if d == T:
            numbered.append(0); qal.append(seq[cy - 1]); cy -= 1
        elif d == L:
            numbered.append(cx); qal.append(98); cx -= 1
        else:
            numbered.append(cx); qal.append(seq[cy - 1]); cx -= 1; cy -= 1
    return dict(H=H, D=Dm, numbered=numbered[::-1], qal=qal[::-1], f=max(H.values()),
                coords=((cx + 1, best[1] + 1), (cy + 1, best[0] + 1)))


# ---- statistics/mod.rs:36-238 walked by hand with scalar math (lists and math.*, no numpy), keeping the Rust scoping: the
# `let (k, lambda) = ...` at :69 lives only inside one loop iteration, so the call's arguments are always the OUTER k, lambda.
import math


def _fexp(x):
    try:
        return math.exp(x)
    except OverflowError:
        return math.inf


def _flog(x, base10=False):
    if x != x:
        return math.nan
    if x < 0:
        return math.nan
    if x == 0:
        return -math.inf
    if x == math.inf:
        return math.inf
    return math.log10(x) if base10 else math.log(x)


def _fdiv(a, b):
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0:
            return math.nan
        return math.copysign(math.inf, a) * math.copysign(1.0, b)


def _evd_nn(ql, ts, k, h):
    out = []
    for t in ts:
        l = _fdiv(_flog(k * ql * t), h)
        out.append((ql - l) * (t - l))
    return out


def _evd_k_lambda(ql, ts, sc, old_k, old_lambda, h, maxiter=10000, thr=1e-4):
    k, lam = old_k, old_lambda
    n = float(len(ts))
    nn = _evd_nn(ql, ts, k, h)
    es = [_fexp(-lam * s) for s in sc]
    ssum = math.fsum(a * b for a, b in zip(nn, es)) if nn else 0.0
    wsum = math.fsum(a * s * b for a, s, b in zip(nn, sc, es)) if nn else 0.0
    for _ in range(maxiter + 1):
        f = _fdiv(1.0, lam) - _fdiv(sum(sc), n) + _fdiv(wsum, ssum)
        fd = -_fdiv(1.0, lam * lam) - _fdiv(sum(a * s * s * b for a, s, b in zip(nn, sc, es)), ssum) + _fdiv(wsum, ssum) ** 2
        if not math.isfinite(f) or not math.isfinite(fd):
            return k, lam
        new_lam = lam - _fdiv(f, fd)
        es = [_fexp(-lam * s) for s in sc]
        ssum = sum(a * b for a, b in zip(nn, es))
        wsum = sum(a * s * b for a, s, b in zip(nn, sc, es))
        new_k = _fdiv(n, ssum)
        if not math.isfinite(new_k) or new_k <= 0:
            return k, lam
        k, lam = new_k, new_lam
        if abs(f) < thr:
            return k, lam
        nn = _evd_nn(ql, ts, k, h)
    return k, lam


def _evd_h(ql, ts, sc, k, lam, old_h, maxiter=10000, thr=1e-4):
    h = old_h
    for _ in range(maxiter + 1):
        g = gd = 0.0
        for t, s in zip(ts, sc):
            l = _fdiv(_flog(k * ql * t), h)
            nn = (ql - l) * (t - l)
            a = 2.0 * l - ql - t
            b = _fdiv(1.0, nn) - k * _fexp(-lam * s)
            c = _fdiv(-l, h)
            g += a * b * c
            gd += 2.0 * b * c * c - _fdiv(a * c, nn) ** 2 - _fdiv(2.0 * a * b * c, h)
        if abs(g) < thr:
            return h
        if gd > 0:
            h = h * 2.0 if g > 0 else h / 2.0
        elif g <= 0:
            h /= 2.0
        else:
            h -= _fdiv(g, gd)
    return h


def evd_params(ql, ts, sc, maxiter=10000):
    """(k, lambda, h, outer iterations run) as statistics/mod.rs:36-123 computes them."""
    ts = [float(t) for t in ts]
    sc = [float(s) for s in sc]
    n = float(len(ts))
    mean = sum(sc) / n
    sd = sum((s - mean) ** 2 for s in sc) / n
    lam0 = 1.0 / sd
    h = 1.0
    nn = [ql * t for t in ts]
    k0 = n / sum(a * _fexp(-lam0 * s) for a, s in zip(nn, sc))
    ll = n * _flog(lam0 * k0) + sum(_flog(a) - lam0 * s - k0 * a * _fexp(-lam0 * s) for a, s in zip(nn, sc))
    act_t, act_s = list(ts), list(sc)
    for it in range(maxiter + 1):
        k, lam = _evd_k_lambda(ql, act_t, act_s, k0, lam0, h, maxiter)          # outer k0 / lam0: the inner binding is scoped to the body
        h = _evd_h(ql, act_t, act_s, k, lam, h, maxiter)
        nn = _evd_nn(ql, ts, k, h)
        ll_new = n * _flog(lam * k, True) + sum(_flog(a, True) - lam * s - k * a * _fexp(-lam * s) for a, s in zip(nn, sc))
        if _fdiv(abs(ll_new - ll), ll) < 1e-6:
            return k, lam, h, it + 1
        ll = ll_new
        act_t, act_s = [], []
        for t, s, a in zip(ts, sc, nn):
            if n * (1.0 - _fexp(-k * a * _fexp(-lam * s))) >= 1.0:
                act_t.append(t); act_s.append(s)
    return k0, lam0, h, maxiter + 1
