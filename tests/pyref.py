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
