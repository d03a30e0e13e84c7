"""Lock-step model of the round-4 LU panel kernel (faer-rs_amd/csrc/getrf.hip, getrf_wpanel_kernel).

Test infrastructure, numpy only.  It checks the ALGEBRA of the kernel's schedule -- logical row labels instead of
interchanges, one header exchange per column, rank-1 updates of the columns right of J + 1 lagging one column behind and
the published candidate rows corrected by the consumer -- against the textbook unblocked elimination the reference runs
(faer/src/linalg/lu/partial_pivoting/factor.rs:19-67): identical pivots and bitwise identical factors when both use the
same floating-point operations (here: an unfused multiply-subtract on both sides).

Per participant ("workgroup" g) the model keeps rows x[i][:] with labels lab[i]; per column J every participant
  publishes  header[J] = {label, a = x_c[J], s = x_c[J+1] (updated through step J-2), l = l_{J-1}[c]} and the candidate row
  sweeps     all headers, picks the winner (largest |a|, smaller label on ties; a zero / NaN-only column keeps row J),
  corrects   u_J[J+1] = s - l * u_{J-1}[J+1], scales column J, updates column J + 1, publishes header J + 1,
  then       fetches the winner's row record, corrects it the same way and applies the rank-1 update to columns >= J + 2.
"""
import numpy as np


def lu_unblocked(a):
    """Reference: unblocked partial-pivot elimination with physical interchanges (factor.rs:19-67)."""
    a = a.copy()
    m, w = a.shape
    piv = []
    for j in range(min(m, w)):
        col = np.abs(a[j:, j])
        best, p = 0.0, j
        for i, v in enumerate(col):
            if v > best:
                best, p = v, j + i
        piv.append(p)
        if p != j:
            a[[j, p], :] = a[[p, j], :]
        with np.errstate(all="ignore"):
            inv = a.dtype.type(1) / a[j, j]
            l = a[j + 1:, j] * inv
            a[j + 1:, j] = l
            for c in range(j + 1, w):
                a[j + 1:, c] = a[j + 1:, c] - l * a[j, c]
    return a, piv


def lu_wpanel_model(a, rows_per_part):
    a = np.array(a, dtype=a.dtype)
    m, w = a.shape
    steps = min(m, w)
    G = (m + rows_per_part - 1) // rows_per_part
    parts = []
    for g in range(G):
        r0, r1 = g * rows_per_part, min(m, (g + 1) * rows_per_part)
        parts.append({"x": a[r0:r1].copy(), "lab": list(range(r0, r1)), "l": np.zeros(r1 - r0, a.dtype),
                      "uprev": np.zeros(w, a.dtype)})
    hdr, rowrec = {}, {}
    piv = []
    zero = a.dtype.type(0)

    def candidate(P, Jn, lprev):
        best, bl, bi = -1.0, None, None
        for i, lb in enumerate(P["lab"]):
            if lb < Jn:
                continue
            v = abs(P["x"][i, Jn])
            cv = v if v > 0 else (0.0 if lb == Jn else -1.0)
            if cv > best or (cv == best and cv >= 0 and lb < bl):
                best, bl, bi = cv, lb, i
        return best, bl, bi

    def publish(g, P, Jn, lprev):
        cv, lb, i = candidate(P, Jn, lprev)
        if cv < 0:
            hdr[(Jn, g)] = None
            return
        s = P["x"][i, Jn + 1] if Jn + 1 < w else zero
        hdr[(Jn, g)] = (lb, P["x"][i, Jn], s, lprev[i])
        rowrec[(Jn, g)] = P["x"][i].copy()

    for g, P in enumerate(parts):
        publish(g, P, 0, np.zeros(len(P["lab"]), a.dtype))
    with np.errstate(all="ignore"):
        for J in range(steps):
            # every participant sweeps the same headers
            best, win = (-1.0, None), None
            for g in range(G):
                h = hdr[(J, g)]
                if h is None:
                    continue
                v = abs(h[1])
                cv = v if v > 0 else 0.0
                if cv > best[0] or (cv == best[0] and h[0] < best[1]):
                    best, win = (cv, h[0]), g
            p, av, sv, lp = hdr[(J, win)]
            piv.append(p)
            for g, P in enumerate(parts):
                uJ1 = sv - lp * P["uprev"][J + 1] if J + 1 < w else zero
                P["lab"] = [p if lb == J else (J if lb == p else lb) for lb in P["lab"]]
                inv = a.dtype.type(1) / av
                act = np.array([lb > J for lb in P["lab"]], dtype=bool)
                P["act"] = act
                P["l"] = np.where(act, P["x"][:, J] * inv, zero)
                P["x"][act, J] = P["l"][act]
                if J + 1 < w:
                    P["x"][act, J + 1] = P["x"][act, J + 1] - P["l"][act] * uJ1
            if J + 1 < steps:
                for g, P in enumerate(parts):
                    publish(g, P, J + 1, P["l"])
            rec = rowrec[(J, win)]
            for g, P in enumerate(parts):
                ucur = rec - lp * P["uprev"]
                act = P["act"]
                for c in range(J + 2, w):
                    P["x"][act, c] = P["x"][act, c] - P["l"][act] * ucur[c]
                P["uprev"] = ucur
    out = np.zeros_like(a)
    for P in parts:
        for i, lb in enumerate(P["lab"]):
            out[lb] = P["x"][i]
    return out, piv


def lu_small_leaf_model(a, rows_per_wave=8):
    """Lock-step model of the single-workgroup leaf (faer-rs_amd/csrc/lu_small_leaf.h, getrf_small_leaf_kernel): every row keeps
    its LABEL instead of moving, the registers of a row are ROTATED left by 8 positions after every group of 8 steps (the column
    being eliminated sits at position J mod 8), every "wavefront" (rows_per_wave rows) publishes its candidate (|a|, label) and the
    candidate's whole row in position order, every row then reads the winner's row by position.  Steps past min(w, m) run on the
    zero padding of the 64 positions and store nothing."""
    a = np.array(a, dtype=a.dtype)
    m, w = a.shape
    W = 64
    assert w <= W and w <= m
    x = np.zeros((m, W), a.dtype)
    x[:, :w] = a
    lab = list(range(m))
    steps = min(w, m)
    piv = []
    rot = 0
    zero = a.dtype.type(0)
    waves = [range(r0, min(m, r0 + rows_per_wave)) for r0 in range(0, m, rows_per_wave)]
    with np.errstate(all="ignore"):
        grp = 0
        while grp * 8 < steps:
            lim = W - grp * 8
            for JJ in range(8):
                J = grp * 8 + JJ
                # per wavefront: candidate and its row (by position)
                cands = []
                for rows in waves:
                    best, bl, bi = -1.0, None, None
                    for i in rows:
                        if lab[i] < J:
                            continue
                        v = abs(float(x[i, JJ]))
                        cv = v if v > 0 else (0.0 if lab[i] == J else -1.0)
                        if cv > best or (cv == best and cv >= 0 and lab[i] < bl):
                            best, bl, bi = cv, lab[i], i
                    cands.append((best, bl, None if bi is None else x[bi].copy()))
                # the workgroup's winner: largest |a|, smallest label on ties
                gbest, gl, grow = -1.0, None, None
                for cv, lb, row in cands:
                    if cv < 0:
                        continue
                    if cv > gbest or (cv == gbest and lb < gl):
                        gbest, gl, grow = cv, lb, row
                if gbest < 0:
                    continue  # nobody has a candidate (padding, J >= m)
                p = gl
                lab = [p if lb == J else (J if lb == p else lb) for lb in lab]
                if J < steps:
                    piv.append(p)
                inv = a.dtype.type(1) / grow[JJ]
                for i in range(m):
                    if lab[i] > J:
                        l = x[i, JJ] * inv
                        x[i, JJ] = l
                        for pos in range(JJ + 1, lim):
                            x[i, pos] = x[i, pos] - l * grow[pos]
            x = np.concatenate([x[:, 8:], x[:, :8]], axis=1)
            rot += 8
            grp += 1
    out = np.zeros((m, w), a.dtype)
    for i in range(m):
        for pos in range(W):
            gc = (pos + rot) % W
            if gc < w:
                out[lab[i], gc] = x[i, pos]
    return out, piv
