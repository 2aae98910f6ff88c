"""TEST INFRASTRUCTURE -- a numpy restatement of what the reference's bundled edlib returns for
edlibAlign(query, target, edlibNewAlignConfig(-1, EDLIB_MODE_HW, EDLIB_TASK_PATH, NULL, 0)) (the stage-5 call sites,
create_new_contigs.cpp:558-629, tools.cpp:515-534), for ANY query length: distance, first end location, its start location
and the alignment, move by move. Only tests/ may import it; pinned against the reference's own edlib through
tests/golden/edlib_path_vectors.json and edlib_long_path_vectors.json.gz (oracle/gen_goldens.py --edlib-long).

What is restated (file:line under /root/reference/src/edlib/src/edlib.cpp), on plain dynamic-programming columns instead of
bit vectors (every score edlib reads inside its band is the exact one, so whole exact columns give the same decisions):
  * HW search, first end location, the padding-row tie with the position "before the target"   :560-700, :664-690
  * start location = last best column of SHW(reversed query, reversed target prefix)             :226-258
  * obtainAlignment: traceback when (2*8 + 4) * blocks * columns + 8 * columns < 1 MB, otherwise Hirschberg    :1166-1219
  * obtainAlignmentHirschberg: target cut in halves, FIRST query row whose left + right scores give the optimum, then the two
    boundary rows, upper-left before lower-right                                                 :1236-1404
  * obtainAlignmentTraceback: up (insertion) before left (deletion) before the diagonal          :947-1140
Moves: 0 '=', 1 insertion (query base alone), 2 deletion, 3 mismatch."""
import numpy as np


def _codes(s):
    return np.frombuffer(s.encode() if isinstance(s, str) else bytes(s), dtype=np.uint8)


def _columns(q, t, top):
    """Yields column j (scores of rows 0..qn-1) of the edit-distance matrix of q (rows) against t (columns).
    top(j) = the score of the boundary row above column j (j = -1 .. tn-1); the boundary column is row + 1."""
    qn = len(q)
    rows = np.arange(0, qn + 1, dtype=np.int64)
    prev = np.arange(1, qn + 1, dtype=np.int64)
    for j in range(len(t)):
        diag = np.empty(qn, dtype=np.int64)
        diag[0] = top(j - 1)
        diag[1:] = prev[:-1]
        diag += (q != t[j])
        cell = np.minimum(diag, prev + 1)
        ext = np.empty(qn + 1, dtype=np.int64)
        ext[0] = top(j)
        ext[1:] = cell
        prev = (np.minimum.accumulate(ext - rows) + rows)[1:]
        yield prev


def _nw_last_column(q, t):
    col = None
    for col in _columns(q, t, lambda j: j + 1):
        pass
    return col


def _traceback(q, t, best):
    """:947-1140 on the whole matrix of a leaf"""
    qn, tn = len(q), len(t)
    m = np.empty((tn, qn), dtype=np.int64)
    for j, col in enumerate(_columns(q, t, lambda j: j + 1)):
        m[j] = col

    def cell(r, c):
        if r < 0:
            return c + 1
        if c < 0:
            return r + 1
        return int(m[c, r])
    r, c, cur, ops = qn - 1, tn - 1, int(m[tn - 1, qn - 1]), []
    assert cur == best, (cur, best)
    while True:
        u = cell(r - 1, c)
        if u + 1 == cur:
            ops.append(1); cur = u; r -= 1
            if r < 0:
                ops += [2] * (c + 1); break
            continue
        l = cell(r, c - 1)
        if l + 1 == cur:
            ops.append(2); cur = l; c -= 1
            if c < 0:
                ops += [1] * (r + 1); break
            continue
        ul = 0 if (r == 0 and c == 0) else cell(r - 1, c - 1)
        ops.append(0 if ul == cur else 3)
        cur = ul; r -= 1; c -= 1
        if c < 0:
            ops += [1] * (r + 1); break
        if r < 0:
            ops += [2] * (c + 1); break
    return ops[::-1]


def _obtain_alignment(q, t, best, stats):
    """:1166-1219 / :1236-1404"""
    qn, tn = len(q), len(t)
    if qn == 0 or tn == 0:
        return [2 if qn == 0 else 1] * (qn + tn)
    blocks = (qn + 63) // 64
    if 20 * blocks * tn + 8 * tn < 1024 * 1024:
        stats["leaves"] += 1
        return _traceback(q, t, best)
    stats["splits"] += 1
    lw = tn // 2
    rw = tn - lw
    left = _nw_last_column(q, t[:lw])                      # left[i]: query[0..i] against the left half
    right = _nw_last_column(q[::-1], t[lw:][::-1])[::-1]   # right[i]: query[i..] against the right half
    cut = None
    hit = np.nonzero(left[:qn - 1] + right[1:] == best)[0]
    if len(hit):
        cut = int(hit[0]); ls, rs = int(left[cut]), int(right[cut + 1])
    elif lw + int(right[0]) == best:
        cut = -1; ls, rs = lw, int(right[0])
    elif int(left[qn - 1]) + rw == best:
        cut = qn - 1; ls, rs = int(left[qn - 1]), rw
    assert cut is not None
    ul = cut + 1
    return _obtain_alignment(q[:ul], t[:lw], ls, stats) + _obtain_alignment(q[ul:], t[lw:], rs, stats)


def hw_align(query, target, path=True, stats=None):
    """-> dict(distance, start, end, ops) as edlibAlign returns them (ops None where edlib has no alignment)."""
    q, t = _codes(query), _codes(target)
    qn, tn = len(q), len(t)
    if stats is None:
        stats = {}
    stats.setdefault("leaves", 0); stats.setdefault("splits", 0)
    if qn == 0 or tn == 0:                                  # :174-191
        return {"distance": qn, "start": -1, "end": -1, "ops": None}
    # HW: the top boundary row is 0 everywhere; best bottom score, first column. With padding rows in the last block the
    # position before the target (score = query length) takes part and wins ties; without, it does not (:664-690).
    best, end = (qn + 1 if qn % 64 == 0 else qn), -1
    for j, col in enumerate(_columns(q, t, lambda j: 0)):
        if col[qn - 1] < best:
            best, end = int(col[qn - 1]), j
    if end < 0:
        return {"distance": best, "start": 0, "end": -1, "ops": ([1] * qn if path else None)}
    last, sbest = -1, None
    for j, col in enumerate(_columns(q[::-1], t[:end + 1][::-1], lambda j: j + 1)):
        s = int(col[qn - 1])
        if sbest is None or s < sbest:
            sbest, last = s, j
        elif s == sbest:
            last = j
    start = end - last
    ops = _obtain_alignment(q, t[start:end + 1], best, stats) if path else None
    return {"distance": best, "start": start, "end": end, "ops": ops}


def cigar(ops):
    if ops is None:
        return "*"
    out, i = [], 0
    while i < len(ops):
        j = i
        while j < len(ops) and ops[j] == ops[i]:
            j += 1
        out.append("%d%s" % (j - i, "=IDX"[ops[i]]))
        i = j
    return "".join(out) if out else "*"
