"""Search schemes (host-side integer tables) — mirrors fmc::search_scheme (search_scheme/ in the reference).

A scheme is a triple (pi, l, u) of uint64 arrays of shape [searches, parts] (the reference's
std::vector<Search{pi,l,u}>, search_scheme/Search.h:19-27, flattened the way the C-ABI takes it).
Pure host code: tiny tables, no device work.
"""
import numpy as np

__all__ = ["h2", "pigeon_opt", "pigeon_trivial", "backtracking", "createUniformPartition", "expand", "limitToHamming",
           "isValid", "isComplete", "nodeCount"]


def _scheme(rows):
    pi = np.array([r[0] for r in rows], dtype=np.uint64).reshape(len(rows), -1)
    l = np.array([r[1] for r in rows], dtype=np.uint64).reshape(len(rows), -1)
    u = np.array([r[2] for r in rows], dtype=np.uint64).reshape(len(rows), -1)
    return pi, l, u


def backtracking(N, minK, K):
    """generator/backtracking.h:14-21 — one search, parts left to right, any error count up to K everywhere"""
    if N <= 0 or minK > K:
        raise ValueError("backtracking(N, minK, K) needs N > 0 and minK <= K")
    l = [0] * N
    l[-1] = minK
    return _scheme([(list(range(N)), l, [K] * N)])


def _pigeon(minK, K, optimised):
    if minK > K:
        raise ValueError("pigeon needs minK <= K")
    N = K + 1
    rows = []
    for i in range(N):
        # start at part i, grow to the left end, then to the right end
        pi = [i] + list(range(i - 1, -1, -1)) + list(range(i + 1, N))
        if optimised:
            # parts left of i must hold at least one error each (else an earlier search covers the case)
            l = [0] + [i - j + 1 for j in range(i, 0, -1)] + [i] * (N - i - 1)
            u = [0] + [K - j + 1 for j in range(i, 0, -1)] + [K] * (N - i - 1)
        else:
            l = [0] * N
            u = [0] + [K] * (N - 1)
        l[-1] = max(l[-1], minK)
        rows.append((pi, l, u))
    return _scheme(rows)


def pigeon_opt(minK, K):
    """generator/pigeon.h:54-102"""
    return _pigeon(minK, K, True)


def pigeon_trivial(minK, K):
    """generator/pigeon.h:14-52"""
    return _pigeon(minK, K, False)


def h2(N, minK, K):
    """generator/h2.h:128-153 — K+1 searches over N parts (the scheme fmc::search uses, with N = K+2).

    Search r starts K-r parts in from the left, runs to the right end, then returns over the skipped parts;
    lower bounds force search r to see exactly r errors in its last K-r+1 parts, upper bounds come from a
    per-part "difference" matrix that is repaired column by column so that no search over- or under-shoots."""
    if N <= K or minK > K or N <= 0:
        raise ValueError("h2(N, minK, K) needs N > K >= minK")
    R = K + 1
    # order of the parts
    pi = np.zeros((R, N), dtype=np.int64)
    for r in range(R):
        skip = K - r
        for n in range(N):
            pi[r, n] = n + skip if n < N - skip else N - n - 1
    # lower bounds: search r demands r errors on its last K-r+1 positions
    l = np.zeros((R, N), dtype=np.int64)
    for r in range(R):
        l[r, N - (K - r + 1):] = r
    # difference matrix (h2.h:39-54) ...
    d = np.zeros((R, N), dtype=np.int64)
    for col in range(N):
        for row in range(R):
            if col >= K:
                d[row, col] = K - row
            elif row < K:
                d[row, col] = (row - col) % K
            else:
                d[row, col] = K
    # ... repaired so that every column is consistent with its row's neighbours (h2.h:56-99)

    def fits(row, col, v):
        if row == col:
            return False
        if row > col:
            return all(d[row, i] >= v for i in range(col))
        return all(d[row, i] <= v for i in range(row + 1, col))

    for col in range(N):
        for row in range(R):
            if col == row or d[row, col] == 0 or fits(row, col, d[row, col]):
                continue
            for other in range(row + 1, R):
                if fits(row, col, d[other, col]) and fits(other, col, d[row, col]):
                    d[row, col], d[other, col] = d[other, col], d[row, col]
                    break
    # upper bounds (h2.h:111-126)
    u = np.zeros((R, N), dtype=np.int64)
    for col in range(1, N):
        for r in range(R - 1, -1, -1):
            u[r, col] = max(u[r, col - 1], l[r, col - 1] + d[K - r, pi[r, col]])
    l[:, -1] = np.maximum(l[:, -1], minK)
    return pi.astype(np.uint64), l.astype(np.uint64), u.astype(np.uint64)


def createUniformPartition(parts, totalSum):
    """expand.h:324-343 — accepts a part count or a scheme"""
    if isinstance(parts, tuple):
        parts = parts[0].shape[1]
    if parts <= 0 or totalSum < parts:
        raise ValueError("createUniformPartition needs 0 < parts <= totalSum")
    base, rest = divmod(totalSum, parts)
    return np.array([base + (1 if i < rest else 0) for i in range(parts)], dtype=np.uint64)


def isValid(scheme):
    """isValid.h:55-93: pi contiguous and reaching part 0, l and u non-decreasing, l <= u"""
    pi, l, u = (np.asarray(x, dtype=np.int64) for x in scheme)
    if pi.ndim != 2 or pi.shape != l.shape or pi.shape != u.shape or pi.shape[1] == 0:
        return False
    for P, L, U in zip(pi, l, u):
        lo = hi = P[0]
        for v in P[1:]:
            if v == hi + 1:
                hi = v
            elif v + 1 == lo:
                lo = v
            else:
                return False
        if lo != 0:
            return False
        if np.any(np.diff(L) < 0) or np.any(np.diff(U) < 0) or np.any(L > U):
            return False
    return True


def expand(scheme, newLen):
    """expand.h:146-165: stretch every search to newLen parts (uniformly), drop searches that become invalid"""
    P = np.asarray(scheme[0]).shape[1]
    counts = [int(c) for c in createUniformPartition(P, newLen)] if newLen >= P else None
    if counts is None:
        base, rest = divmod(newLen, P)
        counts = [base + (1 if i < rest else 0) for i in range(P)]
    return expandByCounts(scheme, counts)


def expandByCounts(scheme, counts):
    """expand.h:167-189: part p of every search becomes counts[p] parts, searches that become invalid are dropped"""
    pi, l, u = (np.asarray(x, dtype=np.int64) for x in scheme)
    S, P = pi.shape
    counts = [int(c) for c in counts]
    newLen = sum(counts)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    rows = []
    for s in range(S):
        npi, nl, nu = [], [], []
        for i in range(P):
            part, cnt = pi[s, i], counts[pi[s, i]]
            forward = (P == 1 or pi[s, 1] > pi[s, 0]) if i == 0 else pi[s, i] > pi[s, i - 1]
            rng = range(starts[part], starts[part] + cnt)
            npi += list(rng) if forward else list(reversed(rng))
            # a part's lower bound only binds on its last character; before that the previous part's bound holds
            if cnt >= 1:
                nl += [l[s, i - 1] if i > 0 else 0] * (cnt - 1) + [l[s, i]]
            elif nl:
                nl[-1] = l[s, i]
            nu += [u[s, i]] * cnt
        if len(npi) == newLen and isValid(_scheme([(npi, nl, nu)])):
            rows.append((npi, nl, nu))
    if not rows:
        z = np.zeros((0, newLen), dtype=np.uint64)
        return z, z.copy(), z.copy()
    return _scheme(rows)


def limitToHamming(scheme):
    """expand.h:301-319: with substitutions only, the error count rises by at most one per entry"""
    pi, l, u = (np.array(x, dtype=np.int64) for x in scheme)
    for L, U in zip(l, u):
        for i in range(len(L) - 1, 0, -1):
            if L[i] == 0:
                break
            L[i - 1] = max(L[i - 1], L[i] - 1)
        for i in range(1, len(U)):
            U[i] = min(U[i], U[i - 1] + 1)
    return pi.astype(np.uint64), l.astype(np.uint64), u.astype(np.uint64)


def isComplete(scheme, minK, maxK):
    """isComplete.h:69-84: every distribution of minK..maxK errors over the parts is covered by some search"""
    pi, l, u = (np.asarray(x, dtype=np.int64) for x in scheme)
    if pi.shape[0] == 0:
        return False
    P = pi.shape[1]

    def covered(cfg):
        for Pi, L, U in zip(pi, l, u):
            acc = np.cumsum(cfg[Pi])
            if np.all((L <= acc) & (acc <= U)):
                return True
        return False

    def rec(cfg, k, start):
        if k >= maxK:
            return True
        for i in range(start, P):
            cfg[i] += 1
            ok = (k + 1 < minK or covered(cfg)) and rec(cfg, k + 1, i)
            cfg[i] -= 1
            if not ok:
                return False
        return True

    cfg = np.zeros(P, dtype=np.int64)
    if minK == 0 and not covered(cfg):
        return False
    return rec(cfg, 0, 0)


def nodeCount(scheme, sigma, edit=False):
    """nodeCount.h:19-57 (Hamming) — expected number of trie nodes a search visits on a full sigma-ary trie"""
    if edit:
        raise NotImplementedError("edit distance is outside the accelerated path")
    pi, l, u = (np.asarray(x, dtype=np.int64) for x in scheme)
    total = 0.0
    for L, U in zip(l, u):
        e = int(U.max())
        last = np.zeros(e + 1, dtype=np.longdouble)
        last[0] = 1
        acc = np.longdouble(0)
        for n in range(len(L)):
            cur = np.zeros(e + 1, dtype=np.longdouble)
            for i in range(e + 1):
                if L[n] <= i <= U[n]:
                    cur[i] = last[i] + ((sigma - 1) * last[i - 1] if i > 0 else 0)
                    acc += cur[i]
            last = cur
        total += float(acc)
    return total


def weightedNodeCount(scheme, sigma, N, edit=False):
    """weightedNodeCount.h:21-69: nodes a search visits when a node of depth n survives with probability min(1, N / sigma^n) — the reference's
    arithmetic: the weight in double, the sums in long double (80 bits on x86-64, numpy.longdouble)"""
    pi, l, u = (np.asarray(x, dtype=np.int64) for x in scheme)
    total = np.longdouble(0)
    for L, U in zip(l, u):
        e = int(U.max())
        last = [np.longdouble(0)] * (e + 1)
        last[0] = np.longdouble(1)
        acc = np.longdouble(0)
        for n in range(1, len(L) + 1):
            with np.errstate(over="ignore"):
                f = float(np.float64(N) / np.power(np.float64(sigma), np.float64(n)))      # (std::pow in double: inf for a deep trie, f = 0)
            if f > 1:
                f = 1.0
            cur = [np.longdouble(0)] * (e + 1)
            for i in range(e + 1):
                if L[n - 1] <= i <= U[n - 1]:
                    v = last[i]
                    if i > 0:
                        v = v + ((sigma - 1) * last[i - 1] + sigma * last[i - 1] + last[i - 1] if edit else (sigma - 1) * last[i - 1])
                    v = v * f
                    cur[i] = v
                    acc = acc + v
            last = cur
        total = total + acc
    return total


def optimizeByWNC(scheme, newLen, sigma, N, edit=False):
    """expand.h:218-241: grow the parts one position at a time, each time where the weighted node count of the expanded scheme is smallest
    (the running best is kept in a double, as there)"""
    P = np.asarray(scheme[0]).shape[1]
    if np.asarray(scheme[0]).shape[0] == 0:
        return []
    counts = [1] * P
    for _ in range(newLen - P):
        best, bestPos = float(np.finfo(np.float64).max), 0
        for j in range(P):
            counts[j] += 1
            f = weightedNodeCount(expandByCounts(scheme, counts), sigma, N, edit)
            counts[j] -= 1
            if f < np.longdouble(best):
                best, bestPos = float(f), j
        counts[bestPos] += 1
    return counts


def expandByWNC(scheme, newLen, sigma, N, edit=False):
    """expand.h:243-247 — what the example's `--gen <name>_dyn` uses (src/example/main.cpp:116, :135: Edit = true, sigma = 4, N = 3e9)"""
    if np.asarray(scheme[0]).shape[0] == 0:
        return scheme
    return expandByCounts(scheme, optimizeByWNC(scheme, newLen, sigma, N, edit))
