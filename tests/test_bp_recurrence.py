"""The bit-parallel column recurrence of bio_amd/csrc/kernels_bp.hip, restated on Python integers (one integer = one
bit vector over the rows, the carry chains are plain additions) and checked cell by cell against a plain DP with the
reference's candidate order (align.go:105-116: up, then left if strictly better, then diagonal if strictly better).
CPU only: this pins the FORMULAS; tests/test_gpu_bitparallel.py pins the kernel against the oracle."""
import random


def plain_dp(s1, s2):
    n, m = len(s1), len(s2)
    S = [[0] * (m + 1) for _ in range(n + 1)]
    mv = [[None] * (m + 1) for _ in range(n + 1)]
    for i in range(1, n + 1):
        S[i][0] = -i
    for j in range(1, m + 1):
        S[0][j] = -j
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            w = 1 if s1[i - 1] == s2[j - 1] else -1
            best, d = S[i - 1][j] - 1, 0
            if S[i][j - 1] - 1 > best:
                best, d = S[i][j - 1] - 1, 1
            if S[i - 1][j - 1] + w > best:
                best, d = S[i - 1][j - 1] + w, 2
            S[i][j] = best
            mv[i][j] = d
    return S, mv


def bit_parallel(s1, s2, front):
    """kernels_bp.hip: ba_bp_bundle / ba_bp_column with `front` padded rows above the matrix."""
    n, m = len(s1), len(s2)
    full = (1 << (n + front)) - 1
    peq = {c: sum(1 << (i + front) for i, a in enumerate(s1) if a == c) for c in set(s2)}
    V1, V2, V3 = (1 << front) - 1, 0, 0
    planes = []
    for j in range(m):
        M = peq[s2[j]]
        X0, nV2, nV3 = ~V1 & full, ~V2 & full, ~V3 & full
        C3 = M & X0
        Q3 = ((X0 + C3) ^ X0 ^ C3) & full
        X1 = V1 & nV2
        A2 = (M & nV2) | (X1 & Q3)
        a2 = A2 | X0
        Q2 = ((a2 + A2) ^ a2 ^ A2) & full
        X2 = V2 & nV3
        P1 = (M & nV3) | X0 | (X1 & Q2) | (X2 & Q3)
        Q1 = (P1 << 1) & full
        U3, U2 = M | V3, M | V2
        nQ1, nQ2, nQ3 = ~Q1 & full, ~Q2 & full, ~Q3 & full
        V3 = nQ1 & U3
        V2 = (nQ1 & U2) | (nQ2 & U3)
        V1 = nQ1 | (nQ2 & U2) | (nQ3 & U3)
        planes.append((V1, P1))
    pc = lambda x: bin(x).count("1")
    return pc(V1) + pc(V2) + pc(V3) - front - n - m, planes


def test_bit_parallel_recurrence_is_the_plain_dp():
    rnd = random.Random(7)
    for it in range(600):
        n, m = rnd.randint(1, 75), rnd.randint(1, 75)
        al = "ACGT"[: rnd.randint(2, 4)]
        s1 = "".join(rnd.choice(al) for _ in range(n))
        if it % 2:
            s2 = "".join(rnd.choice(al) for _ in range(m))
        else:  # an edited copy: long diagonals, the runs the carry chains exist for
            s2 = "".join(c if rnd.random() < 0.85 else rnd.choice(al) for c in s1)[:m]
        front = rnd.randint(0, 31)
        S, mv = plain_dp(s1, s2)
        score, planes = bit_parallel(s1, s2, front)
        assert score == S[n][len(s2)]
        for i in range(1, n + 1):
            t = i - 1 + front
            for j in range(1, len(s2) + 1):
                v, p = planes[j - 1]
                d = 0 if not (v >> t) & 1 else 1 if not (p >> t) & 1 else 2
                assert d == mv[i][j], (s1, s2, i, j)
