import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bio_amd
from bio_amd import synth
from oracle import oracle as O
def pair(n, m, seed):
    s1, o1, s2, o2 = synth.dna_pairs(1, n, m, config=5, first=seed, sub=0.06, ins=0.02, dele=0.02)
    return s1.tobytes(), s2.tobytes()
cases = [((10,-3,5),260,777,302), ((10,-3,5),700,900,302), ((1,-2,2),700,900,302),((1,-2,2),260,777,302), ((1,-2,2),300,310,300), ((1,-2,2),1030,520,301), ((1,-1,1),260,777,302), ((5,-4,3),260,777,302),((1,-2,2),128,777,302),((1,-2,2),129,777,302),((1,-2,2),260,400,302),((1,-2,2),260,600,302)]
for (ma, mi, g), n, m, seed in cases:
    a, oa = bio_amd.MatchAligner(ma, mi), O.MatchAligner(ma, mi)
    x, y = pair(n, m, seed)
    want = O.AlignPair("global", x, y, oa, g)
    for rep in range(3):
        got = bio_amd.AlignPairLong("global", x, y, a, g)
        if got == want:
            print((ma,mi,g), n, m, "rep", rep, "ok")
        else:
            t1, w1 = got[1], want[1]
            # first difference from the END (the walk starts there)
            k = 0
            while k < min(len(t1), len(w1)) and t1[-1-k] == w1[-1-k] and got[2][-1-k] == want[2][-1-k]: k += 1
            # cell where they diverge
            i = n - sum(1 for c in w1[len(w1)-k:] if c != ord('-'))
            j = m - sum(1 for c in want[2][len(w1)-k:] if c != ord('-'))
            print((ma,mi,g), n, m, "rep", rep, "DIFF score", got[0], want[0], "lens", len(t1), len(w1), "diverge after", k, "moves at cell", (i, j))
