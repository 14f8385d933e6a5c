"""Randomised parity run: the HIP path against the oracle over random shapes, alphabets, scorings, gap models and
modes, for a given number of seconds.  Prints one line per batch group and a summary; exits 1 at the first
difference (with the inputs that show it).  Not a test (the oracle is the checker here, as in tests/): a tool to
spend GPU minutes on looking for inputs the fixed tests do not cover.

usage: python tools/fuzz_parity.py --seconds 240 --seed 1
"""
import argparse, collections
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bio_amd  # noqa: E402
from bio_amd import _native, pack_sequences, score_table  # noqa: E402
from oracle import oracle as O  # noqa: E402

MODES = ("global", "local", "fitting", "overlap")


SIZES = {"default": ([1, 3, 64, 65, 200, 700], [8, 40, 100, 200, 330]),
         "big": ([66000, 70000, 140000], [10, 24, 40]),     # several host-driven ranges per device
         "wide": ([1, 5, 64, 90], [700, 1500, 2600])}        # many strips, long rows
PROFILE = "default"


def random_batch(rng):
    kind = rng.choice(["dna", "dna", "protein", "binary"])
    alpha = {"dna": b"ACGT", "protein": b"ARNDCQEGHILKMFPSTWYV", "binary": b"AB"}[kind]
    al = np.frombuffer(alpha, dtype=np.uint8)
    n_pairs = int(rng.choice(SIZES[PROFILE][0]))
    shape = rng.choice(["uniform", "ragged", "few_shapes", "related"])
    hi = int(rng.choice(SIZES[PROFILE][1]))
    s1s, s2s = [], []
    shapes = [(int(rng.integers(0, hi + 1)), int(rng.integers(0, hi + 1))) for _ in range(4)]
    un, um = int(rng.integers(1, hi + 1)), int(rng.integers(1, hi + 1))
    for _ in range(n_pairs):
        if shape == "uniform":
            n, m = un, um
        elif shape == "few_shapes":
            n, m = shapes[int(rng.integers(0, 4))]
        else:
            n, m = int(rng.integers(0, hi + 1)), int(rng.integers(0, hi + 1))
        a = al[rng.integers(0, len(al), n)]
        if shape == "related" and n:
            # s2 = s1 with a few edits, cut to m: long diagonal runs, ties between paths
            b = a.copy()
            for _ in range(int(rng.integers(0, 6))):
                p = int(rng.integers(0, len(b))) if len(b) else 0
                op = rng.integers(0, 3)
                if op == 0 and len(b):
                    b[p] = al[rng.integers(0, len(al))]
                elif op == 1 and len(b):
                    b = np.delete(b, p)
                else:
                    b = np.insert(b, p, al[rng.integers(0, len(al))])
            b = b[:max(m, 1)] if rng.integers(0, 2) else b
        else:
            b = al[rng.integers(0, len(al), m)]
        s1s.append(a.tobytes())
        s2s.append(b.tobytes())
    return kind, s1s, s2s


def random_scoring(rng, kind):
    if kind == "protein" and rng.integers(0, 3):
        name = rng.choice(["Blosum62", "Pam250"])
        pa = getattr(bio_amd, name)
        oa = getattr(O, name)()
        gap = int(rng.integers(1, 12))
        return name, pa, oa, gap
    match, mismatch = int(rng.integers(0, 6)), -int(rng.integers(0, 6))
    gap = int(rng.integers(0, 8))
    return f"match {match} mismatch {mismatch}", bio_amd.MatchAligner(match, mismatch), O.MatchAligner(match, mismatch), gap


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--profile", default="default", choices=sorted(SIZES), help="batch sizes and lengths to draw from")
    ap.add_argument("--fasta", action="store_true", help="the FASTA parser (both readers' rules) on random texts")
    ap.add_argument("--unit", action="store_true", help="only what the bit-parallel kernels take: DNA / two letters, +1 / -1 / gap 1")
    ap.add_argument("--long", action="store_true", help="long mode (bioalign_long) on single pairs instead of batches")
    ap.add_argument("--stripes", type=int, default=0, help="with --long: force this many column stripes")
    ap.add_argument("--long-max", type=int, default=4000, help="with --long: largest sequence length drawn")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    global PROFILE
    PROFILE = args.profile
    if args.fasta:
        pieces = [b">", b">h d", b"ACGT", b"A C", b" ", b"\r", b"\n", b"\n", b"\r\n", b"\n\n", b"N" * 40, b"\t", b"ACGTACGTAC" * 15 + b"\n",
                  b"G" * 59 + b"\n", b">read/1 length=150\n"]
        k = 0
        while time.time() < t_end:
            size = int(rng.choice([0, 5, 60, 3000, 20000, 70000]))
            text = b"".join(pieces[int(rng.integers(0, len(pieces)))] for _ in range(size))
            if rng.integers(0, 4) == 0 and text:
                text = b">first\n" + text
            # ReadFASTA's rules (fasta.go:55-84)
            try:
                want = O.ReadFASTA(text)
            except ValueError as e:
                want = str(e)
            try:
                got = [(f.Header.encode("latin-1"), f.Seq) for f in bio_amd.ReadFASTA(text)]
            except ValueError as e:
                got = str(e)
            if got != want:
                print("MISMATCH ReadFASTA", dict(text=text[:2000]), flush=True)
                sys.exit(1)
            # FASTAReader.ReadSeq's rules (fasta.go:93-142), three buffer sizes
            for limit in (16, 200, 4096):
                r = O.FASTAReader(text, limit)
                want = []
                while True:
                    h, sq, err = r.ReadSeq()
                    if err is not None:
                        want.append(err)
                        break
                    want.append((h, sq))
                g = bio_amd.NewFASTAReader(text, limit)
                got = []
                while True:
                    try:
                        f = g.ReadSeq()
                    except EOFError:
                        got.append("EOF")
                        break
                    except ValueError as e:
                        got.append(str(e))
                        break
                    got.append((f.Header.encode("latin-1"), f.Seq))
                if got != want:
                    print("MISMATCH FASTAReader", dict(limit=limit, text=text[:2000]), flush=True)
                    sys.exit(1)
            k += 1
            if k % 50 == 0:
                print(f"{k} texts, both readers, all equal", flush=True)
        print(f"done: {k} FASTA texts, both readers (buffers 16 / 200 / 4096), all equal (seed {args.seed})", flush=True)
        return
    if args.long:
        # long mode: four modes, linear and affine gaps, DNA / few-letter / protein alphabets; per pair one of: plain,
        # column stripes in one launch, column stripes in host-driven row blocks (linear gaps, not local), linear-space
        # traceback (tiny budget; every mode and gap model)
        k = aff = 0
        variants = collections.Counter()
        al = np.frombuffer(b"ACGT", dtype=np.uint8)
        prot = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
        while time.time() < t_end:
            hi = int(rng.choice([70, 600, 1500, args.long_max]))
            n, m = int(rng.integers(1, hi + 1)), int(rng.integers(1, hi + 1))
            protein = rng.integers(0, 4) == 0
            alpha = prot if protein else al[:int(rng.integers(1, 5))]
            a = alpha[rng.integers(0, len(alpha), n)]
            if rng.integers(0, 2):
                b = a.copy()
                for _ in range(int(rng.integers(0, 30))):
                    p = int(rng.integers(0, max(len(b), 1)))
                    if rng.integers(0, 2) and len(b) > 1:
                        w = int(rng.integers(1, 40)) if rng.integers(0, 3) == 0 else 1  # long gaps too
                        b = np.delete(b, slice(p, p + w)) if len(b) > w else b
                    else:
                        b = np.insert(b, p, alpha[rng.integers(0, len(alpha), int(rng.integers(1, 20)) if rng.integers(0, 3) == 0 else 1)])
                b = b[:max(m, 1)]
            else:
                b = alpha[rng.integers(0, len(alpha), m)]
            x, y = a.tobytes(), b.tobytes()
            if protein and rng.integers(0, 3):
                name = str(rng.choice(["Blosum62", "Pam250"]))
                pa, oa = getattr(bio_amd, name), getattr(O, name)()
                gap = int(rng.integers(1, 12))
            else:
                match, mismatch, gap = int(rng.integers(0, 4)), -int(rng.integers(0, 4)), int(rng.integers(0, 5))
                pa, oa = bio_amd.MatchAligner(match, mismatch), O.MatchAligner(match, mismatch)
            mode = ("global", "local", "fitting", "overlap")[int(rng.integers(0, 4))]
            affine = rng.integers(0, 3) == 0
            ge = int(rng.integers(0, gap + 1)) if affine else gap
            affine = affine and ge != gap
            for var in ("BIOALIGN_LONG_STRIPES", "BIOALIGN_STRIPE_MERGED", "BIOALIGN_STRIPE_BLOCKS", "BIOALIGN_LONG_RESIDUE_BYTES"):
                os.environ.pop(var, None)
            variant = "plain"
            if not affine and mode != "local":
                variant = str(rng.choice(["plain", "merged", "blocks", "linear-space"])) if not args.stripes else "merged"
                if variant in ("merged", "blocks"):
                    os.environ["BIOALIGN_LONG_STRIPES"] = str(args.stripes or int(rng.integers(2, 6)))
                    os.environ["BIOALIGN_STRIPE_MERGED"] = "1" if variant == "merged" else "0"
                    os.environ["BIOALIGN_STRIPE_BLOCKS"] = str(int(rng.integers(1, 5)))
                elif variant == "linear-space":
                    os.environ["BIOALIGN_LONG_RESIDUE_BYTES"] = str(int(rng.choice([1, 400_000])))
            elif rng.integers(0, 2):  # local, affine: unstriped, with or without the whole matrix in memory
                variant = "linear-space"
                os.environ["BIOALIGN_LONG_RESIDUE_BYTES"] = str(int(rng.choice([1, 400_000])))
            try:
                got = bio_amd.AlignAffinePairLong(mode, x, y, pa, gap, ge) if affine else bio_amd.AlignPairLong(mode, x, y, pa, gap)
            except _native.BioAlignError as e:
                if e.code == _native.EUNSUPPORTED:
                    continue
                raise
            want = O.AlignAffine(mode, x, y, oa, gap, ge) if affine else O.AlignPair(mode, x, y, oa, gap)
            if got != want:
                print("MISMATCH", dict(mode=mode, variant=variant, gap=gap, ge=ge, env={v: os.environ.get(v) for v in os.environ if v.startswith("BIOALIGN")},
                                       s1=x, s2=y, want_score=want[0], got_score=got[0]), flush=True)
                sys.exit(1)
            k += 1
            aff += int(affine)
            variants[variant] += 1
            if k % 100 == 0:
                print(f"{k} long pairs, all equal", flush=True)
        print(f"done: {k} long pairs, all equal (seed {args.seed}; {aff} affine; {dict(variants)})", flush=True)
        return
    batches = pairs = unsupported = 0
    eng = bio_amd.engine()
    while time.time() < t_end:
        kind, s1s, s2s = random_batch(rng)
        name, pa, oa, gap = random_scoring(rng, kind)
        affine = bool(rng.integers(0, 3) == 0)
        ext = int(rng.integers(0, gap + 1)) if affine else gap
        if affine and ext == gap:
            affine = False
        mode = MODES[int(rng.integers(0, 4))]
        if args.unit:
            if kind == "protein":
                continue
            name, pa, oa, gap, ext, affine, mode = "match 1 mismatch -1", bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1), 1, 1, False, MODES[int(rng.integers(0, 4))]
        form = rng.choice(["strings", "ops", "scores", "ends"]) if not affine else rng.choice(["strings", "scores", "ends"])
        d1, o1 = pack_sequences(s1s)
        d2, o2 = pack_sequences(s2s)
        tbl, c1, c2 = score_table(pa, np.unique(d1).tolist(), np.unique(d2).tolist())
        flags = {"strings": 0, "ops": _native.TRACE_OPS, "scores": _native.SCORE_ONLY,
                 "ends": _native.SCORE_ONLY | _native.END_CELLS}[form]
        pb = _native.PackedBatch(d1, o1, d2, o2, mode, tbl, c1, c2, gap, ext, flags=flags)
        try:
            scores, t1, t2, t_off, t_len = eng.batch(pb)
        except _native.BioAlignError as e:
            if e.code == _native.EUNSUPPORTED:
                unsupported += 1
                continue
            raise
        for p, (x, y) in enumerate(zip(s1s, s2s)):
            want = O.AlignAffine(mode, x, y, oa, gap, ext) if affine else O.AlignPair(mode, x, y, oa, gap)
            ok = int(scores[p]) == want[0]
            if ok and form == "ends":
                # the end cell: the symbols of each trace are the stretch of its sequence that ends there
                i, j = int(t_off[p]), int(t_len[p])
                a, b = (want[1] or b"").replace(b"-", b""), (want[2] or b"").replace(b"-", b"")
                ok = 0 <= i <= len(x) and 0 <= j <= len(y) and x[i - len(a):i] == a and y[j - len(b):j] == b
                if mode != "local":
                    ok = ok and i == len(x) and (mode != "global" or j == len(y))
            elif ok and form != "scores":
                o, ln = int(t_off[p]), int(t_len[p])
                if form == "ops":
                    g1, g2 = _native.expand_ops(x, y, t1, o, ln)
                else:
                    g1, g2 = t1[o:o + ln].tobytes(), t2[o:o + ln].tobytes()
                ok = (g1 or None, g2 or None) == (want[1], want[2])
            if not ok:
                print("MISMATCH", dict(mode=mode, scoring=name, gap=gap, ext=ext, affine=affine, form=form, pair=p,
                                       s1=x, s2=y, want=want, got_score=int(scores[p])), flush=True)
                sys.exit(1)
        batches += 1
        pairs += len(s1s)
        if batches % (50 if PROFILE == "default" else 5) == 0:
            print(f"{batches} batches, {pairs} pairs, {unsupported} outside the kernels' range, all equal", flush=True)
    print(f"done: {batches} batches, {pairs} pairs, {unsupported} outside the kernels' range, all equal (seed {args.seed})", flush=True)


if __name__ == "__main__":
    main()
