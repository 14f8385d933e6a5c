"""Synthetic inputs of SURVEY.md section 8(d), vectorised over pairs.

PRNG: splitmix64, one stream per pair, seeded 0xB10A11C0 + config*2^32 + k for pair k,
so that any language can regenerate the same batch.  Draw order per pair:
  1. ceil(len1/32) words -> s1 bases (2 bits each, low bits first)   [DNA]
     or one word per residue, (word >> 11) % 20                      [protein]
  2. (fitting reads only) one word -> window offset
  3. one word per source position -> mutation decision:
       u = word & 0xFFFF; u < sub*65536 -> substitute by a different symbol,
       next ins*65536 -> insert a random symbol before it, next del*65536 -> delete it
  4. words for the random tail used when the mutated copy is shorter than the target
"""
from __future__ import annotations

import numpy as np

DNA = np.frombuffer(b"ACGT", dtype=np.uint8)
AA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
SEED0 = 0xB10A11C0
_M64 = (1 << 64) - 1


class SplitMix64:
    """Vectorised splitmix64: one independent stream per element of `seed`."""

    def __init__(self, seed: np.ndarray):
        self.s = seed.astype(np.uint64).copy()

    def next(self) -> np.ndarray:
        with np.errstate(over="ignore"):
            self.s += np.uint64(0x9E3779B97F4A7C15)
            z = self.s.copy()
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))


def _seeds(config: int, n_pairs: int, first: int = 0) -> np.ndarray:
    k = np.arange(first, first + n_pairs, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return np.uint64(SEED0) + (np.uint64(config) << np.uint64(32)) + k


def _random_symbols(rng: SplitMix64, n_pairs: int, length: int, nsym: int) -> np.ndarray:
    out = np.empty((n_pairs, length), dtype=np.uint8)
    if nsym == 4:
        for w in range((length + 31) // 32):
            word = rng.next()
            for q in range(min(32, length - 32 * w)):
                out[:, 32 * w + q] = ((word >> np.uint64(2 * q)) & np.uint64(3)).astype(np.uint8)
    else:
        for i in range(length):
            out[:, i] = ((rng.next() >> np.uint64(11)) % np.uint64(nsym)).astype(np.uint8)
    return out


def _mutate(rng: SplitMix64, src: np.ndarray, target_len: int, nsym: int,
            sub: float, ins: float, dele: float) -> np.ndarray:
    """Mutated copy of src (codes), trimmed / padded with random symbols to target_len."""
    n_pairs, L = src.shape
    t_sub = int(sub * 65536)
    t_ins = t_sub + int(ins * 65536)
    t_del = t_ins + int(dele * 65536)
    cnt = np.ones((n_pairs, L), dtype=np.int32)
    sym = src.copy()
    extra = np.zeros((n_pairs, L), dtype=np.uint8)
    for i in range(L):
        w = rng.next()
        u = (w & np.uint64(0xFFFF)).astype(np.int64)
        hi = w >> np.uint64(16)
        is_sub = u < t_sub
        is_ins = (u >= t_sub) & (u < t_ins)
        is_del = (u >= t_ins) & (u < t_del)
        shift = (hi % np.uint64(nsym - 1)).astype(np.uint8) + 1
        sym[:, i] = np.where(is_sub, (src[:, i] + shift) % nsym, src[:, i])
        extra[:, i] = (hi % np.uint64(nsym)).astype(np.uint8)
        cnt[:, i] = np.where(is_del, 0, np.where(is_ins, 2, 1))
    width = max(target_len, 2 * L) + 2
    out = _random_symbols(rng, n_pairs, width, nsym)  # random tail / padding
    start = np.cumsum(cnt, axis=1) - cnt
    rows = np.repeat(np.arange(n_pairs), L).reshape(n_pairs, L)
    m1 = cnt == 1
    out[rows[m1], start[m1]] = sym[m1]
    m2 = cnt == 2
    out[rows[m2], start[m2]] = extra[m2]
    out[rows[m2], start[m2] + 1] = sym[m2]
    return np.ascontiguousarray(out[:, :target_len])


def _finish(codes1: np.ndarray, codes2: np.ndarray, alphabet: np.ndarray):
    n_pairs, l1 = codes1.shape
    l2 = codes2.shape[1]
    s1 = alphabet[codes1].reshape(-1)
    s2 = alphabet[codes2].reshape(-1)
    o1 = (np.arange(n_pairs + 1, dtype=np.uint64) * np.uint64(l1))
    o2 = (np.arange(n_pairs + 1, dtype=np.uint64) * np.uint64(l2))
    return s1, o1, s2, o2


def dna_pairs(n_pairs: int, len1: int, len2: int, config: int, first: int = 0,
              sub: float = 0.04, ins: float = 0.01, dele: float = 0.01):
    """DNA: s1 i.i.d. uniform; s2 = mutated copy, trimmed / padded to len2.
    Returns (s1 bytes, s1_off, s2 bytes, s2_off)."""
    rng = SplitMix64(_seeds(config, n_pairs, first))
    c1 = _random_symbols(rng, n_pairs, len1, 4)
    c2 = _mutate(rng, c1, len2, 4, sub, ins, dele)
    return _finish(c1, c2, DNA)


def protein_pairs(n_pairs: int, len1: int, len2: int, config: int, first: int = 0,
                  sub: float = 0.20, indel: float = 0.02):
    rng = SplitMix64(_seeds(config, n_pairs, first))
    c1 = _random_symbols(rng, n_pairs, len1, 20)
    c2 = _mutate(rng, c1, len2, 20, sub, indel / 2, indel / 2)
    return _finish(c1, c2, AA)


def fitting_reads(n_pairs: int, read_len: int, window_len: int, config: int, first: int = 0,
                  sub: float = 0.04, ins: float = 0.01, dele: float = 0.01):
    """Config 4: s2 = random window, s1 = read cut from a random offset of it and mutated."""
    rng = SplitMix64(_seeds(config, n_pairs, first))
    win = _random_symbols(rng, n_pairs, window_len, 4)
    off = (rng.next() % np.uint64(window_len - read_len + 1)).astype(np.int64)
    idx = off[:, None] + np.arange(read_len)[None, :]
    cut = np.take_along_axis(win, idx, axis=1)
    read = _mutate(rng, cut, read_len, 4, sub, ins, dele)
    return _finish(read, win, DNA)


def dna_table(match: int = 1, mismatch: int = -1):
    """4x4 DNA scoring as a dense table + code maps for the C ABI."""
    table = np.full((4, 4), mismatch, dtype=np.int32)
    np.fill_diagonal(table, match)
    code = np.zeros(256, dtype=np.uint8)
    for i, b in enumerate(b"ACGT"):
        code[b] = i
    return table, code, code.copy()
