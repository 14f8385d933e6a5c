"""bio_amd -- MI355X (gfx950) batched pairwise alignment behind soniakeys/bio's API.

Only the DP alignment path of the reference lives here:
hand-written HIP kernels + a C ABI (include/bioalign.h) + this thin host mirror.
"""
from .subst_matrix import (Blosum62, MatchAligner, NewSubstMatrix, Pam250,  # noqa: F401
                           SubstMatrix, score_table)
from .align import (AlignAffineBatch, AlignAffinePairLong, AlignGlobal, AlignGlobalAffine, AlignGlobalLS,  # noqa: F401
                    AlignLocal, AlignPair, AlignPairBatch, AlignPairLong, ConstantGap, LinearGap,
                    engine, pack_sequences, set_engine)
from . import _native  # noqa: F401

GapSymbol = ord("-")  # package variable (align.go:3); assign bio_amd.GapSymbol to change it

Seq = bytes  # seq.go:18
from .fasta import FASTAReader, FASTASeq, FastaBatch, NewFASTAReader, ReadFASTA, ReadFASTAFile  # noqa: F401,E402
from .distance import AlignDist, AlignScores, AlignScoresSharded, DistanceMatrix, pair_ranges  # noqa: F401,E402
from .assembly import AlignLongOverlap  # noqa: F401,E402
