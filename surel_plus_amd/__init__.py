"""surel_plus_amd -- MI355X-native SubGAcc hot path of SUREL+ (sample -> SpG -> SpJoin).

Host mirror of the reference's interfaces over a C-ABI HIP library (include/subgacc.h):
    subg_acc.gset_sampler / walk_sampler   <- subg_acc/subg_acc.c (CPython module `subg_acc`)
    spg.subg_matrix, spg.SpG               <- sampler/random_walks.py:74-82
    spjoin.gather / pgather / bgather / hgather  <- train.py:13-111
    ppr.topk_ppr_matrix, ppr.encoding      <- sampler/pprgo.py:85-111, utils.py:35-36 (float64 SpG for the PPR encoder)
"""
from ._lib import SubgAccError, build  # noqa: F401
from .sampler import DeviceCSR, SampledSets, sample_sets  # noqa: F401
from .spg import HeadedSpG, SpG, StridedSpG, np_sampling, rw_matrix, sample_spg, subg_matrix  # noqa: F401
from .spjoin import (attn_stage, bgather, gather, gather_counts, gather_index, gather_many, gather_pairs, hgather, hgather_many, lstm_stage,  # noqa: F401
                     mean_stage, pgather, sample_and_gather, sample_and_gather_many, sjoin, split_batches, StepBuffers)
from .subg_acc import batch_sampler, gset_sampler, walk_join, walk_sampler  # noqa: F401
from .ppr import topk_ppr_matrix  # noqa: F401
from .stepgraph import CapturedJoin, CapturedJoinPool, CapturedStep, CapturedStepPool  # noqa: F401
