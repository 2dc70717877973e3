"""Drop-in module name of the reference's C extension: `from subg_acc import gset_sampler, walk_sampler`
(sampler/random_walks.py:18) resolves to the MI355X implementation when this repo is on sys.path."""
from surel_plus_amd.subg_acc import add, batch_sampler, gset_sampler, sjoin, walk_join, walk_sampler  # noqa: F401
