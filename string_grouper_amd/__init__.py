"""string_grouper_amd -- MI355X-native core for string_grouper's fuzzy-matching hot path.

Public surface = the reference's (string_grouper/__init__.py:1-2): the four module-level functions,
``StringGrouper`` and ``StringGrouperConfig``.  The compute runs in libsg_hip.so (hand-written HIP
kernels for gfx950, C ABI in include/sg_hip.h); there is no CPU fallback."""
from .string_grouper import (StringGrouper, StringGrouperConfig, StringGrouperNotFitException,  # noqa: F401
                             compute_pairwise_similarities, group_similar_strings, match_most_similar,
                             match_strings)

__all__ = ["StringGrouper", "StringGrouperConfig", "StringGrouperNotFitException", "compute_pairwise_similarities",
           "group_similar_strings", "match_most_similar", "match_strings"]
