"""Drop-in alias: ``from string_grouper import match_strings`` resolves to the MI355X package.

A user of Bergvca/string_grouper switches by putting this repository ahead of the reference on
``sys.path``; ``string_grouper.string_grouper`` is the same module object as
``string_grouper_amd.string_grouper`` (so dotted-path patches in existing tests keep working)."""
import sys as _sys

import string_grouper_amd.string_grouper as _impl

_sys.modules[__name__ + ".string_grouper"] = _impl
string_grouper = _impl

from string_grouper_amd.string_grouper import (  # noqa: E402,F401
    StringGrouper, StringGrouperConfig, StringGrouperNotFitException, compute_pairwise_similarities,
    group_similar_strings, match_most_similar, match_strings)
