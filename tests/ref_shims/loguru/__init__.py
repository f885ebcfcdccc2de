"""Two-line stand-in for ``loguru`` (absent in this image); test infrastructure only."""
import logging

logger = logging.getLogger("string_grouper")
