"""``trlda.models`` (reference python/models/__init__.py:1-5): the same five names."""
from trlda_amd.models import Distribution, LDA, OnlineLDA, BatchLDA, CumulativeLDA  # noqa: F401

__all__ = ["Distribution", "LDA", "OnlineLDA", "BatchLDA", "CumulativeLDA"]
