"""``trlda.utils`` (reference python/utils/__init__.py): the loader on the accelerated path
(python/utils/load_documents.py:6-69)."""
from trlda_amd.utils import load_documents, load_documents_csr  # noqa: F401

__all__ = ["load_documents", "load_documents_csr"]
