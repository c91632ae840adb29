"""``trlda_amd.utils`` -- the data-format helpers on either side of the E-step."""
from .load_documents import load_documents, load_documents_csr  # noqa: F401
from .synthetic import make_corpus, csr_to_docs, docs_to_csr  # noqa: F401

__all__ = ["load_documents", "load_documents_csr", "make_corpus", "csr_to_docs", "docs_to_csr"]
