"""``trlda`` -- the reference's import path, served by the MI355X implementation.

The reference package (python/__init__.py:1-9) exposes ``trlda.seed`` and the subpackages
``trlda.models`` (python/models/__init__.py:1-5) and ``trlda.utils``
(python/utils/__init__.py); each name here is the corresponding object of ``trlda_amd``, so a
script written against the reference -- its README example, say -- runs unchanged:

    from trlda.models import OnlineLDA
    from trlda.utils import load_documents

Only the accelerated path exists: Gibbs inference, ``sample`` and the
``load_users`` / ``random_select`` / ``sample_dirichlet`` helpers are not part of it.
"""
__license__ = 'MIT License <http://www.opensource.org/licenses/mit-license.php>'
__docformat__ = 'epytext'

from trlda_amd import __version__, seed  # noqa: F401

__all__ = ["seed", "models", "utils"]
