"""trlda_amd -- MI355X (gfx950) implementation of trlda's variational E-step.

Drop-in for the accelerated path of the reference ``trlda`` package
(python/__init__.py:1-9): ``trlda_amd.seed``, ``trlda_amd.models.OnlineLDA`` /
``BatchLDA`` and ``trlda_amd.utils.load_documents``.  All computation goes through
the C ABI of ``libtrlda_hip.so`` (include/trlda_hip.h) into hand-written HIP
kernels; there is no CPU fallback.
"""
__version__ = "0.1.0"


def seed(value):
    """``trlda.seed``: ``srand(value)`` (reference python/src/module.cpp:332-342).

    Pins the libc ``rand()`` stream that draws the initial lambda (lda.cpp:71) and
    every default gamma initialisation (lda.cpp:135)."""
    from . import _ffi
    if isinstance(value, float):
        raise TypeError("integer argument expected, got float")
    _ffi.lib().trlda_seed(int(value) & 0xFFFFFFFF)


from . import models  # noqa: E402,F401
from . import utils  # noqa: E402,F401
