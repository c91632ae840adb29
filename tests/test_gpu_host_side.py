"""The host-side parity checks once more under the `gpu` mark, so that they also run on the GPU
box's toolchain, libc and host CPU: the text loader against the fixture from
the reference's own loader (tests/test_loader.py) and the libc-`rand()` stream of
`sampleGamma` (a4, tests/test_boundary.py).  The same functions, collected a second time: the CPU
suite (`-m "not gpu"`) keeps running them from their own files."""
import pytest

from test_boundary import (test_host_special_functions,                     # noqa: F401
                           test_jump_cache_eviction_keeps_the_stream,
                           test_lock_free_generator_is_glibc_rand,
                           test_parallel_gamma_draw_is_the_serial_stream,
                           test_sampler_matches_reference_stream)
from test_loader import (fixture, test_a_loaded_batch_is_a_mutable_sequence,  # noqa: F401
                         test_bounded_windows_lazy_batches_and_the_python_form,
                         test_list_flattening_extension, test_loader_matches_the_reference_loader,
                         test_malformed_corpus_raises_like_the_reference,
                         test_parser_threads_and_line_endings)

pytestmark = pytest.mark.gpu
