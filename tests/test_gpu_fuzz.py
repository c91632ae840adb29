"""A short run of the randomised E-step check (tests/fuzz_estep.py): random K, V, batch sizes,
document lengths across every tier of the launch, duplicate ids, zero counts, iteration limits
and thresholds; both statistics modes, split documents on and off; gamma, statistics and
per-document iteration counts against the oracle."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(hip_lib):
    from trlda_amd import _ffi
    assert _ffi.device_count() >= 1, "GPU tests need a visible MI355X"
    return hip_lib


@pytest.mark.parametrize("seed", [3, 11])
def test_random_shapes_agree_with_the_oracle(hip, seed):
    import fuzz_estep
    worst_g, worst_s = fuzz_estep.main(["--cases", "16", "--seed", str(seed)])   # exits non-zero on a mismatch
    assert worst_g < 1e-8 and worst_s < 1e-7


def test_a_batch_after_a_closed_model_leaves_the_callers_arrays_alone(hip):
    """Round 4's long fuzz run (seed 101): a model that owns its stream is closed, a later model's first
    batch reuses a device allocation whose guard event that stream had recorded -- the runtime followed
    the event into the destroyed stream object and incremented a word of freed host memory, which by
    then was the next case's `cnts` array (one document count 1 -> 2; statistics of that word doubled
    from the second call on).  The library no longer keeps such events (trlda_hip.hip, batch_settle /
    purge_stream_guards); the fuzzers compare their input arrays with copies at the end of every case."""
    import fuzz_estep
    worst_g, worst_s = fuzz_estep.main(["--cases", "13", "--seed", "101", "--run", "6,12"])
    assert worst_g < 1e-8 and worst_s < 1e-7
    worst_g, worst_s = fuzz_estep.main(["--cases", "13", "--seed", "101", "--run", "10,11,12"])
    assert worst_g < 1e-8 and worst_s < 1e-7


def test_models_and_batches_with_overlapping_lifetimes(hip):
    """tests/fuzz_lifecycle.py: up to three models alive at once on streams of their own, torch's
    current stream or a torch side stream, resident batches that outlive the model that last read
    them, E-steps and update calls interleaved -- every result against the oracle, every host array
    handed over (and blocks of freshly taken host memory after each closed model) unchanged."""
    import fuzz_lifecycle
    assert fuzz_lifecycle.main(["--steps", "120", "--seed", "2"]) < 1e-8


def test_random_update_calls_agree_with_the_oracle(hip):
    import fuzz_update
    assert fuzz_update.main(["--cases", "8", "--seed", "5"]) < 1e-8


def test_random_update_calls_agree_with_the_reference_itself(hip):
    """OnlineLDA (empirical Bayes, adaptive rate), BatchLDA (line searches) and CumulativeLDA calls on
    random inputs against the reference's unmodified C++ core (oracle/_ref, built where
    /root/reference exists; the library travels with the tree)."""
    from oracle.pyoracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libtrlda_ref.so not built (needs /root/reference at build time)")
    import fuzz_reference
    assert fuzz_reference.main(["--cases", "15", "--seed", "8"]) < 1e-7


def test_random_deferred_streams_equal_the_plain_ones(hip):
    """tests/fuzz_deferred.py: random streams of announced E-steps with deferred statistics, shapes
    inside and outside the stage, and everything that must flush in between -- bitwise the outputs
    of the same stream with the switch off, at the oracle's values."""
    import fuzz_deferred
    assert fuzz_deferred.main(["--cases", "6", "--seed", "4"]) < 1e-8


def test_random_streams_through_two_lanes_equal_the_plain_ones(hip):
    """tests/fuzz_deferred.py --lanes: the same random streams through trlda_model_estep_io_ahead with
    two stream lanes (consecutive calls in flight at once), announcements two calls ahead right,
    wrong or missing, the lanes switched off and on, and everything that must join them in between."""
    import fuzz_deferred
    assert fuzz_deferred.main(["--cases", "6", "--seed", "9", "--lanes"]) < 1e-8
