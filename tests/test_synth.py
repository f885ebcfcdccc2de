"""SynthNames-v1 (string_grouper_amd/synth.py): the survey's generator is deterministic, and the round-6 knob that thins its
repeats (bench.py --dup-frac, the `low_duplicates` block) leaves every other name alone."""
from string_grouper_amd.synth import _row_key, synth_names


def _repeats(names):
    seen, n = set(), 0
    for s in names:
        k = _row_key(s)
        n += k in seen
        seen.add(k)
    return n


def test_the_generator_is_deterministic_and_repeats_itself():
    a, b = synth_names(30000, 1234), synth_names(30000, 1234)
    assert a == b and a != synth_names(30000, 1235)
    assert 0.08 * len(a) < _repeats(a) < 0.2 * len(a)          # (16.5 % at 663 000 names, fewer on short lists)


def test_dup_frac_keeps_that_share_of_repeats_and_touches_nothing_else():
    base = synth_names(30000, 1234)
    thin = synth_names(30000, 1234, dup_frac=0.003)
    assert thin == synth_names(30000, 1234, dup_frac=0.003)
    assert _repeats(thin) == int(0.003 * 30000)
    changed = [i for i, (x, y) in enumerate(zip(base, thin)) if x != y]
    assert len(changed) == _repeats(base) - int(0.003 * 30000)
    seen = set()
    for i, s in enumerate(base):                               # every changed name WAS a repeat of an earlier one
        k = _row_key(s)
        if i in set(changed[:50]):
            assert k in seen
        seen.add(k)
    assert synth_names(2000, 7, dup_frac=1.0) == synth_names(2000, 7)      # nothing to thin
