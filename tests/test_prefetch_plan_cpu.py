"""Host logic of the next-weight prefetch plan (ops.pf_*, round 6): the sequence of wave-split-K products recorded in one eager pass is replayed during capture; every launch
gets the packed weight of the launch behind it, a deviation from the recorded plan switches the hints off, row-major (not packed) weights neither give nor take hints."""
import ctypes as C

from sd_lora_trainer_amd import ops


def _p(v):
    return C.c_void_p(v)


def test_record_then_replay_hands_out_the_next_packed_weight():
    ops.pf_record_begin()
    assert ops._pf_hint(_p(0x1000), 0, 1280, 1280) is None            # recording: no hints
    assert ops._pf_hint(_p(0x2000), 0, 1280, 5120) is None
    assert ops._pf_hint(_p(0x3000), 1280, 1280, 1280) is None         # row-major weight (ldw != 0)
    assert ops._pf_hint(_p(0x4000), 0, 1280, 1280) is None
    seq = ops.pf_record_end()
    assert seq == [(0x1000, 1280, 1280), (0x2000, 1280, 5120), (0, 1280, 1280), (0x4000, 1280, 1280)]
    assert ops._pf_hint(_p(0x1000), 0, 1280, 1280) is None            # outside a replay: nothing
    ops.pf_replay_begin(seq)
    try:
        assert ops._pf_hint(_p(0x1000), 0, 1280, 1280) == (0x2000, 1280, 5120)
        assert ops._pf_hint(_p(0x2000), 0, 1280, 5120) is None        # the next product reads a row-major weight: no hint
        assert ops._pf_hint(_p(0x3000), 1280, 1280, 1280) == (0x4000, 1280, 1280)
        assert ops._pf_hint(_p(0x4000), 0, 1280, 1280) is None        # last of the plan
    finally:
        ops.pf_replay_end()


def test_a_deviation_from_the_recorded_plan_switches_the_hints_off():
    seq = [(0x1000, 1280, 1280), (0x2000, 1280, 1280), (0x3000, 1280, 1280)]
    ops.pf_replay_begin(seq)
    try:
        assert ops._pf_hint(_p(0x1000), 0, 1280, 1280) == seq[1]
        assert ops._pf_hint(_p(0x9000), 0, 1280, 1280) is None        # not what was recorded
        assert ops._pf_hint(_p(0x3000), 0, 1280, 1280) is None        # ... and nothing after it either
    finally:
        ops.pf_replay_end()
    ops.pf_replay_begin([])                                           # an empty plan (no wave-split-K product in the graph): off
    assert ops._pf_hint(_p(0x1000), 0, 1280, 1280) is None
    ops.pf_replay_end()


def test_the_same_weight_twice_in_a_row_needs_no_hint():
    seq = [(0x1000, 1280, 1280), (0x1000, 1280, 1280), (0x2000, 1280, 1280)]
    ops.pf_replay_begin(seq)
    try:
        assert ops._pf_hint(_p(0x1000), 0, 1280, 1280) is None        # the next launch reads the weight this one has just streamed
        assert ops._pf_hint(_p(0x1000), 0, 1280, 1280) == seq[2]
    finally:
        ops.pf_replay_end()
