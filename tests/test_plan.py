"""The launch plan (grail_plan_blocks, include/grail_hip.h) as a pure host function: how a batch is cut into blocks
with a kernel family each, so that the time of a batch is not a step function of its size.  No GPU needed."""
import numpy as np
import pytest

import grail_hip as G

SPAN = 96006          # 2 s at 48 kHz as the f32 Sequencer clock counts them


def _cost(plan):
    return sum(b.model_ms for b in plan) + 0.05 * (len(plan) - 1)


def _one(n, fast, formants=4, cus=256, span=SPAN):
    return G.plan_blocks(n, span, fast, formants, compute_units=cus)


@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("formants", [4, 8])
def test_blocks_cover_the_batch_and_no_row_twice(fast, formants):
    for n in (1, 15, 16, 17, 4096, 4097, 65535, 65536, 65537, 70000, 98304, 131073, 200000, 524288):
        plan = _one(n, fast, formants)
        assert sum(b.rows for b in plan) == n
        assert all(b.rows > 0 for b in plan)
        assert len(plan) <= 8


@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("formants", [4, 8])
def test_cost_is_not_a_step_function_above_one_round(fast, formants):
    """T(n) <= floor(n / 65536) T(65536) + T_best(n mod 65536) + a launch."""
    full = _cost(_one(65536, fast, formants))
    assert len(_one(65536, fast, formants)) == 1
    for n in (65537, 70000, 81920, 98304, 131073, 200000):
        rest = n % 65536
        single_rest = G.plan_blocks(rest, SPAN, fast, formants)          # (itself possibly composite: at most as dear)
        bound = (n // 65536) * full + _cost(single_rest) + 0.3
        assert _cost(_one(n, fast, formants)) <= bound, n
        # and far below what one more full round costs when the rest is small
        if rest <= 8192:
            assert _cost(_one(n, fast, formants)) <= (n // 65536) * full + 0.3 * full, n


def test_exact_headline_sizes_keep_their_single_launch():
    for n, fam in ((4096, "pipe4r32"), (8192, "pipe4r16"), (16384, "exactL4"), (32768, "exactL2"), (65536, "exactL1"),
                   (131072, "exactL1")):
        plan = _one(n, 0)
        assert [(b.rows, b.family()) for b in plan] == [(n, fam)]
    for n, fam in ((2048, "pipe8r32"), (4096, "pipe8r16"), (8192, "exactL8"), (65536, "exactL1")):
        assert [(b.rows, b.family()) for b in _one(n, 0, 8)] == [(n, fam)]


def test_fast_families_by_size_at_two_seconds():
    assert _one(256, 1)[0].family() == "scan3"
    assert _one(1024, 1)[0].family() == "scan3"
    assert _one(4096, 1)[0].family() == "split16"
    assert _one(32768, 1)[0].family() == "split2"
    assert _one(65536, 1)[0].family() == "fastL1"
    # the gap between half a machine and a whole one: a time-split head and a time-split rest
    plan = _one(40000, 1)
    assert sorted(b.rows for b in plan) == [7232, 32768] and all(b.chunks for b in plan)
    assert _cost(plan) < _cost(_one(65536, 1)) - 1.0


def test_capacities_follow_the_compute_unit_count():
    """A partitioned MI355X (CPX: 32 CUs) is planned for 32 CUs: every capacity is 1/8 of the whole device's."""
    for n256, n32 in ((4096, 512), (8192, 1024), (16384, 2048), (32768, 4096), (65536, 8192), (70000, 8750 - 6)):
        a = [(b.family()) for b in _one(n256, 0, 4, 256)]
        b = [(b.family()) for b in _one(n32, 0, 4, 32)]
        assert a == b, (n256, n32, a, b)
    assert _one(8192, 0, 4, 32)[0].family() == "exactL1"
    assert _one(8192, 1, 4, 32)[0].family() == "fastL1"
    assert [b.rows for b in sorted(_one(8193, 0, 4, 32), key=lambda b: -b.rows)] == [8192, 1]


def test_short_utterances_move_the_scan_split_crossover_up():
    """A chunk's warm-up (3904 samples for voices::generic() at 48 kHz) does not shrink with the utterance: at 0.25 s
    the time-split kernels pay a third of the utterance per chunk and the scan kernel keeps batches that would be
    time-split at 2 s."""
    assert _one(2048, 1, span=SPAN)[0].chunks > 0
    assert _one(2048, 1, span=12000)[0].scan > 0
    # long utterances: the same crossover as at 2 s, or lower
    assert _one(2048, 1, span=30 * 48000)[0].chunks > 0


def test_second_tier_takes_the_exact_kernels_where_they_are_faster():
    """Sharp voices (arithmetic 2: the reference's own coefficients at every sample, 0.8 of the exact one-lane kernel):
    time-split for mid-size batches, one lane per utterance for a whole machine, and the exact kernels' wider mappings
    where those are faster — a few hundred utterances, or a voice that does not qualify for time-splitting."""
    assert _one(65536, 2)[0].family() == "fastL1" and _one(65536, 2)[0].fast == 2
    assert _one(16384, 2)[0].chunks == 4 and _one(16384, 2)[0].fast == 2
    assert _one(256, 2)[0].chunks > 16 and _one(256, 2)[0].fast == 2
    for n, fam in ((256, "pipe4r32"), (16384, "exactL4"), (32768, "exactL2")):
        no_split = G.plan_blocks(n, SPAN, 2, 4, warmup=0)            # a voice whose filters ring too long to time-split
        assert [(b.family(), b.fast) for b in no_split] == [(fam, 0)]
    assert _cost(_one(65536, 2)) < _cost(_one(65536, 0)) < 1.3 * _cost(_one(65536, 2))


def test_bad_arguments():
    with pytest.raises(G.GrailError):
        G.plan_blocks(10, SPAN, 0, 5)
    with pytest.raises(G.GrailError):
        G.plan_blocks(10, SPAN, 3, 4)
    with pytest.raises(G.GrailError):
        G.plan_blocks(10, SPAN, 0, 4, compute_units=0)
    assert G.plan_blocks(0, SPAN) == []


@pytest.mark.parametrize("cus", [256, 32, 20])
def test_time_split_grids_count_waves_not_lanes(cus):
    """A wave of a time-split launch holds 64 utterances at ONE chunk index, so a block of n rows takes ceil(n / 64)
    waves per chunk whatever n modulo 64 is: 5 000 utterances are 79 waves per chunk and get 12 chunks on 1 024 SIMDs —
    65 536 / 5 000 = 13 would be 1 027 waves, three of them in a second round (2.75 ms instead of 1.58, measured)."""
    simds = 4 * cus
    assert [b.chunks for b in _one(5000, 1)] == [12]
    for span in (12000, 24000, SPAN):
        for n in list(range(65, 6000, 61)) + [4999, 5000, 5001, 8191, 8193, 9000, 20000, 30001]:
            for b in _one(n, 1, 4, cus, span) + _one(n, 2, 8, cus, span):
                if b.chunks:
                    waves = -(-b.rows // 64) * b.chunks
                    # (whole rounds only: a block that needs a second round for a few waves is what the planner avoids)
                    assert waves <= simds or waves % simds == 0 or waves > 2 * simds, (cus, span, n, b.rows, b.chunks, waves)


def _speech_like_rows(n, seed=7):
    """Lengths, segments and kinks of utterances of 8 - 32 phonemes of 40 - 160 ms (blends of 30 - 80 ms), longest first."""
    rng = np.random.default_rng(seed)
    counts = rng.integers(8, 33, n)
    offs = np.concatenate([[0], np.cumsum(counts)])
    length = rng.uniform(0.04, 0.16, offs[-1])
    blend = rng.uniform(0.03, 0.08, offs[-1])
    samples = np.add.reduceat(length, offs[:-1]) * 48000.0
    kinks = np.add.reduceat((blend < length).astype(np.int64), offs[:-1])
    order = np.argsort(-samples, kind="stable")
    return samples[order].astype(np.uint32), counts[order].astype(np.uint32), kinks[order].astype(np.uint32)


@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("formants", [4, 8])
def test_ragged_batches_take_wider_mappings_in_several_rounds(fast, formants):
    """A machine's worth of utterances that differ in length by a factor of seven (profiles/r04_ragged_plan.txt): one wave
    per SIMD lasts as long as the longest utterance of all; two or four lanes per utterance in two or four rounds are
    shorter by the model, as they are on the device (exact 71 against 90 ms, fast 74 against 88)."""
    samples, segs, kinks = _speech_like_rows(65536)
    plan = G.plan_ragged_blocks(samples, segs, kinks, arithmetic=fast, live_formants=formants)
    assert sum(b.rows for b in plan) == 65536 and len(plan) == 1
    assert plan[0].lanes_per_utterance in (2, 4) and plan[0].chunks == 0 and plan[0].scan == 0
    # ... against the one-round plan of the same batch priced by the same rows
    one_round = G.plan_blocks(65536, int(samples[0]), fast, formants)
    assert len(one_round) == 1 and one_round[0].lanes_per_utterance == 1
    aligned = np.full(65536, samples[0], dtype=np.uint32)
    worst = G.plan_ragged_blocks(aligned, segs, kinks, arithmetic=fast, live_formants=formants)
    assert sum(b.model_ms for b in plan) < 0.9 * sum(b.model_ms for b in worst)


@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("formants", [4, 8])
@pytest.mark.parametrize("rows", [300, 4096, 20000, 65536, 70000, 131072])
def test_aligned_rows_keep_the_plan_of_aligned_batches(rows, fast, formants):
    """All utterances of one length, four segments each (the bench corpus): nothing to gain from further rounds — the
    ragged planner returns the cut grail_plan_blocks makes."""
    samples = np.full(rows, 96006, dtype=np.uint32)
    four = np.full(rows, 4, dtype=np.uint32)
    ragged = G.plan_ragged_blocks(samples, four, four if fast else None, arithmetic=fast, live_formants=formants)
    plain = G.plan_blocks(rows, 96006, fast, formants)
    assert [(b.rows, b.lanes_per_utterance, b.pipelined, b.chunks, b.scan, b.fast) for b in ragged] == \
           [(b.rows, b.lanes_per_utterance, b.pipelined, b.chunks, b.scan, b.fast) for b in plain]


def test_ragged_planner_bad_and_hostile_arguments():
    assert G.plan_ragged_blocks(np.zeros(0, dtype=np.uint32)) == []
    huge = np.full(1000, 0xFFFFFFFF, dtype=np.uint32)
    plan = G.plan_ragged_blocks(huge, huge, huge, arithmetic=1, live_formants=8)
    assert sum(b.rows for b in plan) == 1000
    with pytest.raises(G.GrailError):
        G.plan_ragged_blocks(np.ones(10, dtype=np.uint32), compute_units=0)
    # ascending instead of descending lengths, no events: still a plan of all rows
    plan = G.plan_ragged_blocks(np.arange(1, 100001, dtype=np.uint32), arithmetic=0)
    assert sum(b.rows for b in plan) == 100000


def test_dense_events_serve_a_fast_request_with_the_exact_cut():
    """Phonemes of 4 - 16 ms: an event of some lane in nearly every tile costs the fast kernels more than they save
    (4 096 utterances: 12.7 ms time-split, 5.1 ms exact; 65 536: 30.7 against 10.5) — the planner prices both and takes the
    exact families, whose bits satisfy the tolerance trivially.  A few utterances keep the scan kernel; the speech-like
    corpus itself (phonemes of 40 - 160 ms) takes the scan kernel at 4 096 and the time-split kernels at 16 384."""
    rng = np.random.default_rng(3)
    def rows(n, scale):
        counts = rng.integers(8, 33, n)
        offs = np.concatenate([[0], np.cumsum(counts)])
        length = rng.uniform(0.04, 0.16, offs[-1]) * scale
        blend = rng.uniform(0.03, 0.08, offs[-1]) * scale
        samples = np.add.reduceat(length, offs[:-1]) * 48000.0
        kinks = np.add.reduceat((blend < length).astype(np.int64), offs[:-1])
        o = np.argsort(-samples, kind="stable")
        return samples[o].astype(np.uint32), counts[o].astype(np.uint32), kinks[o].astype(np.uint32)
    for n in (20000, 65536):
        plan = G.plan_ragged_blocks(*rows(n, 0.1), arithmetic=1, live_formants=4)
        assert sum(b.rows for b in plan) == n and all(b.fast == 0 and b.chunks == 0 and b.scan == 0 for b in plan), n
    # (up to 16 384 utterances of different lengths the scan kernel is a candidate, and such rows are its own ground: a workgroup
    # per utterance, lanes = time — 4 096 such utterances 1.2 ms where the exact kernels take 4.1)
    for n in (4096, 8192, 16384):
        plan = G.plan_ragged_blocks(*rows(n, 0.1), arithmetic=1, live_formants=4)
        assert len(plan) == 1 and plan[0].rows == n and plan[0].scan != 0, n
    # (200 000 of them are six waves per SIMD on two lanes per utterance: there the fast kernels' two waves per SIMD draw level
    # with the exact one-lane kernels — 27.6 against 26.2 - 27.2 ms measured — and either plan is a good one)
    plan = G.plan_ragged_blocks(*rows(200000, 0.1), arithmetic=1, live_formants=4)
    assert sum(b.rows for b in plan) == 200000 and all(b.chunks == 0 and b.scan == 0 for b in plan)
    few = G.plan_ragged_blocks(*rows(256, 0.1), arithmetic=1, live_formants=4)
    assert len(few) == 1 and few[0].scan and few[0].fast == 1
    # the speech-like corpus itself (phonemes of 40 - 160 ms): the scan kernel up to 8 704 utterances (4 096: 6.7 ms, 15.5 on the
    # time-split kernels, which fast-forward through 64 utterances' events and wait for the longest), time-split beyond
    speech = G.plan_ragged_blocks(*rows(4096, 1.0), arithmetic=1, live_formants=4)
    assert len(speech) == 1 and speech[0].scan != 0 and speech[0].fast == 1
    speech = G.plan_ragged_blocks(*rows(16384, 1.0), arithmetic=1, live_formants=4)
    assert len(speech) == 1 and speech[0].chunks >= 5 and speech[0].fast == 1
