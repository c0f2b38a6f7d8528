"""GPU: the persistent decode engine (csrc/decode_engine.hip: the 24 blocks of a token step as ONE launch, <= 6 rows,
bf16) against the launch path it replaces (gemv_bf16_kernel + decode_attn2_kernel, five launches a layer): the engine
repeats the launch path's arithmetic operation for operation, so ids AND logits have to be bit-identical - at 1 to 4 (5 - 6 rows: the launch path is MFMA there, tolerance + row-independence tests)
rows, under graph replay and eager launches, through the register window of the cache attention and past it.
Reference hot loop: indextts/gpt/model.py:115-192."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from itts_hip import config as icfg  # noqa: E402
from itts_hip import engine as ieng  # noqa: E402
from itts_hip import synth  # noqa: E402

CFG = icfg.indextts_1_5()


@pytest.fixture(scope="module")
def eng16():
    return ieng.build_engine(CFG, "bf16", parts=("gpt",))


@pytest.fixture(scope="module")
def cond(eng16):
    return eng16.conditioning(torch.from_numpy(synth.prompt_mel(511, seed=7)))


def run(eng, cond, text, steps, no_engine, no_graph=False, chunk=8):
    eng.debug(no_engine=no_engine, engine=not no_engine, no_graph=no_graph)
    try:
        eng.prefill(cond, text, steps, 10.0, True)
        done = 1
        while done < steps:
            n = min(chunk, steps - done)
            eng.decode(n)
            done += n
        codes, lg = eng.fetch(logits=True)
        eng._exit()
    finally:
        eng.debug()
    return codes, lg


@pytest.mark.parametrize("rows", [1, 2, 3, 4])
def test_engine_equals_launch_path_bitwise(eng16, cond, rows):
    text = np.stack([synth.text_ids(105, 11 + r, CFG.gpt.number_text_tokens) for r in range(rows)]).astype(np.int32)
    ref_codes, ref_lg = run(eng16, cond, text, 64, no_engine=True)
    codes, lg = run(eng16, cond, text, 64, no_engine=False)
    assert np.array_equal(codes, ref_codes)
    assert np.array_equal(lg.view(np.uint32), ref_lg.view(np.uint32)), float(np.abs(lg - ref_lg).max())
    # eager launches of the same kernel (no graph): same bits again
    codes2, lg2 = run(eng16, cond, text, 24, no_engine=False, no_graph=True)
    assert np.array_equal(codes2, ref_codes[:, :24])


@pytest.mark.parametrize("rows", [2, 4])
def test_engine_ragged_rows_and_long_sequence(eng16, cond, rows):
    """Rows of different text lengths (left padding -> kv_start) decoded far past the 768-key register window of the
    cache attention (prefix 139 + 700 steps): the streaming part of every key split runs.  4 rows: the two-slot LDS map."""
    stop = CFG.gpt.stop_text_token
    text = np.full((rows, 105), stop, np.int32)  # start/stop text ids are stripped and left-padded by the prefix builder
    for r, n in enumerate([105, 60, 33, 90][:rows]):
        text[r, :n] = synth.text_ids(n, 21 + r, CFG.gpt.number_text_tokens)
    ref_codes, ref_lg = run(eng16, cond, text, 700, no_engine=True, chunk=64)
    codes, lg = run(eng16, cond, text, 700, no_engine=False, chunk=64)
    assert np.array_equal(codes, ref_codes)
    assert np.array_equal(lg.view(np.uint32), ref_lg.view(np.uint32))


@pytest.mark.parametrize("items,nb,sample", [(1, 3, True), (1, 3, False), (2, 2, True), (1, 4, True), (1, 2, False)])  # (2 x 3 = 6 rows: tolerance test below)
def test_engine_beam_rows_equal_launch_path(eng16, cond, items, nb, sample):
    """The reference's default generate() mode (num_beams 3, model.py:698-703) one sentence at a time = 3 rows: the
    engine's cache attention gathers every key through the beam's ancestry row exactly as decode_attn2_kernel<.., ANC>
    does (the cache is never re-ordered), so the beam sampler sees the same logits bit for bit and returns the same ids,
    past the 768-key register window (prefix 139 + 700 steps) and under graph replay."""
    text = np.stack([synth.text_ids(105, 51 + r, CFG.gpt.number_text_tokens) for r in range(items)]).astype(np.int32)
    n = 700 if (items, nb, sample) == (1, 3, True) else 96
    u = np.random.default_rng(9).random((n, items, 2 * nb), dtype=np.float32)
    kw = dict(do_sample=sample, num_beams=nb, top_k=30, top_p=0.8, temperature=1.0, uniforms=u, suppress_stop=True)
    res, modes = [], []
    for no_engine in (True, False):
        eng16.debug(no_engine=no_engine, engine=not no_engine)
        try:
            res.append(eng16.generate(cond, text, n, **kw))
            modes.append(eng16.decode_mode())
        finally:
            eng16.debug()
    assert modes == [0, 1], modes
    assert res[0].shape == (items, n) and np.array_equal(res[0], res[1])


@pytest.mark.parametrize("sample", [True, False])
def test_engine_six_beam_rows_equal_the_three_row_runs(eng16, cond, sample):
    """2 sentences x 3 beams = 6 rows on the engine (two-half mlp.c_proj slots, ancestry gather with nb = 3 of 6 rows): rows
    are independent and the engine's arithmetic per row does not depend on the row count, so every sentence's ids equal
    the ids of that sentence decoded alone (3 rows on the engine = bit-identical to the launch path, above)."""
    text = np.stack([synth.text_ids(105, 91 + r, CFG.gpt.number_text_tokens) for r in range(2)]).astype(np.int32)
    n, nb = 64, 3
    u = np.random.default_rng(17).random((n, 2, 2 * nb), dtype=np.float32)
    kw = dict(do_sample=sample, num_beams=nb, top_k=30, top_p=0.8, temperature=1.0, suppress_stop=True)
    eng16.debug(engine=True)
    try:
        both = eng16.generate(cond, text, n, uniforms=u, **kw)
        assert eng16.decode_mode() == 1
        for i in range(2):
            one = eng16.generate(cond, text[i:i + 1], n, uniforms=np.ascontiguousarray(u[:, i:i + 1]), **kw)
            assert np.array_equal(one[0], both[i]), i
    finally:
        eng16.debug()


@pytest.fixture(scope="module")
def eng16s():  # the well-conditioned synthetic checkpoint of tests/test_gpu_bf16_accuracy.py: bf16 noise is not amplified
    return ieng.build_engine(CFG, "bf16", parts=("gpt",), state_dicts={"gpt": synth.gpt_state_dict(CFG, 1234, profile="smooth")})


@pytest.mark.parametrize("rows", [5, 6])
def test_engine_5_6_rows_track_the_mfma_launch_path(eng16s, rows):
    """5 - 6 rows (2 sentences x 3 beams: the reference's default mode on a two-sentence text): the engine keeps the GEMV
    arithmetic (v_dot2c chains), the launch path at these row counts runs on the matrix cores - different summation orders,
    so the comparison is a tolerance: teacher-forced on the launch path's own ids, every step's logits within 1e-2 relative
    RMS (measured ~3e-3: two bf16 evaluations of the same step), the arg-max equal wherever the margin is not a near-tie."""
    eng16 = eng16s
    cond = eng16.conditioning(torch.from_numpy(synth.prompt_mel(511, seed=7)))
    stop = CFG.gpt.stop_text_token
    padded = np.full((rows, 105), stop, np.int32)  # ragged rows (left padding -> kv_start)
    for r in range(rows):
        padded[r, :105 - 7 * r] = synth.text_ids(105 - 7 * r, 71 + r, CFG.gpt.number_text_tokens)
    n = 40
    ref_codes, _ = run(eng16, cond, padded, n, no_engine=True)
    traces, modes = [], []
    try:
        eng16.set_forced(ref_codes[:, :n])
        for no_engine in (True, False):
            eng16.debug(no_engine=no_engine, engine=not no_engine)
            eng16.prefill(cond, padded, n, 10.0, True)
            lgs = []
            for k in range(n):
                lgs.append(eng16.fetch(logits=True)[1].copy())
                if k + 1 < n:
                    eng16.decode(1)
            modes.append(eng16.decode_mode())
            eng16._exit()
            traces.append(np.stack(lgs))
    finally:
        eng16.set_forced(None)
        eng16.debug()
    assert modes == [0, 1], modes
    a, b = traces
    worst = 0.0
    for k in range(n):
        for r in range(rows):
            d = float(np.sqrt(((a[k, r] - b[k, r]) ** 2).mean()) / np.sqrt((a[k, r] ** 2).mean()))
            worst = max(worst, d)
            top = np.sort(a[k, r])[-2:]
            if top[1] - top[0] > 0.25:
                assert int(a[k, r].argmax()) == int(b[k, r].argmax()), (k, r)
    print(f"engine vs MFMA launch path, {rows} rows: worst per-step logits rel-RMS {worst:.2e}")
    assert worst < 1e-2, worst


@pytest.mark.parametrize("rows", [2, 6])
def test_engine_repeats_are_identical(eng16, cond, rows):
    """The hand-offs are polled, their timing differs from run to run - the results must not: three full-length generations
    (479 engine launches each, graph replay) return the same ids and the same final logits, bit for bit."""
    text = np.stack([synth.text_ids(105, 131 + r, CFG.gpt.number_text_tokens) for r in range(rows)]).astype(np.int32)
    runs = [run(eng16, cond, text, 480, no_engine=False, chunk=64) for _ in range(3)]
    for codes, lg in runs[1:]:
        assert np.array_equal(codes, runs[0][0]) and np.array_equal(lg.view(np.uint32), runs[0][1].view(np.uint32))


def test_engine_status_reports_no_timeout(eng16, cond):
    text = synth.text_ids(105, 31, CFG.gpt.number_text_tokens).reshape(1, -1).astype(np.int32)
    eng16.debug(engine=True)
    try:
        codes = eng16.generate(cond, text, 40)  # eos enabled: status() every 16 steps reads the engine's abort word
    finally:
        eng16.debug()
    assert codes.shape[0] == 1 and codes.shape[1] >= 1


def test_engine_timeout_is_reported_and_the_launch_path_takes_over(cond, monkeypatch):
    """Failure detection: every in-launch wait is bounded by a wall clock (20 ms).  With the bound forced to 10 ns every
    hand-off gives up, the launch drains (the in-launch sampler commits NOTHING when its candidates are missing - no id outside
    the vocabulary reaches the id buffer, the seen-bitmap or the embedding gather), the abort word is set: status() / fetch()
    return ITTS_E_HANDOFF (lib.HandoffTimeout), the engine object disables its persistent engine, and Engine.generate redoes the
    utterance once on the five-launches-per-block path with a logged RuntimeWarning - the caller gets the right ids."""
    from itts_hip import lib as ilib

    eng = ieng.build_engine(CFG, "bf16", parts=("gpt",))
    text = synth.text_ids(105, 61, CFG.gpt.number_text_tokens).reshape(1, -1).astype(np.int32)
    eng.debug(no_engine=True)
    want = eng.generate(cond, text, 12, suppress_stop=True)
    monkeypatch.setenv("ITTS_ENGINE_TIMEOUT_TICKS", "1")
    eng.debug(engine=True, no_graph=True)
    # the C ABI level: the failed generation is reported, not returned
    eng.prefill(cond, text, 12, 10.0, True)
    eng.decode(11)
    with pytest.raises(ilib.HandoffTimeout, match="hand-off timed out"):
        eng.fetch()
    eng._exit()
    assert issubclass(ilib.HandoffTimeout, RuntimeError)
    # the product level: a second engine object, same forced timeout - generate() downgrades, warns and returns the right ids
    eng2 = ieng.build_engine(CFG, "bf16", parts=("gpt",))
    eng2.debug(engine=True, no_graph=True)
    with pytest.warns(RuntimeWarning, match="persistent decode engine timed out"):
        got2 = eng2.generate(cond, text, 12, suppress_stop=True)
    assert eng2.decode_mode() == 0 and np.array_equal(got2, want)
    ids_ok = (got2 >= 0) & (got2 < CFG.gpt.number_mel_codes)
    assert ids_ok.all()
    monkeypatch.delenv("ITTS_ENGINE_TIMEOUT_TICKS")
    eng.debug(engine=True)
    got = eng.generate(cond, text, 12, suppress_stop=True)
    assert eng.decode_mode() == 0  # disabled for this engine object after the failure
    assert np.array_equal(got, want)
    eng.debug()
    eng2.debug()


def test_two_engine_objects_interleaved_on_one_device(eng16, cond):
    """Two Engine objects (two HIP streams) decoding at the same time on one GPU, both on the persistent engine: its launches
    keep one workgroup on every CU and wait for each other, so two in flight together could starve each other until the
    20 ms bound fires - the library chains the engine launches of a device across streams.  Interleaved decode calls
    without any synchronisation in between give each engine the ids it produces alone."""
    other = ieng.build_engine(CFG, "bf16", parts=("gpt",))
    texts = [synth.text_ids(105, 81 + i, CFG.gpt.number_text_tokens).reshape(1, -1).astype(np.int32) for i in range(2)]
    n = 96
    alone = []
    for e, t in ((eng16, texts[0]), (other, texts[1])):
        e.debug(engine=True)
        alone.append(e.generate(cond, t, n, suppress_stop=True))
    try:
        for e, t in ((eng16, texts[0]), (other, texts[1])):
            e.prefill(cond, t, n, 10.0, True)
        for k in range(0, n - 1, 5):
            for e in (eng16, other):
                e.decode(min(5, n - 1 - k))
        got = []
        for e in (eng16, other):
            assert e.decode_mode() == 1
            got.append(e.fetch().astype(np.int64))
            e._exit()
    finally:
        eng16.debug()
        other.debug()
    assert np.array_equal(got[0], alone[0]) and np.array_equal(got[1], alone[1])


def test_launch_path_eager_full_length_equals_graph(eng16, cond):
    """The 122-launches-per-step path at the bench's full length WITHOUT graph capture (58 k eager launches: the run the
    r02 rocprofv3 --pmc pass died in) against its graph replay: same ids, same logits - the eager path's scratch / state
    indexing holds for 479 steps (the older graph-vs-eager test stops at 24)."""
    text = np.stack([synth.text_ids(105, 41 + r, CFG.gpt.number_text_tokens) for r in range(2)]).astype(np.int32)
    ref_codes, ref_lg = run(eng16, cond, text, 480, no_engine=True, chunk=64)
    codes, lg = run(eng16, cond, text, 480, no_engine=True, no_graph=True, chunk=64)
    assert np.array_equal(codes, ref_codes) and np.array_equal(lg.view(np.uint32), ref_lg.view(np.uint32))
