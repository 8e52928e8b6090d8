"""A stream of batches (c4gpu_stage, include/c4gpu.h): the next batch is staged -- gathered into page-locked memory, copied
over the link, coded, its splice arrays built -- on a stream of its own while the current one is aligned, then swapped into
the batch that keeps the engine, the launch lanes and their buffers.  What the reference does per pair before its first cell
(Sequence_strncpy sequence.c:588, Intron_Data's splice prediction intron.c:259-269) must come out the same whether a batch
was created in one piece or arrived through a stage, alone or while another batch was running."""
import threading

import pytest

import exonerate_amd as ex
from exonerate_amd import workloads
import oracle_lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = ex.Engine(0)
    yield e
    e.close()


def _export(b):
    return b.export().tolist()


def test_two_batches_in_flight_give_the_single_batch_results(eng):
    """Batch k runs while batch k + 1 is staged by another thread, six batches of different sizes in a row (the device
    arrays and page-locked buffers of batch k - 1 are reused by batch k + 1, larger and smaller): every result stream equals
    the one a batch created in one piece gives, and a sample of alignments equals the oracle's."""
    model = ex.Model("est2genome")
    sizes = [(48, 700, 36000), (64, 1000, 40000), (20, 400, 60000), (64, 1000, 40000), (3, 90, 32000), (33, 1023, 45000)]
    batches = [workloads.est2genome_pairs(n, q, t, first=1000 * k) for k, (n, q, t) in enumerate(sizes)]
    want = []
    for pairs in batches:
        b = ex.ResidentBatch(eng, model, pairs)
        b.run(2)
        want.append(_export(b))
        b.close()
    stage = ex.Stage(eng, model)
    stage.load(batches[0])
    batch = ex.ResidentBatch(eng, model, batches[0][:2])          # any batch: it is swapped out before the first run
    for k, pairs in enumerate(batches):
        batch.swap(stage)
        err = []
        nxt = batches[k + 1] if k + 1 < len(batches) else None

        def bg():
            try:
                if nxt is not None:
                    stage.load(nxt)
            except Exception as e:                                  # noqa: BLE001 (reported in the main thread)
                err.append(e)

        th = threading.Thread(target=bg)
        th.start()
        batch.run(2)
        got = _export(batch)
        th.join()
        assert not err, err
        assert got == want[k], "batch %d differs from the batch created in one piece" % k
        if len(pairs[0][0]) <= 400:                                 # (the oracle takes seconds per 10^7 cells)
            q, t = pairs[-1]
            a = batch.alignment(len(pairs) - 1)
            assert a is not None and a.as_dict() == oracle_lib.find_path(model.c, model.params, q, t), k
    batch.close()
    stage.close()


def test_stage_for_other_families_and_a_shared_contig(eng):
    """protein2genome against one shared contig (one device copy, phase arrays) and affine:local through a stage."""
    proteins, contig, _ = workloads.protein_vs_contig(12, 120, 80000, seed=77, introns=True)
    for name, pairs in (("protein2genome", [(p, contig) for p in proteins]),
                        ("affine:local", workloads.affine_dna_pairs(40, 300))):
        model = ex.Model(name)
        ref = ex.ResidentBatch(eng, model, pairs)
        ref.run(2)
        stage = ex.Stage(eng, model)
        stage.load(pairs)
        b = ex.ResidentBatch(eng, model, pairs[:1])
        b.swap(stage)
        b.run(2)
        assert _export(b) == _export(ref), name
        stage.load(pairs[::-1])                                 # into the buffers the batch handed back
        b.swap(stage)
        b.run(2)
        exp = [ref.alignment(i) for i in range(len(pairs))][::-1]
        for i, e in enumerate(exp):
            a = b.alignment(i)
            assert (a is None) == (e is None) and (a is None or a.as_dict() == e.as_dict()), (name, i)
        b.close(); ref.close(); stage.close()


def test_swap_refuses_an_unloaded_stage_and_another_model(eng):
    m1, m2 = ex.Model("est2genome"), ex.Model("affine:local")
    pairs = workloads.affine_dna_pairs(4, 100)
    b = ex.ResidentBatch(eng, m1, pairs)
    st = ex.Stage(eng, m1)
    with pytest.raises(ex.C4GpuError):
        b.swap(st)                                               # nothing loaded
    st2 = ex.Stage(eng, m2)
    st2.load(pairs)
    with pytest.raises(ex.C4GpuError):
        b.swap(st2)                                              # another model
    st.load(pairs)
    b.swap(st)
    with pytest.raises(ex.C4GpuError):
        b.swap(st)                                               # a load is handed over once
    b.close(); st.close(); st2.close()


def test_tiled_splice_arrays_equal_the_oracle_at_every_tile_boundary(eng, lib):
    """SplicePredictor_predict_array_int (splice.c:383-397) by the tiled kernel (1 024 positions per workgroup, four per
    thread, a sliding window of model columns): targets whose ends fall on, before and behind tile and thread boundaries, with
    N, lower case and non-IUPAC residues, default parameters and --forcegtag; every array against the oracle's (which is
    pinned on the reference's arrays, tests/test_oracle_golden.py)."""
    import ctypes as C
    import random
    from exonerate_amd import _abi
    olib = oracle_lib.load()
    rng = random.Random(5)
    for gtag in (0, 1):
        params = ex.default_params()
        lib.c4gpu_params_set_forcegtag(params, gtag)
        for n in (1, 2, 3, 4, 5, 19, 20, 21, 40, 1023, 1024, 1025, 1027, 1040, 2047, 2048, 2051, 5000, 100003):
            t = "".join(rng.choice("ACGTACGTACGTNacgtnRY") for _ in range(n))
            if n > 30:
                t = t[:7] + "GTAAGT" + t[13:n - 9] + "TTTCAGG" + t[n - 2:]
            got = eng.splice_predict(params, t)
            for k in range(4):
                out = (C.c_int32 * n)()
                olib.oracle_splice_predict(params.splice[k], t.encode(), n, out)
                assert got[k] == list(out), (gtag, n, k)


def test_the_splice_kernel_writes_the_packed_array_ss16_kernel_builds(eng, monkeypatch):
    """A stage lets the splice kernel write the packed passes' array (four clamped 16-bit values per position, pre-splice
    constants folded in) instead of building it from the int arrays on the first packed launch: both forms, position for
    position (C4GPU_SS16_CHECK), for default and non-default intron penalties."""
    monkeypatch.setenv("C4GPU_SS16_CHECK", "1")
    for penalty in (-30, -3, -200):
        params = ex.default_params()
        params.intron_open_penalty = penalty
        model = ex.Model("est2genome", params=params)
        st = ex.Stage(eng, model)
        st.load(workloads.est2genome_pairs(9, 300, 33000, first=7))
        st.close()
