"""SURVEY 8f-4 on the device: the printers of alignment.c -- Alignment_display (alignment.c:234-1380), the GFF2 dumps
(alignment.c:2710-3236) and --ryo (alignment.c:1781-2669) -- fed with the alignments the MI355X makes
(c4gpu_optimal_find_path_batch through the C ABI) and compared byte for byte with the stdout of the reference's own compiled
exonerate (oracle/_ref/exonerate-compiled, travelling as a binary), five models, both strands.  The cases and the comparison
are those of tests/test_printers.py (where the alignments come from the oracle and no device is needed); here only the
source of the alignment changes, and every alignment is also checked against the vulgar line the reference printed."""
import os
import pytest

import exonerate_amd as ex
import test_printers as tp

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(tp.CPU_EXE), reason="the reference binary is built in the build container")]


@pytest.fixture(scope="module")
def eng():
    e = ex.Engine(0)
    yield e
    e.close()


@pytest.fixture()
def device_alignments(eng, monkeypatch):
    served = []

    def align(model, qq, tt, threshold):
        a = eng.find_path(model, [(qq, tt)], dpmemory=32, threshold=threshold)[0]
        assert a is not None, "the device found no alignment at the reference's score"
        served.append(a.score)
        return a
    monkeypatch.setattr(tp, "_align", align)
    yield served
    assert served, "no alignment came from the device"


@pytest.mark.parametrize("model_type", ["est2genome", "protein2genome", "affine:local", "protein2dna", "affine:local:protein"])
@pytest.mark.parametrize("flip", [False, True])
def test_gff_dump_of_device_alignments(tmp_path, device_alignments, model_type, flip):
    tp.test_gff_dump_is_the_reference_s(tmp_path, model_type, flip)


@pytest.mark.parametrize("model_type", ["est2genome", "protein2genome", "affine:local", "protein2dna", "affine:local:protein"])
@pytest.mark.parametrize("flip", [False, True])
def test_alignment_display_of_device_alignments(tmp_path, device_alignments, model_type, flip):
    tp.test_alignment_display_is_the_reference_s(tmp_path, model_type, flip, 80)


@pytest.mark.parametrize("model_type", ["est2genome", "protein2genome", "protein2dna", "affine:local", "affine:global"])
def test_printers_on_random_cases_with_device_alignments(tmp_path, device_alignments, model_type):
    tp.test_printers_on_random_cases(tmp_path, model_type)


@pytest.mark.parametrize("model_type", ["est2genome", "protein2genome", "protein2dna", "affine:local", "affine:local:protein"])
def test_ryo_of_device_alignments(tmp_path, device_alignments, model_type):
    tp.test_ryo_is_the_reference_s(tmp_path, model_type)
