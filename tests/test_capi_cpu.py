"""CPU tier: the C-ABI library loads, exports every symbol include/parasuite_hip.h declares, and
fails loudly (no CPU fallback) when asked to compute without a HIP device."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "parasuite_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", text)))


def test_exports_match_header():
    import capi
    lib = capi.lib()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(capi.EXPORTS) == names
    assert lib.ps_version().startswith(b"parasuite-hip")


def test_struct_layouts():
    import ctypes as C
    import capi
    assert capi.ALN_DTYPE.itemsize == 32 and capi.HIT_DTYPE.itemsize == 128
    assert C.sizeof(capi.IndexInfo) == 8 * 11 + 4 * 4 + 8 + 4 * 2
    assert C.sizeof(capi.KStats) == 64


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="checks the no-device behaviour")
def test_compute_fails_loudly_without_device(example, workdir):
    import capi
    with pytest.raises(capi.PsError, match="no HIP device"):
        capi.Ctx.build(example["fa"])
    with pytest.raises(capi.PsError):
        capi.ps_index(example["fa"])
    with pytest.raises(capi.PsError):
        capi.ps_map(1, "2", None, None, example["fa"], os.path.join(workdir, "x.fq"), os.path.join(workdir, "x.sam"))


@pytest.mark.skipif(not _no_gpu(), reason="checks the no-device behaviour")
def test_mapping_mirror_error_contract(example, workdir):
    """non-zero status from the library surfaces as ExternalCallErrorException carrying the command
    (Mapping.java:170-172,189-197)"""
    import __graft_entry__ as ge
    mod = ge.load_package()
    m = mod.mapping.PARAsuiteMapping()
    m.setErrorProfileFilename(os.path.join(workdir, "none.errorprofile"))
    m.setIndelProfileFilename(os.path.join(workdir, "none.indelprofile"))
    with pytest.raises(mod.mapping.ExternalCallErrorException) as ei:
        m.executeMapping(4, example["fa"], os.path.join(workdir, "x.fq"), os.path.join(workdir, "out"), 10, "-1")
    assert "bwa" in ei.value.getMappingCommand()


def test_mapq_filter_on_sam(tmp_path):
    import __graft_entry__ as ge
    mod = ge.load_package()
    src = tmp_path / "a.sam"
    src.write_text("@SQ\tSN:c\tLN:10\nr1\t0\tc\t1\t37\t5M\t*\t0\t0\tACGTA\tIIIII\nr2\t0\tc\t2\t0\t5M\t*\t0\t0\tACGTA\tIIIII\n"
                   "r3\t4\t*\t0\t0\t*\t*\t0\t0\tACGTA\tIIIII\n")
    mod.mapping.Mapping.filter_sam_mapq(str(src), str(tmp_path / "b.sam"), 10)
    assert [l.split("\t")[0] for l in open(tmp_path / "b.sam")] == ["@SQ", "r1"]
