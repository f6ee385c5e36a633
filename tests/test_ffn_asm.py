"""The generated code of the panel kernel (veto_amd/csrc/ffn_fused.hip) is part of its correctness: its MFMAs are inline asm, so the
compiler pads no MFMA hazard and counts none of its LDS-DMA.  This test compiles the file for gfx950 (no GPU needed) and runs the
audit of veto_amd/asmcheck.py on the assembly: no hazard, every inline-asm MFMA opened by its pad, no scratch traffic, <= 256
registers in all three modes, M0 untouched by compiler code.  A second build WITHOUT the pads is the negative control: the same
audit must reject it."""
import pytest

from veto_amd import asmcheck


@pytest.fixture(scope="module")
def shipped(tmp_path_factory):
    return asmcheck.compile_asm(str(tmp_path_factory.mktemp("ffn_asm")))


def test_generated_code_of_the_panel_kernel_passes_its_audit(shipped):
    assert asmcheck.hazards(shipped) == []
    assert asmcheck.unpadded(shipped) == []
    assert asmcheck.m0_users(shipped) == []
    st = asmcheck.stats(shipped)
    assert sorted(st) == [0, 1, 2]
    for mode, k in st.items():
        assert k["scratch_ops"] == 0 and k["scratch_bytes"] == 0, (mode, k)
        assert k["vgprs"] is not None and k["vgprs"] <= asmcheck.MAX_VGPRS, (mode, k)
        assert k["barriers"] > 30, (mode, k)          # (the stage stream is there: the parser looked at the right kernels)
    assert asmcheck.problems(shipped) == []


def test_audit_rejects_a_build_without_the_mfma_pads(tmp_path):
    bad = asmcheck.compile_asm(str(tmp_path), extra_flags=['-DFFN_MMA_NOP=""'])
    missing = asmcheck.unpadded(bad)
    assert len(missing) > 1000, len(missing)           # every MFMA statement of the three kernels
    assert len(asmcheck.problems(bad)) >= len(missing)   # (what check() -- and with it __graft_entry__.build() -- raises on)
