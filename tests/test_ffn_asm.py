"""The generated code of the panel kernel (veto_amd/csrc/ffn_fused.hip) and of the fused QKV + attention kernel
(veto_amd/csrc/qkv_attn_fused.hip) is part of their correctness: their GEMM MFMAs are inline asm, so the
compiler pads no MFMA hazard and counts none of their LDS-DMA.  This test compiles the file for gfx950 (no GPU needed) and runs the
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
    assert sorted(st) == [0, 1, 2, 2001, 2100, 2110]      # (2100 / 2110: the layer tail on 3-byte residual rows in / in and out; 2001: its single-pass form)
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


def test_generated_code_of_the_fused_qkv_attention_kernel_passes_its_audit(tmp_path):
    src = "qkv_attn_fused.hip"
    path = asmcheck.compile_asm(str(tmp_path), source=src)
    assert asmcheck.hazards(path) == [] and asmcheck.unpadded(path) == [] and asmcheck.m0_users(path) == []
    st = asmcheck.stats(path, asmcheck.KERNELS[src][0])
    assert sorted(st) == [72, 96, 721, 961]      # (721 / 961: the single-pass forms)
    k = st[72]      # the eight-head form (the default path): no scratch at all -- a reload inside the main loop drains the DMA queue
    assert k["scratch_ops"] == 0 and k["scratch_bytes"] == 0 and k["vgprs"] <= asmcheck.MAX_VGPRS and k["barriers"] >= 10, k
    assert st[96]["vgprs"] <= asmcheck.MAX_VGPRS
    # the six-head form is allowed the spill it has today and no more (a reload inside the stage loop drains the wave's DMA queue)
    for k96 in (96, 961):
        assert st[k96]["vgprs"] <= asmcheck.MAX_VGPRS and st[k96]["scratch_bytes"] <= 64 and st[k96]["scratch_ops"] <= 13, st[k96]
    assert st[721]["scratch_ops"] == 0 and st[721]["scratch_bytes"] == 0 and st[721]["vgprs"] <= asmcheck.MAX_VGPRS
    assert asmcheck.problems(path, source=src, scratch_ok=(96, 961)) == []
    bad = asmcheck.compile_asm(str(tmp_path), extra_flags=["-DQA_NO_PADS"], source=src)
    assert len(asmcheck.unpadded(bad)) > 100


def test_auditor_on_hand_written_snippets(tmp_path):
    """The auditor itself, on small assembly texts: what it must flag and what it must let through."""
    def write(name, body):
        p = tmp_path / name
        p.write_text(body)
        return str(p)

    mfma = "v_mfma_f32_16x16x32_f16 v[0:3], v[8:11], v[12:15], v[0:3]"
    ok = write("ok.s", "\n".join([";;#ASMSTART", "s_nop 1", mfma, ";;#ASMEND", "s_nop 15", "s_nop 15", "v_add_f32 v4, v0, v1", "s_endpgm"]))
    assert asmcheck.hazards(ok) == [] and asmcheck.unpadded(ok) == [] and asmcheck.m0_users(ok) == []
    # an accumulator read 1 state behind the MFMA that writes it
    early = write("early.s", "\n".join([";;#ASMSTART", "s_nop 1", mfma, ";;#ASMEND", "v_add_f32 v4, v0, v1", "s_endpgm"]))
    assert len(asmcheck.hazards(early)) == 1
    # a vector write of an MFMA operand right in front of an unpadded MFMA
    fresh = write("fresh.s", "\n".join(["v_mov_b32 v8, v20", ";;#ASMSTART", mfma, ";;#ASMEND", "s_endpgm"]))
    assert any("vector write of v8" in f for f in asmcheck.hazards(fresh)) and len(asmcheck.unpadded(fresh)) == 1
    # ... which the pad inside the statement cures
    padded = write("padded.s", "\n".join(["v_mov_b32 v8, v20", ";;#ASMSTART", "s_nop 1", mfma, ";;#ASMEND", "s_endpgm"]))
    assert asmcheck.hazards(padded) == [] and asmcheck.unpadded(padded) == []
    # a hazard across a loop back-edge: the MFMA at the bottom of the loop, the reader at its top
    loop = write("loop.s", "\n".join([".LBB0_1:", "v_add_f32 v4, v0, v1", "s_nop 15", "s_nop 15", ";;#ASMSTART", "s_nop 1", mfma, ";;#ASMEND",
                                       "s_cbranch_scc1 .LBB0_1", "s_endpgm"]))
    assert any("v_add_f32" in f for f in asmcheck.hazards(loop))
    # an inline-asm MFMA result as the A operand of the next MFMA (not the tied accumulator)
    opnd = write("opnd.s", "\n".join([";;#ASMSTART", "s_nop 1", mfma, ";;#ASMEND", ";;#ASMSTART", "s_nop 1",
                                       "v_mfma_f32_16x16x32_f16 v[16:19], v[0:3], v[12:15], v[16:19]", ";;#ASMEND", "s_endpgm"]))
    assert any("MFMA operand" in f for f in asmcheck.hazards(opnd))
    # an MFMA the compiler emitted itself gets its wait states from the compiler: not tracked
    own = write("own.s", "\n".join(["v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]", "v_add_f32 v40, v0, v1", "s_endpgm"]))
    assert asmcheck.hazards(own) == []
    # the accumulate chain (next MFMA takes the result whole as C) needs no wait; compiler code that touches M0 is reported
    chain = write("chain.s", "\n".join([";;#ASMSTART", "s_nop 1", mfma, ";;#ASMEND", ";;#ASMSTART", "s_nop 1", mfma, ";;#ASMEND", "s_mov_b32 m0, s4", "s_endpgm"]))
    assert asmcheck.hazards(chain) == [] and asmcheck.m0_users(chain) == ["s_mov_b32 m0, s4"]
