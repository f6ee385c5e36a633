"""CPU-side checks of the C-ABI library: it builds/loads, exports every entry point that
include/veto_amd.h declares, and rejects bad configurations before touching the GPU."""
import ctypes
import os
import re

import pytest

from veto_amd import native

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "veto_amd.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(veto_[a-z_]+)\s*\(", text)))


def test_header_and_binding_list_the_same_entry_points():
    assert _declared() == sorted(native.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = native.load_library()
    for name in _declared():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.veto_version()


def test_struct_sizes_match_the_header_layout():
    # 13 int32 fields; 4 ints + 3 ptrs + 2 ints + 5 ptrs; 2 ints + 4 ptrs
    assert ctypes.sizeof(native.VetoConfig) == 52
    assert ctypes.sizeof(native.VetoInputs) == 16 + 3 * 8 + 8 + 6 * 8
    assert ctypes.sizeof(native.VetoDebugOutputs) == 8 + 4 * 8
    assert ctypes.sizeof(native.VetoPostMeetArgs) == 6 * 4 + 11 * 8
    assert ctypes.sizeof(native.VetoPostVoteArgs) == 8 * 4 + 12 * 8


def test_postprocess_vote_rejects_bad_arguments_without_a_gpu():
    lib = native.load_library()
    a = native.VetoPostVoteArgs()
    assert lib.veto_postprocess_vote(None, ctypes.byref(a), ctypes.c_void_p(8), 0) == -1
    assert b"size mismatch" in lib.veto_last_error()
    a.struct_size = ctypes.sizeof(native.VetoPostVoteArgs)
    a.n_obj, a.n_pair, a.n_groups, a.n_rel_cls, a.n_obj_cls, a.voting = 3, 6, 5, 51, 151, 2
    assert lib.veto_postprocess_vote(None, ctypes.byref(a), ctypes.c_void_p(8), 0) == -1
    assert b"voting" in lib.veto_last_error()


def test_create_rejects_bad_configs_without_a_gpu():
    lib = native.load_library()
    h = ctypes.c_void_p()

    def cfg(**kw):
        base = dict(struct_size=ctypes.sizeof(native.VetoConfig), dim=576, layers=4, heads=8, patch=2, channels=256,
                    resolution=8, num_obj_cls=151, embed_dim=200, num_out=51, precision=0, device=0,
                    max_chunk_pairs=0)
        base.update(kw)
        return native.VetoConfig(**base)

    for bad, needle in ((dict(dim=512), b"576"), (dict(heads=7), b"NHEADS"), (dict(patch=4), b"PATCH_SIZE"),
                        (dict(struct_size=8), b"size mismatch"), (dict(layers=0), b"ENC_LAYERS"),
                        (dict(precision=5), b"precision")):
        rc = lib.veto_create(ctypes.byref(cfg(**bad)), ctypes.byref(h))
        assert rc == -1, bad
        assert needle in lib.veto_last_error(), (bad, lib.veto_last_error())
    assert lib.veto_workspace_bytes(None, 10, 90) == 0
    assert lib.veto_debug_gemm_workspace_bytes(256, 192, 32) == 2 * 256 * 32 * 2 + 2 * 192 * 32 * 2 + 256   # + the weight exponent of the mixed mode


def test_public_header_is_plain_c(tmp_path):
    """include/veto_amd.h is the drop-in boundary: it must compile as C99 (no C++-isms, no torch types)."""
    import os
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include "veto_amd.h"\nint main(void) { veto_config_t c; c.struct_size = (int32_t)sizeof(c); return c.struct_size == 0; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I", inc, str(src)], check=True)
