"""CPU: the C-ABI library loads and exports every symbol include/reftr_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "reftr_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(rt_[a-z0-9_]+)\s*\(", src)))


def declared_structs():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"}\s*(rt_[a-z0-9_]+_(?:desc|job))\s*;", src)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from reftr_amd import hip
    return hip


def test_every_declared_symbol_is_exported_and_bound(built):
    hip = built
    L = hip.lib()
    decl = declared_functions()
    assert len(decl) >= 25
    assert sorted(hip.exported_symbols()) == decl, set(hip.exported_symbols()) ^ set(decl)
    for name in decl:
        assert getattr(L, name) is not None
    assert L.rt_abi_version() == hip.ABI_VERSION


def test_struct_layouts_match_header(built):
    """ctypes.sizeof of every binding struct equals the C compiler's sizeof for the header's struct."""
    import subprocess
    import tempfile
    hip = built
    names = declared_structs()
    binding = {"rt_conv_gemm_desc": hip.ConvGemmDesc, "rt_conv_wgrad_desc": hip.ConvWgradDesc,
               "rt_layernorm_desc": hip.LayerNormDesc, "rt_layernorm_bwd_desc": hip.LayerNormBwdDesc,
               "rt_groupnorm_desc": hip.GroupNormDesc, "rt_groupnorm_bwd_desc": hip.GroupNormBwdDesc,
               "rt_attn_desc": hip.AttnDesc, "rt_attn_bwd_desc": hip.AttnBwdDesc,
               "rt_mask_posenc_desc": hip.MaskPosencDesc, "rt_rows_add_desc": hip.RowsAddDesc,
               "rt_box_loss_desc": hip.BoxLossDesc, "rt_adamw_desc": hip.AdamWDesc,
               "rt_gn_nhwc_desc": hip.GnNhwcDesc, "rt_gn_nhwc_bwd_desc": hip.GnNhwcBwdDesc,
               "rt_upsample_add_desc": hip.UpsampleAddDesc, "rt_upsample_add_bwd_desc": hip.UpsampleAddBwdDesc,
               "rt_attn_map_desc": hip.AttnMapDesc, "rt_attn_map_bwd_desc": hip.AttnMapBwdDesc,
               "rt_seg_concat_desc": hip.SegConcatDesc, "rt_mask_loss_desc": hip.MaskLossDesc,
               "rt_small_wgrad_job": hip.SmallWgradJob, "rt_ln_pg_job": hip.LnPgJob,
               "rt_mask_post_desc": hip.MaskPostDesc, "rt_box_post_desc": hip.BoxPostDesc, "rt_cem_desc": hip.CemDesc,
               "rt_decoder_fwd_desc": hip.DecoderFwdDesc, "rt_decoder_bwd_desc": hip.DecoderBwdDesc,
               "rt_bottleneck_desc": hip.BottleneckDesc, "rt_qenc_fwd_desc": hip.QencFwdDesc,
               "rt_head_loss_desc": hip.HeadLossDesc, "rt_qenc_bwd_desc": hip.QencBwdDesc, "rt_finish_desc": hip.FinishDesc}
    assert sorted(binding) == names
    prog = '#include <stdio.h>\n#include "reftr_hip.h"\nint main(){' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c"); exe = os.path.join(td, "s")
        open(c, "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe], text=True)
    for line in out.strip().splitlines():
        n, sz = line.split()
        assert ctypes.sizeof(binding[n]) == int(sz), n


def test_library_missing_fails_loudly(monkeypatch):
    from reftr_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/libreftr_hip.so")
    with pytest.raises(RuntimeError, match="not built"):
        hip.lib()
