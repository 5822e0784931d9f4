"""The hand-placed tile loops of the attention forward (csrc/attention_fwd_p3_asm.inc.h) and of dK/dV (csrc/attention_dkdv_p2_asm.inc.h) are GENERATED: the
committed files must be what tools/gen_attn_p3_asm.py / gen_attn_dkdv_asm.py write, and the generators' hazard padding / LDS-queue simulation must catch
what hipcc does not pad inside an asm statement."""
import importlib.util
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _gen():
    spec = importlib.util.spec_from_file_location("gen_attn_p3_asm", ROOT / "tools" / "gen_attn_p3_asm.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_stream_is_the_generators_output(tmp_path):
    subprocess.run([sys.executable, str(ROOT / "tools" / "gen_attn_p3_asm.py"), "--out", str(tmp_path)], check=True, capture_output=True)
    want = (tmp_path / "attention_fwd_p3_asm.inc.h").read_text()
    have = (ROOT / "prime_rl_amd" / "csrc" / "attention_fwd_p3_asm.inc.h").read_text()
    assert have == want, "attention_fwd_p3_asm.inc.h is stale: run python tools/gen_attn_p3_asm.py"


def test_hazard_padding_rules():
    g = _gen()
    acc, frag, x = 192, 104, 160
    m = g.mfma(acc, frag, x, acc)
    # MFMA D -> VALU reader: 12 wait states appear between them
    seq = g.pad_hazards([m, g.valu(f"v_mul_f32 v{acc}, v{acc}, v90", [acc, 90], [acc])], tail_mfma_pad=False)
    assert [i.kind for i in seq] == ["mfma", "nop", "valu"] and seq[1].text == "s_nop 11"
    # VALU write -> MFMA operand: 2 wait states
    seq = g.pad_hazards([g.valu(f"v_cvt_pk_bf16_f32 v{x}, v128, v129", [128, 129], [x]), m], tail_mfma_pad=False)
    assert [i.text for i in seq][1] == "s_nop 1"
    # accumulate chain on the same tuple: no padding
    seq = g.pad_hazards([m, g.mfma(acc, frag, x + 4, acc)], tail_mfma_pad=False)
    assert [i.kind for i in seq] == ["mfma", "mfma"]
    # permlane needs 2 states after the write of its operands and 2 before a reader of its results
    a, b = 80, 81
    seq = g.pad_hazards([g.valu(f"v_mov_b32 v{b}, v{a}", [a], [b]), g.valu(f"v_permlane32_swap_b32 v{a}, v{b}", [a, b], [a, b], kind="perm"),
                         g.valu(f"v_max_f32 v{a}, v{a}, v{b}", [a, b], [a])], tail_mfma_pad=False)
    assert [i.kind for i in seq] == ["valu", "nop", "perm", "nop", "valu"]
    # a block that ends on an MFMA hands its result to the next block padded
    assert g.pad_hazards([m])[-1].text == "s_nop 11"


def test_lds_queue_simulation_rejects_a_fragment_used_before_its_wait():
    g = _gen()
    rd = g.ds_b128(104, "%[ak0]", 0)
    rd2 = g.ds_b128(108, "%[ak0]", 4096)
    use = g.mfma(160, 104, "%[q00]", None)
    with pytest.raises(SystemExit, match="before its wait"):
        g.pad_hazards([rd, rd2, g.wait_lgkm(2), use])
    assert g.pad_hazards([rd, rd2, g.wait_lgkm(1), use, g.wait_lgkm(0)])  # the first read has returned at lgkmcnt(1)
    with pytest.raises(SystemExit, match="in flight"):
        g.pad_hazards([rd])


def test_every_variant_builds_and_streams_have_the_expected_mix():
    g = _gen()
    for opts in g.VARIANTS:
        lines = g.build(opts)
        assert lines[-1].startswith("v_mov_b32 %[l1]")
    lines = g.build({})
    n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
    # per parity: slow 64 + fast 64 + pv 32 + qk 32
    assert n_mfma == 2 * (64 + 64 + 32 + 32)


def _gen_dkdv():
    spec = importlib.util.spec_from_file_location("gen_attn_dkdv_asm", ROOT / "tools" / "gen_attn_dkdv_asm.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_dkdv_stream_is_the_generators_output(tmp_path):
    subprocess.run([sys.executable, str(ROOT / "tools" / "gen_attn_dkdv_asm.py"), "--out", str(tmp_path)], check=True, capture_output=True)
    want = (tmp_path / "attention_dkdv_p2_asm.inc.h").read_text()
    have = (ROOT / "prime_rl_amd" / "csrc" / "attention_dkdv_p2_asm.inc.h").read_text()
    assert have == want, "attention_dkdv_p2_asm.inc.h is stale: run python tools/gen_attn_dkdv_asm.py"


def test_dkdv_stream_mix_and_accumulator_operands():
    g = _gen_dkdv()
    for opts in g.VARIANTS:
        assert g.build(opts)[-1] == "s_waitcnt lgkmcnt(0)"
    lines = g.build({})
    mf = [l for l in lines if l.startswith("v_mfma")]
    # per copy of the body (2 score sets x 3 ring slots): the interleaved block and the sequential blocks each hold S / dP (64) and dV / dK (64)
    assert len(mf) == 6 * (64 + 64 + 64 + 64)
    assert len([l for l in g.build(dict(unroll6=False)) if l.startswith("v_mfma")]) == 2 * 256  # the rotating-slot form (A/B build)
    # S / dP products: D in the architectural file, B (the K / V fragment) in the accumulator file; dV / dK: D in the accumulator file, B = packed P / dS
    p1 = [l for l in mf if l.split()[1].startswith("v[")]
    p2 = [l for l in mf if l.split()[1].startswith("a[")]
    assert len(p1) == len(p2) == 768
    assert all(l.split(", ")[2].startswith("a[") for l in p1) and all(l.split(", ")[2].startswith("v[") for l in p2)
    # a K / V fragment or an accumulator tuple is never written by anything but its own MFMA chain / the prologue
    writers = [l for l in lines if (l.startswith("ds_read") or l.startswith("v_cvt") or l.startswith("v_mul")) and l.split()[1].startswith("a")]
    assert not writers
