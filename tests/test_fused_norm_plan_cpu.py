"""CPU suite: the plan of the epilogue-fused gradient norm (GRPOEngine._fs_build) — which parameters get per-tile partials from their weight-gradient GEMM,
which ranges of the flat gradient buffer the segmented / streaming kernels cover — must account for EVERY parameter element exactly once, and the norm
assembled from those pieces (stand-in kernels) must equal the norm of the whole buffer."""

import dataclasses

import pytest
import torch


def _engine(tied: bool, bias: bool):
    import tests.cpu_standin_ops as standin
    from prime_rl_amd.trainer import engine as E

    E.ops = standin
    dims = dataclasses.replace(E.ModelDims.tiny(layers=3, vocab=1024), tie_word_embeddings=tied, attention_bias=bias, qk_norm=not bias)
    return E.GRPOEngine(dims, device="cpu", seed=1)


@pytest.mark.parametrize("tied,bias", [(False, False), (True, False), (False, True)])
def test_every_gradient_element_is_covered_exactly_once(tied, bias):
    eng = _engine(tied, bias)
    plan = eng._fs_build()
    cover = torch.zeros(eng.store.total, dtype=torch.int32)
    base = eng.store.main_grad.data_ptr()
    for name in plan["slots"]:
        g = eng.store.g[name]
        off = (g.data_ptr() - base) // 4
        cover[off:off + g.numel()] += 1
    for o, n in plan["big"]:
        cover[o:o + n] += 1
    for o, n in plan["segments"].tolist():
        assert 0 < n <= eng._FS_SEGMENT
        cover[o:o + n] += 1
    is_param = torch.zeros_like(cover)
    for b in eng.store.buckets:
        for pname, off, shape in b.params:
            n = 1
            for d in shape:
                n *= d
            is_param[off:off + n] = 1
    assert torch.equal(cover, is_param)  # parameters once, padding never
    # the matrices are exactly the 2-D weights that a weight-gradient GEMM writes last (not the embedding; not a tied head)
    want = {n for b in eng.store.buckets for n, _, s in b.params if len(s) == 2 and n != "model.embed_tokens.weight" and not (tied and n == "lm_head.weight")}
    assert set(plan["slots"]) == want and len(want) == 4 * 3 + (0 if tied else 1)
    slots = sorted(plan["slots"].values())
    assert slots[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(slots, slots[1:])) and slots[-1][0] + slots[-1][1] <= plan["tiles"].numel()


def test_norm_from_the_pieces_equals_the_norm_of_the_buffer():
    import tests.cpu_standin_ops as standin

    eng = _engine(False, False)
    torch.manual_seed(0)
    eng.store.main_grad.normal_()
    pad = torch.ones(eng.store.total, dtype=torch.bool)
    for b in eng.store.buckets:
        for pname, off, shape in b.params:
            n = 1
            for d in shape:
                n *= d
            pad[off:off + n] = False
    eng.store.main_grad[pad] = 0.0  # padding never holds gradient
    plan = eng._fs_build()
    eng._fs_plan = plan
    for name, (o, n) in plan["slots"].items():  # what the weight-gradient epilogues would have left: per-matrix sums in some slots
        plan["tiles"][o] = eng.store.g[name].float().pow(2).sum()
        eng._fs_covered.add(name)
    out = torch.zeros(1)
    eng._fs_norm(out)
    want = eng.store.main_grad.double().pow(2).sum().item()
    assert abs(out.item() - want) <= 1e-5 * want and not eng._fs_covered
    # a matrix whose last weight-gradient call did not run on the own core falls back to the streaming pass over its range
    eng._fs_covered.update(n for n in plan["slots"] if "o_proj" not in n)
    out2 = torch.zeros(1)
    eng._fs_norm(out2)
    assert abs(out2.item() - want) <= 1e-5 * want
    assert standin.gemm_sumsq_slots(256, 256) == 8
