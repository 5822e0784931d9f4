"""GPU parity of the hand-written MFMA GEMM core (csrc/gemm.hip, `pm_gemm_bf16`) through the C ABI: the three operand layouts of the trainer step
(forward, dgrad on the native [out, in] weight, wgrad with f32 accumulation) against an f32 torch matmul of the same bf16 operands.

Reference ops: F.linear of trainer/models/layers/mlp.py:27-29 / attn.py:60-84 / lm_head.py:118-148 and their autograd gradients.  Tolerances: the
kernel accumulates exact bf16 x bf16 products in f32 like the reference's cuBLAS call; only the summation ORDER differs, so an f32 result is held
to 2e-5 x sqrt(K) relative to the operand scale and a bf16 result to one bf16 rounding step."""

import pytest
import torch

from prime_rl_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _variants_build() -> bool:
    """Is this a `make VARIANTS=1` library (measured-and-not-adopted A/B builds compiled in)?  Setting the schedule word needs no GPU."""
    from prime_rl_amd import _lib

    lib = _lib.load()
    ok = lib.pm_gemm_debug_set_variant(65536 + 7) == 0
    lib.pm_gemm_debug_set_variant(-1)  # (a VARIANTS=1 library keeps a process-wide override: clear it)
    return ok


# 3 / 7: the shipped K loop, one workgroup per tile / persistent grid.  Only in a VARIANTS=1 library: 65536 + .: the four-phase loop of rounds 3-4; 64 / 68: the
# one-wave-per-SIMD kernel (per tile / persistent); 128 / 132: with register staging; 256 / 260: its fine schedule
AB = _variants_build()
V_ALL = [3, 7] + ([65536 + 3, 65536 + 7, 64, 68, 128, 132, 256, 260] if AB else [])
V_SOME = [3, 7] + ([65536 + 7, 64, 68, 132, 260] if AB else [])
V_FEW = [3, 7] + ([65536 + 7, 68, 132, 260] if AB else [])


def _set_variant(v):
    """Select a schedule word of the core; the measured-and-not-adopted builds (four-phase loop, one-wave-per-SIMD kernel) exist only in a
    `make VARIANTS=1` library: skip on the product build."""
    from prime_rl_amd._lib import PrimeMI355XError

    try:
        ops.gemm_set_variant(v)
    except PrimeMI355XError as e:
        if "VARIANTS=1" in str(e):
            pytest.skip(f"variant {v}: only in a `make VARIANTS=1` build")
        raise


def _rand(rows, cols, scale=1.0, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(rows, cols, device=DEV, generator=g) * scale).to(torch.bfloat16)


def _check(out, ref, k, bf16_out):
    ref = ref.float()
    scale = ref.abs().max().item() + 1e-6
    err = (out.float() - ref).abs().max().item()
    tol = (2.0 ** -8 if bf16_out else 2e-6 * k ** 0.5) * scale
    assert err <= tol, f"max abs err {err:.3e} > {tol:.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("variant", V_ALL)
@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (512, 768, 256), (300, 520, 384), (1000, 256, 1024), (77, 1000, 128)])
def test_gemm_forward_layout(m, n, k, variant):
    _set_variant(variant)
    try:
        x, w = _rand(m, k, 1.0, 1), _rand(n, k, 0.05, 2)
        ref = x.float() @ w.float().t()
        _check(ops.gemm(x, w), ref, k, True)
        _check(ops.gemm(x, w, out_dtype=torch.float32), ref, k, False)
    finally:
        ops.gemm_set_variant(7)


def test_gemm_bias_and_strided_operands():
    m, n, k = 384, 640, 256
    x_full, w = _rand(m, 3 * k, 1.0, 3), _rand(n, k, 0.05, 4)
    x = x_full[:, k:2 * k]  # a column slice: row stride 3k (the v-slice of a fused qkv buffer is such a view)
    bias = _rand(1, n, 1.0, 5).reshape(n)
    ref = x.float() @ w.float().t() + bias.float()
    _check(ops.gemm(x, w, bias=bias), ref, k, True)
    out = torch.zeros(m, 2 * n, dtype=torch.bfloat16, device=DEV)
    ops.gemm(x, w, out=out[:, n:], bias=bias)  # strided output
    _check(out[:, n:], ref, k, True)
    assert torch.count_nonzero(out[:, :n]) == 0


@pytest.mark.parametrize("variant", V_SOME)
@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (512, 384, 256), (300, 520, 640), (77, 1024, 128)])
def test_gemm_dgrad_layout(m, n, k, variant):
    """dx[m, in] = dy[m, out] @ W[out, in]: the weight is read K-major (no transposed copy)."""
    _set_variant(variant)
    try:
        dy, w = _rand(m, k, 1.0, 6), _rand(k, n, 0.05, 7)  # w stored [out = K, in = N]
        ref = dy.float() @ w.float()
        _check(ops.gemm(dy, w.t()), ref, k, True)
    finally:
        ops.gemm_set_variant(7)


@pytest.mark.parametrize("variant", V_SOME)
@pytest.mark.parametrize("tokens,n_out,n_in", [(128, 256, 256), (1024, 512, 384), (1000, 264, 520), (77, 256, 128), (4096, 256, 256)])
def test_gemm_wgrad_layout(tokens, n_out, n_in, variant):
    """dw[out, in] (+)= dy[tok, out]^T @ x[tok, in], f32, any token count (rows past the last token read as zeros)."""
    _set_variant(variant)
    try:
        dy, x = _rand(tokens, n_out, 1.0, 8), _rand(tokens, n_in, 1.0, 9)
        ref = dy.float().t() @ x.float()
        g = torch.full((n_out, n_in), 7.0, dtype=torch.float32, device=DEV)
        ops.gemm(dy.t(), x.t(), out=g, accumulate=False)
        _check(g, ref, tokens, False)
        ops.gemm(dy.t(), x.t(), out=g, accumulate=True)
        _check(g, 2 * ref, tokens, False)
    finally:
        ops.gemm_set_variant(7)


@pytest.mark.parametrize("variant", V_FEW)
@pytest.mark.parametrize("tokens,n_out,n_in", [(256, 256, 256), (1024, 768, 512), (1000, 264, 520), (2048, 4096, 1024)])
def test_gemm_wgrad_tile_sumsq(tokens, n_out, n_in, variant):
    """The gradient-norm pass fused into the weight-gradient epilogue: the per-(tile, wave) partials add up to sum(dw^2) of the FINAL values (also
    after an accumulating call), bit-identically from run to run; slots of tiles that do not exist are not written."""
    _set_variant(variant)
    try:
        dy, x = _rand(tokens, n_out, 1.0, 12), _rand(tokens, n_in, 1.0, 13)
        g = torch.empty((n_out, n_in), dtype=torch.float32, device=DEV)
        slots = torch.full((ops.gemm_sumsq_slots(n_out, n_in),), -1.0, dtype=torch.float32, device=DEV)
        ops.gemm(dy.t(), x.t(), out=g, accumulate=False, tile_sumsq=slots)
        written = slots[slots >= 0]
        want = g.double().pow(2).sum().item()
        assert abs(written.double().sum().item() - want) <= 1e-5 * want
        n_tiles = ((n_out + 255) // 256) * ((n_in + 255) // 256)
        split = _lib_ws_bytes(n_out, n_in, tokens) > 0  # a split last round also resets the slots of its K-part units (zeros)
        assert written.numel() >= 8 * n_tiles if split else written.numel() == 8 * n_tiles
        ref = dy.float().t() @ x.float()
        _check(g, ref, tokens, False)  # the values themselves are unchanged by the extra output
        first = slots.clone()
        ops.gemm(dy.t(), x.t(), out=g, accumulate=False, tile_sumsq=slots)
        assert torch.equal(first, slots)
        ops.gemm(dy.t(), x.t(), out=g, accumulate=True, tile_sumsq=slots)  # g = 2 ref: the partials follow the stored values, not the product
        want2 = g.double().pow(2).sum().item()
        assert abs(slots[slots >= 0].double().sum().item() - want2) <= 1e-5 * want2
        total = torch.zeros(1, dtype=torch.float32, device=DEV)
        slots.clamp_(min=0)
        ops.sum_partials(slots, total)
        assert abs(total.item() - want2) <= 1e-5 * want2
    finally:
        ops.gemm_set_variant(7)


@pytest.mark.parametrize("variant", [3, 7])
@pytest.mark.parametrize("tokens,n_out,n_in", [(8192, 6144, 4096), (8192, 2048, 1024), (9000, 520, 264), (16384, 3072, 2048), (4096, 6144, 4096)])
def test_gemm_wgrad_split_k_of_the_last_round(tokens, n_out, n_in, variant):
    """Tile counts that leave at most half of the last round of 256 tiles occupied (24 x 16 = 384 tiles: the q|k|v weight gradient of Qwen3-8B; a single
    partial round for small products) run that round split along K with a workspace + fixed-order reduction: same values as the f32 reference, overwrite
    and accumulate, norm partials of the FINAL values, bit-identical from run to run and between the persistent and the per-tile grid."""
    from prime_rl_amd import _lib

    lib = _lib.load()
    dy, x = _rand(tokens, n_out, 1.0, 30), _rand(tokens, n_in, 1.0, 31)
    ref = dy.float().t() @ x.float()
    runs = []
    for v in (variant, 10 - variant):  # 3 <-> 7: the other grid form
        _set_variant(v)
        try:
            g = torch.full((n_out, n_in), 3.0, dtype=torch.float32, device=DEV)
            slots = torch.zeros(ops.gemm_sumsq_slots(n_out, n_in), dtype=torch.float32, device=DEV)
            ops.gemm(dy.t(), x.t(), out=g, accumulate=False, tile_sumsq=slots)
            _check(g, ref, tokens, False)
            want = g.double().pow(2).sum().item()
            assert abs(slots.double().sum().item() - want) <= 1e-5 * want
            first = g.clone()
            ops.gemm(dy.t(), x.t(), out=g, accumulate=True, tile_sumsq=slots)
            _check(g, 2 * ref, tokens, False)
            want2 = g.double().pow(2).sum().item()
            assert abs(slots.double().sum().item() - want2) <= 1e-5 * want2
            runs.append((first, g.clone()))
        finally:
            ops.gemm_set_variant(7)
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    if lib.pm_gemm_workspace_bytes(n_out, n_in, tokens) == 0:
        pytest.skip("this shape has a full (or mostly full) last round: no split, covered as a plain product")


def test_gemm_wgrad_with_a_k_major_span_beyond_2_gib():
    """The LM-head weight gradient's situation (16 384 tokens x 151 936 columns of dlogits = 5 GB): a K-major operand whose k * ld reaches past the 32-bit
    buffer offsets.  The contraction then runs as chained parts of every tile's accumulation inside ONE launch (descriptors re-based per part, accumulators
    kept in registers) — here on a 384-column window of a [8192, 151936] matrix (2.5 GB), overwrite and accumulate, with the norm partials."""
    tokens, ld, n_out, n_in = 8192, 151936, 384, 256
    g0 = torch.Generator(device=DEV).manual_seed(40)
    big = torch.empty(tokens, ld, dtype=torch.bfloat16, device=DEV)
    big[:, 4096:4096 + n_out] = torch.randn(tokens, n_out, device=DEV, generator=g0).to(torch.bfloat16)
    dy = big[:, 4096:4096 + n_out]  # row stride 151936: k * ld * 2 B = 2.49 GB
    x = _rand(tokens, n_in, 1.0, 41)
    ref = dy.float().t() @ x.float()
    g = torch.full((n_out, n_in), -5.0, dtype=torch.float32, device=DEV)
    slots = torch.zeros(ops.gemm_sumsq_slots(n_out, n_in), dtype=torch.float32, device=DEV)
    ops.gemm(dy.t(), x.t(), out=g, accumulate=False, tile_sumsq=slots)
    _check(g, ref, tokens, False)
    want = g.double().pow(2).sum().item()
    assert abs(slots.double().sum().item() - want) <= 1e-5 * want
    ops.gemm(dy.t(), x.t(), out=g, accumulate=True)
    _check(g, 2 * ref, tokens, False)
    del big


def test_sumsq_segments():
    base = torch.randn(300_000, device=DEV)
    segs = torch.tensor([[0, 4096], [4224, 128], [8192, 1], [10_000, 250_001], [299_999, 1]], dtype=torch.int64, device=DEV)
    part = torch.zeros(5, dtype=torch.float32, device=DEV)
    ops.sumsq_segments(base, segs, part)
    for i, (o, n) in enumerate(segs.tolist()):
        want = base[o:o + n].double().pow(2).sum().item()
        assert abs(part[i].item() - want) <= 1e-5 * max(want, 1e-6), (i, part[i].item(), want)


def _lib_ws_bytes(m, n, k):
    from prime_rl_amd import _lib

    return _lib.load().pm_gemm_workspace_bytes(m, n, k)


def test_gemm_is_deterministic_and_matches_itself_across_variants():
    x, w = _rand(512, 512, 1.0, 10), _rand(512, 512, 0.05, 11)
    outs = []
    from prime_rl_amd._lib import PrimeMI355XError

    for v in (3, 7, 65536 + 3, 65536 + 7, 64, 68, 128, 132, 256, 260):  # every K loop walks K in the same order per output element: same bits
        try:
            ops.gemm_set_variant(v)
        except PrimeMI355XError:
            continue  # a `make VARIANTS=1` build only
        outs.append(ops.gemm(x, w, out_dtype=torch.float32))
    ops.gemm_set_variant(7)
    assert len(outs) >= 2
    for o in outs[1:]:
        assert torch.equal(o, outs[0])  # same summation order in every schedule variant
    assert torch.equal(ops.gemm(x, w, out_dtype=torch.float32), outs[1])


def test_gemm_identity_with_asymmetric_operand():
    """A = I against an asymmetric W: catches a row <-> column swap of the C fragment map (guide §3)."""
    k = 256
    eye = torch.eye(k, device=DEV).to(torch.bfloat16)
    w = (torch.arange(512 * k, device=DEV).reshape(512, k) % 251).to(torch.bfloat16)
    assert torch.equal(ops.gemm(eye, w, out_dtype=torch.float32), w.float().t().contiguous())
    wk = w[:256].t().contiguous()  # stored [K, N]: the K-major path
    assert torch.equal(ops.gemm(eye, wk.t(), out_dtype=torch.float32), w[:256].float().t().contiguous())


@pytest.mark.parametrize("variant", V_SOME)
def test_gemm_many_tiles_per_workgroup(variant):
    """More output tiles than CUs: the persistent form (variant bit 2) walks several tiles per workgroup, staging the next tile's first K-tiles under
    the current tile's last ones; results must equal the one-workgroup-per-tile form bit for bit."""
    _set_variant(variant)
    try:
        m, n, k = 2048 + 77, 10240, 256  # 9 x 40 = 360 tiles
        x, w = _rand(m, k, 1.0, 21), _rand(n, k, 0.05, 22)
        out = ops.gemm(x, w)
        ref = x.float() @ w.float().t()
        _check(out, ref, k, True)
        dy, xx = _rand(1000, 4096, 1.0, 23), _rand(1000, 5120, 1.0, 24)  # wgrad: 16 x 20 = 320 tiles, ragged token count
        g = torch.zeros(4096, 5120, dtype=torch.float32, device=DEV)
        ops.gemm(dy.t(), xx.t(), out=g)
        _check(g, dy.float().t() @ xx.float(), 1000, False)
    finally:
        ops.gemm_set_variant(7)


@pytest.mark.parametrize("m,inter,k", [(256, 128, 128), (1000, 384, 256), (77, 256, 640), (2048, 1024, 512), (4096, 12288, 1024)])
def test_gemm_swiglu_epilogue_equals_gemm_then_swiglu(m, inter, k):
    """pm_gemm_swiglu_bf16 (mlp.py:27-29: act_fn(gate_proj(x)) * up_proj(x) in the epilogue of the fused gate|up product): gate_up must equal the plain
    product and act must equal pm_swiglu_fwd on it BIT FOR BIT (same accumulation order per element, same activation function on the rounded values);
    ragged row counts, strided outputs."""
    x, w = _rand(m, k, 1.0, 11), _rand(2 * inter, k, 0.05, 12)
    ref_gu = ops.gemm(x, w)
    ref_act = ops.swiglu_fwd(ref_gu)
    gu, act = ops.gemm_swiglu(x, w)
    assert torch.equal(gu, ref_gu)
    assert torch.equal(act, ref_act)
    _check(gu, x.float() @ w.float().t(), k, True)
    # strided outputs (row strides beyond the logical width), untouched padding
    gu2 = torch.zeros(m, 2 * inter + 64, dtype=torch.bfloat16, device=DEV)
    act2 = torch.zeros(m, inter + 8, dtype=torch.bfloat16, device=DEV)
    ops.gemm_swiglu(x, w, gate_up=gu2[:, :2 * inter], act=act2[:, :inter])
    assert torch.equal(gu2[:, :2 * inter], ref_gu) and torch.equal(act2[:, :inter], ref_act)
    assert torch.count_nonzero(gu2[:, 2 * inter:]) == 0 and torch.count_nonzero(act2[:, inter:]) == 0
    assert not ops.gemm_swiglu_supported(x, _rand(2 * inter + 2, k, 0.05, 13))  # inter must be a multiple of 128


@pytest.mark.parametrize("m,inter,k", [(256, 256, 128), (1000, 384, 256), (77, 1096, 640), (2048, 1024, 512), (4096, 12288, 1024), (300, 136, 128)])
def test_gemm_swiglu_bwd_epilogue_equals_gemm_then_swiglu_bwd(m, inter, k):
    """pm_gemm_swiglu_bwd_bf16 (the backward of mlp.py:27-29 in the epilogue of the down projection's row-gradient product): [d_gate | d_up] must equal
    pm_swiglu_bwd on the plain product BIT FOR BIT (the product is rounded to bf16 exactly as it would be stored, then the same function); ragged row and
    column counts, strided operands, in place over gate_up."""
    dy, w_down = _rand(m, k, 1.0, 21), _rand(k, inter, 0.05, 22)  # w_down as stored: [hidden, inter]
    gu = _rand(m, 2 * inter, 1.5, 23)
    ref = ops.swiglu_bwd(ops.gemm(dy, w_down.t()), gu)
    got = ops.gemm_swiglu_bwd(dy, w_down, gu)
    assert torch.equal(got, ref)
    g = gu[:, :inter].float()
    d_up = (dy.float() @ w_down.float()) * g * torch.sigmoid(g)  # f32 arithmetic on the same bf16 operands; the kernel rounds the product, then d_up
    assert ((got[:, inter:].float() - d_up).abs() <= 2.0 ** -7 * d_up.abs() + 2.0 ** -7 * d_up.abs().mean()).all()
    # strided gate_up / out (row strides beyond the logical width), untouched padding
    gu2 = torch.zeros(m, 2 * inter + 64, dtype=torch.bfloat16, device=DEV)
    gu2[:, :2 * inter] = gu
    out2 = torch.zeros(m, 2 * inter + 8, dtype=torch.bfloat16, device=DEV)
    ops.gemm_swiglu_bwd(dy, w_down, gu2[:, :2 * inter], out=out2[:, :2 * inter])
    assert torch.equal(out2[:, :2 * inter], ref) and torch.count_nonzero(out2[:, 2 * inter:]) == 0
    # in place: every element is read before it is overwritten
    inplace = gu.clone()
    assert ops.gemm_swiglu_bwd(dy, w_down, inplace, out=inplace) is inplace
    assert torch.equal(inplace, ref)
    assert not ops.gemm_swiglu_bwd_supported(dy, _rand(k, inter + 4, 0.05, 24))  # inter must be a multiple of 8

