#!/usr/bin/env python3
"""Generate prime_rl_amd/csrc/attention_fwd_p3_asm.inc.h: the tile loop of the three-stage pipelined attention forward (attention_fwd_p3.hip.h)
as ONE hand-placed gfx950 instruction stream.

    python tools/gen_attn_p3_asm.py            # writes the .inc.h
    python tools/gen_attn_p3_asm.py --stats    # + per-block instruction mix of the product stream

Why a generator: the stream is ~2500 instructions over fixed physical registers; hipcc neither schedules nor pads the hazards inside an asm statement
(cdna_hip_programming.md 5.7), so every producer -> consumer distance that needs wait states is checked HERE, on the final instruction order, and padded
with s_nop where the placement alone does not provide it.  The checker knows the rules this stream can trip:

    MFMA D          -> any non-MFMA reader / writer, or an MFMA reading it as A / B          >= 12 wait states (8-pass XDL)
    VALU write      -> MFMA reading it (A, B or C)                                            >= 2
    VALU write      -> v_permlane*_swap reading / writing it                                  >= 2
    v_permlane*     -> VALU reading its results                                               >= 2
    v_exp (trans)   -> VALU reading its result                                                >= 2
    MFMA reads C    -> a non-MFMA overwrites that register (WAR)                               >= 6
    s_add m0        -> buffer_load ... lds                                                    >= 1
    ds_read result  -> first reader: a counted s_waitcnt lgkmcnt must have retired it (verified by simulating the in-order LDS return queue)

Register plan (VGPRs named literally are listed as clobbers of the asm statement; everything else is an operand hipcc allocates below v80):
    v[192:255]  O^T accumulators   [cb][nb] -> 192 + (cb * 8 + nb) * 4
    v[160:191]  score set A        [cb][nb] -> 160 + (cb * 4 + nb) * 4      (even iterations: S(i) lands here, odd: the softmax works here)
    v[128:159]  score set B
    v[104:127]  fragment ring: 6 slots x 4 registers (V^T fragments = two transpose reads, K fragments = one b128 read)
    v[80:103]   softmax state and temporaries

The interleaved block of an iteration (all three stages live, tile i - 1 unmasked):
    [reads of fragment groups 0-2] [lane maxima of S(i-1): 16 v_max3 + the test "did any row maximum of the wave move?"]
      -> SLOW stream (some maximum moved): cross-row reduction, new maxima, alpha, exp2, sums, l = l alpha + sum, pack     | each: 32 P.V MFMAs of tile
      -> FAST stream (none moved: alpha = 1 exactly): exp2 against the kept maxima, sums, l = l + sum, pack               | i-2, then 32 QK^T MFMAs of
    per group of two fragments: [counted lgkmcnt] [4 MFMAs, VALU fillers after each] [reads of group + 3]; the 8 LDS-DMA    | tile i, fillers between
    instructions of K(i+1) / V(i-1) ride behind the first MFMAs instead of in a burst after the barrier.
Both streams produce the bits of fwd_softmax<15> (attention_kernels.hip.h): when no maximum moves, m_new = m_old, alpha = exp2(0) = 1 and
fma(l, 1, sum) = l + sum exactly.
"""
import sys

O_BASE, XA_BASE, XB_BASE, FR_BASE, T_BASE = 192, 160, 128, 104, 80
RING = 6
KOFF0, VOFF0, SLOT = 0, 32768, 16384
TNAMES = ["mx", "t", "muse", "m", "l", "al", "s0", "s1", "s2", "s3"]


def T(name, cb):
    return T_BASE + cb * len(TNAMES) + TNAMES.index(name)


V_NINF = T_BASE + 2 * len(TNAMES)  # holds -inf for the mask selects (a literal and vcc together exceed the constant bus)
NEG_FLT_MAX = "0xff7fffff"
NEG_INF = "0xff800000"

# physical scalar registers owned by the stream
S_I, S_KBASE, S_VBASE, S_KVSM, S_TMP, S_LIMIT, S_RS, S_KDMA, S_PAIR = 80, 81, 82, 83, 84, 85, 86, 87, 88  # S_PAIR: s[88:89]
S_LAST = 89


def v(n):
    return f"v{n}"


def vt(n, w):
    return f"v[{n}:{n + w - 1}]"


def s(n):
    return f"s{n}"


class Ins:
    __slots__ = ("text", "kind", "reads", "writes", "creads", "wait")

    def __init__(self, text, kind, reads=(), writes=(), creads=(), wait=None):
        self.text, self.kind = text, kind
        self.reads, self.writes, self.creads = set(reads), set(writes), set(creads)  # creads: registers an MFMA reads as C (same tuple as D: no wait)
        self.wait = wait


def R(base, n):
    return range(base, base + n)


# ---- instruction constructors -------------------------------------------------------------------------------------------------------------
def mfma(d, a, b, c):
    """d, a, b: register bases (4-tuples) or operand strings for a / b; c: base or None (C = 0)."""
    astr, areads = (a, ()) if isinstance(a, str) else (vt(a, 4), R(a, 4))
    bstr, breads = (b, ()) if isinstance(b, str) else (vt(b, 4), R(b, 4))
    cstr = "0" if c is None else vt(c, 4)
    creads = () if c is None else R(c, 4)
    return Ins(f"v_mfma_f32_16x16x32_bf16 {vt(d, 4)}, {astr}, {bstr}, {cstr}", "mfma", reads=list(areads) + list(breads), writes=R(d, 4), creads=creads)


def valu(text, reads, writes, kind="valu"):
    return Ins(text, kind, reads=reads, writes=writes)


def ds_tr(dst, addr_op, off):
    return Ins(f"ds_read_b64_tr_b16 {vt(dst, 2)}, {addr_op} offset:{off}", "ds", writes=R(dst, 2))


def ds_b128(dst, addr_op, off):
    return Ins(f"ds_read_b128 {vt(dst, 4)}, {addr_op} offset:{off}", "ds", writes=R(dst, 4))


def wait_lgkm(n):
    return Ins(f"s_waitcnt lgkmcnt({n})", "wait", wait=n)


def salu(text):
    return Ins(text, "salu")


def nop(n):
    return Ins(f"s_nop {n - 1}", "nop")


MARK = Ins("", "marker")


# ---- the pieces ------------------------------------------------------------------------------------------------------------------------------
def frag_slot(j):
    return FR_BASE + (j % RING) * 4


def issue_frag(j, par):
    """fragment j of an iteration: 0..15 = V^T fragments (kk2 = j // 8, nb = j % 8), 16..31 = K fragments (kk = (j - 16) // 4, nb = (j - 16) % 4)"""
    r = frag_slot(j)
    if j < 16:
        kk2, nb = divmod(j, 8)
        off = VOFF0 + par * SLOT + kk2 * 8192
        return [ds_tr(r, f"%[av{nb}]", off), ds_tr(r + 2, f"%[av{nb}]", off + 4096)]
    kk, nb = divmod(j - 16, 4)
    return [ds_b128(r, f"%[ak{kk}]", KOFF0 + par * SLOT + nb * 4096)]


def frag_mfmas(j, X):
    r = frag_slot(j)
    if j < 16:
        kk2, nb = divmod(j, 8)
        return [mfma(O_BASE + (cb * 8 + nb) * 4, r, X + (cb * 4 + kk2) * 4, O_BASE + (cb * 8 + nb) * 4) for cb in (0, 1)]
    kk, nb = divmod(j - 16, 4)
    return [mfma(X + (cb * 4 + nb) * 4, r, f"%[q{cb}{kk}]", None if kk == 0 else X + (cb * 4 + nb) * 4) for cb in (0, 1)]


def merge(a, b):
    res = []
    for i in range(max(len(a), len(b))):
        if i < len(a):
            res.append(a[i])
        if i < len(b):
            res.append(b[i])
    return res


def Yreg(Y, cb, nb, r):
    return Y + (cb * 4 + nb) * 4 + r


def lane_max(Y):
    """mx[cb] = sc2 * max over the lane's 16 raw scores of q block cb (the v_max3 chain of row_max16_xrow; scaling before the cross-row reduction gives
    the same bits as after it: rounding is monotone and sc2 > 0)"""
    per = []
    for cb in (0, 1):
        y = lambda nb, r: Yreg(Y, cb, nb, r)
        mx = T("mx", cb)
        L = [valu(f"v_max3_f32 {v(mx)}, {v(y(0, 0))}, {v(y(0, 1))}, {v(y(0, 2))}", [y(0, 0), y(0, 1), y(0, 2)], [mx]),
             valu(f"v_max_f32 {v(mx)}, {v(mx)}, {v(y(0, 3))}", [mx, y(0, 3)], [mx])]
        for nb in (1, 2, 3):
            L.append(valu(f"v_max3_f32 {v(mx)}, {v(mx)}, {v(y(nb, 0))}, {v(y(nb, 1))}", [mx, y(nb, 0), y(nb, 1)], [mx]))
            L.append(valu(f"v_max3_f32 {v(mx)}, {v(mx)}, {v(y(nb, 2))}, {v(y(nb, 3))}", [mx, y(nb, 2), y(nb, 3)], [mx]))
        L.append(valu(f"v_mul_f32 {v(mx)}, %[sc2], {v(mx)}", [mx], [mx]))
        per.append(L)
    return merge(*per)


def moved_test(target):
    """wave-uniform: did any lane's scaled maximum exceed the running maximum of its row?  No -> branch to `target`."""
    return [valu(f"v_cmp_gt_f32 vcc, {v(T('mx', 0))}, {v(T('m', 0))}", [T("mx", 0), T("m", 0)], []),
            valu(f"v_cmp_gt_f32 s[{S_PAIR}:{S_PAIR + 1}], {v(T('mx', 1))}, {v(T('m', 1))}", [T("mx", 1), T("m", 1)], []),
            salu(f"s_or_b64 vcc, vcc, s[{S_PAIR}:{S_PAIR + 1}]"),
            salu(f"s_cbranch_vccz {target}")]


def xreduce(merged):
    """max over the four 16-lane rows of the wave, both q blocks.  merged: ONE swap sequence for both (the halves of the first v_permlane32_swap carry
    q block 0 and 1): 7 instructions instead of 12."""
    m0, m1, t0, t1 = T("mx", 0), T("mx", 1), T("t", 0), T("t", 1)
    if merged:
        return [valu(f"v_permlane32_swap_b32 {v(m0)}, {v(m1)}", [m0, m1], [m0, m1], kind="perm"),  # m0 = [x.r0 x.r1 y.r0 y.r1], m1 = [x.r2 x.r3 y.r2 y.r3]
                valu(f"v_max_f32 {v(m0)}, {v(m0)}, {v(m1)}", [m0, m1], [m0]),
                valu(f"v_mov_b32 {v(t0)}, {v(m0)}", [m0], [t0]),
                valu(f"v_permlane16_swap_b32 {v(m0)}, {v(t0)}", [m0, t0], [m0, t0], kind="perm"),
                valu(f"v_max_f32 {v(m0)}, {v(m0)}, {v(t0)}", [m0, t0], [m0]),                      # [M0 M0 M1 M1]
                valu(f"v_mov_b32 {v(m1)}, {v(m0)}", [m0], [m1]),
                valu(f"v_permlane32_swap_b32 {v(m0)}, {v(m1)}", [m0, m1], [m0, m1], kind="perm")]  # m0 = M0 everywhere, m1 = M1 everywhere
    per = []
    for mx, t in ((m0, t0), (m1, t1)):
        per.append([valu(f"v_mov_b32 {v(t)}, {v(mx)}", [mx], [t]),
                    valu(f"v_permlane32_swap_b32 {v(mx)}, {v(t)}", [mx, t], [mx, t], kind="perm"),
                    valu(f"v_max_f32 {v(mx)}, {v(mx)}, {v(t)}", [mx, t], [mx]),
                    valu(f"v_mov_b32 {v(t)}, {v(mx)}", [mx], [t]),
                    valu(f"v_permlane16_swap_b32 {v(mx)}, {v(t)}", [mx, t], [mx, t], kind="perm"),
                    valu(f"v_max_f32 {v(mx)}, {v(mx)}, {v(t)}", [mx, t], [mx])])
    return merge(*per)


def newmax():
    per = []
    for cb in (0, 1):
        mx, t, muse, m, al = (T(n, cb) for n in ("mx", "t", "muse", "m", "al"))
        per.append([valu(f"v_max_f32 {v(mx)}, {v(m)}, {v(mx)}", [m, mx], [mx]),                       # m_new
                    valu(f"v_max_f32 {v(muse)}, {NEG_FLT_MAX}, {v(mx)}", [mx], [muse]),                 # the value the exponent subtracts (-inf -> -FLT_MAX)
                    valu(f"v_sub_f32 {v(t)}, {v(m)}, {v(muse)}", [m, muse], [t]),
                    valu(f"v_exp_f32 {v(al)}, {v(t)}", [t], [al], kind="trans"),                      # alpha = exp2(m_old - m_use)
                    valu(f"v_mov_b32 {v(m)}, {v(mx)}", [mx], [m])])
    return merge(*per)


def expsum(Y, slow):
    """p = exp2(s sc2 - m_use) in place, per-lane row sum into l, pack to bf16 in place (tuples 0 / 1 of the q block's four)"""
    per = []
    for cb in (0, 1):
        y = lambda nb, r: Yreg(Y, cb, nb, r)
        muse, l, al = (T(n, cb) for n in ("muse", "l", "al"))
        sm = [T(f"s{r}", cb) for r in range(4)]
        sc = [y(nb, r) for nb in range(4) for r in range(4)]
        fm = [valu(f"v_fma_f32 {v(x)}, %[sc2], {v(x)}, -{v(muse)}", [x, muse], [x]) for x in sc]
        ex = [valu(f"v_exp_f32 {v(x)}, {v(x)}", [x], [x], kind="trans") for x in sc]
        L = [fm[0], fm[1]]
        for k in range(16):  # the exp of element k trails its fma by two instructions of this list
            L.append(ex[k])
            if k + 2 < 16:
                L.append(fm[k + 2])
        # sum4 = p(nb 0); sum4 += p(nb 1..3) element-wise; l = l * alpha + ((s0 + s1) + (s2 + s3))
        for r in range(4):
            L.append(valu(f"v_add_f32 {v(sm[r])}, {v(y(0, r))}, {v(y(1, r))}", [y(0, r), y(1, r)], [sm[r]]))
        for nb in (2, 3):
            for r in range(4):
                L.append(valu(f"v_add_f32 {v(sm[r])}, {v(sm[r])}, {v(y(nb, r))}", [sm[r], y(nb, r)], [sm[r]]))
        L.append(valu(f"v_add_f32 {v(sm[0])}, {v(sm[0])}, {v(sm[1])}", [sm[0], sm[1]], [sm[0]]))
        L.append(valu(f"v_add_f32 {v(sm[2])}, {v(sm[2])}, {v(sm[3])}", [sm[2], sm[3]], [sm[2]]))
        L.append(valu(f"v_add_f32 {v(sm[0])}, {v(sm[0])}, {v(sm[2])}", [sm[0], sm[2]], [sm[0]]))
        if slow:
            L.append(valu(f"v_fma_f32 {v(l)}, {v(l)}, {v(al)}, {v(sm[0])}", [l, al, sm[0]], [l]))
        else:
            L.append(valu(f"v_add_f32 {v(l)}, {v(l)}, {v(sm[0])}", [l, sm[0]], [l]))
        # P(kk2) = {pk(c0[0], c0[1]), pk(c0[2], c0[3]), pk(c1[0], c1[1]), pk(c1[2], c1[3])}, c0 = block 2 kk2, c1 = block 2 kk2 + 1, into tuple kk2
        for kk2 in (0, 1):
            dst = Y + (cb * 4 + kk2) * 4
            src = [(y(2 * kk2, 0), y(2 * kk2, 1)), (y(2 * kk2, 2), y(2 * kk2, 3)), (y(2 * kk2 + 1, 0), y(2 * kk2 + 1, 1)), (y(2 * kk2 + 1, 2), y(2 * kk2 + 1, 3))]
            for jj, (a, b) in enumerate(src):
                L.append(valu(f"v_cvt_pk_bf16_f32 {v(dst + jj)}, {v(a)}, {v(b)}", [a, b], [dst + jj]))
        per.append(L)
    return merge(*per)


def mask_list(Y):
    """score (cb, nb, r) is visible iff  nb * 16 + r <= hi[cb]  and  nb * 16 + r >= lw[cb],  hi = qrel - kv_sm, lw = lrel - kv_sm  (qrel = q row - 4 lq,
    lrel = first visible key - 4 lq: the lane's 4 lq part of the kv index is folded into the bounds)"""
    L = []
    for cb in (0, 1):
        hi, lw = T("s0", cb), T("s1", cb)
        L.append(valu(f"v_subrev_u32 {v(hi)}, {s(S_KVSM)}, %[qrel{cb}]", [], [hi]))
        L.append(valu(f"v_subrev_u32 {v(lw)}, {s(S_KVSM)}, %[lrel{cb}]", [], [lw]))
    for cb in (0, 1):
        hi, lw = T("s0", cb), T("s1", cb)
        for nb in range(4):
            for r in range(4):
                x = Yreg(Y, cb, nb, r)
                c = nb * 16 + r
                L.append(valu(f"v_cmp_le_i32 vcc, {c}, {v(hi)}", [hi], []))
                L.append(valu(f"v_cndmask_b32 {v(x)}, {v(V_NINF)}, {v(x)}, vcc", [x, V_NINF], [x]))
                L.append(valu(f"v_cmp_ge_i32 vcc, {c}, {v(lw)}", [lw], []))
                L.append(valu(f"v_cndmask_b32 {v(x)}, {v(V_NINF)}, {v(x)}, vcc", [x, V_NINF], [x]))
    return L


def dma_ins(par, which, ksrc=S_KBASE):
    """the LDS-DMA of K(i + 1) ("k") or V(i - 1) ("v") into the slots of the other parity: [m0 write, 4 loads]"""
    if which == "k":
        return [salu(f"s_add_u32 m0, %[ldsw], {KOFF0 + (1 - par) * SLOT}")] + [
            Ins(f"buffer_load_dwordx4 %[ko{j}], %[krsrc], {s(ksrc)} offen" + (f" offset:{j * 1024}" if j else "") + " lds", "vmem") for j in range(4)]
    return [salu(f"s_add_u32 m0, %[ldsw], {VOFF0 + (1 - par) * SLOT}")] + [
        Ins(f"buffer_load_dwordx4 %[vo{j}], %[vrsrc], {s(S_VBASE)} offen" + (f" offset:{j * 1024}" if j else "") + " lds", "vmem") for j in range(4)]


def pipelined(frags, X, par, pre, fillers, extras=None, ablate=()):
    """[reads of 3 groups][pre][MARK] then per group of two fragments: [counted wait][4 MFMAs; after the k-th: extras[k], then its share of the fillers]
    [reads of group + 3]; leftover fillers at the end."""
    seq = []
    groups = [frags[i:i + 2] for i in range(0, len(frags), 2)]
    for g in groups[:3]:
        for j in g:
            seq += issue_frag(j, par)
    seq += list(pre)
    seq.append(MARK)
    fl = list(fillers)
    n_mfma = 2 * len(frags)
    base, extra = divmod(len(fl), n_mfma)
    k = 0
    for gi, g in enumerate(groups):
        later = sum(len(issue_frag(j, par)) for gg in groups[gi + 1:gi + 3] for j in gg)
        seq.append(wait_lgkm(later))
        ms = [m for j in g for m in frag_mfmas(j, X)]
        for mi, m in enumerate(ms):
            seq.append(m)
            if mi == len(ms) - 1 and gi + 3 < len(groups):
                for j in groups[gi + 3]:
                    seq += issue_frag(j, par)
            if extras and k in extras:
                seq += extras[k]
            n = base + (1 if k < extra else 0)
            seq += fl[:n]
            fl = fl[n:]
            k += 1
    seq += fl
    if "nolds" in ablate:
        seq = [i for i in seq if i.kind not in ("ds", "wait")]
    if "nomfma" in ablate:
        seq = [i for i in seq if i.kind != "mfma"]
    return seq


# ---- hazard padding + verification ------------------------------------------------------------------------------------------------------------
def pad_hazards(seq, tail_mfma_pad=True):
    """Insert s_nop so that every rule of the module docstring holds; simulate the LDS return queue.  Returns the padded list."""
    out = []
    hist = []  # (kind, reads, writes, creads, wait states, text) of emitted instructions, most recent last

    def states_since(pred, limit):
        d = 0
        for h in reversed(hist):
            if pred(h):
                return d
            d += h[4]
            if d >= limit:
                return limit
        return limit

    pending = []  # outstanding ds ops: the registers each writes, in issue order
    for ins in seq:
        if ins.kind == "marker":
            out.append(ins)
            continue
        need = 0
        rw = ins.reads | ins.writes | ins.creads
        if ins.kind == "mfma":
            ab = ins.reads
            need = max(need, 12 - states_since(lambda h: h[0] == "mfma" and (h[2] & ab), 12))
            need = max(need, 2 - states_since(lambda h: h[0] in ("valu", "trans", "perm") and (h[2] & (ab | ins.creads)), 2))
        elif ins.kind in ("valu", "trans", "perm", "ds"):
            need = max(need, 12 - states_since(lambda h: h[0] == "mfma" and (h[2] & rw), 12))
            need = max(need, 6 - states_since(lambda h: h[0] == "mfma" and (h[3] & ins.writes), 6))
            if ins.kind == "perm":
                need = max(need, 2 - states_since(lambda h: h[0] in ("valu", "trans", "perm") and (h[2] & rw), 2))
            if ins.kind in ("valu", "trans", "perm"):
                need = max(need, 2 - states_since(lambda h: h[0] == "perm" and (h[2] & ins.reads), 2))
                need = max(need, 2 - states_since(lambda h: h[0] == "trans" and (h[2] & ins.reads), 2))
        elif ins.kind == "vmem":
            need = max(need, 1 - states_since(lambda h: h[0] == "salu" and " m0," in h[5], 1))
        if need > 0:
            out.append(nop(need))
            hist.append(("nop", set(), set(), set(), need, ""))
        if ins.kind == "wait":
            while len(pending) > ins.wait:
                pending.pop(0)
        else:
            touched = ins.reads | ins.writes | ins.creads
            for p in pending:
                if p & touched and ins.kind != "ds":
                    raise SystemExit(f"LDS result used before its wait: {ins.text}")
                if p & ins.writes and ins.kind == "ds":
                    raise SystemExit(f"LDS read into a register with a pending read: {ins.text}")
        if ins.kind == "ds":
            pending.append(set(ins.writes))
            if len(pending) > 15:
                raise SystemExit("more than 15 LDS operations in flight (lgkmcnt saturates)")
        out.append(ins)
        hist.append((ins.kind, ins.reads, ins.writes, ins.creads, 1, ins.text))
    if pending:
        raise SystemExit("block ends with LDS reads in flight")
    if tail_mfma_pad:  # a block must not hand MFMA results to the next block's first instructions unpadded
        d = 0
        for h in reversed(hist):
            if h[0] == "mfma":
                break
            d += h[4]
        if d < 12:
            out.append(nop(12 - d))
    return out


def split_mark(seq):
    i = next(k for k, x in enumerate(seq) if x.kind == "marker")
    return seq[:i], seq[i + 1:]


def stats(name, seq):
    c = {}
    for i in seq:
        if i.kind != "marker":
            c[i.kind] = c.get(i.kind, 0) + 1
    nops = sum(int(i.text.split()[1]) + 1 for i in seq if i.kind == "nop")
    print(f"  {name}: " + ", ".join(f"{k} {n}" for k, n in sorted(c.items())) + f"  (wait states in s_nop: {nops})", file=sys.stderr)


# ---- the whole loop ---------------------------------------------------------------------------------------------------------------------------
def build(opts, want_stats=False):
    ablate = opts.get("ablate", ())
    fast, merged, dma_in = opts.get("fast", True), opts.get("merged", True), opts.get("dma_in", True)
    lines = []

    def emit(x):
        lines.append(x.text if isinstance(x, Ins) else x)

    def emit_block(name, seq):
        if want_stats:
            stats(name, seq)
        for i in seq:
            if i.kind != "marker":
                emit(i)

    L = lambda n: f".Lp3_{n}_%="
    for r in range(O_BASE, O_BASE + 64):
        emit(f"v_mov_b32 {v(r)}, 0")
    for cb in (0, 1):
        emit(f"v_mov_b32 {v(T('m', cb))}, {NEG_INF}")
        emit(f"v_mov_b32 {v(T('muse', cb))}, {NEG_FLT_MAX}")
        emit(f"v_mov_b32 {v(T('l', cb))}, 0")
        emit(f"v_mov_b32 {v(T('al', cb))}, 1.0")
    emit(f"v_mov_b32 {v(V_NINF)}, {NEG_INF}")
    emit(f"s_mov_b32 {s(S_I)}, 0")
    emit(f"s_mov_b32 {s(S_RS)}, 0")
    emit(f"s_mov_b32 {s(S_KBASE)}, %[kbase]")
    emit(f"s_mov_b32 {s(S_VBASE)}, %[vbase]")
    emit(f"s_mov_b32 {s(S_KVSM)}, %[kvsm]")
    emit(f"s_add_u32 {s(S_LIMIT)}, %[nit], 2")
    for par in (0, 1):
        X, Y = (XA_BASE, XB_BASE) if par == 0 else (XB_BASE, XA_BASE)
        P = lambda n: L(f"{n}{par}")
        n_rs = [0]

        def rescale():  # O *= alpha(i - 2) unless no maximum moved in that softmax (flag) or alpha is 1 for the whole wave anyway
            k = n_rs[0]
            n_rs[0] += 1
            emit(f"s_cmp_eq_u32 {s(S_RS)}, 0")
            emit(f"s_cbranch_scc1 {L(f'nors{par}x{k}')}")
            for cb in (0, 1):
                al = T("al", cb)
                emit(f"v_cmp_eq_f32 vcc, 1.0, {v(al)}")
                emit("s_cmp_eq_u64 vcc, exec")
                emit(f"s_cbranch_scc1 {L(f'nors{par}x{k}c{cb}')}")
                for r in range(O_BASE + cb * 32, O_BASE + cb * 32 + 32):
                    emit(f"v_mul_f32 {v(r)}, {v(r)}, {v(al)}")
                emit("s_nop 1")
                emit(f"{L(f'nors{par}x{k}c{cb}')}:")
            emit(f"{L(f'nors{par}x{k}')}:")

        emit(f"{P('top')}:")
        if "nosync" not in ablate and "nowait" not in ablate:
            emit("s_waitcnt vmcnt(0)")
        if "nosync" not in ablate and "nobarrier" not in ablate:
            emit("s_barrier")
        emit(f"s_add_u32 {s(S_KBASE)}, {s(S_KBASE)}, %[krow64]")
        emit(f"s_add_u32 {s(S_VBASE)}, {s(S_VBASE)}, %[vrow64]")
        emit(f"s_add_u32 {s(S_KVSM)}, {s(S_KVSM)}, 64")
        # the interleaved block: 2 <= i < nt_w (all three stages live) and tile i - 1 needs no mask
        emit(f"s_cmp_lt_i32 {s(S_I)}, 2")
        emit(f"s_cbranch_scc1 {P('generic')}")
        emit(f"s_cmp_ge_i32 {s(S_I)}, %[ntw]")
        emit(f"s_cbranch_scc1 {P('generic')}")
        emit(f"s_add_u32 {s(S_TMP)}, {s(S_KVSM)}, 63")
        emit(f"s_cmp_gt_i32 {s(S_TMP)}, %[q0]")
        emit(f"s_cbranch_scc1 {P('generic')}")
        emit(f"s_cmp_lt_i32 {s(S_KVSM)}, %[lomax]")
        emit(f"s_cbranch_scc1 {P('generic')}")
        # the block always issues a K DMA: K(i + 1), or — for the last tile of the workgroup — K(i) once more (an address inside the tensor; that slot is not read again)
        emit(f"s_add_u32 {s(S_TMP)}, {s(S_I)}, 1")
        emit(f"s_sub_u32 {s(S_KDMA)}, {s(S_KBASE)}, %[krow64]")
        emit(f"s_cmp_lt_i32 {s(S_TMP)}, %[nit]")
        emit(f"s_cselect_b32 {s(S_KDMA)}, {s(S_KBASE)}, {s(S_KDMA)}")
        if not dma_in and "nosync" not in ablate:  # burst right after the barrier
            for x in pad_hazards(dma_ins(par, "k", S_KDMA) + dma_ins(par, "v"), tail_mfma_pad=False):
                emit(x)
        rescale()
        extras = None
        if dma_in and "nosync" not in ablate and "nodma" not in ablate:
            d = dma_ins(par, "k", S_KDMA) + dma_ins(par, "v")
            extras = {k: [d[k]] for k in range(len(d))}
        novalu = "novalu" in ablate
        pre = [] if novalu else lane_max(Y) + (moved_test(P("fast")) if fast else [])
        slow_f = [] if novalu else xreduce(merged) + newmax() + expsum(Y, True)
        slow = pad_hazards(pipelined(list(range(32)), X, par, pre, slow_f + [salu(f"s_mov_b32 {s(S_RS)}, 1")], extras, ablate))
        head, slow_rest = split_mark(slow)
        emit_block(f"full[{par}] head", head)
        emit_block(f"full[{par}] slow", slow_rest)
        emit(f"s_branch {P('end')}")
        if fast and not novalu:
            fst = pad_hazards(pipelined(list(range(32)), X, par, pre, expsum(Y, False) + [salu(f"s_mov_b32 {s(S_RS)}, 0")], extras, ablate))
            head2, fast_rest = split_mark(fst)
            assert [i.text for i in head2] == [i.text for i in head]
            emit(f"{P('fast')}:")
            emit_block(f"full[{par}] fast", fast_rest)
            emit(f"s_branch {P('end')}")
        # every other iteration (fill, drain, masked tile): the stages one after the other
        emit(f"{P('generic')}:")
        if "nosync" not in ablate:
            emit(f"s_add_u32 {s(S_TMP)}, {s(S_I)}, 1")
            emit(f"s_cmp_ge_i32 {s(S_TMP)}, %[nit]")
            emit(f"s_cbranch_scc1 {P('nok')}")
            for x in pad_hazards(dma_ins(par, "k"), tail_mfma_pad=False):
                emit(x)
            emit(f"{P('nok')}:")
            emit(f"s_cmp_lt_i32 {s(S_I)}, 1")
            emit(f"s_cbranch_scc1 {P('nov')}")
            emit(f"s_cmp_gt_i32 {s(S_I)}, %[nit]")
            emit(f"s_cbranch_scc1 {P('nov')}")
            for x in pad_hazards(dma_ins(par, "v"), tail_mfma_pad=False):
                emit(x)
            emit(f"{P('nov')}:")
        # do_pv: 2 <= i < nt_w + 2
        emit(f"s_cmp_lt_i32 {s(S_I)}, 2")
        emit(f"s_cbranch_scc1 {P('nopv')}")
        emit(f"s_sub_u32 {s(S_TMP)}, {s(S_I)}, 2")
        emit(f"s_cmp_ge_i32 {s(S_TMP)}, %[ntw]")
        emit(f"s_cbranch_scc1 {P('nopv')}")
        rescale()
        emit_block(f"pv[{par}]", pad_hazards(pipelined(list(range(16)), X, par, [], [])))
        emit(f"{P('nopv')}:")
        emit(f"s_cmp_ge_i32 {s(S_I)}, %[ntw]")
        emit(f"s_cbranch_scc1 {P('noqk')}")
        emit_block(f"qk[{par}]", pad_hazards(pipelined(list(range(16, 32)), X, par, [], [])))
        emit(f"{P('noqk')}:")
        # do_sm: 1 <= i < nt_w + 1
        emit(f"s_cmp_lt_i32 {s(S_I)}, 1")
        emit(f"s_cbranch_scc1 {P('end')}")
        emit(f"s_sub_u32 {s(S_TMP)}, {s(S_I)}, 1")
        emit(f"s_cmp_ge_i32 {s(S_TMP)}, %[ntw]")
        emit(f"s_cbranch_scc1 {P('end')}")
        emit(f"s_add_u32 {s(S_TMP)}, {s(S_KVSM)}, 63")
        emit(f"s_cmp_gt_i32 {s(S_TMP)}, %[q0]")
        emit(f"s_cbranch_scc1 {P('mask')}")
        emit(f"s_cmp_ge_i32 {s(S_KVSM)}, %[lomax]")
        emit(f"s_cbranch_scc1 {P('sm')}")
        emit(f"{P('mask')}:")
        emit_block(f"mask[{par}]", pad_hazards(mask_list(Y), tail_mfma_pad=False))
        emit(f"{P('sm')}:")
        emit_block(f"sm[{par}]", pad_hazards(lane_max(Y) + xreduce(merged) + newmax() + expsum(Y, True), tail_mfma_pad=False))
        emit(f"s_mov_b32 {s(S_RS)}, 1")
        emit(f"{P('end')}:")
        emit(f"s_add_u32 {s(S_I)}, {s(S_I)}, 1")
        emit(f"s_cmp_ge_i32 {s(S_I)}, {s(S_LIMIT)}")
        if par == 0:
            emit(f"s_cbranch_scc1 {L('done')}")
        else:
            emit(f"s_cbranch_scc0 {L('top0')}")
    emit(f"{L('done')}:")
    emit("s_nop 7")
    for i in range(64):
        emit(f"v_mov_b32 %[o{i}], {v(O_BASE + i)}")
    for cb in (0, 1):
        emit(f"v_mov_b32 %[m{cb}], {v(T('m', cb))}")
        emit(f"v_mov_b32 %[l{cb}], {v(T('l', cb))}")
    return lines


HEADER = '''// attention_fwd_p3_asm.inc.h — GENERATED by tools/gen_attn_p3_asm.py (do not edit; edit the generator).  The tile loop of attn_fwd_p3a_kernel as one
// hand-placed gfx950 instruction stream: fixed physical registers (v80-v255, s80-s89), counted lgkmcnt waits, hazard distances checked by the generator.
#pragma once
'''
SIG = ("const u32x4 (&qf)[2][4], const unsigned (&aK)[4], const unsigned (&aV)[8], const int (&kop)[4], const int (&vop)[4],\n"
       "                                            const int (&qrel)[2], const int (&lrel)[2], __amdgpu_buffer_rsrc_t krsrc, __amdgpu_buffer_rsrc_t vrsrc, int sc2_bits,\n"
       "                                            int n_iters, int nt_w, int kvsm0, int q0, int lo_max, int krow64, int vrow64, int kbase0, int vbase0, int ldsw,\n"
       "                                            float (&o)[64], float (&m)[2], float (&l)[2]")
DOC = """
// One wave's whole tile loop (n_iters + 2 iterations).  In: Q fragments, per-lane LDS read addresses, per-lane DMA source offsets (piece j pre-decremented
// by j KiB: the instruction offset moves the LDS and the memory address together), mask bounds; scalars as named.  Out: the 64 O^T accumulators of the lane
// (index (cb * 8 + nb) * 4 + r), the running base-2 maxima and the per-lane partial row sums.  V = 0 is the product stream; V >= 1 are A/B builds and
// timing ablations (the latter with WRONG results) of `make VARIANTS=1` libraries: see VARIANTS in the generator.
"""
# function p3_loop_asm<V>: 0 = the product; 1-4 = A/B builds of the placement options; 5-8 = timing ablations of the interleaved block (wrong results)
VARIANTS = [dict(), dict(fast=False, merged=False, dma_in=False), dict(fast=False), dict(dma_in=False), dict(merged=False),
            dict(ablate=("novalu",)), dict(ablate=("nolds",)), dict(ablate=("nomfma",)), dict(ablate=("nosync",)),
            dict(ablate=("nowait",)), dict(ablate=("nobarrier",)), dict(ablate=("nodma",))]


def render(vi, opts, want_stats=False):
    lines = build(opts, want_stats)
    body = "\\n\\t\"\n      \"".join(l.replace('"', '\\"') for l in lines)
    outs = [f'[o{i}] "=v"(o[{i}])' for i in range(64)] + [f'[m{c}] "=v"(m[{c}])' for c in (0, 1)] + [f'[l{c}] "=v"(l[{c}])' for c in (0, 1)]
    ins = ([f'[q{cb}{kk}] "v"(qf[{cb}][{kk}])' for cb in (0, 1) for kk in range(4)] + [f'[ak{k}] "v"(aK[{k}])' for k in range(4)] +
           [f'[av{k}] "v"(aV[{k}])' for k in range(8)] + [f'[ko{j}] "v"(kop[{j}])' for j in range(4)] + [f'[vo{j}] "v"(vop[{j}])' for j in range(4)] +
           [f'[qrel{c}] "v"(qrel[{c}])' for c in (0, 1)] + [f'[lrel{c}] "v"(lrel[{c}])' for c in (0, 1)] +
           ['[krsrc] "s"(krsrc)', '[vrsrc] "s"(vrsrc)', '[sc2] "s"(sc2_bits)', '[nit] "s"(n_iters)', '[ntw] "s"(nt_w)', '[kvsm] "s"(kvsm0)', '[q0] "s"(q0)',
            '[lomax] "s"(lo_max)', '[krow64] "s"(krow64)', '[vrow64] "s"(vrow64)', '[kbase] "s"(kbase0)', '[vbase] "s"(vbase0)', '[ldsw] "s"(ldsw)'])
    clob = [f'"v{r}"' for r in range(T_BASE, 256)] + [f'"s{r}"' for r in range(S_I, S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
    note = "the product stream" if not opts else str(opts)
    return (f"template <>\n__device__ __forceinline__ void p3_loop_asm<{vi}>({SIG}) {{  // {note}\n  asm volatile(\n      \"" + body + "\"\n      : " +
            ", ".join(outs) + "\n      : " + ", ".join(ins) + "\n      : " + ", ".join(clob) + ");\n}\n"), len(lines)


def main():
    """attention_fwd_p3_asm.inc.h: the declaration + the product stream (committed).  --variants: also attention_fwd_p3_asm_variants.inc.h with the A/B and
    ablation builds (generated by `make VARIANTS=1`, not committed).  --out DIR: write there instead of prime_rl_amd/csrc (tests)."""
    out_dir = __file__.rsplit("/", 2)[0] + "/prime_rl_amd/csrc"
    if "--out" in sys.argv:
        out_dir = sys.argv[sys.argv.index("--out") + 1]
    fn, n_lines = render(0, VARIANTS[0], "--stats" in sys.argv)
    src = HEADER + DOC + f"template <int V>\n__device__ __forceinline__ void p3_loop_asm({SIG});\n\n" + fn
    open(out_dir + "/attention_fwd_p3_asm.inc.h", "w").write(src)
    print(f"wrote {out_dir}/attention_fwd_p3_asm.inc.h: {n_lines} asm lines", file=sys.stderr)
    if "--variants" in sys.argv:
        head = ("// attention_fwd_p3_asm_variants.inc.h — GENERATED by tools/gen_attn_p3_asm.py --variants (`make VARIANTS=1`; not committed): the placement A/B builds and the\n"
                "// timing ablations (WRONG results) of p3_loop_asm, selected by pm_attn_debug_set_variant(0, 129 + V).\n#pragma once\n\n")
        open(out_dir + "/attention_fwd_p3_asm_variants.inc.h", "w").write(head + "\n".join(render(vi, o)[0] for vi, o in enumerate(VARIANTS) if vi > 0))
        print(f"wrote {out_dir}/attention_fwd_p3_asm_variants.inc.h: {len(VARIANTS) - 1} variants", file=sys.stderr)


if __name__ == "__main__":
    main()
