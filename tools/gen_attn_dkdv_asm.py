#!/usr/bin/env python3
"""Generate prime_rl_amd/csrc/attention_dkdv_p2_asm.inc.h: the tile loop of the dK/dV kernel with 32 kv rows per wave (attention_dkdv_p2.hip.h) as ONE
hand-placed gfx950 instruction stream.  Same tool chain as tools/gen_attn_p3_asm.py (instruction records, hazard padding, LDS-queue simulation are imported
from it); what differs is the kernel:

  * one wave per SIMD (512 registers): dK^T / dV^T of the wave's two 16-row kv blocks (128 registers) and its K / V fragments (64) live in the ACCUMULATOR
    file, the two score sets (S, dP of the tile being produced / of the tile being turned into P, dS: 2 x 64) in the architectural file;
  * every Q / dO fragment read from LDS feeds TWO MFMAs (the shipped kernel: one — it is bounded by the LDS, profiles/r06_attention.md);
  * two-stage software pipeline inside the wave: iteration j issues S / dP of (q tile, head) pair j (64 MFMAs) with the exp / multiply / pack VALU of pair
    j - 1 between them, then dV / dK of pair j - 1 (64 MFMAs); the Q / dO tiles stay three iterations in a 3-slot LDS ring (one workgroup per CU);
  * same arithmetic in the same order as attn_bwd_dkdv_kernel: the results are bit-identical.

    python tools/gen_attn_dkdv_asm.py [--stats] [--out DIR]
"""
import importlib.util
import os
import sys

_spec = importlib.util.spec_from_file_location("gen_attn_p3_asm", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gen_attn_p3_asm.py"))
B = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(B)
Ins, pad_hazards, wait_lgkm, salu, nop, MARK, split_mark, valu, stats = B.Ins, B.pad_hazards, B.wait_lgkm, B.salu, B.nop, B.MARK, B.split_mark, B.valu, B.stats

A0 = 1000                                   # accumulator register n is id A0 + n in the hazard sets
SETS = {0: (128, 160), 1: (192, 224)}       # score sets: (S base, dP base); tuple (b, nb) at base + (b * 4 + nb) * 4
FR = 104                                    # fragment ring: 3 groups x 8 registers
AQ, AT, AR = 80, 84, 92                     # this iteration's LDS read addresses: phase 1 [kk], phase 2 [n0], the statistics rows of pair j
TM = 93                                     # v93..v99: mask temporaries
AR2 = 100                                   # ... the statistics rows of pair j - 1 (its seq_start row feeds the mask)
DK, DV, KF, VF = 0, 64, 128, 160            # accumulator file: dK^T / dV^T [b][nb] -> (b * 8 + nb) * 4, K / V fragments [b][kk] -> (b * 4 + kk) * 4
SLOTB = 33792                               # one LDS slot: Q tile 16 KiB, dO tile 16 KiB, rows -lse/scale | -delta | seq_start (3 x 256 B) + pad
# scalar registers owned by the stream
(S_J, S_HEAD, S_BASE, S_ROFF, S_QN, S_Q1, S_Q2, S_ON, S_O1, S_O2, S_LOLD, S_LO2, S_TMP, S_TMP2, S_DB, S_DR, S_P0, S_P1) = range(70, 88)
S_LAST = 89  # s[88:89] = the compare pair
S_SC2 = 68   # s[68:69]: sc2 as the (even-aligned) scalar source of the packed multiplies (op_sel_hi broadcasts its low half)
PK_LAG = (1, 3)
PK = False   # packed f32 arithmetic (v_pk_mul_f32: two elements per instruction, the same IEEE multiply per element) in the fully visible pairs: 96 instead
             # of 128 VALU instructions per pair, bit-identical, and 1-2 % SLOWER at every producer-consumer distance tried (profiles/r06_attention.md): off


def v(n):
    return f"v{n}"


def s(n):
    return f"s{n}"


def vt(n, w):
    return f"v[{n}:{n + w - 1}]"


def at(n, w):
    return f"a[{n}:{n + w - 1}]"


def R(b, n):
    return range(b, b + n)


def mfma_va(d_is_a, d, a_ring, b_is_a, b, c_same=True):
    """D / C: tuple `d` in the accumulator (d_is_a) or the architectural file; A: ring registers; B: accumulator or architectural tuple."""
    dstr = at(d, 4) if d_is_a else vt(d, 4)
    did = R(A0 + d, 4) if d_is_a else R(d, 4)
    bstr = at(b, 4) if b_is_a else vt(b, 4)
    bid = list(R(A0 + b, 4) if b_is_a else R(b, 4))
    return Ins(f"v_mfma_f32_16x16x32_bf16 {dstr}, {vt(a_ring, 4)}, {bstr}, {dstr}", "mfma", reads=list(R(a_ring, 4)) + bid, writes=did, creads=did)


def ring(g):
    return FR + (g % 3) * 8


class Addr:
    """LDS read addresses of an iteration.  Rotating form: per-iteration registers v80.. (slot offsets added at the top of the iteration).  Unrolled form
    (6 copies of the body = 2 score sets x 3 ring slots): the slot is an immediate on the operand registers; slot 2 (beyond the 16-bit offset field) has its own
    pre-added operands."""

    def __init__(self, slot1=None, slot2=None):
        self.s1, self.s2 = slot1, slot2

    def _op(self, name, idx, slot, reg):
        if slot is None:
            return v(reg), 0
        return (f"%[{name}2_{idx}]" if slot == 2 else f"%[{name}{idx}]"), (0 if slot == 2 else slot * SLOTB)

    def q(self, kk):
        return self._op("aq", kk, self.s1, AQ + kk)

    def t(self, n0):
        return self._op("at", n0, self.s2, AT + n0)

    def r1(self):
        return self._op("arow", "", self.s1, AR)

    def r2(self):
        return self._op("arow", "", self.s2, AR2)


def p1_reads(g, par, A=Addr()):
    """Q and dO fragment of (k-step kk, q block nb); with the first k-step also the block's accumulator starts: S at -lse / scale, dP at -delta of its q rows
    (rows 0 / 1 of the slot's statistics block; both kv blocks of the wave start from the same values)"""
    kk, nb = divmod(g, 4)
    r = ring(g)
    out = []
    if kk == 0:
        sb, pb = SETS[par]
        for b in (0, 1):
            ro, ri = A.r1()
            out.append(Ins(f"ds_read_b128 {vt(sb + (b * 4 + nb) * 4, 4)}, {ro} offset:{ri + nb * 64}", "ds", writes=R(sb + (b * 4 + nb) * 4, 4)))
            out.append(Ins(f"ds_read_b128 {vt(pb + (b * 4 + nb) * 4, 4)}, {ro} offset:{ri + 256 + nb * 64}", "ds", writes=R(pb + (b * 4 + nb) * 4, 4)))
    qo, qi = A.q(kk)
    return out + [Ins(f"ds_read_b128 {vt(r, 4)}, {qo} offset:{qi + nb * 4096}", "ds", writes=R(r, 4)),
                  Ins(f"ds_read_b128 {vt(r + 4, 4)}, {qo} offset:{qi + 16384 + nb * 4096}", "ds", writes=R(r + 4, 4))]


def p1_mfmas(g, par):
    kk, nb = divmod(g, 4)
    r = ring(g)
    sb, pb = SETS[par]
    out = []
    for b in (0, 1):
        out.append(mfma_va(False, sb + (b * 4 + nb) * 4, r, True, KF + (b * 4 + kk) * 4))
        out.append(mfma_va(False, pb + (b * 4 + nb) * 4, r + 4, True, VF + (b * 4 + kk) * 4))
    return out


def p2_reads(g, A=Addr()):
    h, n0 = divmod(g - 16, 8)
    r = ring(g)
    to, ti = A.t(n0)
    o = ti + h * 8192
    return [Ins(f"ds_read_b64_tr_b16 {vt(r, 2)}, {to} offset:{16384 + o}", "ds", writes=R(r, 2)),
            Ins(f"ds_read_b64_tr_b16 {vt(r + 2, 2)}, {to} offset:{16384 + o + 4096}", "ds", writes=R(r + 2, 2)),
            Ins(f"ds_read_b64_tr_b16 {vt(r + 4, 2)}, {to} offset:{o}", "ds", writes=R(r + 4, 2)),
            Ins(f"ds_read_b64_tr_b16 {vt(r + 6, 2)}, {to} offset:{o + 4096}", "ds", writes=R(r + 6, 2))]


def p2_mfmas(g, par):  # par: the set that holds P / dS of the pair
    h, n0 = divmod(g - 16, 8)
    r = ring(g)
    sb, pb = SETS[par]
    out = []
    for b in (0, 1):
        out.append(mfma_va(True, DV + (b * 8 + n0) * 4, r, False, sb + (b * 4 + h) * 4))      # dV^T += dO^T . P
        out.append(mfma_va(True, DK + (b * 8 + n0) * 4, r + 4, False, pb + (b * 4 + h) * 4))  # dK^T += Q^T . dS
    return out


def pack_list(par):
    sb, pb = SETS[par]
    L = []
    for base in (sb, pb):
        for b in (0, 1):
            y = lambda nb, r: base + (b * 4 + nb) * 4 + r
            for h in (0, 1):
                dst = base + (b * 4 + h) * 4
                src = [(y(2 * h, 0), y(2 * h, 1)), (y(2 * h, 2), y(2 * h, 3)), (y(2 * h + 1, 0), y(2 * h + 1, 1)), (y(2 * h + 1, 2), y(2 * h + 1, 3))]
                for jj, (a, c) in enumerate(src):
                    L.append(valu(f"v_cvt_pk_bf16_f32 {v(dst + jj)}, {v(a)}, {v(c)}", [a, c], [dst + jj]))
    return L


def softmax_bwd_list(par):
    """fully visible pair: P = exp2(S sc2) (S started at -lse / scale), dS / scale = P * dP (dP started at -delta), both packed to bf16 in place"""
    sb, pb = SETS[par]
    sc = [(sb + k, pb + k) for k in range(32)]
    ex = [valu(f"v_exp_f32 {v(a)}, {v(a)}", [a], [a], kind="trans") for a, _ in sc]
    if PK:
        # element pairs (2 j, 2 j + 1) are even-aligned register pairs of the score sets: one v_pk_mul_f32 scales two scores (sc2 from s[68:69], low half
        # for both lanes), one multiplies two probabilities into their dP; the exponentials stay one per element (96 instead of 128 VALU issues per pair)
        ms = [valu(f"v_pk_mul_f32 {vt(sb + 2 * j, 2)}, {vt(sb + 2 * j, 2)}, s[{S_SC2}:{S_SC2 + 1}] op_sel_hi:[1,0]", [sb + 2 * j, sb + 2 * j + 1], [sb + 2 * j, sb + 2 * j + 1])
              for j in range(16)]
        md = [valu(f"v_pk_mul_f32 {vt(pb + 2 * j, 2)}, {vt(pb + 2 * j, 2)}, {vt(sb + 2 * j, 2)}", [pb + 2 * j, pb + 2 * j + 1, sb + 2 * j, sb + 2 * j + 1], [pb + 2 * j, pb + 2 * j + 1])
              for j in range(16)]
        L = []
        d1, d2 = PK_LAG
        for j in range(16 + d2):  # pair j is scaled, pair j - d1 exponentiated (two instructions), pair j - d2 multiplied into dP
            if j < 16:
                L.append(ms[j])
            if 0 <= j - d1 < 16:
                L += [ex[2 * (j - d1)], ex[2 * (j - d1) + 1]]
            if 0 <= j - d2 < 16:
                L.append(md[j - d2])
        return L + pack_list(par)
    ms = [valu(f"v_mul_f32 {v(a)}, %[sc2], {v(a)}", [a], [a]) for a, _ in sc]
    md = [valu(f"v_mul_f32 {v(d)}, {v(d)}, {v(a)}", [d, a], [d]) for a, d in sc]
    L = []
    for k in range(36):
        if k < 32:
            L.append(ms[k])
        if 0 <= k - 2 < 32:
            L.append(ex[k - 2])
        if 0 <= k - 4 < 32:
            L.append(md[k - 4])
    return L + pack_list(par)


def masked_list(par, A=Addr()):
    """a pair some score of which is hidden (diagonal, document boundary, rows past the end): score (b, nb, r) = (q row qi0 + nb 16 + 4 lq + r, kv row kvi + 16 b)
    counts iff  (unsigned)(q - kv) < n - kv  and  kv >= seq_start[q]  (dkdv_tile of attention_kernels.hip.h); P = 0 otherwise."""
    sb, pb = SETS[par]
    LO = FR  # the pair's seq_start row values: 16 registers of the (idle) fragment ring
    kvb_, nku, base = TM, TM + 2, TM + 4  # [b]
    tmp = TM + 6
    ro, ri = A.r2()
    L = [Ins(f"ds_read_b128 {vt(LO + nb * 4, 4)}, {ro} offset:{ri + 512 + nb * 64}", "ds", writes=R(LO + nb * 4, 4)) for nb in range(4)]
    for b in (0, 1):
        L.append(valu(f"v_add_u32 {v(kvb_ + b)}, {16 * b}, %[kvi]", [], [kvb_ + b]))
        L.append(valu(f"v_sub_u32 {v(nku + b)}, %[n], {v(kvb_ + b)}", [kvb_ + b], [nku + b]))
        L.append(valu(f"v_sub_u32 {v(base + b)}, {s(S_Q2)}, {v(kvb_ + b)}", [kvb_ + b], [base + b]))
        L.append(valu(f"v_add_u32 {v(base + b)}, {v(base + b)}, %[lq4]", [base + b], [base + b]))
    L.append(wait_lgkm(0))
    for b in (0, 1):
        for nb in range(4):
            for r in range(4):
                a, d = sb + (b * 4 + nb) * 4 + r, pb + (b * 4 + nb) * 4 + r
                L.append(valu(f"v_add_u32 {v(tmp)}, {nb * 16 + r}, {v(base + b)}", [base + b], [tmp]))
                L.append(valu(f"v_cmp_lt_u32 vcc, {v(tmp)}, {v(nku + b)}", [tmp, nku + b], []))
                L.append(valu(f"v_cmp_ge_i32 s[88:89], {v(kvb_ + b)}, {v(LO + nb * 4 + r)}", [kvb_ + b, LO + nb * 4 + r], []))
                L.append(salu("s_and_b64 vcc, vcc, s[88:89]"))
                L.append(valu(f"v_mul_f32 {v(a)}, %[sc2], {v(a)}", [a], [a]))
                L.append(valu(f"v_exp_f32 {v(a)}, {v(a)}", [a], [a], kind="trans"))
                L.append(valu(f"v_cndmask_b32 {v(a)}, 0, {v(a)}, vcc", [a], [a]))
                L.append(valu(f"v_mul_f32 {v(d)}, {v(d)}, {v(a)}", [d, a], [d]))
    return L + pack_list(par)


def dma_ins(slot=None):
    """LDS-DMA of the NEXT pair's Q and dO tiles (4 one-KiB pieces per wave each) and of its statistics row (wave 0: -lse / scale, 1: -delta, 2 and 3: seq_start)
    into slot S_ON (rotating form) / `slot` (unrolled form); source offsets S_DB / S_DR (the next pair's, or this pair's once more after the last)."""
    son = s(S_ON) if slot is None else str(slot * SLOTB)
    out = [salu(f"s_add_u32 m0, %[ldsw], {son}")]
    out += [Ins(f"buffer_load_dwordx4 %[vo{j}], %[qrsrc], {s(S_DB)} offen" + (f" offset:{j * 1024}" if j else "") + " lds", "vmem") for j in range(4)]
    out += [salu("s_add_u32 m0, m0, 16384")]
    out += [Ins(f"buffer_load_dwordx4 %[vo{j}], %[dorsrc], {s(S_DB)} offen" + (f" offset:{j * 1024}" if j else "") + " lds", "vmem") for j in range(4)]
    out += [salu(f"s_add_u32 m0, %[rowm0], {son}")]
    out += [Ins(f"buffer_load_dword %[lane4], %[rowrsrc], {s(S_DR)} offen lds", "vmem")]
    return out


def pipelined(groups, reads, mfmas, pre, fillers, extras=None):
    """[reads of 2 groups][pre][per group: counted wait, 4 MFMAs with fillers after each, reads of group + 2]; leftover fillers at the end"""
    seq = []
    for g in groups[:2]:
        seq += reads(g)
    seq += list(pre)
    fl = list(fillers)
    n_m = 4 * len(groups)
    base, extra = divmod(len(fl), n_m) if n_m else (0, 0)
    k = 0
    for gi, g in enumerate(groups):
        later = sum(len(reads(x)) for x in groups[gi + 1:gi + 2])
        seq.append(wait_lgkm(later))
        ms = mfmas(g)
        for mi, m in enumerate(ms):
            seq.append(m)
            if mi == len(ms) - 1 and gi + 2 < len(groups):
                seq += reads(groups[gi + 2])
            if extras and k in extras:
                seq += extras[k]
            nf = base + (1 if k < extra else 0)
            seq += fl[:nf]
            fl = fl[nf:]
            k += 1
    seq += fl
    return seq


def build(opts, want_stats=False):
    global PK, PK_LAG
    PK = opts.get("pk", False)
    PK_LAG = opts.get("pk_lag", (1, 3))
    lines = []
    abl = set(opts.get("ablate", ()))  # timing ablations (WRONG results): "nolds", "novalu", "nomfma", "nodma", "nobarrier"

    def emit(x):
        lines.append(x.text if isinstance(x, Ins) else x)

    def emit_block(name, seq):
        if "nolds" in abl:
            seq = [i for i in seq if i.kind not in ("ds", "wait")]
        if "novalu" in abl:
            seq = [i for i in seq if i.kind not in ("valu", "trans")]
        if "nomfma" in abl:
            seq = [i for i in seq if i.kind != "mfma"]
        if "nodma" in abl:
            seq = [i for i in seq if i.kind != "vmem"]
        if want_stats:
            stats(name, seq)
        for i in seq:
            if i.kind != "marker":
                emit(i)

    L = lambda n: f".Ldk_{n}_%="
    # ---- prologue: accumulators, K / V fragments, the first pair's tiles
    for r in range(128):
        emit(f"v_accvgpr_write_b32 a{r}, 0")
    for b in (0, 1):
        for kk in range(4):
            emit(f"global_load_dwordx4 {at(KF + (b * 4 + kk) * 4, 4)}, %[kp{b}], off offset:{kk * 64}")
            emit(f"global_load_dwordx4 {at(VF + (b * 4 + kk) * 4, 4)}, %[vp{b}], off offset:{kk * 64}")
    if PK:
        emit(f"v_readfirstlane_b32 {s(S_SC2)}, %[sc2v]")  # (the "s" operand may arrive in a vector register: hipcc does not always prove it uniform)
        emit(f"s_mov_b32 {s(S_SC2 + 1)}, {s(S_SC2)}")
    emit(f"s_mov_b32 {s(S_J)}, 0")
    emit(f"s_mov_b32 {s(S_HEAD)}, %[head0]")
    emit(f"s_mov_b32 {s(S_BASE)}, %[base0]")
    emit(f"s_mov_b32 {s(S_ROFF)}, %[roff0]")
    emit(f"s_mov_b32 {s(S_QN)}, %[qi00]")
    emit(f"s_mov_b32 {s(S_Q1)}, %[qi00]")
    emit(f"s_mov_b32 {s(S_Q2)}, %[qi00]")
    unroll6 = opts.get("unroll6", True)  # six copies of the body (2 score sets x 3 ring slots): slot offsets are immediates, no per-iteration address registers
    if not unroll6:
        emit(f"s_mov_b32 {s(S_O1)}, 0")
        emit(f"s_mov_b32 {s(S_ON)}, 0")  # the first pair lands in slot 0 ...
        emit(f"s_mov_b32 {s(S_O2)}, {2 * SLOTB}")
    emit(f"s_mov_b32 {s(S_DB)}, {s(S_BASE)}")
    emit(f"s_mov_b32 {s(S_DR)}, {s(S_ROFF)}")
    emit(f"s_mov_b32 {s(S_LO2)}, 0")
    emit(f"s_mov_b32 {s(S_LOLD)}, 0")
    emit(f"s_cmp_lt_i32 {s(S_J)}, %[total]")
    emit(f"s_cbranch_scc0 {L('done')}")
    for x in pad_hazards(dma_ins(0 if unroll6 else None), tail_mfma_pad=False):
        emit(x)
    if not unroll6:
        emit(f"s_mov_b32 {s(S_ON)}, {SLOTB}")  # ... the second in slot 1
    bodies = list(range(6)) if unroll6 else [0, 1]
    for u in bodies:
        par = u % 2
        A = Addr(u % 3, (u + 2) % 3) if unroll6 else Addr()
        slot_n = (u + 1) % 3 if unroll6 else None
        P = lambda nme: L(f"{nme}{u}")
        emit(f"{P('top')}:")
        emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
        if "nobarrier" not in abl:
            emit("s_barrier")
        # the statistic the mask decision of pair j - 1 needs came with the scalar load of the previous iteration
        emit(f"s_mov_b32 {s(S_LO2)}, {s(S_LOLD)}")
        # seq_start of the last q row of pair j (clamped to the tensor): consumed in the next iteration
        emit(f"s_add_u32 {s(S_TMP)}, {s(S_Q1)}, 63")
        emit(f"s_sub_u32 {s(S_TMP2)}, %[n], 1")
        emit(f"s_min_i32 {s(S_TMP)}, {s(S_TMP)}, {s(S_TMP2)}")
        emit(f"s_lshl_b32 {s(S_TMP)}, {s(S_TMP)}, 2")
        emit(f"s_load_dword {s(S_LOLD)}, %[seqptr], {s(S_TMP)}")
        # advance the issue cursor to pair j + 1 (heads inner, q tiles from the last one down); after the last pair: this pair's sources once more
        emit(f"s_add_u32 {s(S_TMP)}, {s(S_J)}, 1")
        emit(f"s_cmp_ge_i32 {s(S_TMP)}, %[total]")
        emit(f"s_cbranch_scc1 {P('nonext')}")
        emit(f"s_sub_u32 {s(S_HEAD)}, {s(S_HEAD)}, 1")
        emit(f"s_cmp_lt_i32 {s(S_HEAD)}, 0")
        emit(f"s_cbranch_scc1 {P('tilestep')}")
        emit(f"s_sub_u32 {s(S_BASE)}, {s(S_BASE)}, 256")
        emit(f"s_add_u32 {s(S_ROFF)}, {s(S_ROFF)}, %[rowdh]")
        emit(f"s_branch {P('nonext')}")
        emit(f"{P('tilestep')}:")
        emit(f"s_mov_b32 {s(S_HEAD)}, %[head0]")
        emit(f"s_add_u32 {s(S_BASE)}, {s(S_BASE)}, %[qdt]")
        emit(f"s_add_u32 {s(S_ROFF)}, {s(S_ROFF)}, %[rowdt]")
        emit(f"s_sub_u32 {s(S_QN)}, {s(S_QN)}, 64")
        emit(f"{P('nonext')}:")
        emit(f"s_mov_b32 {s(S_DB)}, {s(S_BASE)}")
        emit(f"s_mov_b32 {s(S_DR)}, {s(S_ROFF)}")
        if not unroll6:  # this iteration's LDS read addresses
            for kk in range(4):
                emit(f"v_add_u32 {v(AQ + kk)}, {s(S_O1)}, %[aq{kk}]")
            emit(f"v_add_u32 {v(AR)}, {s(S_O1)}, %[arow]")
            emit(f"v_add_u32 {v(AR2)}, {s(S_O2)}, %[arow]")
            for n0 in range(8):
                emit(f"v_add_u32 {v(AT + n0)}, {s(S_O2)}, %[at{n0}]")
        # the interleaved block: pair j and pair j - 1 both computed by this wave, pair j - 1 fully visible
        emit(f"s_cmp_lt_i32 {s(S_J)}, 1")
        emit(f"s_cbranch_scc1 {P('generic')}")
        emit(f"s_cmp_ge_i32 {s(S_J)}, %[ntw]")
        emit(f"s_cbranch_scc1 {P('generic')}")
        emit(f"s_add_u32 {s(S_TMP)}, %[kvb], 31")
        emit(f"s_cmp_lt_i32 {s(S_Q2)}, {s(S_TMP)}")
        emit(f"s_cbranch_scc1 {P('generic')}")
        emit(f"s_add_u32 {s(S_TMP)}, {s(S_Q2)}, 64")
        emit(f"s_cmp_gt_i32 {s(S_TMP)}, %[n]")
        emit(f"s_cbranch_scc1 {P('generic')}")
        emit(f"s_cmp_gt_i32 {s(S_LO2)}, %[kvb]")
        emit(f"s_cbranch_scc1 {P('generic')}")
        d = dma_ins(slot_n)
        extras = {k: [d[k]] for k in range(len(d))}
        p1 = pipelined(list(range(16)), lambda g: p1_reads(g, par, A), lambda g: p1_mfmas(g, par), [], softmax_bwd_list(par ^ 1), extras)
        p2 = pipelined(list(range(16, 32)), lambda g: p2_reads(g, A), lambda g: p2_mfmas(g, par ^ 1), [], [])
        emit_block(f"full[{u}] S/dP + softmax-bwd", pad_hazards(p1, tail_mfma_pad=False))
        emit_block(f"full[{u}] dV/dK", pad_hazards(p2, tail_mfma_pad=False))
        emit(f"s_branch {P('end')}")
        emit(f"{P('generic')}:")
        emit(f"s_add_u32 {s(S_TMP)}, {s(S_J)}, 1")
        emit(f"s_cmp_ge_i32 {s(S_TMP)}, %[total]")
        emit(f"s_cbranch_scc1 {P('nodma')}")
        if "nodma" not in abl:
            for x in pad_hazards(dma_ins(slot_n), tail_mfma_pad=False):
                emit(x)
        emit(f"{P('nodma')}:")
        emit(f"s_cmp_ge_i32 {s(S_J)}, %[ntw]")
        emit(f"s_cbranch_scc1 {P('nop1')}")
        emit_block(f"p1[{u}]", pad_hazards(pipelined(list(range(16)), lambda g: p1_reads(g, par, A), lambda g: p1_mfmas(g, par), [], []), tail_mfma_pad=False))
        emit(f"{P('nop1')}:")
        # pair j - 1: 1 <= j <= nt_w
        emit(f"s_cmp_lt_i32 {s(S_J)}, 1")
        emit(f"s_cbranch_scc1 {P('end')}")
        emit(f"s_cmp_gt_i32 {s(S_J)}, %[ntw]")
        emit(f"s_cbranch_scc1 {P('end')}")
        emit(f"s_add_u32 {s(S_TMP)}, %[kvb], 31")
        emit(f"s_cmp_lt_i32 {s(S_Q2)}, {s(S_TMP)}")
        emit(f"s_cbranch_scc1 {P('mask')}")
        emit(f"s_add_u32 {s(S_TMP)}, {s(S_Q2)}, 64")
        emit(f"s_cmp_gt_i32 {s(S_TMP)}, %[n]")
        emit(f"s_cbranch_scc1 {P('mask')}")
        emit(f"s_cmp_gt_i32 {s(S_LO2)}, %[kvb]")
        emit(f"s_cbranch_scc1 {P('mask')}")
        emit_block(f"softmax-bwd[{u}]", pad_hazards(softmax_bwd_list(par ^ 1), tail_mfma_pad=False))
        emit(f"s_branch {P('p2')}")
        emit(f"{P('mask')}:")
        emit_block(f"masked[{u}]", pad_hazards(masked_list(par ^ 1, A), tail_mfma_pad=False))
        emit(f"{P('p2')}:")
        emit_block(f"p2[{u}]", pad_hazards(pipelined(list(range(16, 32)), lambda g: p2_reads(g, A), lambda g: p2_mfmas(g, par ^ 1), [], []), tail_mfma_pad=False))
        emit(f"{P('end')}:")
        if not unroll6:  # rotate the slots
            emit(f"s_mov_b32 {s(S_TMP)}, {s(S_O2)}")
            emit(f"s_mov_b32 {s(S_O2)}, {s(S_O1)}")
            emit(f"s_mov_b32 {s(S_O1)}, {s(S_ON)}")
            emit(f"s_mov_b32 {s(S_ON)}, {s(S_TMP)}")
        emit(f"s_mov_b32 {s(S_Q2)}, {s(S_Q1)}")
        emit(f"s_mov_b32 {s(S_Q1)}, {s(S_QN)}")
        emit(f"s_add_u32 {s(S_J)}, {s(S_J)}, 1")
        emit(f"s_cmp_gt_i32 {s(S_J)}, %[total]")
        if u != bodies[-1]:
            emit(f"s_cbranch_scc1 {L('done')}")
        else:
            emit(f"s_cbranch_scc0 {L('top0')}")
    emit(f"{L('done')}:")
    # hand the accumulators to the C++ epilogue through LDS (after every wave has finished reading the tiles): tuple t of the wave at t KiB + lane x 16
    emit("s_nop 15")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    emit("s_barrier")
    for t in range(32):
        emit(f"ds_write_b128 %[aout], {at(t * 4, 4)} offset:{t * 1024}")
    emit("s_waitcnt lgkmcnt(0)")
    return lines


HEADER = '''// attention_dkdv_p2_asm.inc.h — GENERATED by tools/gen_attn_dkdv_asm.py (do not edit; edit the generator).  The tile loop of attn_bwd_dkdv_p2_kernel as one
// hand-placed gfx950 instruction stream: v80-v255, a0-a191, s70-s89 are owned by the stream; hazard distances and counted waits checked by the generator.
#pragma once
'''
SIG = ("const void* kp0, const void* kp1, const void* vp0, const void* vp1, const unsigned (&aQ)[4], const unsigned (&aT)[8], unsigned arow, const unsigned (&aQ2)[4],\n"
       "    const unsigned (&aT2)[8], unsigned arow2, const int (&vo)[4],\n"
       "    int lane4, int kvi, int lq4, unsigned aout, __amdgpu_buffer_rsrc_t qrsrc, __amdgpu_buffer_rsrc_t dorsrc, __amdgpu_buffer_rsrc_t rowrsrc, const int* seqptr,\n"
       "    int sc2_bits, int n, int kvb, int total, int nt_w, int head0, int base0, int roff0, int qi00, int qdt, int rowdh, int rowdt, int ldsw, int rowm0")


VARIANTS = [dict(), dict(unroll6=False), dict(ablate=("novalu",)), dict(ablate=("nolds",)), dict(ablate=("nomfma",)), dict(ablate=("nodma",)), dict(ablate=("nobarrier",)),
            dict(pk=True),  # 7: packed multiplies (measured, not adopted)
            dict(pk=True, pk_lag=(2, 5)), dict(pk=True, pk_lag=(3, 7)), dict(pk=True, pk_lag=(4, 9))]  # 8-10: longer distances between a packed multiply and its consumers


def render(vi, opts, want_stats=False):
    lines = build(opts, want_stats)
    body = "\\n\\t\"\n      \"".join(l.replace('"', '\\"') for l in lines)
    ins = (['[kp0] "v"(kp0)', '[kp1] "v"(kp1)', '[vp0] "v"(vp0)', '[vp1] "v"(vp1)'] + [f'[aq{k}] "v"(aQ[{k}])' for k in range(4)] +
           [f'[at{k}] "v"(aT[{k}])' for k in range(8)] + ['[arow] "v"(arow)'] + [f'[aq2_{k}] "v"(aQ2[{k}])' for k in range(4)] +
           [f'[at2_{k}] "v"(aT2[{k}])' for k in range(8)] + ['[arow2_] "v"(arow2)'] + [f'[vo{j}] "v"(vo[{j}])' for j in range(4)] +
           ['[lane4] "v"(lane4)', '[kvi] "v"(kvi)', '[lq4] "v"(lq4)', '[aout] "v"(aout)', '[qrsrc] "s"(qrsrc)', '[dorsrc] "s"(dorsrc)', '[rowrsrc] "s"(rowrsrc)',
            '[seqptr] "s"(seqptr)', '[sc2] "s"(sc2_bits)'] + (['[sc2v] "v"(sc2_bits)'] if opts.get("pk") else []) + [ '[n] "s"(n)', '[kvb] "s"(kvb)', '[total] "s"(total)', '[ntw] "s"(nt_w)', '[head0] "s"(head0)',
            '[base0] "s"(base0)', '[roff0] "s"(roff0)', '[qi00] "s"(qi00)', '[qdt] "s"(qdt)', '[rowdh] "s"(rowdh)', '[rowdt] "s"(rowdt)', '[ldsw] "s"(ldsw)',
            '[rowm0] "s"(rowm0)'])
    clob = [f'"v{r}"' for r in range(80, 256)] + [f'"a{r}"' for r in range(192)] + [f'"s{r}"' for r in range(S_SC2 if opts.get("pk") else S_J, S_LAST + 1)] + ['"vcc"', '"scc"', '"memory"']
    note = "the product stream" if not opts else str(opts)
    return (f"template <>\n__device__ __forceinline__ void dkdv_p2_loop_asm<{vi}>({SIG}) {{  // {note}\n  asm volatile(\n      \"" + body + "\"\n      :\n      : " +
            ", ".join(ins) + "\n      : " + ", ".join(clob) + ");\n}\n"), len(lines)


def main():
    """attention_dkdv_p2_asm.inc.h: the declaration + the product stream (committed).  --variants: also attention_dkdv_p2_asm_variants.inc.h with the timing
    ablations (generated by `make VARIANTS=1`, not committed)."""
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "prime_rl_amd", "csrc")
    if "--out" in sys.argv:
        out_dir = sys.argv[sys.argv.index("--out") + 1]
    fn, n_lines = render(0, VARIANTS[0], "--stats" in sys.argv)
    doc = ("\n// One wave's whole loop over its (q tile, head) pairs: total + 1 iterations.  In: per-lane global pointers of the wave's K / V fragment rows, per-lane LDS\n"
           "// read addresses (slot 0), DMA source offsets, mask operands; scalars as named.  Out: the 32 accumulator tuples (dK^T [b][nb] then dV^T [b][nb]) in\n"
           "// LDS at aout + t KiB, lane-linear, after a workgroup barrier.  V = 0: the product stream; V >= 1: timing ablations (`make VARIANTS=1`).\n")
    open(os.path.join(out_dir, "attention_dkdv_p2_asm.inc.h"), "w").write(HEADER + doc + f"template <int V>\n__device__ __forceinline__ void dkdv_p2_loop_asm({SIG});\n\n" + fn)
    print(f"wrote {out_dir}/attention_dkdv_p2_asm.inc.h: {n_lines} asm lines", file=sys.stderr)
    if "--variants" in sys.argv:
        head = ("// attention_dkdv_p2_asm_variants.inc.h — GENERATED by tools/gen_attn_dkdv_asm.py --variants (`make VARIANTS=1`; not committed): timing ablations (WRONG\n"
                "// results) of dkdv_p2_loop_asm, selected by pm_attn_debug_set_variant(1, 12 + 32 V).\n#pragma once\n\n")
        open(os.path.join(out_dir, "attention_dkdv_p2_asm_variants.inc.h"), "w").write(head + "\n".join(render(vi, o)[0] for vi, o in enumerate(VARIANTS) if vi > 0))
        print(f"wrote {out_dir}/attention_dkdv_p2_asm_variants.inc.h: {len(VARIANTS) - 1} variants", file=sys.stderr)


if __name__ == "__main__":
    main()
