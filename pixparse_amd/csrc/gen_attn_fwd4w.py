#!/usr/bin/env python3
"""Generates attn_fwd2x_body.inc / attn_fwd4w_body.inc: the hand-placed gfx950 instruction stream of the attention forward (attention.hip:
attn_fwd4w_kernel<OCC>; non-causal, no dropout, head_dim 64, q prescaled by scale * log2 e) at 256 / 512 registers per wave.

Workgroup = 256 queries of one (batch, head) = 4 waves; a wave owns 64 queries = two 32-query blocks qb, so every K row fragment and every V^T
fragment it reads from LDS feeds TWO v_mfma_f32_32x32x16_bf16 (the 32-queries-per-wave kernel reads 1 KiB of LDS per MFMA = the whole LDS bandwidth
of the CU at the MFMA rate; this one half of it).  Per 64-key tile a wave issues
  16 MFMAs  S^T = K . Q^T + seed   (query on the lane, the 32 keys of a half kh in the 16 accumulator registers; seed = -m in every register)
  64 v_exp_f32 (in place) + 32 v_cvt_pk_bf16_f32 -> the P^T fragments (accumulator order = the k order of the transposed V reads: no lane exchange)
  16 MFMAs  O^T += V^T . P^T      (O in a[0:63])
   8 v_mfma_f32_16x16x32_bf16  L += ONES . P^T: the row sums on the matrix pipe (a 16-cycle MFMA instead of 8 VALU adds of 4 cycles: a lone wave's
     stream is bound by VALU ISSUE, not by the pipe).  A lane of the 16x16x32 B operand holds query (lane & 15) + 16 (k-group & 1), so the ONES
     operand has row 1 = 1 on k-groups 0 / 2, row 2 = 1 on k-groups 1 / 3: L[1][n] = row sum of query n, L[2][n] = of query n + 16 (a[64:71]).
The reference m is the exact row maximum over the FIRST key tile (the wrapper computes it) and never moves: probabilities are 2^(s - m), up to
2^127 -- bf16 and fp32 have the exponent range, the relative precision of p, l and O does not depend on it.  A row whose later scores exceed
its first tile's maximum by more than 127 (or whose sums overflow) ends with a non-finite l or O: the wrapper detects that at the end of the
block and the WORKGROUP re-runs the block with the moving-maximum kernel body (fwd_pre_block: never seen outside the forced test).

Software pipeline over quarters k = 4 t + j (tile t, key half kh = j >> 1, query block qb = j & 1), one step per quarter:
  step k:  MFMAs  QK(k) [4]  |  PV(k - 2) [4] + L(k - 2) [2 small]      VALU: exp / cvt of quarter k - 1 (two v_exp + one cvt per 32-cycle gap; a
           conversion never sits in the gap of a v_exp that feeds it: the transcendental's result is not forwarded)
MFMA order QK0 PVa QK1 PVb La QK2 PVc QK3 PVd Lb: no MFMA follows one that writes its accumulator.
K / V tiles (8 KiB + 8 KiB, the swizzled 64 x 64 image of attn_frag.h) arrive by LDS-DMA two tiles ahead into a ring of four 16-KiB slots (64 KiB:
every ds_read offset is an immediate); each wave issues 4 of the 16 pieces of a tile, one per step; one barrier per tile (tile t + 1 landed for
every wave, slot of tile t - 2 free).  Tiles past the end arrive as zeros (bounds-checked descriptor).  The loop is unrolled four-fold (ring
period); the first tile is peeled (nothing to exponentiate / accumulate yet); the LAST tile runs in a tail with register-based slot addressing whose
S chains start from seed-or-minus-infinity tuples (keys >= Nk masked: built in the tail, 32 VALU per quarter), followed by the drain of the pipeline.

TWO FORMS of this pipeline (generate(occ2)); measured: profiles/r5_attn_fwd_stream.txt:
 * 512 registers, one workgroup per CU (attn_fwd4w_body.inc).  Fragment registers are double sets indexed by kh; the LDS reads of a set are issued
   while the other set is in use (step 4t: K[kh1](t) ks 0,1 + V[kh0](t) s 0 | 4t + 1: ks 2,3 + s 1 | 4t + 2: V[kh1](t) s 0 | 4t + 3: BARRIER(t + 1),
   K[kh0](t + 1), V[kh1](t) s 1), placed by the PATTERN tables.  v[0:31] K fragments [kh][ks] x 4, v[32:63] V^T fragments [kh][2 s + db] x 4, v[64:95]
   S / P accumulators (two sets), v[96:111] P^T fragments (two sets x [s] x 4), v[112:143] masked seed tuples (tail), v[144:151] slot-relative lane
   addresses (tail); Q fragments, seeds, ONES are the wrapper's register-tuple operands.  A LONE WAVE ISSUES IN ORDER: its exp / cvt fill the gaps
   and every further instruction costs its ~5 issue cycles: 1500 cycles per tile against 1152 of matrix-pipe time, wherever the fillers sit.
 * 256 registers, TWO workgroups per CU (attn_fwd2x_body.inc; the default): the second wave on each SIMD issues into the first one's stalls and
   the pair runs at the pipe's rate (2304 cycles per tile for the two).  Single fragment sets, each register re-read right behind its last use (odd
   steps, ten MFMAs ahead of the next use); LLVM splits a 256-register budget 128 + 128, so v[0:15] K fragments, v[16:47] S / P sets, v[48:63] P^T
   fragments, v[64:95] seed tuples (C and D of an MFMA share a register file and the VALU exponentiates S), a[72:103] Q fragments (loaded by the
   stream itself from memory), a[104:107] ONES, a[108:123] V^T fragments; the tail masks straight into the S set and advances the lane addresses in
   place.  No register-tuple operands at all (hipcc's allocator does not terminate on the tuple form at this budget).
Both: a[0:63] O^T [qb][db] x 16, a[64:71] L [qb] x 4.  Same arithmetic, bit-identical results.

Options (timing experiments: scripts/ab_f4w.sh): F4W_OPTS=pat=..,pat3=..,pat2=..,pat23=.. filler placement tables; lsum=valu row sums as fp32 VALU adds
of the unrounded probabilities (+4 % time); lsum=mfma4 the 4x4x4 MFMA form (experiment only, see below); fragacc=1 (512-register form: fragments in
the accumulator half, nothing).  F4W_DROP=exp,cvt,lsum,lgkwait and G4W_DROP=dsread,dma,barrier,vmwait: timing-only builds with WRONG results.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_gemm4w import I, Hazards, op, vregs  # noqa: E402

OPTS = dict(kv.split('=') if '=' in kv else (kv, '1') for kv in filter(None, os.environ.get('F4W_OPTS', '').split(',')))
FDROP = set(filter(None, os.environ.get('F4W_DROP', '').split(',')))      # timing-only builds (WRONG results): exp cvt dsread dma lsum

V_KF, V_VF, V_S, V_P, V_CM, V_TA, N_HAND = 0, 32, 64, 96, 112, 144, 152
A_O, A_L = 0, 64
OCC2 = False        # set by generate(occ2=True): the two-waves-per-SIMD form (256 registers per wave) -- see the header
FRAG_ACC = OPTS.get('fragacc', '0') == '1'      # K / V fragments in the accumulator half of the register file (a[72:135]) instead of v[0:63]
FP = 'a' if FRAG_ACC else 'v'
# lsum=mfma4: row sums by v_mfma_f32_4x4x4_16b_bf16 (1/8 of the 16x16x32 form's multiply-adds).  EXPERIMENT ONLY: the instruction takes 16 cycles of the pipe
# on gfx950, not 8 (2531 against 2304 cycles per key tile), and the lane mapping assumed here is not the hardware's (sums come out wrong)
LSUM4 = OPTS.get('lsum', 'mfma') == 'mfma4'
LSUM_VALU = OPTS.get('lsum', 'mfma') == 'valu'      # row sums as fp32 VALU adds of the unrounded probabilities (two chains per query block) instead of MFMAs
if FRAG_ACC:
    V_KF, V_VF = 72, 104


def set_map(occ2):
    """register map of the form being generated"""
    global OCC2, V_KF, V_VF, V_S, V_P, V_CM, V_TA, N_HAND
    OCC2 = occ2
    if occ2:
        V_KF, V_VF, V_S, V_P, V_CM, V_TA, N_HAND = 0, A_VF2, 16, 48, None, None, 96
    else:
        V_KF, V_VF, V_S, V_P, V_CM, V_TA, N_HAND = (72 if FRAG_ACC else 0), (104 if FRAG_ACC else 32), 64, 96, 112, 144, 152
RING = 4
SLOT = 16384


def KF(kh, ks):
    return V_KF + (0 if OCC2 else 16 * kh) + 4 * ks


def VF(kh, f):
    return V_VF + (0 if OCC2 else 16 * kh) + 4 * f


def SB(b):
    return V_S + 16 * (b & 1)


def PB(b, s):
    return V_P + 8 * (b & 1) + 4 * s


def AO(qb, db):
    return A_O + 16 * (2 * qb + db)


def AL(qb):
    return A_L + 4 * qb


def vt(b, n):
    return f'v[{b}:{b + n - 1}]'


def ft(b, n, which='k'):
    """fragment registers: K ('k') / V^T ('v') sets; the two-waves form keeps V^T in the accumulator half (LLVM splits a 256-register budget 128 + 128)"""
    pre = ('a' if which == 'v' else 'v') if OCC2 else FP
    return f'{pre}[{b}:{b + n - 1}]'


def fregs(b, n, which='k'):
    pre = ('a' if which == 'v' else 'v') if OCC2 else FP
    return {f'{pre}{i}' for i in range(b, b + n)}


def at(b, n):
    return f'a[{b}:{b + n - 1}]'


# ---------------------------------------------------------------------------------------------------------------- instruction builders
A_Q, A_ONES, A_VF2, A_END = 72, 104, 108, 124      # two-waves form: Q fragments, ONES and the V^T fragment set in hand-allocated accumulator registers
V_SEED2 = 64                           # ... its seed tuples (C and D of an MFMA share a register file: S is exponentiated by the VALU) v[64:95]


def q_op(qb, ks):
    return at(A_Q + 16 * qb + 4 * ks, 4) if OCC2 else op(f'q{qb}{ks}')


def seed_op(qb):
    return vt(V_SEED2 + 16 * qb, 16) if OCC2 else op(f'seed{qb}')


def ones_op(n=4):
    if OCC2:
        return at(A_ONES, n)
    return op('ones') if n == 4 else op('ones2')


def mfma_qk(k, ks, cm=None):
    """S[k & 1] (+)= K[kh][ks] . Q[qb][ks]^T; the chain starts from the seed tuple of the query block (or the masked tuple `cm`, tail)"""
    j = k & 3
    kh, qb = j >> 1, j & 1
    d = SB(k)
    if ks == 0:
        c = vt(cm, 16) if cm is not None else seed_op(qb)
    else:
        c = vt(d, 16)
    rd = fregs(KF(kh, ks), 4) | (vregs(cm, 16) if (cm is not None and ks == 0) else set())
    return I(f'v_mfma_f32_32x32x16_bf16 {vt(d, 16)}, {ft(KF(kh, ks), 4)}, {q_op(qb, ks)}, {c}', 'mfma', reads=rd, writes=vregs(d, 16))


def mfma_pv(k, s, db):
    """O[qb][db] += V^T[kh][s][db] . P(k)[s]"""
    j = k & 3
    kh, qb = j >> 1, j & 1
    o = AO(qb, db)
    return I(f'v_mfma_f32_32x32x16_bf16 {at(o, 16)}, {ft(VF(kh, 2 * s + db), 4, "v")}, {vt(PB(k, s), 4)}, {at(o, 16)}', 'mfma',
             reads=fregs(VF(kh, 2 * s + db), 4, 'v') | vregs(PB(k, s), 4))


def mfma_l4(k, s, half):
    """L[qb][0..3] += the lane's own four probabilities P(k)[s][4 half .. 4 half + 3] (ONES . B per 4 x 4 block: every row of D gets the column sum)"""
    qb = k & 1
    b = PB(k, s) + 2 * half
    return I(f'v_mfma_f32_4x4x4_16b_bf16 {at(AL(qb), 4)}, {ones_op(2)}, {vt(b, 2)}, {at(AL(qb), 4)}', 'mfma', reads=vregs(b, 2))


def mfma_l(k, s):
    qb = k & 1
    return I(f'v_mfma_f32_16x16x32_bf16 {at(AL(qb), 4)}, {ones_op()}, {vt(PB(k, s), 4)}, {at(AL(qb), 4)}', 'mfma', reads=vregs(PB(k, s), 4))


def v_exp(k, r):
    x = SB(k) + r
    return I(f'v_exp_f32 v{x}, v{x}', 'valu', reads={f'v{x}'}, writes={f'v{x}'}, tag='exp')


def v_cvt(k, pair):
    x = SB(k) + 2 * pair
    d = PB(k, pair >> 2) + (pair & 3)
    return I(f'v_cvt_pk_bf16_f32 v{d}, v{x}, v{x + 1}', 'valu', reads={f'v{x}', f'v{x + 1}'}, writes={f'v{d}'}, tag='cvt')


def v_sum(k, r):
    """psum[qb][r & 1] += p(k)[r]"""
    x = SB(k) + r
    acc = op(f'ps{k & 1}{r & 1}')
    return I(f'v_add_f32 {acc}, {acc}, v{x}', 'valu', reads={f'v{x}'}, tag='sum')


class Addr:
    """LDS addressing of a tile: in the unrolled loop the wrapper's lane addresses + slot immediates, in the tail the slot-relative copies"""
    def __init__(self, slot=None):
        self.slot = slot

    def k(self, ks):
        if self.slot is not None:
            return op(f'akr{ks}'), self.slot * SLOT
        return (op(f'akr{ks}'), 0) if OCC2 else (f'v{V_TA + ks}', 0)      # tail: the two-waves form advances the operands themselves

    def v(self, db, half):
        if self.slot is not None:
            return op(f'avt{db}{half}'), self.slot * SLOT
        return (op(f'avt{db}{half}'), 0) if OCC2 else (f'v{V_TA + 4 + 2 * db + half}', 0)


def read_k(addr, kh_src, kh_dst, ks, tag):
    reg, imm = addr.k(ks)
    b = KF(kh_dst, ks)
    return [I(f'ds_read_b128 {ft(b, 4)}, {reg} offset:{imm + 4096 * kh_src}', 'ds', writes=fregs(b, 4), tag=tag)]


def read_v(addr, kh, s, db, tag):
    b = VF(kh, 2 * s + db)
    out = []
    for half in range(2):
        reg, imm = addr.v(db, half)
        out.append(I(f'ds_read_b64_tr_b16 {ft(b + 2 * half, 2, "v")}, {reg} offset:{imm + 8192 + 4096 * kh + 2048 * s}', 'ds', writes=fregs(b + 2 * half, 2, 'v'), tag=tag))
    return out


def dma_piece(slot, n, tag):
    """piece n of the wave's four of a tile: 0 / 1 = K rows 8 w.. / 32 + 8 w.., 2 / 3 = V likewise"""
    o = 'k' if n < 2 else 'v'
    out = [I(f's_add_u32 m0, {op("s_ldsw")}, {slot * SLOT + (8192 if n >= 2 else 0) + (4096 if n & 1 else 0)}', 'salu')]
    if n & 1:
        out.append(I(f's_add_u32 {op("s_t")}, {op(f"s_{o}off")}, {op(f"s_{o}32")}', 'salu'))
        soff = op('s_t')
    else:
        out.append(I('s_nop 0', 'salu'))
        soff = op(f's_{o}off')
    out.append(I(f'buffer_load_dwordx4 {op(f"voff{o.upper()}")}, {op(f"srd{o.upper()}")}, {soff} offen lds', 'dma', tag=tag))
    if n & 1:
        out.append(I(f's_add_u32 {op(f"s_{o}off")}, {op(f"s_{o}off")}, {op(f"s_{o}step")}', 'salu'))
    return out


# MFMA positions of a step: QK0 PVa QK1 PVb L QK2 [L] PVc QK3 [L] PVd L  (four 8-cycle row-sum MFMAs in the 4x4x4 form, two 16-cycle ones otherwise)
if LSUM4:
    NPOS, QKPOS, PVPOS, LPOS = 12, (0, 2, 5, 8), (1, 3, 7, 10), ((4, 0, 0), (6, 0, 1), (9, 1, 0), (11, 1, 1))
else:
    NPOS, QKPOS, PVPOS, LPOS = 10, (0, 2, 5, 7), (1, 3, 6, 8), ((4, 0, None), (9, 1, None))


def pv_l(k):
    """{position: MFMA} of PV(k) and L(k)"""
    out = {PVPOS[0]: mfma_pv(k, 0, 0), PVPOS[1]: mfma_pv(k, 0, 1), PVPOS[2]: mfma_pv(k, 1, 0), PVPOS[3]: mfma_pv(k, 1, 1)}
    if not ('lsum' in FDROP or LSUM_VALU):
        for pos, s_, half in LPOS:
            out[pos] = mfma_l4(k, s_, half) if LSUM4 else mfma_l(k, s_)
    return out


# what each of the ten MFMA gaps of a step takes, in order: e = next v_exp, c = next conversion (of v_exp two gaps back at the latest), l = next LDS
# read, d = the step's LDS-DMA piece, b = the barrier group, s = two row-sum adds (lsum=valu).  Issue model (gen_attn_bwd_sp.py): a gap costs
# max(MFMA time, fillers + ~5) with v_exp 8, plain VALU 5, an LDS instruction 9: [e e l] = 25 + 5, [e e c c] = 26 + 5 fill a 32-cycle gap.
if LSUM4:
    PATTERN2 = {0: OPTS.get('pat2', 'ee-eec-eec-eec-d-eec--eec-eec--eec-c').split('-'),
                3: OPTS.get('pat23', 'eeb-eec-eec-eec-d-eec--eec-eec--eec-c').split('-')}
else:
    PATTERN2 = {0: OPTS.get('pat2', 'ee-eec-eec-eec-d-eec-eec-eec-eec-c').split('-'),
                3: OPTS.get('pat23', 'eeb-eec-eec-eec-d-eec-eec-eec-eec-c').split('-')}
if LSUM4:
    PATTERN = {0: OPTS.get('pat', 'eel-eecl-eel-eecc-l-eel--eecc-eel--eecc-cd').split('-'),
               3: OPTS.get('pat3', 'eeb-eecl-eel-eeccl-l-eel--eeccl-eel-l-eeccl-cd').split('-')}
else:
    PATTERN = {0: OPTS.get('pat', 'eel-eecl-eel-eecc-l-eel-eecc-eel-eecc-cd').split('-'),
               3: OPTS.get('pat3', 'eeb-eecl-eel-eeccl-l-eel-eeccl-eel-eeccl-cd').split('-')}


def step(k, pos, has_exp, has_pv, tail=False, last_body=False, cm=None, addr_cur=None, addr_next=None, dma=True, tag_t=0):
    """one pipeline step = quarter k = 4 t + j.  pos = ring slot of tile t.  Returns (mfmas[10], gaps[10])"""
    j = k & 3
    mf = [None] * NPOS
    for ks in range(4):
        mf[QKPOS[ks]] = mfma_qk(k, ks, cm if ks == 0 else None)
    if has_pv:
        for pos_, ins in pv_l(k - 2).items():
            mf[pos_] = ins
    gaps = [[] for _ in range(NPOS)]
    exps = [v_exp(k - 1, r) for r in range(16)] if (has_exp and 'exp' not in FDROP) else []
    cvts = [v_cvt(k - 1, p) for p in range(8)] if (has_exp and 'cvt' not in FDROP) else []
    sums = [v_sum(k - 1, r) for r in range(16)] if (has_exp and LSUM_VALU and 'lsum' not in FDROP) else []
    # ---- LDS reads of this step, in issue order (see the header), the barrier group (step 4 t + 3) and the step's LDS-DMA piece
    T_ = tag_t
    lds, bar, piece = [], [], []
    if j == 0:
        lds = read_k(addr_cur, 1, 1, 0, (T_, 'K1')) + read_k(addr_cur, 1, 1, 1, (T_, 'K1')) + read_v(addr_cur, 0, 0, 0, (T_, 'V0')) + read_v(addr_cur, 0, 0, 1, (T_, 'V0'))
    elif j == 1:
        lds = read_k(addr_cur, 1, 1, 2, (T_, 'K1')) + read_k(addr_cur, 1, 1, 3, (T_, 'K1')) + read_v(addr_cur, 0, 1, 0, (T_, 'V0')) + read_v(addr_cur, 0, 1, 1, (T_, 'V0'))
    elif j == 2:
        lds = read_v(addr_cur, 1, 0, 0, (T_, 'V1')) + read_v(addr_cur, 1, 0, 1, (T_, 'V1'))
    else:
        if not tail:
            # BARRIER(t + 1): this wave's pieces of tile t + 1 have landed (the three of tile t + 2 issued so far stay in flight)
            bar = [I('s_waitcnt vmcnt(3)', 'vmwait'), I('s_barrier', 'barrier')]
            for ks in range(4):
                lds += read_k(addr_next, 0, 0, ks, (T_ + 1, 'K0'))
        lds += read_v(addr_cur, 1, 1, 0, (T_, 'V1')) + read_v(addr_cur, 1, 1, 1, (T_, 'V1'))
    if dma and not tail:
        piece = dma_piece((pos + 2) % RING, j, None)
    fixed = {}
    if OCC2:
        # single fragment sets: every register is re-read right behind its last use (odd steps: query block 1 has used it too), ten MFMAs ahead of
        # its next use; K of the next key half behind the QK MFMAs, V^T of the key half after next behind the PV MFMAs
        lds = []
        if j & 1:
            if j == 1:
                for ks, g in enumerate(QKPOS):
                    fixed[g] = read_k(addr_cur, 1, 1, ks, (T_, 'K1'))
            elif not tail:
                for ks, g in enumerate(QKPOS):
                    fixed[g] = read_k(addr_next, 0, 0, ks, (T_ + 1, 'K0'))
            for f, g in enumerate(PVPOS):
                fixed[g] = read_v(addr_cur, j >> 1, f >> 1, f & 1, (T_, f'V{j >> 1}'))
    pat = PATTERN2[3 if (j == 3 and not tail) else 0] if OCC2 else PATTERN[3 if (j == 3 and not tail) else 0]
    done_e = 0
    for g in range(NPOS):
        e_before = done_e
        for ch in pat[g]:
            if ch == 'e' and exps:
                gaps[g].append(exps.pop(0))
                done_e += 1
            elif ch == 'c' and cvts:
                assert not has_exp or 'exp' in FDROP or 2 * (8 - len(cvts)) + 2 <= e_before, 'a conversion in the gap of its own v_exp'
                gaps[g].append(cvts.pop(0))
            elif ch == 's' and sums:
                gaps[g] += [sums.pop(0), sums.pop(0)]
            elif ch == 'l' and lds:
                gaps[g].append(lds.pop(0))
            elif ch == 'd' and piece:
                gaps[g] += piece
                piece = []
            elif ch == 'b' and bar:
                gaps[g] += bar
                bar = []
        gaps[g] += fixed.get(g, [])
    assert not exps and not cvts and not bar, (len(exps), len(cvts), len(bar))
    gaps[NPOS - 1] += lds + piece + sums
    return mf, gaps


def emit_step(E, mf, gaps):
    for m in range(NPOS):
        if mf[m] is not None:
            E(mf[m])
        for ins in gaps[m]:
            E(ins)


def mask_tuple(E, k, buf):
    """tail: the seed tuple of quarter k with -inf where the key (32 kh + row of register r, + 4 hh: folded into limlane) lies past the end"""
    j = k & 3
    kh, qb = j >> 1, j & 1
    b = SB(k) if OCC2 else V_CM + 16 * buf      # two-waves form: straight into the S set of the quarter (the chain then starts from its own destination)
    for r in range(16):
        row = 32 * kh + (r & 3) + 8 * (r >> 2)
        E(I(f'v_cmp_lt_i32 vcc, {row}, {op("limlane")}', 'valu'))
        if OCC2:
            E(I(f'v_mov_b32 v{b + r}, 0xff800000', 'valu', writes={f'v{b + r}'}))
            E(I(f'v_cndmask_b32 v{b + r}, v{b + r}, {op(f"negm{qb}")}, vcc', 'valu', writes={f'v{b + r}'}))
        else:
            E(I(f'v_cndmask_b32 v{b + r}, {op("neginf")}, {op(f"negm{qb}")}, vcc', 'valu', writes={f'v{b + r}'}))
    E(I('s_nop 1', 'nop'))
    return b


def generate(occ2=False):
    set_map(occ2)
    H = Hazards()
    H.OLD = 1 << 30      # exact counted waits: the merge-over-old-reads shortcut of the GEMM generator makes the pending set depend on MFMA counts, which differ in the peeled first tile
    E = H.emit
    # ---- entry: tiles 0 and 1 are in flight / landed (the wrapper waited for tile 0 behind a barrier to take the row maxima), s_koff / s_voff point at tile 2
    a0 = Addr(0)
    if occ2:
        # Q fragments straight from memory into the accumulator half (8 x 16 bytes per lane; the wrapper's own copy died with the row maxima), seeds and
        # ONES written there from one register each: the statement has no register-tuple operands left
        for qb in range(2):
            for ks in range(4):
                E(I(f'buffer_load_dwordx4 {at(A_Q + 16 * qb + 4 * ks, 4)}, {op(f"voffQ{qb}")}, {op("srdQ")}, 0 offen offset:{32 * ks}', 'vmem'))
        for qb in range(2):
            for r in range(16):
                E(I(f'v_mov_b32 v{V_SEED2 + 16 * qb + r}, {op(f"negm{qb}")}', 'valu'))
        for r in range(2 if LSUM4 else 4):
            E(I(f'v_accvgpr_write_b32 a{A_ONES + r}, {op("onesv")}', 'valu'))
    for ks in range(4):
        for ins in read_k(a0, 0, 0, ks, (0, 'K0')):
            E(ins)
    if occ2:
        H.out.append(I('s_waitcnt vmcnt(0)', 'wait'))      # Q (and with it tile 1: the queue retires in order)
        H.out.append(I('s_nop 1', 'nop'))

    def body(t_tag, pos, first):
        cur, nxt = Addr(pos), Addr((pos + 1) % RING)
        for j in range(4):
            k = 4 * t_tag + j
            has_exp = not (first and j == 0)
            has_pv = not (first and j < 2)
            mf, gaps = step(k, pos, has_exp, has_pv, addr_cur=cur, addr_next=nxt, tag_t=t_tag)
            emit_step(E, mf, gaps)
        E(I(f's_sub_u32 {op("s_cnt")}, {op("s_cnt")}, 1', 'salu'))
        E(I(f's_mov_b32 {op("s_slot")}, {((pos + 1) % RING) * SLOT}', 'salu'))
        E(I(f's_cmp_eq_u32 {op("s_cnt")}, 0', 'salu'))
        E(I('s_cbranch_scc1 TAIL%=', 'branch'))

    H.in_loop = True
    body(0, 0, True)
    s1 = H.state(0)
    H.out.append(I('LOOP%=:', 'label'))
    for n, pos in enumerate((1, 2, 3, 0)):
        body(1 + n, pos, False)
    assert H.state(RING) == s1, 'loop-carried LDS state differs'
    E(I('s_branch LOOP%=', 'branch'))
    H.in_loop = False
    # ---- tail: the last tile (slot in s_slot) with masked seeds, then the drain of the pipeline.  Entered from any ring position with the SAME pending
    # LDS reads (K[kh0] of the last tile), which the state assertion above guarantees.
    H.out.append(I('TAIL%=:', 'label'))
    names = [f'akr{ks}' for ks in range(4)] + [f'avt{db}{h}' for db in range(2) for h in range(2)]
    for n, nm in enumerate(names):
        if OCC2:
            E(I(f'v_add_u32 {op(nm)}, {op("s_slot")}, {op(nm)}', 'valu'))
        else:
            E(I(f'v_add_u32 v{V_TA + n}, {op("s_slot")}, {op(nm)}', 'valu', writes={f'v{V_TA + n}'}))
    ta = Addr(None)
    T = 8            # tag only: any tile number whose quarter parity matches (4 T + j)
    for j in range(4):
        k = 4 * T + j
        cm = mask_tuple(E, k, j & 1)
        mf, gaps = step(k, 0, True, True, tail=True, cm=cm, addr_cur=ta, tag_t=T)
        emit_step(E, mf, gaps)
    for j in range(2):     # drain: exp / cvt of the last quarter, PV / L of the last two
        k = 4 * T + 4 + j
        mf = [None] * NPOS
        for pos_, ins in pv_l(k - 2).items():
            mf[pos_] = ins
        gaps = [[] for _ in range(NPOS)]
        if j == 0:
            for r in range(16):
                gaps[0].append(v_exp(k - 1, r))
            for p in range(8):
                gaps[0].append(v_cvt(k - 1, p))
            if LSUM_VALU:
                for r in range(16):
                    gaps[0].append(v_sum(k - 1, r))
            gaps[0].append(I('s_nop 1', 'nop'))
        emit_step(E, mf, gaps)
    H.drain('s_waitcnt vmcnt(0) lgkmcnt(0)')
    H.out.append(I('s_nop 7\n\ts_nop 7', 'nop'))
    if 'lgkwait' in FDROP:      # timing-only: LDS reads without their waits
        return [i for i in H.out[:-2] if not (i.kind == 'wait' and 'lgkmcnt' in i.text)] + H.out[-2:]
    return H.out


def render(stream, occ2=False):
    lines = []
    for ins in stream:
        lines += ins.text.split('\n\t')
    body = '\n'.join(f'    "{ln}\\n\\t"' for ln in lines)
    addr_names = [f'akr{ks}' for ks in range(4)] + [f'avt{db}{h}' for db in range(2) for h in range(2)]
    outs = ', '.join(f'"+{{a[{AO(qb, db)}:{AO(qb, db) + 15}]}}"(o{qb}{db})' for qb in range(2) for db in range(2)) + ',\n      ' + \
        (', '.join(f'[ps{qb}{c}] "+v"(ps{qb}{c})' for qb in range(2) for c in range(2)) if LSUM_VALU else
         ', '.join(f'"+{{a[{AL(qb)}:{AL(qb) + 3}]}}"(lsum{qb})' for qb in range(2))) + ',\n      ' + \
        ', '.join(f'[{n}] "+&s"({n})' for n in ('s_koff', 's_voff', 's_cnt')) + ', [s_t] "=&s"(s_t), [s_slot] "=&s"(s_slot)' + \
        ((',\n      ' + ', '.join(f'[{n}] "+v"({n})' for n in addr_names)) if occ2 else '')
    if occ2:
        vin = ['voffK', 'voffV', 'voffQ0', 'voffQ1', 'limlane', 'negm0', 'negm1', 'onesv']
        sin = ['srdK', 'srdV', 'srdQ', 's_ldsw', 's_k32', 's_v32', 's_kstep', 's_vstep']
    else:
        vin = [f'q{qb}{ks}' for qb in range(2) for ks in range(4)] + ['seed0', 'seed1', 'ones2' if LSUM4 else 'ones'] + addr_names + ['voffK', 'voffV', 'limlane', 'negm0', 'negm1', 'neginf']
        sin = ['srdK', 'srdV', 's_ldsw', 's_k32', 's_v32', 's_kstep', 's_vstep']
    ins_ = ', '.join(f'[{n}] "v"({n})' for n in vin) + ',\n      ' + ', '.join(f'[{n}] "s"({n})' for n in sin)
    clob = ', '.join(f'"v{i}"' for i in range(64 if FRAG_ACC and not occ2 else 0, N_HAND)) + \
        (', ' + ', '.join(f'"a{i}"' for i in range(72, 136)) if FRAG_ACC and not occ2 else '') + \
        (', ' + ', '.join(f'"a{i}"' for i in range(A_Q, A_END)) if occ2 else '') + ', "vcc", "scc", "memory"'
    # the ONES operand of the row-sum MFMAs, built by the wrapper's compiler: all ones (4x4x4 form) or rows 1 / 2 on alternate k-groups (16x16x32 form)
    sel = '1' if LSUM4 else '(((lane & 15) == 1 && ((lane >> 4) & 1) == 0) || ((lane & 15) == 2 && ((lane >> 4) & 1) == 1))'
    if occ2:
        prelude = f'const uint32_t onesv = {sel} ? 0x3f803f80u : 0u;\n'
    elif LSUM4:
        prelude = 'bf16x4 ones2;\nfor (int j_ = 0; j_ < 4; ++j_) ones2[j_] = (__bf16)1.0f;\n'
    else:
        prelude = ''
    return ('// GENERATED by gen_attn_fwd4w.py -- do not edit; see that file for the register map and the schedule\n' +
            '#undef F4W_LSUM_VALU\n#undef F4W_LSUM4\n' + ('#define F4W_LSUM_VALU 1\n' if LSUM_VALU else '#define F4W_LSUM_VALU 0\n') + ('#define F4W_LSUM4 1\n' if LSUM4 else '#define F4W_LSUM4 0\n') +
            prelude + 'asm volatile(\n' + body + '\n    : ' + outs + '\n    : ' + ins_ + '\n    : ' + clob + ');\n')


FILES = ['attn_fwd4w_body.inc', 'attn_fwd2x_body.inc']


def generate_all(outdir):
    for name, occ2 in zip(FILES, (False, True)):
        stream = generate(occ2)
        with open(os.path.join(outdir, name), 'w') as f:
            f.write(render(stream, occ2))
        yield name, stream


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    for name, stream in generate_all(here):
        if '-v' in sys.argv:
            kinds = {}
            for ins in stream:
                kinds[ins.kind] = kinds.get(ins.kind, 0) + 1
            print(name, kinds)
