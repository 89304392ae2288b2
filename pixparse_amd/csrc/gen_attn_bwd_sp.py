#!/usr/bin/env python3
"""Generates attn_bwd_sp_body.inc: the hand-placed gfx950 instruction stream of the single-pass attention backward
(attention.hip: attn_bwd_spx_kernel; attn_bwd_sp_kernel is the C++ form of the same algorithm, its bit-exact reference).  One workgroup = 4 waves = 256 keys of one (batch, head), one wave per SIMD with the whole
512-entry register file; the stream is ONE inline-asm statement whose registers are allocated here:

  a[0:63]     dV^T accumulators  [kb][db] x 16        a[64:127]   dK^T accumulators
  a[128:159]  K row fragments    [kb][ks] x 4         a[160:191]  V row fragments
  a[192:255]  K^T fragments of the workgroup's 256 keys for this wave's dQ block  [16] x 4
  v[0:63]     S / dP accumulators, two sets (one per key block kb)
  v[64:95]    P / dS as bf16 MFMA operands, two sets
  v[96:127]   row constants (-lse log2e, -delta) of the current 32-query block: the C operand of the first MFMA of every S / dP chain
  v[128:159]  Q / dO row fragments of the current 32-query block (each feeds both key blocks)
  v[160:191]  Q^T / dO^T transposed fragments of the current 32-query block (each feeds both key blocks)
  v[192:207]  dS^T fragments for the dQ product       v[208:223]  dQ^T accumulator
  v[224:255]  left to the compiler: the 26 per-lane LDS / global offsets it passes in as operands (the K / V fragment and dK / dV store
              offsets are computed in the stream from the lane id: the wrapper's key-block loop needs the rest of these 32 registers)

Per 64-query tile a wave issues 80 MFMAs (32 S / dP, 32 dV / dK, 16 dQ of the PREVIOUS tile) as a fixed backbone; everything else --
114 LDS instructions, 64 v_exp, 64 multiplies, 72 conversions, 16 unpack operations, 7 LDS-DMA pieces, 2 slab stores (16 bytes per lane
after a v_permlane32_swap exchange), counted waits -- is assigned to one of the 80 MFMA gaps by the tables below.
CHAINS: the wrapper runs this statement once per key block of a chain (attention.hip: attn_bwd_spx_kernel); the dQ of a tile starts from
the running partial the previous key blocks of the chain left in the slab: DMA'd into a ring of three 8-KiB buffers above the dS buffers
two tiles ahead (dma_part: the youngest memory operations of a pass, the rendezvous leaves them in flight), read back at the end of the
pass before its use through addresses that undo the store's lane exchange (load_part), unpacked to fp32 in the first gaps of the next
pass (unpack_part) = the C operand of the tile's first dQ MFMA.  The first key block of a chain reads through a descriptor without records.  The loop is unrolled six-fold
(ring of 3 x dS double buffer): every LDS address of a pass is a per-lane base register + an immediate.  Software pipeline over 32 x 32 blocks n = (qb, kb):
  slot n:  MFMAs  S / dP of block n + 1  |  dV / dK of block n - 1  |  a quarter of dQ(t - 1)     VALU: exp / mul / cvt of block n
so no MFMA ever waits for VALU work of its own slot.  One s_barrier per tile (top of the iteration): dS of tile t - 1 complete, ring
slot of tile t - 1 and dS buffer of tile t - 2 free.  Q / dO tiles arrive by LDS-DMA two tiles ahead (ring of 3).  Behind the loop a
drain does what a further pass would still owe (dV / dK of the last block, dQ of the last tile); then dK / dV leave as bf16.

Issue model the placement is priced with (measured: DESIGN.md "Round 4"): a wave alone on its SIMD pays max(32, sum of filler issue
costs + ~5) cycles per MFMA gap -- v_exp ~8, plain VALU ~5, an LDS instruction ~9; nothing else hides anything.
Diagnostics: SPX_OPTS=stamps (with -DSPX_STAMPS: six s_memtime stamps per pass + four in the prologue, written out by scalar stores),
SPX_DROP=class,... (timing-only builds without an instruction class: WRONG results), SPX_OPTS=barrier_at=n; scripts/ab_spx.sh drives them.

The hazard pass below inserts what hipcc does not do for asm: counted s_waitcnt lgkmcnt for every LDS read before its first consumer,
s_nop between a VALU write and an MFMA read of the same register, and it CHECKS (does not fix) the MFMA-result -> VALU distance.
"""
import os
import sys

# ----------------------------------------------------------------------------------------------- registers
A_DV, A_DK, A_KF, A_VF, A_KA = 0, 64, 128, 160, 192
V_S = [0, 32]
V_D = [16, 48]
V_P = [64, 80]
V_DS = [72, 88]
V_SEEDL, V_SEEDD = 96, 112
V_QF, V_DOF = 128, 144
V_TF = 160
V_DSF = 192
V_DQ = 208
N_HAND = 224

SLOT = 16384 + 1024            # ring slot: Q tile | dO tile | 4 x 256 B row-constant vectors (waves 0 / 1: -lse', -delta; 2 / 3: unused copies)
RING = 3
DS_BASE = 65536                # two dS buffers of 32 KiB: 65536, 98304 (toggle = xor 0x8000); K is staged through the first one
PART_BASE = DS_BASE + 2 * 32768   # running partial dQ tiles of the chain: ring of three buffers (tile mod 3) x 4 waves x 2 pieces of 1 KiB
PBUF = 8192
LDS_BYTES = PART_BASE + 3 * PBUF


def vr(b, n=1):
    return f'v{b}' if n == 1 else f'v[{b}:{b + n - 1}]'


def ar(b, n=1):
    return f'a{b}' if n == 1 else f'a[{b}:{b + n - 1}]'


def regs(prefix, b, n):
    return [f'{prefix}{i}' for i in range(b, b + n)]


class I:
    """one instruction: text + what the hazard pass needs"""
    def __init__(self, text, kind, reads=(), writes=(), cost=4):
        self.text, self.kind, self.reads, self.writes, self.cost = text, kind, tuple(reads), tuple(writes), cost

    def __repr__(self):
        return self.text


def op(name):
    return f'%[{name}]'


# ----------------------------------------------------------------------------------------------- instruction builders
def mfma(d, a, b, c, dn, an, bn, cn):
    """d/a/b/c: (prefix, base); dn.. register counts; c may be the string '0'"""
    ds_ = (ar if d[0] == 'a' else vr)(d[1], 16)
    as_ = (ar if a[0] == 'a' else vr)(a[1], 4)
    bs_ = (ar if b[0] == 'a' else vr)(b[1], 4)
    if c == '0':
        cs, cr = '0', []
    else:
        cs, cr = (ar if c[0] == 'a' else vr)(c[1], 16), regs(c[0], c[1], 16)
    return I(f'v_mfma_f32_32x32x16_bf16 {ds_}, {as_}, {bs_}, {cs}', 'mfma',
             reads=regs(a[0], a[1], 4) + regs(b[0], b[1], 4) + cr, writes=regs(d[0], d[1], 16), cost=8)


def ds_read_b128(dst, addr_op, off):
    return I(f'ds_read_b128 {vr(dst, 4)}, {op(addr_op)} offset:{off}', 'ds', reads=[addr_op], writes=regs('v', dst, 4))


def ds_read_tr(dst_prefix, dst, addr_op, off):
    d = (ar if dst_prefix == 'a' else vr)(dst, 2)
    return I(f'ds_read_b64_tr_b16 {d}, {op(addr_op)} offset:{off}', 'ds', reads=[addr_op], writes=regs(dst_prefix, dst, 2))


def ds_write_b64(addr_op, src, off):
    return I(f'ds_write_b64 {op(addr_op)}, {vr(src, 2)} offset:{off}', 'ds', reads=[addr_op] + regs('v', src, 2))


def valu(text, reads, writes, cost=4, kind='valu'):
    return I(text, kind, reads=reads, writes=writes, cost=cost)


def v_exp(r):
    return valu(f'v_exp_f32 v{r}, v{r}', [f'v{r}'], [f'v{r}'], cost=8, kind='trans')


def v_mul(d, a, b):
    return valu(f'v_mul_f32 v{d}, v{a}, v{b}', [f'v{a}', f'v{b}'], [f'v{d}'])


def v_cvt(d, a, b):
    return valu(f'v_cvt_pk_bf16_f32 v{d}, v{a}, v{b}', [f'v{a}', f'v{b}'], [f'v{d}'])


def salu(text):
    return I(text, 'salu')


# ----------------------------------------------------------------------------------------------- the pieces of a tile
def m1_block(qb_unused, kb, first_c_seed=True):
    """S / dP of one 32 x 32 block into accumulator set kb: 8 MFMAs (Q / dO row fragments x K / V row fragments of key block kb)"""
    out = []
    for ks in range(4):
        cS = ('v', V_SEEDL) if ks == 0 else ('v', V_S[kb])
        cD = ('v', V_SEEDD) if ks == 0 else ('v', V_D[kb])
        out.append(mfma(('v', V_S[kb]), ('v', V_QF + 4 * ks), ('a', A_KF + 4 * (4 * kb + ks)), cS, 16, 4, 4, 16))
        out.append(mfma(('v', V_D[kb]), ('v', V_DOF + 4 * ks), ('a', A_VF + 4 * (4 * kb + ks)), cD, 16, 4, 4, 16))
    return out


def tf(ss, f):
    """transposed fragment register base: f = 0 dO^T d-block 0, 1 dO^T d-block 1, 2 Q^T d-block 0, 3 Q^T d-block 1"""
    return V_TF + 16 * ss + 4 * f


def m2_block(kb):
    """dV^T / dK^T of key block kb += (dO^T | Q^T) . (P | dS): 8 MFMAs"""
    out = []
    for ss in range(2):
        p, d = V_P[kb] + 4 * ss, V_DS[kb] + 4 * ss
        for db in range(2):
            acc = A_DV + 16 * (2 * kb + db)
            out.append(mfma(('a', acc), ('v', tf(ss, db)), ('v', p), ('a', acc), 16, 4, 4, 16))
        for db in range(2):
            acc = A_DK + 16 * (2 * kb + db)
            out.append(mfma(('a', acc), ('v', tf(ss, 2 + db)), ('v', d), ('a', acc), 16, 4, 4, 16))
    return out


def dq_quarter(w):
    """a quarter (64 keys) of dQ^T of the previous tile; the chain starts from the running partial of the key blocks before this one
    (load_part / unpack_part: zeros for the first key block of a chain)"""
    out = []
    for kk in range(4):
        c = '0' if (FIRST[0] and w == 0 and kk == 0) else ('v', V_DQ)
        out.append(mfma(('v', V_DQ), ('a', A_KA + 4 * (4 * w + kk)), ('v', V_DSF + 4 * kk), c, 16, 4, 4, 16))
    return out


def valu_block(qb, kb, par=0):
    """exp / mul / cvt of block (qb, kb) and the four dS stores; returns dict name -> list of instructions in dependency order"""
    S, D, P, DS = V_S[kb], V_D[kb], V_P[kb], V_DS[kb]
    ex = [v_exp(S + r) for r in range(16)]
    mu = [v_mul(D + r, S + r, D + r) for r in range(16)]
    cp = [v_cvt(P + j, S + 2 * j, S + 2 * j + 1) for j in range(8)]
    cd = [v_cvt(DS + j, D + 2 * j, D + 2 * j + 1) for j in range(8)]
    wr = []
    for ss in range(2):
        c0 = 4 * qb + 2 * ss
        wr.append(ds_write_b64(f'adsw{c0}', DS + 4 * ss, 4096 * kb + 32768 * par))
        wr.append(ds_write_b64(f'adsw{c0 + 1}', DS + 4 * ss + 2, 4096 * kb + 32768 * par))
    return dict(exp=ex, mul=mu, cvtp=cp, cvtd=cd, wr=wr)


def load_seeds(qb, ring=0):
    out = []
    for g in range(4):
        out.append(ds_read_b128(V_SEEDL + 4 * g, 'aseed', ring * SLOT + 128 * qb + 32 * g))
        out.append(ds_read_b128(V_SEEDD + 4 * g, 'aseed', ring * SLOT + 256 + 128 * qb + 32 * g))
    return out


def load_qf(qb, ring=0):
    out = []
    for ks in range(4):
        out.append(ds_read_b128(V_QF + 4 * ks, f'arow{ks}', ring * SLOT + 4096 * qb))
        out.append(ds_read_b128(V_DOF + 4 * ks, f'arow{ks}', ring * SLOT + 8192 + 4096 * qb))
    return out


def load_tf(qb, ss, ring=0):
    out = []
    off = ring * SLOT + (32 * qb + 16 * ss) * 128
    for f in range(4):
        db, tile = f & 1, (8192 if f < 2 else 0)
        out.append(ds_read_tr('v', tf(ss, f), f'atr{2 * db}', tile + off))
        out.append(ds_read_tr('v', tf(ss, f) + 2, f'atr{2 * db + 1}', tile + off))
    return out


def load_dsf(w, par=0):
    out = []
    for kk in range(4):
        out.append(ds_read_tr('v', V_DSF + 4 * kk, 'atrs0', 32768 * par + w * 8192 + kk * 2048))
        out.append(ds_read_tr('v', V_DSF + 4 * kk + 2, 'atrs1', 32768 * par + w * 8192 + kk * 2048))
    return out


def load_part(par):
    """the running partial of the tile whose dQ the NEXT pass accumulates, as the previous key block of the chain stored it (dq_store: 16
    bytes per lane AFTER the lane-half exchange), DMA'd into partial buffer `par` (tile mod 3) two passes ago.  Read back through addresses that undo the
    exchange: lanes 0..31 take their own first 8 bytes and those of lane + 32, lanes 32..63 the second 8 bytes of lane - 32 and their
    own (512 bytes apart: one ds_read2st64_b64 per piece)"""
    out = []
    for gp in range(2):
        dst = V_DQ + 8 * gp + 4
        o = 16 * par + 2 * gp
        out.append(I(f'ds_read2st64_b64 {vr(dst, 4)}, {op("apart")} offset0:{o} offset1:{o + 1}', 'ds', reads=['apart'], writes=regs('v', dst, 4)))
    return out


def unpack_part():
    """bf16 pairs -> the fp32 C operand of the first dQ MFMA, in place (sources v[b+4 : b+7] are consumed in order)"""
    out = []
    for gp in range(2):
        b = V_DQ + 8 * gp
        for j in range(4):
            src = b + 4 + j
            out.append(valu(f'v_lshlrev_b32 v{b + 2 * j}, 16, v{src}', [f'v{src}'], [f'v{b + 2 * j}']))
            out.append(valu(f'v_and_b32 v{b + 2 * j + 1}, 0xffff0000, v{src}', [f'v{src}'], [f'v{b + 2 * j + 1}']))
    return out


def dq_store():
    """dQ^T block -> bf16 in place -> slab (rows past Nq, and the whole tile -1, fall off the descriptor).  Pairs of 4-channel groups
    are exchanged between the lane halves (v_permlane32_swap, guide T21) so that every lane owns 16 contiguous bytes: two 16-byte
    stores per lane instead of four 8-byte ones (a store instruction costs the wave ~70 cycles here, whatever its width)."""
    cv, st = [], []
    for g in (0, 2):
        b = V_DQ + 4 * g
        cv.append(v_cvt(b, b, b + 1))
        cv.append(v_cvt(b + 1, b + 2, b + 3))
        cv.append(v_cvt(b + 2, b + 4, b + 5))
        cv.append(v_cvt(b + 3, b + 6, b + 7))
    for g in (0, 2):
        b = V_DQ + 4 * g
        cv.append(valu(f'v_permlane32_swap_b32 v{b}, v{b + 2}', [f'v{b}', f'v{b + 2}'], [f'v{b}', f'v{b + 2}'], kind='perm'))
        cv.append(valu(f'v_permlane32_swap_b32 v{b + 1}, v{b + 3}', [f'v{b + 1}', f'v{b + 3}'], [f'v{b + 1}', f'v{b + 3}'], kind='perm'))
        st.append(I(f'buffer_store_dwordx4 {vr(b, 4)}, {op("slabv")}, {op("rslab")}, {op("s_slaboff")} offen offset:{16 * g}\n\ts_nop 1', 'vmem',
                    reads=regs('v', b, 4) + ['slabv']))
    return cv, st


def dma_tile(ring):
    """the five LDS-DMA pieces of one wave for tile t + 2 into ring slot `ring`: two of Q, two of dO, one row-constant vector"""
    out = []
    def piece(setup, srd, voff, soff, width):
        ins = list(setup)
        ins.append(I(f's_nop 0\n\tbuffer_load_{width} {op(voff)}, {op(srd)}, {soff} offen lds', 'vmem', reads=[voff], cost=12))
        return ins
    base = ring * SLOT
    out += piece([salu(f's_add_u32 m0, {op("s_m0q")}, {base}')], 'rq', 'sqv', op('s_qoff'), 'dwordx4')
    out += piece([salu(f's_add_u32 {op("s_tmp1")}, {op("s_qoff")}, {op("s_q32")}'), salu(f's_add_u32 m0, {op("s_m0q")}, {base + 4096}')], 'rq', 'sqv', op('s_tmp1'), 'dwordx4')
    out += piece([salu(f's_add_u32 m0, {op("s_m0q")}, {base + 8192}')], 'rdo', 'sdov', op('s_dooff'), 'dwordx4')
    out += piece([salu(f's_add_u32 {op("s_tmp1")}, {op("s_dooff")}, {op("s_do32")}'), salu(f's_add_u32 m0, {op("s_m0q")}, {base + 12288}')], 'rdo', 'sdov', op('s_tmp1'), 'dwordx4')
    out += piece([salu(f's_add_u32 m0, {op("s_m0rc")}, {base}')], 'rrc', 'rcv', op('s_rcoff'), 'dword')
    return out


def dma_part(ring):
    """the running partial of tile t + 2 (this pass stores tile t - 1: three slab steps ahead of the store offset) into partial buffer
    `ring`: the two 16-byte pieces per lane exactly as dq_store wrote them.  rprev has zero records for the first key block of a chain
    (zeros arrive).  These loads come from HBM (a workgroup reads back what it wrote a whole key block earlier): they are the YOUNGEST
    memory operations of a pass, so that the rendezvous of the next pass (vmcnt(4): two slab stores + these two) does not wait for them --
    the pass after that does."""
    out = [salu(f's_add_u32 {op("s_tmp1")}, {op("s_slaboff")}, {op("s_slab3")}'), salu(f's_add_u32 m0, {op("s_m0p")}, {ring * PBUF}'),
           I(f's_nop 0\n\tbuffer_load_dwordx4 {op("slabv")}, {op("rprev")}, {op("s_tmp1")} offen lds', 'vmem', reads=['slabv'], cost=12),
           salu(f's_add_u32 {op("s_tmp1")}, {op("s_tmp1")}, 32'), salu(f's_add_u32 m0, {op("s_m0p")}, {ring * PBUF + 1024}'),
           I(f's_nop 0\n\tbuffer_load_dwordx4 {op("slabv")}, {op("rprev")}, {op("s_tmp1")} offen lds', 'vmem', reads=['slabv'], cost=12)]
    return out


# ----------------------------------------------------------------------------------------------- placement
class Gaps:
    def __init__(self, n):
        self.g = [[] for _ in range(n)]

    def put(self, gap, ins):
        self.g[gap].append(ins)

    def spread(self, lo, hi, inss, per_gap=None):
        """instructions in order, evenly over gaps lo..hi (inclusive)"""
        n = len(inss)
        if n == 0:
            return
        width = hi - lo + 1
        for j, ins in enumerate(inss):
            self.g[lo + (j * width) // n].append(ins)

    def fill(self, lo, hi, inss):
        """instructions in order into gaps lo..hi, cheapest gaps first: the lowest level L such that topping every gap up to L (left to
        right, order kept) takes them all"""
        for level in range(0, 400, 4):
            room = [max(0, (level - self.cost(g)) // 4) for g in range(lo, hi + 1)]
            if sum(room) >= len(inss):
                k = 0
                for g, r in zip(range(lo, hi + 1), room):
                    for _ in range(r):
                        if k < len(inss):
                            self.g[g].append(inss[k]); k += 1
                return
        raise RuntimeError('fill: no room')

    def cost(self, gap):
        return sum(i.cost for i in self.g[gap])


def build_iteration(p):
    """pass p of the six-fold unrolled loop (tile t = p mod 6): returns (backbone: 80 MFMAs, gaps: 80 lists of fillers); gap i = the
    instructions issued right after MFMA i.  Ring slot of tile t = p mod 3, dS buffer of tile t = p mod 2: every LDS address of the
    pass is a per-lane base register + an immediate."""
    bb = []
    G = Gaps(80)
    ring_t, ring_n, par_w, par_r = p % 3, (p + 1) % 3, p & 1, (p + 1) & 1
    for s in range(4):
        qb, kb = s >> 1, s & 1
        g0 = 20 * s
        nkb = 1 - kb                                  # key block of the next block (= accumulator set it uses)
        m1 = m1_block(None, nkb)
        m2 = m2_block(nkb)                            # previous block has the other kb as well
        dq = dq_quarter(s)
        if s == 3:
            bb += m1 + dq + m2                        # the last quarter of dQ early: its conversion + stores fit behind it in this slot
        else:
            bb += m1 + m2 + dq
        # ---- VALU of block (qb, kb): exp r in gap r, mul one gap later, conversions trailing
        vb = valu_block(qb, kb, par_w)
        for r in range(16):
            G.put(g0 + r, vb['exp'][r])
            G.put(g0 + r + 1, vb['mul'][r])
        for j in range(8):
            G.put(g0 + 2 * j + 3, vb['cvtp'][j])
            G.put(g0 + 2 * j + 4, vb['cvtd'][j])
        G.put(g0 + 11, vb['wr'][0]); G.put(g0 + 12, vb['wr'][1])
        G.put(g0 + 19, vb['wr'][2]); G.put(g0 + 19, vb['wr'][3])
        # ---- dS^T fragments of this slot's dQ quarter
        if s == 3:
            G.spread(g0 + 0, g0 + 6, load_dsf(s, par_r))     # used from MFMA 8 of the slot
        else:
            G.spread(g0 + 4, g0 + 11, load_dsf(s, par_r))    # used from MFMA 16
        if kb == 0:
            # row constants and row fragments of the NEXT 32-query block (this slot's M1 was the last user of the current ones)
            nqb = 1 - qb
            nring = ring_t if qb == 0 else ring_n     # slot 0 loads (t, qb 1), slot 2 loads (t + 1, qb 0)
            G.spread(g0 + 2, g0 + 9, load_seeds(nqb, nring))
            G.spread(g0 + 8, g0 + 15, load_qf(nqb, nring))
            # transposed fragments of THIS query block, first half (ss = 0: free once MFMA 11 of the slot has issued)
            G.spread(g0 + 12, g0 + 19, load_tf(qb, 0, ring_t))
        else:
            G.spread(g0 + 0, g0 + 6, load_tf(qb, 1, ring_t))  # ss = 1 (free since MFMA 15 of the previous slot); used from MFMA 12 / 16 of this slot
    # ---- slot 1: LDS-DMA of tile t + 2
    g0 = 20
    G.spread(g0 + 7, g0 + 16, dma_tile((p + 2) % 3))
    # ---- slot 0: the running partial of tile t - 1 -> C operand of its first dQ MFMA (MFMA 16 of the pass)
    # (its two LDS reads were issued in the last gap of the previous pass -- see below -- and are covered by the rendezvous)
    if not FIRST[0]:
        G.fill(0, 15, unpack_part())
    # ---- slot 3: dQ conversion + stores (dQ MFMAs are 8..11 of the slot), address toggles, ring advance of the transposed addresses
    g0 = 60
    cv, st = dq_store()
    G.spread(g0 + 14, g0 + 18, cv)
    for k in range(2):
        G.put(g0 + 19, st[k])
    # the running partial of tile t (DMA'd one pass ago, landed at this pass's rendezvous) for the NEXT pass, into the upper halves of the
    # accumulator block, which the conversions above have just released
    if not FIRST[0]:
        for ins in load_part(ring_t) + dma_part((p + 2) % 3):
            G.put(g0 + 19, ins)
    return bb, G


def scalar_tail():
    """per-pass scalar bookkeeping: the running DMA source offsets (tile t + 2 -> t + 3)"""
    return [salu(f's_add_u32 {op("s_qoff")}, {op("s_qoff")}, {op("s_qstep")}'),
            salu(f's_add_u32 {op("s_dooff")}, {op("s_dooff")}, {op("s_dostep")}'),
            salu(f's_add_u32 {op("s_rcoff")}, {op("s_rcoff")}, 256')]


# timing-only experiments (WRONG RESULTS): SPX_DROP=class,class,... removes instruction classes from the loop body
#   exp mul cvt dsread dswrite dma store barrier addr salu
DROP = set(filter(None, os.environ.get('SPX_DROP', '').split(',')))
OPTS = dict(kv.split('=') if '=' in kv else (kv, '1') for kv in filter(None, os.environ.get('SPX_OPTS', '').split(',')))
BARRIER_AT = int(OPTS.get('barrier_at', 0))     # number of MFMAs of a pass issued before its wait + s_barrier (0 = at the top)
STAMPS = 'stamps' in OPTS     # diagnostic build (-DSPX_STAMPS): six s_memtime stamps per pass, written to a trace by scalar stores
STAMP_AT = {19: 2, 39: 3, 59: 4}  # after MFMA n -> stamp k   (0: after the wait at the top, 1: after the barrier, 5: end of the pass)
in_loop = [False]
# FIRST[0]: generate the variant for the FIRST key block of a chain (attn_bwd_sp_body_first.inc, round 6).  Its running partial is zero by
# definition, so the whole running-tile machinery goes: no read-back (load_part), no unpack (16 VALU per pass), no LDS-DMA of the partial
# (2 pieces per pass); the tile's first dQ MFMA starts from the inline constant 0 -- the same +0.0 the unpacked zero bits gave, so dQ / dK / dV
# are bit-identical to the general form fed a descriptor without records (which is what round 4/5 ran for these blocks: +3 % of the stream
# for nothing on 7 of every 25 key blocks at chain 4).  The rendezvous then leaves 2 operations in flight (the slab stores), not 4.
FIRST = [False]


def drop_class(ins):
    t = ins.text
    if ins.kind == 'trans':
        return 'exp'
    if t.startswith('v_mul_f32'):
        return 'mul'
    if t.startswith('v_cvt_pk'):
        return 'cvt'
    if t.startswith('ds_read'):
        return 'dsread'
    if t.startswith('ds_write'):
        return 'dswrite'
    if 'lds' in t and 'buffer_load' in t:
        return 'dma'
    if t.startswith('buffer_store'):
        return 'store'
    if t.startswith('s_barrier'):
        return 'barrier'
    if t.startswith('v_add_u32') or t.startswith('v_xor_b32'):
        return 'addr'
    if ins.kind == 'salu' and 's_cnt' not in t and 'm0' not in t:
        return 'salu'
    return ins.kind


# ----------------------------------------------------------------------------------------------- hazard / wait pass
class Hazards:
    """walks a linear stream; inserts counted lgkmcnt waits and VALU->MFMA nops; checks MFMA->VALU distance"""
    def __init__(self):
        self.pending = []          # outstanding DS ops in issue order: (serial, set(written regs))
        self.serial = 0
        self.out = []
        self.recent_valu = []      # [(age, written regs)] of the last VALU instructions
        self.mfma_written = {}     # reg -> index (in MFMA count) of the MFMA that last wrote it
        self.n_mfma = 0
        self.last_trans = None     # registers written by the directly preceding transcendental

    def emit(self, ins):
        if DROP and in_loop[0] and drop_class(ins) in DROP:
            return
        used = set(ins.reads) | set(ins.writes)
        # ---- LDS results: wait for the youngest pending read that touches a used register
        need = None
        for k, (ser, wr) in enumerate(self.pending):
            if wr & used:
                need = k
        if need is not None:
            outstanding_after = len(self.pending) - 1 - need
            if not (in_loop[0] and 'wait' in DROP):
                self.out.append(I(f's_waitcnt lgkmcnt({min(outstanding_after, 15)})', 'wait'))
            self.pending = self.pending[need + 1:] if outstanding_after <= 15 else []
            self.recent_valu = []          # a wait is an instruction between the VALU and the consumer
            self.last_trans = None
        # ---- VALU result -> MFMA operand: two wait states
        if ins.kind in ('mfma', 'perm'):
            hazard = 0
            for age, wr in self.recent_valu:
                if wr & set(ins.reads):
                    hazard = max(hazard, 3 - age)
            if hazard > 0:
                self.out.append(I(f's_nop {hazard - 1}', 'nop'))
                self.recent_valu = [(a + hazard, w) for a, w in self.recent_valu]
        # ---- MFMA result -> VALU / LDS / store reading it: needs the MFMA (8 passes) finished; two later MFMAs guarantee that
        if ins.kind in ('valu', 'trans', 'ds', 'vmem', 'perm'):
            for r in ins.reads:
                if r in self.mfma_written and self.n_mfma - self.mfma_written[r] < 3:
                    raise RuntimeError(f'MFMA result {r} read too early by: {ins.text}')
            for r in ins.writes:
                if r in self.mfma_written and self.n_mfma - self.mfma_written[r] < 3 and ins.kind != 'ds':
                    raise RuntimeError(f'MFMA result {r} overwritten too early by: {ins.text}')
        # ---- transcendental result -> next VALU: one wait state (gfx940+ VALUTransUseHazard)
        if self.last_trans and ins.kind in ('valu', 'trans') and (self.last_trans & set(ins.reads)):
            self.out.append(I('s_nop 0', 'nop'))
        self.out.append(ins)
        # ---- bookkeeping
        self.recent_valu = [(a + 1, w) for a, w in self.recent_valu if a + 1 < 3]
        if ins.kind in ('valu', 'trans', 'perm'):
            self.recent_valu.append((0, set(ins.writes)))
        self.last_trans = set(ins.writes) if ins.kind == 'trans' else None
        if ins.kind == 'ds':
            self.serial += 1
            self.pending.append((self.serial, set(ins.writes)))
        if ins.kind == 'mfma':
            self.n_mfma += 1
            for r in ins.writes:
                self.mfma_written[r] = self.n_mfma
            for r in ins.writes:
                pass

    def drain(self, text):
        """an explicit full wait (lgkmcnt(0) inside `text`) resets the LDS bookkeeping"""
        self.out.append(I(text, 'wait'))
        self.pending = []
        self.recent_valu = []
        self.last_trans = None

    def settle_mfma(self):
        self.mfma_written = {}


# ----------------------------------------------------------------------------------------------- whole statement
VOPS = ['arow0', 'arow1', 'arow2', 'arow3', 'atr0', 'atr1', 'atr2', 'atr3', 'aseed'] + [f'adsw{i}' for i in range(8)] + \
       ['atrs0', 'atrs1', 'sqv', 'sdov', 'rcv', 'slabv', 'atrk0', 'atrk1', 'apart']
SRDS = ['rq', 'rdo', 'rrc', 'rk', 'rv', 'rdk', 'rdv', 'rslab', 'rprev']
SIN = ['s_qstep', 's_dostep', 's_q32', 's_do32', 's_slabstep', 's_dk32', 's_dv32', 's_dkscale', 's_iters', 's_m0q', 's_m0rc', 's_slab3', 's_m0p',
       's_key0', 's_nkm1', 's_krs2', 's_vrs2', 's_dkrs2', 's_dvrs2']
SRW = ['s_qoff', 's_dooff', 's_rcoff', 's_slaboff']
STMP = ['s_tmp1', 's_cnt']


def generate():
    H = Hazards()
    E = H.emit
    # ================================================================= prologue
    if STAMPS:
        H.out.append(I('s_memtime %[st2]', 'salu'))
    # K / V row fragments of this wave's 64 keys straight into the accumulator file (rows past Nk are clamped to the last key): issued FIRST,
    # their latency and that of the staging DMA runs under the ~200 register initialisations below
    # (per-lane offsets from the lane id in rcv = 4 lane: key row ki = lane & 31 (+ 32 kb), clamped to the last key; 16 bytes hh = lane >> 5.
    # Computed here, in registers the stream does not use yet, instead of being handed in: the compiler has 32 vector registers for ALL operands)
    for t in [f'v_lshrrev_b32 v0, 2, {op("rcv")}', 'v_and_b32 v0, 31, v0', f'v_lshrrev_b32 v1, 7, {op("rcv")}', 'v_lshlrev_b32 v1, 4, v1',
              f'v_add_u32 v2, {op("s_key0")}, v0', 'v_add_u32 v3, 32, v2', f'v_min_u32 v2, {op("s_nkm1")}, v2', f'v_min_u32 v3, {op("s_nkm1")}, v3',
              f'v_mul_lo_u32 v6, v2, {op("s_krs2")}', f'v_mul_lo_u32 v7, v3, {op("s_krs2")}',
              f'v_mul_lo_u32 v8, v2, {op("s_vrs2")}', f'v_mul_lo_u32 v9, v3, {op("s_vrs2")}',
              'v_add_u32 v6, v6, v1', 'v_add_u32 v7, v7, v1', 'v_add_u32 v8, v8, v1', 'v_add_u32 v9, v9, v1']:
        H.out.append(I(t, 'valu'))
    for kb in range(2):
        for ks in range(4):
            E(I(f'buffer_load_dwordx4 {ar(A_KF + 4 * (4 * kb + ks), 4)}, v{6 + kb}, {op("rk")}, 0 offen offset:{32 * ks}', 'vmem', reads=[f'v{6 + kb}']))
            E(I(f'buffer_load_dwordx4 {ar(A_VF + 4 * (4 * kb + ks), 4)}, v{8 + kb}, {op("rv")}, 0 offen offset:{32 * ks}', 'vmem', reads=[f'v{8 + kb}']))
    for i in range(128):
        E(valu(f'v_accvgpr_write_b32 a{i}, 0', [], [f'a{i}']))
    for b in (V_P[0], V_P[1]):
        for i in range(16):
            E(valu(f'v_mov_b32 v{b + i}, 0', [], [f'v{b + i}']))
    for i in range(32):
        E(valu(f'v_mov_b32 v{V_TF + i}, 0', [], [f'v{V_TF + i}']))
    for i in range(16):
        E(valu(f'v_mov_b32 v{V_DSF + i}, 0', [], [f'v{V_DSF + i}']))
    H.drain('s_waitcnt vmcnt(0)')        # K staging, tiles 0 and 1 (issued by the C++ part), the fragments above
    if STAMPS:
        H.out.append(I('s_memtime %[st3]', 'salu'))
    E(I('s_barrier', 'barrier'))
    # K^T fragments [32 d of this wave's d block][256 keys] from the staged K tiles
    for kk in range(16):
        off = (kk >> 2) * 8192 + (kk & 3) * 2048
        E(ds_read_tr('a', A_KA + 4 * kk, 'atrk0', off))
        E(ds_read_tr('a', A_KA + 4 * kk + 2, 'atrk1', off))
    for ins in load_seeds(0) + load_qf(0) + ([] if FIRST[0] else load_part(2)):     # pass 0 accumulates dQ of "tile -1" (dS = 0, stored nowhere): any seed will do
        E(ins)
    H.drain('s_waitcnt lgkmcnt(0)')
    if STAMPS:
        H.out.append(I('s_memtime %[st4]', 'salu'))
    for m in m1_block(None, 0):          # S / dP of block (tile 0, qb 0, kb 0)
        E(m)
    E(I('s_nop 7\n\ts_nop 7', 'nop'))
    H.settle_mfma()
    E(salu(f's_mov_b32 {op("s_cnt")}, {op("s_iters")}'))
    # ================================================================= loop, unrolled over six passes (ring of 3 x dS double buffer)
    if STAMPS:
        H.out.append(I('s_memtime %[st5]\n\ts_mov_b64 %[st0], 0\n\ts_mov_b64 %[st1], 0', 'salu'))
    H.out.append(I('LOOP%=:', 'label'))
    in_loop[0] = True
    Gs = []
    for p in range(6):
        bb, G = build_iteration(p)
        Gs.append(G)
        for k, ins in enumerate(scalar_tail()):
            G.put(72 + 2 * k, ins)
        # the slab offset advances BEFORE this pass's stores (they come in the last gaps): it starts two tiles back
        G.put(1, salu(f's_add_u32 {op("s_slaboff")}, {op("s_slaboff")}, {op("s_slabstep")}'))
        # the pass's rendezvous: tile t + 1 landed (the two slab stores of the previous pass may still be in flight), this wave's dS stores and
        # every LDS read of the previous pass are complete.  It sits BEHIND the first BARRIER_AT MFMAs of the pass (S / dP of the next block from
        # operands already in registers; their gaps hold VALU work only), so the wait overlaps matrix work instead of draining the pipe.
        def rendezvous():
            H.drain('s_waitcnt vmcnt(2) lgkmcnt(0)' if FIRST[0] else 's_waitcnt vmcnt(4) lgkmcnt(0)')
            if STAMPS:
                # the six stamps of the previous pass have arrived (lgkmcnt(0) above): write them out, then stamp this pass
                H.out.append(I('\n\t'.join(f's_store_dwordx2 %[st{k}], %[dbg{k}], %[s_dbgoff]' for k in range(6)) +
                               '\n\ts_add_u32 %[s_dbgoff], %[s_dbgoff], 8\n\ts_memtime %[st0]', 'salu'))
            E(I('s_barrier', 'barrier'))
            if STAMPS:
                H.out.append(I('s_memtime %[st1]', 'salu'))
        if BARRIER_AT == 0:
            rendezvous()
        for i, m in enumerate(bb):
            E(m)
            for ins in G.g[i]:
                E(ins)
            if i + 1 == BARRIER_AT:
                rendezvous()
            if STAMPS and i in STAMP_AT:
                H.out.append(I(f's_memtime %[st{STAMP_AT[i]}]', 'salu'))
        if STAMPS:
            H.out.append(I('s_memtime %[st5]', 'salu'))
        E(salu(f's_sub_u32 {op("s_cnt")}, {op("s_cnt")}, 1'))
        E(salu(f's_cmp_eq_u32 {op("s_cnt")}, 0'))
        E(I(f's_cbranch_scc1 DRAIN{p & 1}%=', 'branch'))
        if p == 5:
            E(I('s_branch LOOP%=', 'branch'))
    in_loop[0] = False
    # ================================================================= drain: what the pass after the last tile would still have to do --
    # dV / dK of the last block (operands are in registers) and dQ of the last tile -- without the 56 MFMAs and the VALU work it would
    # spend on a tile of zero rows.  The pass that follows pass p reads the dS buffer of parity p & 1: one copy per parity.
    for par in range(2):
        H.out.append(I(f'DRAIN{par}%=:', 'label'))
        H.drain('s_waitcnt vmcnt(2) lgkmcnt(0)' if FIRST[0] else 's_waitcnt vmcnt(4) lgkmcnt(0)')
        E(I('s_barrier', 'barrier'))
        E(salu(f's_add_u32 {op("s_slaboff")}, {op("s_slaboff")}, {op("s_slabstep")}'))
        H.settle_mfma()                       # the S / dP results of the last pass are not read any more
        for ins in load_dsf(0, par):
            E(ins)
        for m in m2_block(1):
            E(m)
        if not FIRST[0]:
            for ins in unpack_part():
                E(ins)
        for w in range(4):                    # one quarter of dS^T at a time (the fragment registers are shared); once per workgroup
            if w:
                for ins in load_dsf(w, par):
                    E(ins)
            for m in dq_quarter(w):
                E(m)
        E(I('s_nop 7\n\ts_nop 7', 'nop'))
        H.settle_mfma()
        cv, st = dq_store()
        for ins in cv + st:
            E(ins)
        if par == 0:
            E(I('s_branch DONE%=', 'branch'))
    H.out.append(I('DONE%=:', 'label'))
    G = Gs[0]
    # ================================================================= epilogue: dK (scaled), dV -> bf16 -> bounds-checked stores
    H.drain('s_waitcnt vmcnt(0) lgkmcnt(0)')
    if STAMPS:
        H.out.append(I('\n\t'.join(f's_store_dwordx2 %[st{k}], %[dbg{k}], %[s_dbgoff]' for k in range(6)) + '\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)', 'salu'))
    E(I('s_nop 7\n\ts_nop 7', 'nop'))
    H.settle_mfma()
    # per-lane store offsets (16 bytes per lane): row key0 + ki of dK / dV, channel 8 hh; rows past Nk fall off the descriptors
    for t in [f'v_lshrrev_b32 v216, 2, {op("rcv")}', 'v_and_b32 v216, 31, v216', f'v_lshrrev_b32 v218, 7, {op("rcv")}', 'v_lshlrev_b32 v218, 4, v218',
              f'v_add_u32 v216, {op("s_key0")}, v216', f'v_mul_lo_u32 v217, v216, {op("s_dvrs2")}', f'v_mul_lo_u32 v216, v216, {op("s_dkrs2")}',
              'v_add_u32 v216, v216, v218', 'v_add_u32 v217, v217, v218']:
        H.out.append(I(t, 'valu'))
    tmp = 0
    for which, abase, voff, srd, s32 in (('dk', A_DK, 'v216', 'rdk', 's_dk32'), ('dv', A_DV, 'v217', 'rdv', 's_dv32')):
        for kb in range(2):
            for db in range(2):
                acc = abase + 16 * (2 * kb + db)
                for g in (0, 2):              # groups g, g + 1 -> bf16 -> exchanged between the lane halves -> one 16-byte store per lane
                    r = tmp
                    tmp = (tmp + 8) % 216
                    for j in range(8):
                        E(valu(f'v_accvgpr_read_b32 v{r + j}, a{acc + 4 * g + j}', [f'a{acc + 4 * g + j}'], [f'v{r + j}']))
                    if which == 'dk':
                        for j in range(8):
                            E(valu(f'v_mul_f32 v{r + j}, {op("s_dkscale")}, v{r + j}', [f'v{r + j}'], [f'v{r + j}']))
                    E(v_cvt(r, r, r + 1))
                    E(v_cvt(r + 1, r + 2, r + 3))
                    E(v_cvt(r + 2, r + 4, r + 5))
                    E(v_cvt(r + 3, r + 6, r + 7))
                    E(valu(f'v_permlane32_swap_b32 v{r}, v{r + 2}', [f'v{r}', f'v{r + 2}'], [f'v{r}', f'v{r + 2}'], kind='perm'))
                    E(valu(f'v_permlane32_swap_b32 v{r + 1}, v{r + 3}', [f'v{r + 1}', f'v{r + 3}'], [f'v{r + 1}', f'v{r + 3}'], kind='perm'))
                    soff = op(s32) if kb else '0'
                    E(I(f'buffer_store_dwordx4 {vr(r, 4)}, {voff}, {op(srd)}, {soff} offen offset:{64 * db + 16 * g}\n\ts_nop 1', 'vmem', reads=regs('v', r, 4)))
    return H.out, G


def render(stream, first_stream=None):
    """one asm statement; with first_stream: BOTH variants in it, selected by the scalar operand s_first (1 = first key block of a chain) -- one
    operand set for the compiler (two statements behind an if / else made hipcc spill 112 bytes per lane around them)"""
    lines = []
    if first_stream is not None:
        lines += ['s_cmp_eq_u32 %[s_first], 1', 's_cbranch_scc1 FIRSTV%=']
    for ins in stream:
        for ln in ins.text.split('\n\t'):
            lines.append(ln)
    if first_stream is not None:
        lines += ['s_branch BOTHDONE%=', 'FIRSTV%=:']
        for ins in first_stream:
            for ln in ins.text.split('\n\t'):
                for lab in ('LOOP%=', 'DRAIN0%=', 'DRAIN1%=', 'DONE%='):
                    ln = ln.replace(lab, 'F_' + lab)
                lines.append(ln)
        lines += ['BOTHDONE%=:']
    body = '\n'.join(f'    "{ln}\\n\\t"' for ln in lines)
    srw = SRW + (['s_dbgoff'] if STAMPS else [])
    outs = ', '.join(f'[{n}] "+&v"({n})' for n in VOPS) + ',\n      ' + ', '.join(f'[{n}] "+&s"({n})' for n in srw) + ',\n      ' + \
        ', '.join(f'[{n}] "=&s"({n})' for n in STMP + ([f'st{k}' for k in range(6)] if STAMPS else []))
    ins_ = ', '.join(f'[{n}] "s"({n})' for n in SRDS) + ',\n      ' + ', '.join(f'[{n}] "s"({n})' for n in SIN + (['s_first'] if first_stream is not None else []) + ([f'dbg{k}' for k in range(6)] if STAMPS else []))
    clob = ', '.join(f'"v{i}"' for i in range(N_HAND)) + ',\n      ' + ', '.join(f'"a{i}"' for i in range(256)) + ', "vcc", "scc", "memory"'
    return ('// GENERATED by gen_attn_bwd_sp.py -- do not edit; see that file for the register map and the schedule\n'
            'asm volatile(\n' + body + '\n    : ' + outs + '\n    : ' + ins_ + '\n    : ' + clob + ');\n')


def stats(stream, G):
    kinds = {}
    for ins in stream:
        kinds[ins.kind] = kinds.get(ins.kind, 0) + 1
    costs = [G.cost(i) for i in range(80)]
    return kinds, costs


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    FIRST[0] = True
    first_stream, _ = generate()
    FIRST[0] = False
    stream, G = generate()
    with open(os.path.join(here, 'attn_bwd_sp_body.inc'), 'w') as f:
        f.write(render(stream, None if 'nofirst' in OPTS else first_stream))
    kinds, costs = stats(stream, G)
    if '-v' in sys.argv:
        print(kinds)
        print('filler issue cycles per MFMA gap (budget 24):')
        for s in range(4):
            print('  slot', s, costs[20 * s:20 * s + 20])
