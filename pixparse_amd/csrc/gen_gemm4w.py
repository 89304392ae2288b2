#!/usr/bin/env python3
"""Generates gemm4w_body_{nt,nn,tn}.inc: the hand-placed gfx950 main loop of the one-wave-per-SIMD bf16 GEMM (gemm4w.hip).

Workgroup = 256x256 output tile, 256 threads = 4 waves, ONE WAVE PER SIMD with the whole 512-entry register file; wave (wr, wc) owns the
128x128 quadrant (rows 128 wr.., columns 128 wc..): 64 accumulator tiles of 16x16 = a[0:255], tile (i, j) = a[4 (8 i + j) : +3], lane
(li = lane & 15, lq = lane >> 4) holding C[16 i + li][16 j + 4 lq + 0..3].  Per K tile of 64 a wave issues 128 v_mfma_f32_16x16x32_bf16
and reads ONE A half-tile + ONE B half-tile from LDS: 32 ds_read_b128 per 128 MFMAs -- 1.5 x fewer LDS bytes per FLOP than the 8-wave
kernel of gemm256.hip (24 per 64), the lever rule 28 of the guide ranks for a loop that is held down by power, and the structure of the
vendor kernel that sustains 1.64 PF/s on the box where gemm256 sustains 1.24 (profiles/r5_yardstick.txt).

Registers (hand-allocated: v[0 : N_HAND); everything above is left to the compiler):
  v[0:31]   A fragments A_i (i = 0..7: rows 16 i.. of the wave's quadrant) of the current k-step of 32
  v[32:63]  B fragments B_j (j = 0..7)
  v[64:..]  the fragment-read base addresses + 64 KiB and + 128 KiB (a ds_read offset has 16 bits, the ring is 160 KiB)
  a[0:255]  accumulators (physical-register asm outputs: the C++ epilogue reads them)
ONE fragment set: the MFMA order of a k-step is  [i = 0..7] x [j = 0..3]  then  [i = 0..7] x [j = 4..7], so A_i is free after MFMA 35 + 4 i,
B_j (j < 4) after MFMA 28 + j and B_j (j >= 4) after MFMA 60 + j - 4; each fragment of the NEXT k-step is re-read right behind the last use of
its register, 22..31 MFMAs (>= 350 cycles) before its first use.

LDS = a ring of FIVE pair-slots of 32 KiB (all 160 KiB): the A pair [rows 0..127 | rows 128..255] and the B pair [cols 0..127 | cols 128..255] of a
K tile are two consecutive ring units, A(t) in slot 2 t mod 5, B(t) in slot (2 t + 1) mod 5; half-tile images as in gemm256.hip (KM: [128 rows][64
k], 16-byte slots XOR-swizzled; TR: [64 k][128 cols] for k-major operands, read with ds_read_b64_tr_b16).  The slot pattern repeats every five K
tiles: the loop is unrolled five-fold so that every LDS address is a per-lane base register + an immediate.  Staging is LDS-DMA
(buffer_load_dwordx4 ... lds), 16 pieces of 1 KiB per wave and K tile, issued EVENLY over the K tile: measured (profiles/r5_gemm4w_*.txt), a
burst of pieces from all four waves behind a barrier saturates the CU's address path and stalls the only wave each SIMD has (1.0 PF/s), the
same 16 pieces spread out cost 18 % of the MFMA-only rate (1.46 vs 1.82 PF/s at 8192^3).  Two K tiles of look-ahead:
  tile t, k-step 0:  the 8 A pieces of tile t + 2 -> slot of B(t - 1), released at BETA of tile t - 1
  tile t, k-step 1:  [ALPHA, gap 6: every wave has read A(t)]  the 8 B pieces of tile t + 2 -> slot of A(t);
                     BETA, gap 31: B(t) is read, tile t + 1 has landed (counted vmcnt: the pieces of tile t + 2 stay in flight) -> reads of tile t + 1
Every piece has >= 99 MFMAs (~1.7 k cycles) to land, most have 160+.  K tiles past the end of the contraction arrive as zeros (the
descriptors' num_records are zeroed in the stream: s_live counts the tiles left to stage).

The statement is entered with K tiles 0 and 1 of the output tile staged by the C++ wrapper (it issues them under the previous tile's epilogue)
and left with every accumulator complete, no memory operation outstanding and every wave behind a barrier (the LDS is free).
Diagnostics: G4W_DROP=class,... (timing-only builds, WRONG results: dsread dma barrier vmwait), G4W_OPTS=alpha=0 (no ALPHA barrier: the B pieces
follow BETA), G4W_OPTS=nostage0 (overlapped form, timing only: no staging of K tiles 0 and 1 at the entry -- the ceiling of a cross-tile prefetch,
profiles/r6_gemm_tile_boundary_pricing.txt); scripts/ab_g4w.sh builds such variants into separate libraries.
"""
import os
import sys

FA, FB, VX = 0, 32, 64
KINDS = {'nt': ('km', 'km'), 'nn': ('km', 'tr'), 'tn': ('tr', 'tr')}     # (A, B): km = k-contiguous rows, tr = k-major (transposed read)
SRD = {'A': 36, 'B': 40}          # pinned SGPR tuples s[36:39], s[40:43] (the stream zeroes word 2 = num_records at the end of the contraction)
OPTS = dict(kv.split('=') if '=' in kv else (kv, '1') for kv in filter(None, os.environ.get('G4W_OPTS', '').split(',')))
DROP = set(filter(None, os.environ.get('G4W_DROP', '').split(',')))
USE_ALPHA = OPTS.get('alpha', '1') != '0'
ALPHA_GAP, BETA_GAP = 6, 31
A_GAPS = [2, 10, 18, 26, 34, 42, 50, 58]                       # k-step 0: the A pieces of tile t + 2
B_GAPS = [8, 14, 20, 26, 32, 38, 44, 50] if USE_ALPHA else [32, 36, 40, 44, 48, 52, 56, 60]      # k-step 1: the B pieces of tile t + 2
NB_BEFORE_BETA = sum(1 for g in B_GAPS if g < BETA_GAP)
RING = 5


def op(name):
    return f'%[{name}]'


def vregs(b, n):
    return {f'v{i}' for i in range(b, b + n)}


class I:
    def __init__(self, text, kind, reads=(), writes=(), tag=None):
        self.text, self.kind, self.reads, self.writes, self.tag = text, kind, set(reads), set(writes), tag


def acc(i, j):
    return 4 * (8 * i + j)


def mfma(i, j, zero_c):
    a = acc(i, j)
    c = '0' if zero_c else f'a[{a}:{a + 3}]'
    return I(f'v_mfma_f32_16x16x32_bf16 a[{a}:{a + 3}], v[{FB + 4 * j}:{FB + 4 * j + 3}], v[{FA + 4 * i}:{FA + 4 * i + 3}], {c}', 'mfma',
             reads=vregs(FA + 4 * i, 4) | vregs(FB + 4 * j, 4))


def kstep_order():
    return [(i, j) for i in range(8) for j in range(4)] + [(i, j) for i in range(8) for j in range(4, 8)]


def slot_of(opnd, t):
    return (2 * t + (1 if opnd == 'B' else 0)) % RING


class Gen:
    def __init__(self, layout, ovl=False, cs=False):
        self.layout = layout
        self.ovl = ovl
        self.cs = cs          # weight-gradient layout only: also the column sums of the A operand (= the bias gradient), see colsum_block
        self.kind = dict(zip('AB', KINDS[layout]))
        # base operands handed in by the wrapper (ring offset 0) and their hand-allocated copies at + 64 KiB / + 128 KiB
        self.bases = []
        for o in 'AB':
            self.bases += [f'ar{o}k0', f'ar{o}k1'] if self.kind[o] == 'km' else [f'ar{o}t{j}' for j in range(8)]
        self.hi = {}
        r = VX
        for n in self.bases:
            self.hi[n] = (r, r + 1)
            r += 2
        # overlapped epilogue: the previous output tile's results, packed to bf16 -- tile (i, j) = 2 registers at P + 2 (8 i + j)
        self.P = (r + 3) & ~3
        self.n_hand = self.P + 128 if ovl else r
        if cs:
            assert layout == 'tn' and not ovl
            self.ONES = (r + 3) & ~3          # a fragment of bf16 ones
            self.CS = self.ONES + 4           # eight column-sum accumulators (one per A fragment i): asm OUTPUTS, not clobbers
            self.n_hand = self.CS

    def addr(self, name, slot):
        """(register text, immediate part) of ring slot `slot` for base operand `name`"""
        k = slot >> 1
        reg = op(name) if k == 0 else f'v{self.hi[name][k - 1]}'
        return reg, (slot & 1) * 32768

    def entry_code(self):
        out = []
        for n in self.bases:
            out.append(I(f'v_add_u32 v{self.hi[n][0]}, 0x10000, {op(n)}', 'valu'))
            out.append(I(f'v_add_u32 v{self.hi[n][1]}, 0x20000, {op(n)}', 'valu'))
        if self.cs:
            out += [I(f'v_mov_b32 v{self.ONES + r}, 0x3f803f80', 'valu') for r in range(4)]
            out += [I(f'v_mov_b32 v{self.CS + r}, 0', 'valu') for r in range(32)]
        return out

    # ---- fragment reads of K tile t (ring position), k-step ks
    def read_frag(self, opnd, idx, ks, t, tag):
        base = (FA if opnd == 'A' else FB) + 4 * idx
        slot = slot_of(opnd, t % RING)
        if self.kind[opnd] == 'km':
            reg, imm = self.addr(f'ar{opnd}k{ks}', slot)
            return [I(f'ds_read_b128 v[{base}:{base + 3}], {reg} offset:{imm + idx * 2048}', 'ds', writes=vregs(base, 4), tag=tag)]
        reg, imm = self.addr(f'ar{opnd}t{idx}', slot)
        off = imm + ks * 8192
        return [I(f'ds_read_b64_tr_b16 v[{base}:{base + 1}], {reg} offset:{off}', 'ds', writes=vregs(base, 2), tag=tag),
                I(f'ds_read_b64_tr_b16 v[{base + 2}:{base + 3}], {reg} offset:{off + 1024}', 'ds', writes=vregs(base + 2, 2), tag=tag)]

    def colsum_block(self, n):
        """Bias gradient on the matrix pipe (round 6).  In the weight-gradient layout the A operand is dY, k-major: its column sums over the tokens are
        the bias gradient of the Linear.  MFMA with a fragment of ONES in the other operand: D[m][n] = sum_k A[k][n] for every m, i.e. lane (li, lq)
        ends up with the sum of row 16 i + li of the wave's A half in all four registers of accumulator i.  Eight extra MFMAs per k-step, issued in
        front of the first re-read of an A fragment (gap 35: every A_i of the current k-step is still in its registers), and only on the K tiles
        this wave is on duty for (s_csgo: one K tile in `period` = 2 x (column tiles sharing the A operand), so that the two waves and the <= 4
        workgroups that read the same A panel split the work: +1.6 % MFMAs per wave instead of +12.5 %)."""
        out = [I(f's_cmp_eq_u32 {op("s_csgo")}, 1', 'salu'), I(f's_cbranch_scc0 CSK{n}_%=', 'branch')]
        for i in range(8):
            c = self.CS + 4 * i
            out.append(I(f'v_mfma_f32_16x16x32_bf16 v[{c}:{c + 3}], v[{self.ONES}:{self.ONES + 3}], v[{FA + 4 * i}:{FA + 4 * i + 3}], v[{c}:{c + 3}]', 'mfma',
                         reads=vregs(FA + 4 * i, 4)))
        out.append(I(f'CSK{n}_%=:', 'label'))
        return out

    def colsum_duty(self):
        """once per K tile: on duty iff the phase counter is zero; the counter runs down modulo the period"""
        return [I(f's_cmp_eq_u32 {op("s_csk")}, 0', 'salu'), I(f's_cselect_b32 {op("s_csgo")}, 1, 0', 'salu'),
                I(f's_sub_u32 {op("s_csk")}, {op("s_csk")}, 1', 'salu'), I(f's_and_b32 {op("s_csk")}, {op("s_csk")}, {op("s_csmask")}', 'salu')]

    def vops(self):
        return self.bases + ['voffA', 'voffB'] + (['voffBias', 'voffC', 'colv'] if self.ovl else [])

    # ---- overlapped epilogue of the PREVIOUS output tile (plain bf16: bias, column scale, bf16 rounding; gemm_epilogue.h operation for operation)
    def readout(self):
        """serial, at statement entry, while the staging DMA of this tile's first two K tiles is in flight: the lane's 32 bias values (rounded to
        bf16 like autocast) into v[0:31], its 8 column scales into v[32 + 2 j], then accumulators -> + bias -> x scale -> bf16 pairs in P"""
        out = []
        for j in range(8):
            out.append(I(f'buffer_load_dwordx4 v[{4 * j}:{4 * j + 3}], {op("voffBias")}, {op("srdBias")}, 0 offen offset:{64 * j}', 'vmem', tag='bias'))
        return out

    def readout_math(self):
        out = [I(f'v_mov_b32 v56, {op("s_cscale")}', 'valu')]
        for j in range(8):
            out += [I(f'v_add_u32 v57, {16 * j}, {op("colv")}', 'valu'), I(f'v_cmp_gt_u32 vcc, {op("s_cscols")}, v57', 'valu'),
                    I(f'v_cndmask_b32 v{32 + 2 * j}, 1.0, v56, vcc', 'valu')]
        for q in range(32):      # round_bf(bias)
            out += [I(f'v_cvt_pk_bf16_f32 v57, v{q}, v{q}', 'valu'), I(f'v_lshlrev_b32 v{q}, 16, v57', 'valu')]
        for i in range(8):
            for j in range(8):
                n = 8 * i + j
                t = 48 + 4 * (n & 1)
                a = acc(i, j)
                out += [I(f'v_accvgpr_read_b32 v{t + r}, a{a + r}', 'valu') for r in range(4)]
                out += [I(f'v_pk_add_f32 v[{t}:{t + 1}], v[{t}:{t + 1}], v[{4 * j}:{4 * j + 1}]', 'valu'),
                        I(f'v_pk_add_f32 v[{t + 2}:{t + 3}], v[{t + 2}:{t + 3}], v[{4 * j + 2}:{4 * j + 3}]', 'valu'),
                        I(f'v_pk_mul_f32 v[{t}:{t + 1}], v[{t}:{t + 1}], v[{32 + 2 * j}:{33 + 2 * j}] op_sel_hi:[1,0]', 'valu'),
                        I(f'v_pk_mul_f32 v[{t + 2}:{t + 3}], v[{t + 2}:{t + 3}], v[{32 + 2 * j}:{33 + 2 * j}] op_sel_hi:[1,0]', 'valu'),
                        I(f'v_cvt_pk_bf16_f32 v{self.P + 2 * n}, v{t}, v{t + 1}', 'valu'),
                        I(f'v_cvt_pk_bf16_f32 v{self.P + 2 * n + 1}, v{t + 2}, v{t + 3}', 'valu')]
        return out

    def store_unit(self, u):
        """row group i = u >> 2, strip pair pr = u & 3: the two 16-column strips of the pair exchange their odd / even 16-lane rows
        (swap_strips of gemm_epilogue.h) so that a lane owns 8 consecutive columns -> one 16-byte store"""
        r = self.P + 4 * u
        out = [I(f'v_permlane16_swap_b32 v{r}, v{r + 2}', 'valu'), I(f'v_permlane16_swap_b32 v{r + 1}, v{r + 3}', 'valu'),
               I(f'buffer_store_dwordx4 v[{r}:{r + 3}], {op("voffC")}, {op("srdC")}, {op("s_crow")} offen offset:{64 * (u & 3)}', 'vmem', tag='store')]
        if (u & 3) == 3:
            out.append(I(f's_add_u32 {op("s_crow")}, {op("s_crow")}, {op("s_cstep")}', 'salu'))
        return out

    # ---- the 8 LDS-DMA pieces of one operand pair of K tile t (ring position): half h at + 16 KiB, piece `it` at + 4 KiB
    def dma_group(self, opnd, t, vtag=None):
        out = []
        srd = SRD[opnd]
        slot = slot_of(opnd, t % RING)
        for h in range(2):
            for it in range(4):
                m0 = slot * 32768 + h * 16384 + it * 4096
                piece = [I(f's_add_u32 m0, {op("s_ldsw")}, {m0}', 'salu')]
                if h == 0 and it == 0:
                    piece.append(I('s_nop 0', 'salu'))
                    soff = op(f's_off{opnd}')
                else:
                    if it == 0:
                        piece.append(I(f's_add_u32 {op("s_t")}, {op(f"s_off{opnd}")}, {op(f"s_half{opnd}")}', 'salu'))
                    elif it == 1 and h == 0:
                        piece.append(I(f's_add_u32 {op("s_t")}, {op(f"s_off{opnd}")}, {op(f"s_it{opnd}")}', 'salu'))
                    else:
                        piece.append(I(f's_add_u32 {op("s_t")}, {op("s_t")}, {op(f"s_it{opnd}")}', 'salu'))
                    soff = op('s_t')
                piece.append(I(f'buffer_load_dwordx4 {op(f"voff{opnd}")}, s[{srd}:{srd + 3}], {soff} offen lds', 'dma', tag=vtag))
                out.append(piece)
        return out

    # ---- one K tile = two k-steps of 64 MFMAs + fillers.  t = ring position (tile index mod 5), tag = running tile number (hazard tags)
    def tile(self, t, tag, first, fillers=None):
        """fillers: {(ks, gap): [instructions]} -- the overlapped epilogue pieces placed in this tile"""
        out = []
        for ks in range(2):
            bb = [mfma(i, j, first and ks == 0) for (i, j) in kstep_order()]
            gaps = [[] for _ in range(64)]
            nt, nks, ntag = (t, 1, tag) if ks == 0 else (t + 1, 0, tag + 1)
            for jj, j in enumerate(range(4, 8)):                    # B_j (j >= 4) of THIS k-step (their registers were busy until MFMA 60.. of the previous one)
                gaps[1 + 4 * jj] += self.read_frag('B', j, ks, t, (tag, 'B'))
            for j in range(4):
                gaps[33 + 4 * j] += self.read_frag('B', j, nks, nt, (ntag, 'B'))
            if self.cs:
                if ks == 0:
                    gaps[0] += self.colsum_duty()
                self.ncs = getattr(self, 'ncs', 0) + 1
                gaps[35] += self.colsum_block(self.ncs)
            for i in range(8):
                gaps[35 + 4 * i] += self.read_frag('A', i, nks, nt, (ntag, 'A'))
            if ks == 0:
                # the A pair of tile t + 2 -> the slot B(t - 1) left at BETA of the previous tile
                gaps[A_GAPS[0] - 1] += [I(f's_cmp_gt_i32 {op("s_live")}, 0', 'salu'),
                                        I(f's_cselect_b32 s{SRD["A"] + 2}, s{SRD["A"] + 2}, 0', 'salu')]
                for g, piece in zip(A_GAPS, self.dma_group('A', t + 2, tag + 2)):
                    gaps[g] += piece
                gaps[A_GAPS[-1]] += [I(f's_add_u32 {op("s_offA")}, {op("s_offA")}, {op("s_ktA")}', 'salu')]
            else:
                pre = [I(f's_cmp_gt_i32 {op("s_live")}, 0', 'salu'),
                       I(f's_cselect_b32 s{SRD["B"] + 2}, s{SRD["B"] + 2}, 0', 'salu'),
                       I(f's_sub_u32 {op("s_live")}, {op("s_live")}, 1', 'salu')]
                if USE_ALPHA:
                    # ALPHA: the A pair of this tile is read by every wave -> the B pair of tile t + 2 goes into its slot
                    gaps[ALPHA_GAP] += [I('WAIT_TAG', 'waittag', tag=[(tag, 'A')])] + pre + [I('s_barrier', 'barrier')]
                # BETA: the B pair is read too, the next tile has landed (the pieces of tile t + 2 issued so far stay in flight)
                gaps[BETA_GAP] += [I('WAIT_TAG', 'waittag', tag=[(tag, 'B'), (tag, 'A')]),
                                   I(f's_waitcnt vmcnt({8 + NB_BEFORE_BETA})', 'vmwait', tag=tag + 1)] + ([] if USE_ALPHA else pre) + [I('s_barrier', 'barrier')]
                for g, piece in zip(B_GAPS, self.dma_group('B', t + 2, tag + 2)):
                    gaps[g] += piece
                gaps[B_GAPS[-1]] += [I(f's_add_u32 {op("s_offB")}, {op("s_offB")}, {op("s_ktB")}', 'salu')]
            for (fks, g), inss in (fillers or {}).items():
                if fks == ks:
                    gaps[g] += inss
            out.append((bb, gaps))
        return out


class Hazards:
    """linear walk: counted lgkmcnt waits for LDS reads before their consumers (merged: a wait also covers every older-than-OLD read)"""
    OLD = 12      # a read issued >= OLD MFMAs ago has certainly returned: waiting for it costs nothing
    in_loop = False

    def __init__(self):
        self.out = []
        self.pending = []      # (writes, tag, mfma index at issue)
        self.n_mfma = 0
        self.vmq = []          # tags of the vector-memory operations issued since the last full drain, in issue order (vmcnt retires in order)
        self.exact_vm = False  # peeled code: compute counted vmcnt waits from the queue; loop body: the steady-state constant

    def _wait_for(self, k):
        """wait until pending[0..k] have returned; extend k over reads that are old anyway"""
        while k + 1 < len(self.pending) and self.n_mfma - self.pending[k + 1][2] >= self.OLD:
            k += 1
        after = len(self.pending) - 1 - k
        self.out.append(I(f's_waitcnt lgkmcnt({min(after, 15)})', 'wait'))
        self.pending = self.pending[k + 1:] if after <= 15 else []

    def emit(self, ins):
        cls = {'ds': 'dsread', 'dma': 'dma', 'barrier': 'barrier', 'vmwait': 'vmwait'}.get(ins.kind)
        if cls in DROP and self.in_loop:
            return
        if ins.kind == 'waittag':
            need = None
            for k, (wr, tag, _) in enumerate(self.pending):
                if tag in ins.tag:
                    need = k
            if need is not None:
                self._wait_for(need)
            return
        if ins.kind == 'vmwait' and self.exact_vm:
            # all but the operations issued after the last one tagged ins.tag may stay in flight (stores of the overlapped epilogue included)
            idx = [k for k, t in enumerate(self.vmq) if t == ins.tag]
            if not idx:
                return
            n = len(self.vmq) - 1 - idx[-1]
            assert n <= 63
            self.out.append(I(f's_waitcnt vmcnt({n})', 'vmwait'))
            self.vmq = self.vmq[len(self.vmq) - n:] if n else []
            return
        if ins.kind in ('dma', 'vmem'):
            self.vmq.append(ins.tag)
        used = ins.reads | ins.writes
        need = None
        for k, (wr, tag, _) in enumerate(self.pending):
            if wr & used:
                need = k
        if need is not None:
            self._wait_for(need)
        self.out.append(ins)
        if ins.kind == 'ds':
            self.pending.append((ins.writes, ins.tag, self.n_mfma))
        if ins.kind == 'mfma':
            self.n_mfma += 1

    def drain(self, text):
        self.out.append(I(text, 'wait'))
        self.pending = []
        if 'vmcnt(0)' in text:
            self.vmq = []

    def state(self, shift=0):
        return [(sorted(w), (t[0] - shift, t[1]), n - self.n_mfma) for (w, t, n) in self.pending]


def generate(layout, ovl=None, cs=False):
    G = Gen(layout, ovl, cs)
    H = Hazards()
    E = H.emit
    for ins in G.entry_code():
        E(ins)
    if ovl:
        # ---- entry of the overlapped form: the LDS is free (the previous statement ended behind a barrier) -> this tile's K tiles 0 and 1 are
        # staged HERE, behind the bias loads, and the previous tile's accumulators are packed to bf16 while they travel
        for ins in G.readout():
            E(ins)
        for tile in range(2):
            for opnd in 'AB':
                E(I(f's_cmp_gt_i32 {op("s_live")}, 0', 'salu'))
                E(I(f's_cselect_b32 s{SRD[opnd] + 2}, s{SRD[opnd] + 2}, 0', 'salu'))
                for piece in G.dma_group(opnd, tile, tile):
                    for ins in piece:
                        if OPTS.get('nostage0') and ins.kind == 'dma':       # timing-only (wrong results): what would a tile boundary cost if K tiles 0 and 1 were already there?
                            continue
                        E(ins)
                E(I(f's_add_u32 {op(f"s_off{opnd}")}, {op(f"s_off{opnd}")}, {op(f"s_kt{opnd}")}', 'salu'))
            E(I(f's_sub_u32 {op("s_live")}, {op("s_live")}, 1', 'salu'))
        H.out.append(I('s_waitcnt vmcnt(0)' if OPTS.get('nostage0') else 's_waitcnt vmcnt(32)', 'wait'))        # the 8 bias loads (oldest) have landed, the 32 staging pieces stay in flight
        for ins in G.readout_math():
            E(ins)
    # ---- K tiles 0 and 1 have landed for every wave -> barrier -> fragments of (tile 0, k-step 0)
    H.drain('s_waitcnt vmcnt(0) lgkmcnt(0)')
    E(I('s_barrier', 'barrier'))
    for j in range(4):
        for ins in G.read_frag('B', j, 0, 0, (0, 'B')):
            E(ins)
    for i in range(8):
        for ins in G.read_frag('A', i, 0, 0, (0, 'A')):
            E(ins)

    def emit_tile(t, tag, first, fillers=None):
        for bb, gaps in G.tile(t, tag, first, fillers):
            for g, m in enumerate(bb):
                E(m)
                for ins in gaps[g]:
                    E(ins)
        E(I(f's_sub_u32 {op("s_cnt")}, {op("s_cnt")}, 1', 'salu'))
        E(I(f's_cmp_eq_u32 {op("s_cnt")}, 0', 'salu'))
        E(I('s_cbranch_scc1 EXIT%=', 'branch'))

    H.in_loop = True
    H.exact_vm = True
    if ovl:
        # the first FIVE K tiles (one period of the ring) are peeled: they carry the previous tile's 32 stores (two lane exchanges + one 16-byte
        # store each), one every 18 MFMAs.  The wrapper takes this form only for contractions of >= OVL_MIN_NK K tiles.
        fill = {}

        def put(g, inss):
            fill.setdefault(g // 128, {}).setdefault(((g % 128) // 64, g % 64), []).extend(inss)
        for u in range(32):
            put(24 + 18 * u, G.store_unit(u))
        for t in range(5):
            emit_tile(t, t, t == 0, fill.get(t))
        first_loop = 5
    else:
        emit_tile(0, 0, True)                 # the first K tile of an output tile starts the accumulators (C = 0)
        first_loop = 1
    H.exact_vm = False
    s1 = H.state(0)
    H.out.append(I('LOOP%=:', 'label'))   # memory order of the loop body: ring positions 1, 2, 3, 4, 0 (0 .. 4 behind the peeled period)
    for t in range(first_loop, first_loop + 5):
        emit_tile(t % RING, t, False)
    assert H.state(RING) == s1, 'loop-carried LDS state differs'
    E(I('s_branch LOOP%=', 'branch'))
    H.in_loop = False
    H.out.append(I('EXIT%=:', 'label'))
    H.drain('s_waitcnt vmcnt(0) lgkmcnt(0)')
    H.out.append(I('s_barrier', 'barrier'))          # every wave has left the LDS: the next output tile may be staged
    H.out.append(I('s_nop 7\n\ts_nop 7', 'nop'))
    return H.out, G


def generate_drain(ovl='bf16'):
    """the epilogue of a workgroup's LAST output tile in the overlapped form: the same read-out and the same stores, back to back"""
    G = Gen('nt', ovl)
    H = Hazards()
    H.exact_vm = True
    for ins in G.readout():
        H.emit(ins)
    H.drain('s_waitcnt vmcnt(0)')
    for ins in G.readout_math():
        H.emit(ins)
    for u in range(32):
        for ins in G.store_unit(u):
            H.emit(ins)
    H.drain('s_waitcnt vmcnt(0)')
    return H.out, G


OVL_MIN_NK = 6
ACC_IO = ', '.join(f'"+{{a[{acc(i, j)}:{acc(i, j) + 3}]}}"(c[{i}][{j}])' for i in range(8) for j in range(8))
ACC_OUT = ', '.join(f'"={{a[{acc(i, j)}:{acc(i, j) + 3}]}}"(c[{i}][{j}])' for i in range(8) for j in range(8))


def render(stream, G):
    lines = []
    for ins in stream:
        lines += ins.text.split('\n\t')
    body = '\n'.join(f'    "{ln}\\n\\t"' for ln in lines)
    srw = ['s_offA', 's_offB', 's_live', 's_cnt'] + (['s_crow'] if G.ovl else []) + (['s_csk'] if G.cs else [])
    outs = (ACC_IO if G.ovl else ACC_OUT) + ',\n      ' + f'"+{{s[{SRD["A"]}:{SRD["A"] + 3}]}}"(srdA), "+{{s[{SRD["B"]}:{SRD["B"] + 3}]}}"(srdB),\n      ' + \
        ', '.join(f'[{n}] "+&s"({n})' for n in srw) + ', [s_t] "=&s"(s_t)'
    if G.cs:      # the column-sum accumulators leave as early-clobber physical-register outputs (no input operand may be given one of them)
        outs += ', [s_csgo] "=&s"(s_csgo),\n      ' + ', '.join(f'"=&{{v[{G.CS + 4 * i}:{G.CS + 4 * i + 3}]}}"(cs[{i}])' for i in range(8))
    sin = ['s_ldsw', 's_itA', 's_halfA', 's_ktA', 's_itB', 's_halfB', 's_ktB'] + (['srdBias', 'srdC', 's_cstep', 's_cscols', 's_cscale'] if G.ovl else []) + \
        (['s_csmask'] if G.cs else [])
    ins_ = ', '.join(f'[{n}] "v"({n})' for n in G.vops()) + ',\n      ' + ', '.join(f'[{n}] "s"({n})' for n in sin)
    clob = ', '.join(f'"v{i}"' for i in range(G.n_hand)) + ', "vcc", "scc", "memory"'
    return (f'// GENERATED by gen_gemm4w.py ({G.layout}{", overlapped " + str(G.ovl) + " epilogue" if G.ovl else ""}{", + column sums of A" if G.cs else ""}) -- do not edit; see that file for the register map and the schedule\n'
            'asm volatile(\n' + body + '\n    : ' + outs + '\n    : ' + ins_ + '\n    : ' + clob + ');\n')


def render_drain(stream, G):
    lines = []
    for ins in stream:
        lines += ins.text.split('\n\t')
    body = '\n'.join(f'    "{ln}\\n\\t"' for ln in lines)
    outs = ACC_IO + ',\n      [s_crow] "+&s"(s_crow)'
    ins_ = ', '.join(f'[{n}] "v"({n})' for n in ['voffBias', 'voffC', 'colv']) + ',\n      ' + \
        ', '.join(f'[{n}] "s"({n})' for n in ['srdBias', 'srdC', 's_cstep', 's_cscols', 's_cscale'])
    clob = ', '.join(f'"v{i}"' for i in range(G.n_hand)) + ', "vcc", "scc", "memory"'
    return ('// GENERATED by gen_gemm4w.py (epilogue of the last output tile, overlapped form) -- do not edit\n'
            'asm volatile(\n' + body + '\n    : ' + outs + '\n    : ' + ins_ + '\n    : ' + clob + ');\n')


FILES = [(f'gemm4w_body_{l}.inc', l, False, False) for l in KINDS] + [(f'gemm4w_body_{l}_ovl.inc', l, 'bf16', False) for l in ('nt', 'nn')] + \
    [('gemm4w_body_tn_cs.inc', 'tn', False, True)]


def generate_all(outdir):
    for name, layout, ovl, cs in FILES:
        stream, G = generate(layout, ovl, cs)
        with open(os.path.join(outdir, name), 'w') as f:
            f.write(render(stream, G))
        yield name, stream
    for name, kind in (('gemm4w_drain_ovl.inc', 'bf16'),):
        stream, G = generate_drain(kind)
        with open(os.path.join(outdir, name), 'w') as f:
            f.write(render_drain(stream, G))
        yield name, stream


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    for name, stream in generate_all(here):
        if '-v' in sys.argv:
            kinds = {}
            for ins in stream:
                kinds[ins.kind] = kinds.get(ins.kind, 0) + 1
            print(name, kinds)
