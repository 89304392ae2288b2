#!/usr/bin/env python3
"""PRICE the two-waves-per-SIMD forms of the single-pass attention backward before building one.

Writes scripts/ubench_bwd_forms.hip: timing-only kernels (no global-memory traffic, WRONG arithmetic) whose waves issue, per 64-query x 256-key
tile of one workgroup, exactly the instruction multiset each candidate structure would issue -- MFMAs (v_mfma_f32_32x32x16_bf16, operands fed
by the LDS reads through counted lgkmcnt waits, as in csrc/gen_attn_bwd_sp.py whose instruction builders and hazard pass are reused), LDS reads /
writes of the right widths on conflict-free addresses, v_exp / v_mul / v_cvt_pk, the barriers -- evenly interleaved into the MFMA gaps:

  A   today's stream: 4 waves x 64 keys, ONE wave per SIMD (512 registers):            80 MFMA | 32 b128 + 64 b64_tr + 2 read2 + 16 writes | 64 exp 64 mul 72 cvt 16 unpack | 1 barrier
  B   VERDICT (ii): 8 waves x 32 keys in a 256-key workgroup, two waves per SIMD:       40 MFMA | 32 b128 + 48 b64_tr + 8 writes            | 32 exp 32 mul 36 cvt  8 unpack | 1 barrier
      (every wave still needs the row constants, the Q / dO row fragments and the Q^T / dO^T fragments of ALL 64 queries for half the keys)
  C   VERDICT (i): role split, per SIMD one wave owns S / dP / exp / dS, one the dV^T / dK^T / dQ products; P and dS cross the LDS:
      softmax wave   32 MFMA | 32 b128 + 32 writes (P and dS)          | 64 exp 64 mul 64 cvt          | 2 barriers
      gradient wave  48 MFMA | 16 b128 (P / dS rows) + 64 b64_tr + 2   |  8 cvt 16 unpack              | 2 barriers
  A2  control: stream A's per-wave work at TWO waves per SIMD (8 waves, the register file would have to be twice as large): what sharing
      a SIMD buys when nothing else changes -- the upper bound any two-wave form is chasing

Each kernel runs `tiles` tile passes per wave on all 256 CUs and stamps s_memtime / s_memrealtime around the loop.  The host part prints, per
form: cycles per tile pass of a CU, the share of that time the matrix pipe is busy (MFMAs per SIMD and tile x 32 cycles), the in-kernel clock,
and the LDS bytes per tile.  Register FEASIBILITY is a separate question (table in profiles/r6_attn_bwd_two_waves_pricing.txt); this prices issue + LDS.
"""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('gen_sp', os.path.join(HERE, '..', 'pixparse_amd', 'csrc', 'gen_attn_bwd_sp.py'))
G = importlib.util.module_from_spec(spec)
spec.loader.exec_module(G)
I, Hazards = G.I, G.Hazards

# register blocks (the same in every form, so that a form fits 128 VGPR + 128 AGPR): fragments F (12 tuples of 4), S-type accumulators (2 tuples of 16),
# VALU work registers W (32), gradient-type accumulators in the AGPR file (8 tuples of 16)
NF = 12
F0, S0, W0 = 0, 48, 80
N_HAND = 112


def mfma(m, kind):
    """MFMA m reads fragment tuples m mod 12 (A) and m + 6 mod 12 (B): every tuple is consumed every sixth MFMA"""
    a = ('v', F0 + 4 * (m % NF))
    b = ('v', F0 + 4 * ((m + NF // 2) % NF))
    if kind == 'S':
        d = ('v', S0 + 16 * (m % 2))
    else:
        d = ('a', 16 * (m % 8))
    return G.mfma(d, a, b, d, 16, 4, 4, 16)


class Fill:
    """the fillers of one tile pass as DESCRIPTORS; an LDS read becomes an instruction when it is placed (its destination depends on the gap:
    a read issued behind MFMA i refills the tuple MFMA i has just consumed, whose next reader is six MFMAs = ~200 cycles away -- the real stream's
    prefetch distances are of that order, so no read is waited for right behind its issue)"""
    def __init__(self):
        self.w = 0       # running work register
        self.o = 0       # running LDS offset
        self.nr = 0      # running LDS read: consecutive reads refill the A tuple and the B tuple of the MFMA in front of them in turn

    def make(self, desc, m):
        kind = desc[0]
        self.o += 1
        if kind in ('b128', 'b64tr'):
            self.nr += 1
        if kind == 'b128':
            slot = (m + (NF // 2) * (self.nr & 1)) % NF
            return G.ds_read_b128(F0 + 4 * slot, 'adbc' if desc[1] else 'ad128', 1024 * (self.o % 16))
        if kind == 'b64tr':
            slot = (m + (NF // 2) * (self.nr & 1)) % NF
            return G.ds_read_tr('v', F0 + 4 * slot + 2 * ((self.nr >> 1) & 1), 'ad64', 512 * (self.o % 32))
        if kind == 'read2':
            return I(f'ds_read2st64_b64 v[{W0 + 28}:{W0 + 31}], %[ad64] offset0:{2 * (self.o % 8)} offset1:{2 * (self.o % 8) + 1}', 'ds', reads=['ad64'],
                     writes=G.regs('v', W0 + 28, 4))
        if kind == 'write':
            return G.ds_write_b64('adw', W0 + 2 * (self.o % 8), 32768 + 512 * (self.o % 32))
        r = W0 + self.w % 24
        self.w += 1
        if kind == 'exp':
            return G.v_exp(r)
        if kind == 'mul':
            return G.v_mul(r, r, W0 + (self.w + 7) % 24)
        if kind == 'cvt':
            return G.v_cvt(W0 + 2 * (self.w % 8), r, W0 + (self.w + 1) % 24)
        src = W0 + (self.w + 9) % 24        # (the real stream unpacks a running tile it read a whole pass earlier: no wait behind the read)
        return G.valu(f'v_lshlrev_b32 v{r}, 16, v{src}', [f'v{src}'], [f'v{r}'])


def interleave(lists):
    """round-robin merge keeping each list's order: an even mix of instruction classes along the pass"""
    lists = [l for l in lists if l]
    out, idx = [], [0] * len(lists)
    total = sum(len(l) for l in lists)
    while len(out) < total:
        best, bl = None, -1.0
        for j, l in enumerate(lists):
            if idx[j] < len(l):
                frac = idx[j] / len(l)
                if best is None or frac < bl:
                    best, bl = j, frac
        out.append(lists[best][idx[best]])
        idx[best] += 1
    return out


def mix(b128_bcast=0, b128=0, b64tr=0, read2=0, write=0, exp=0, mul=0, cvt=0, unpack=0):
    return interleave([[('b128', True)] * b128_bcast + [('b128', False)] * b128, [('b64tr',)] * b64tr, [('write',)] * write + [('read2',)] * read2,
                       [('exp',)] * exp, [('mul',)] * mul, [('cvt',)] * cvt, [('unpack',)] * unpack])


def tile_pass(n_s, n_g, descs, barriers, fl, m0):
    """one tile pass: MFMA backbone (S- and gradient-type MFMAs alternating in proportion) with the fillers spread evenly; barriers at equal distances"""
    n = n_s + n_g
    kinds = []
    s_left, g_left = n_s, n_g
    for m in range(n):
        if (s_left * n >= n_s * (n - m) and s_left > 0) or g_left == 0:
            kinds.append('S'); s_left -= 1
        else:
            kinds.append('G'); g_left -= 1
    gaps = [[] for _ in range(n)]
    for j, d in enumerate(descs):
        g = (j * n) // max(1, len(descs))
        gaps[g].append(fl.make(d, m0 + g))
    bar_at = [(b * n) // barriers for b in range(barriers)]
    return kinds, gaps, bar_at


def stream(n_s, n_g, make_fillers, barriers, passes=2):
    H = Hazards()
    E = H.emit
    H.out.append(I('s_memtime %[t0]', 'salu'))
    H.out.append(I('s_memrealtime %[r0]', 'salu'))
    E(G.salu('s_mov_b32 %[s_cnt], %[s_iters]'))
    H.out.append(I('LOOP%=:', 'label'))
    m = 0
    fl = Fill()
    for p in range(passes):
        kinds, gaps, bar_at = tile_pass(n_s, n_g, make_fillers(), barriers, fl, m)
        for i, kd in enumerate(kinds):
            if i in bar_at:
                H.drain('s_waitcnt lgkmcnt(0)')
                E(I('s_barrier', 'barrier'))
            E(mfma(m, kd)); m += 1
            for ins in gaps[i]:
                E(ins)
    E(G.salu('s_sub_u32 %[s_cnt], %[s_cnt], 1'))
    E(G.salu('s_cmp_eq_u32 %[s_cnt], 0'))
    E(I('s_cbranch_scc0 LOOP%=', 'branch'))
    H.drain('s_waitcnt lgkmcnt(0)')
    H.out.append(I('s_nop 7\n\ts_nop 7', 'nop'))
    H.out.append(I('s_memtime %[t1]', 'salu'))
    H.out.append(I('s_memrealtime %[r1]\n\ts_waitcnt lgkmcnt(0)', 'salu'))
    return H.out


def render(st):
    lines = []
    for ins in st:
        lines += ins.text.split('\n\t')
    body = '\n'.join(f'      "{ln}\\n\\t"' for ln in lines)
    clob = ', '.join(f'"v{i}"' for i in range(N_HAND)) + ', ' + ', '.join(f'"a{i}"' for i in range(128)) + ', "vcc", "scc", "memory"'
    return ('    asm volatile(\n' + body + '\n      : [t0] "=&s"(t0), [t1] "=&s"(t1), [r0] "=&s"(r0), [r1] "=&s"(r1), [s_cnt] "=&s"(cnt)\n'
            '      : [ad128] "v"(a128), [adbc] "v"(abc), [ad64] "v"(a64), [adw] "v"(aw), [s_iters] "s"(iters)\n      : ' + clob + ');\n')


# ---- the forms -----------------------------------------------------------------------------------------------------------------
def fill_A():
    return mix(b128_bcast=16, b128=16, b64tr=64, read2=2, write=16, exp=64, mul=64, cvt=72, unpack=16)


def fill_B():
    return mix(b128_bcast=16, b128=16, b64tr=48, read2=1, write=8, exp=32, mul=32, cvt=36, unpack=8)


def fill_C_soft():
    return mix(b128_bcast=16, b128=16, write=32, exp=64, mul=64, cvt=64)


def fill_C_grad():
    return mix(b128=16, b64tr=64, read2=2, cvt=8, unpack=16)


FORMS = [
    # name, threads, [(role predicate, n_s, n_g, fillers, barriers)], MFMAs per SIMD and tile, LDS bytes per CU and tile
    ('A', 256, [('true', 32, 48, fill_A, 1)]),
    ('A2', 512, [('true', 32, 48, fill_A, 1)]),
    ('B', 512, [('true', 16, 24, fill_B, 1)]),
    ('C', 512, [('wave < 4', 32, 0, fill_C_soft, 2), ('wave >= 4', 0, 48, fill_C_grad, 2)]),
]


def lds_bytes(descs):
    return sum({'b128': 1024, 'b64tr': 512, 'write': 512, 'read2': 1024}.get(d[0], 0) for d in descs)


def main():
    out = ['// GENERATED by scripts/gen_ubench_bwd_forms.py -- timing-only instruction streams (see that file)',
           '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>', '#include <vector>', '#include <algorithm>', '#include <cstdint>', '',
           '#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e_)); return 1; } } while (0)', '']
    meta = []
    for name, threads, roles in FORMS:
        out.append(f'__global__ __launch_bounds__({threads}, 1) void form_{name}(unsigned long long* stamps, int iters) {{')
        out.append('  extern __shared__ __attribute__((aligned(16))) char smem[];')
        out.append('  // random bits into the LDS (operand toggling as with real activations)')
        out.append(f'  for (int i = threadIdx.x; i < 163840 / 4; i += {threads}) {{ unsigned x = (unsigned)i * 0x9E3779B1u + blockIdx.x * 0x7F4A7C15u; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;')
        out.append('    ((unsigned*)smem)[i] = (x & 0x807f807fu) | 0x3f003f00u; }   // bf16 pairs in +-[0.5, 1)')
        out.append('  __syncthreads();')
        out.append('  const int lane = threadIdx.x & 63;')
        out.append('  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);')
        out.append('  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;')
        out.append('  unsigned a128 = base + (unsigned)(lane * 16 + (wave & 3) * 16384), abc = base + (unsigned)((lane >> 5) * 16 + (wave & 3) * 16384);')
        out.append('  unsigned a64 = base + 65536u + (unsigned)(lane * 8 + (wave & 3) * 16384), aw = base + 131072u - 32768u + (unsigned)(lane * 8 + (wave & 7) * 2048);')
        out.append('  unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0; unsigned cnt;')
        mf, lb = 0, 0
        for pred, n_s, n_g, fillers, barriers in roles:
            st = stream(n_s, n_g, fillers, barriers)
            out.append(f'  if ({pred}) {{')
            out.append(render(st))
            out.append('  }')
            waves = threads // 64
            nw = waves if pred == 'true' else waves // 2
            mf += (n_s + n_g) * nw
            lb += lds_bytes(fillers()) * nw
        out.append('  if (lane == 0) { unsigned long long* p = stamps + ((size_t)blockIdx.x * 8 + wave) * 4; p[0] = t0; p[1] = t1; p[2] = r0; p[3] = r1; }')
        out.append('}')
        out.append('')
        meta.append((name, threads, mf // 4, lb))
    out.append('int main(int argc, char** argv) {')
    out.append('  const int iters = argc > 1 ? atoi(argv[1]) : 1500;      // loop iterations of two tile passes each')
    out.append('  unsigned long long* d; CK(hipMalloc(&d, 256 * 8 * 4 * sizeof(unsigned long long)));')
    out.append('  std::vector<unsigned long long> h(256 * 8 * 4);')
    out.append('  printf("form  waves/SIMD  MFMA/SIMD/tile  LDS KiB/tile  cycles/tile  pipe busy  LDS B/clk/CU  clock GHz  us/tile\\n");')
    for name, threads, mf, lb in meta:
        out.append('  {')
        out.append(f'    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&form_{name}), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));')
        out.append(f'    for (int rep = 0; rep < 3; ++rep) {{ hipLaunchKernelGGL(form_{name}, dim3(256), dim3({threads}), 163840, 0, d, rep < 2 ? iters : iters); CK(hipDeviceSynchronize()); }}')
        out.append('    CK(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));')
        out.append(f'    const int waves = {threads // 64};')
        out.append('    std::vector<double> cyc, clk;')
        out.append('    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { const unsigned long long* p = &h[((size_t)b * 8 + w) * 4];')
        out.append('      cyc.push_back((double)(p[1] - p[0]) / (2.0 * iters)); clk.push_back((double)(p[1] - p[0]) / (double)(p[3] - p[2]) * 0.1); }')
        out.append('    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());')
        out.append('    const double c = cyc[cyc.size() / 2], f = clk[clk.size() / 2];')
        out.append(f'    printf("{name:4s}  %d           {mf:4d}            {lb / 1024.0:6.1f}      %8.0f     %5.1f %%     %6.1f      %5.2f    %6.3f\\n", waves / 4, c, 100.0 * {mf} * 32.0 / c, {lb}.0 / c, f, c / f / 1000.0);')
        out.append('  }')
    out.append('  return 0;')
    out.append('}')
    with open(os.path.join(HERE, 'ubench_bwd_forms.hip'), 'w') as f:
        f.write('\n'.join(out) + '\n')
    for name, threads, mf, lb in meta:
        print(name, 'threads', threads, 'MFMA per SIMD and tile', mf, 'LDS KiB per CU and tile', lb / 1024.0)


if __name__ == '__main__':
    main()
