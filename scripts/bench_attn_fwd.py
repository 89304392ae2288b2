"""forward attention, hand-placed stream (crl_attn_fwd_set_mode 0) against the compiler-scheduled kernel (mode 1), alternating on one box.
    python scripts/bench_attn_fwd.py [lib.so ...]      # extra libraries = timing-only variants of the stream (scripts/ab_f4w.sh)"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeline(rec, what, per_wg=True):
    """per-CU occupancy of the last launch from the diagnostic build's records: [stream cycles, t_entry, t_end (10-ns ticks), HW_ID | XCC_ID << 32]"""
    import collections
    t0 = int(rec[:, 1].min())
    t_end = int(rec[:, 2].max()) - t0
    cus = collections.defaultdict(list)
    for i in range(rec.shape[0]):
        hw = int(rec[i, 3])
        key = ((hw >> 32) & 0xff, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)       # (xcc, se, sh, cu)
        cus[key].append((int(rec[i, 1]) - t0, int(rec[i, 2]) - t0, i))
    busy2 = busy1 = 0
    last_start = []
    for key, ivs in cus.items():
        ev = sorted([(a, 1) for a, b, _ in ivs] + [(b, -1) for a, b, _ in ivs])
        n, prev = 0, 0
        for t, d in ev:
            if n >= 2: busy2 += t - prev
            elif n == 1: busy1 += t - prev
            n += d; prev = t
        last_start.append(max(a for a, b, _ in ivs))
    pro = ((rec[:, 3] >> 40) & 0xffffff).float()
    dur = (rec[:, 2] - rec[:, 1]).float()
    if per_wg:
      print(f'  per workgroup: entry -> end of stream {float(dur.mean()) / 100:.1f} us (min {float(dur.min()) / 100:.1f}, max {float(dur.max()) / 100:.1f}), of which prologue {float(pro.mean()) / 100:.2f} us '
          f'(max {float(pro.max()) / 100:.2f}); stream {float(rec[:, 0].float().mean()):.0f} s_memtime ticks = {float(rec[:, 0].float().mean()) / (float((dur - pro).mean()) * 10):.3f} ticks per ns')
    xcc = collections.defaultdict(list)
    for key, ivs in cus.items():
        xcc[key[0]] += ivs
    print('  per XCD: ' + '  '.join(f'x{k}: {len(v)} wg, last end {max(b for a, b, _ in v) / 100:.0f} us, mean dur {sum(b - a for a, b, _ in v) / len(v) / 100:.0f}' for k, v in sorted(xcc.items())), flush=True)
    ends = sorted(max(b for a, b, _ in ivs) for ivs in cus.values())
    print(f'  CU finish times: min {ends[0] / 100:.0f} us, 10 % {ends[len(ends) // 10] / 100:.0f}, median {ends[len(ends) // 2] / 100:.0f}, 90 % {ends[9 * len(ends) // 10] / 100:.0f}, max {ends[-1] / 100:.0f}', flush=True)
    ncu = len(cus)
    per = sorted(len(v) for v in cus.values())
    print(f'  timeline ({what}): {ncu} CUs seen, kernel span {t_end / 100:.1f} us; CU time with two workgroups resident {100 * busy2 / (ncu * t_end):.1f} %, with one {100 * busy1 / (ncu * t_end):.1f} %, '
          f'idle {100 * (1 - (busy1 + busy2) / (ncu * t_end)):.1f} %; workgroups per CU min {per[0]} median {per[len(per) // 2]} max {per[-1]}; last workgroup start at {max(last_start) / 100:.1f} us, '
          f'median CU\'s last start {sorted(last_start)[len(last_start) // 2] / 100:.1f} us', flush=True)


NAMES = {0: 'stream (default)       ', 1: '32 q per wave          ', 4: 'stream, 2 waves / SIMD ', 5: 'stream, not persistent '}


def main():
    from pixparse_amd import hip, ops
    hip.load()
    dev = torch.device('cuda:0')
    lib = hip.load()
    stamps = None
    if hasattr(lib, 'crl_debug_f4w_stamps'):      # diagnostic builds: cycles of the stream statement per workgroup
        import ctypes
        stamps = torch.zeros(4 * 8 * 16 * 25, dtype=torch.int64, device=dev)
        lib.crl_debug_f4w_stamps.argtypes = [ctypes.c_void_p]
        lib.crl_debug_f4w_stamps(stamps.data_ptr())
    shapes = [('vit', 8, 16, 6189, 6189), ('cross', 8, 16, 1023, 6189)]
    one = 'one' in sys.argv
    if 'bwdtime' in sys.argv:       # the whole backward (delta + single pass + reduce) with and without the query split of the remainder chains, alternating
        for name, B, H, Nq, Nk in (('vit', 8, 16, 6189, 6189), ('cross', 8, 16, 1023, 6189)):
            D = H * 64
            g = torch.Generator(device=dev).manual_seed(1)
            qp = (torch.randn(B, Nq, D, generator=g, device=dev) * 0.125 * ops.LOG2E).to(torch.bfloat16)
            k, v = (torch.randn(B, Nk, D, generator=g, device=dev).to(torch.bfloat16) for _ in range(2))
            o = torch.empty(B, Nq, D, dtype=torch.bfloat16, device=dev); lse = torch.empty(B, H, Nq, device=dev)
            do = torch.randn(B, Nq, D, generator=g, device=dev).to(torch.bfloat16)
            dq, dk, dv = torch.empty_like(qp), torch.empty_like(k), torch.empty_like(v)
            delta = torch.empty(2, B, H, Nq, device=dev)
            ops.attn_fwd(qp, k, v, o, lse, H, 0.125, False, q_prescaled=True)
            for rnd in range(3):
                for split, persist in ((0, 0), (-1, 0), (-1, 1)):
                    hip.call('crl_attn_bwd_set_qsplit', split)
                    hip.call('crl_attn_bwd_set_persistent', persist)
                    run = lambda: ops.attn_bwd(qp, k, v, o, do, lse, delta, dq, dk, dv, H, 0.125, False, q_prescaled=True)
                    for _ in range(3): run()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20): run()
                    e1.record()
                    torch.cuda.synchronize()
                    print(f'{name:6s} backward, query split {"auto" if split else "off "}, {"persistent (tickets)" if persist else "one workgroup per chain"}: {e0.elapsed_time(e1) / 20:.3f} ms', flush=True)
            hip.call('crl_attn_bwd_set_qsplit', -1)
            hip.call('crl_attn_bwd_set_persistent', 1)
        return
    if 'bwd' in sys.argv and stamps is not None:      # where the workgroups of the single-pass backward ran and when (diagnostic build)
        B, H, N = 8, 16, 6189
        D = H * 64
        g = torch.Generator(device=dev).manual_seed(1)
        qkv = torch.randn(B, N, 3 * D, generator=g, device=dev).to(torch.bfloat16)
        qp = (qkv[:, :, :D].float() * 0.125 * ops.LOG2E).to(torch.bfloat16)
        k, v = qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
        o = torch.empty(B, N, D, dtype=torch.bfloat16, device=dev); lse = torch.empty(B, H, N, device=dev)
        do = torch.randn(B, N, D, generator=g, device=dev).to(torch.bfloat16)
        dqkv = torch.empty(B, N, 3 * D, dtype=torch.bfloat16, device=dev); delta = torch.empty(2, B, H, N, device=dev)
        ops.attn_fwd(qp, k, v, o, lse, H, 0.125, False, q_prescaled=True)
        for rep in range(3):
            stamps.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.attn_bwd(qp, k, v, o, do, lse, delta, dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], H, 0.125, False, q_prescaled=True)
            e1.record()
            torch.cuda.synchronize()
            rec = stamps.view(-1, 4).cpu()
            rec = rec[rec[:, 1] > 0]
            print(f'backward (delta + single pass + reduce) {e0.elapsed_time(e1):.3f} ms, {rec.shape[0]} workgroups of the single pass; chain lengths {sorted(set(rec[:, 0].tolist()))}')
            timeline(rec, 'single-pass backward', per_wg=False)
        return
    if one:
        shapes = shapes[:1]
    for name, B, H, Nq, Nk in shapes:
        D = H * 64
        g = torch.Generator(device=dev).manual_seed(1)
        qkv = torch.randn(B, max(Nq, Nk), 3 * D, generator=g, device=dev).to(torch.bfloat16)
        q, k, v = qkv[:, :Nq, :D], qkv[:, :Nk, D:2 * D], qkv[:, :Nk, 2 * D:]
        qp = (q.float() * 0.125 * ops.LOG2E).to(torch.bfloat16)
        o = torch.empty(B, Nq, D, dtype=torch.bfloat16, device=dev)
        lse = torch.empty(B, H, Nq, device=dev)
        fl = 4.0 * Nq * Nk * 64 * B * H
        ref = None
        for rnd in range(1 if one else 3):
            for mode in ((0, 4) if one else ((1, 0, 5) if 'persist' in sys.argv else (1, 0, 4))):
                hip.call('crl_attn_fwd_set_mode', 0 if mode == 5 else mode)
                hip.call('crl_attn_fwd_set_persistent', 0 if mode == 5 else 1)
                if mode == 4 and os.environ.get('F4W_OCC2', '0') != '1':
                    continue
                for _ in range(5):
                    ops.attn_fwd(qp, k, v, o, lse, H, 0.125, False, q_prescaled=True)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 40
                e0.record()
                for _ in range(n):
                    ops.attn_fwd(qp, k, v, o, lse, H, 0.125, False, q_prescaled=True)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / n
                extra = ''
                if mode == 1:
                    ref = o.clone()
                elif ref is not None:
                    extra = f'  rel diff vs mode 1: {float((o.float() - ref.float()).norm() / ref.float().norm()):.2e}'
                if stamps is not None and mode != 1 and name == 'vit':
                    nwg = B * H * ((Nq + 255) // 256)
                    rec = stamps[:4 * nwg].view(nwg, 4).cpu()
                    rec = rec[rec[:, 1] > 0]              # workgroups that ran the stream (a short last block runs the 32-queries-per-wave body: no record)
                    st = rec[:, 0].float()
                    if 'timeline' in sys.argv:
                        timeline(rec, NAMES[mode])
                    extra += f'  stream cycles per workgroup: mean {float(st.mean()):.0f} min {float(st.min()):.0f} max {float(st.max()):.0f} = {float(st.mean()) / ((Nk + 63) // 64):.0f} per key tile'
                print(f'{name:6s} B{B} H{H} Nq{Nq} Nk{Nk} round {rnd} {NAMES[mode]}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF/s{extra}', flush=True)
        hip.call('crl_attn_fwd_set_mode', 0)
        hip.call('crl_attn_fwd_set_persistent', 1)


if __name__ == '__main__':
    main()
